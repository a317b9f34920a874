#!/bin/bash
# Register / scratch / occupancy figures of every kernel of rayn_amd/csrc/kernels.hip as the product flags compile it (policy 0).
#   tools/kernel_resources.sh [kernels.hip] [extra hipcc flags]
here=$(cd "$(dirname "$0")/.." && pwd)
src=${1:-$here/rayn_amd/csrc/kernels.hip}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -align-all-nofallthru-blocks=5 -fPIC \
  -I"$here/rayn_amd/csrc" -DRAYN_FMA_POLICY=0 -DRAYN_KNS=rayn_p0 "$@" -Rpass-analysis=kernel-resource-usage -c -o /tmp/kernel_resources.o "$src" 2>&1 | python3 -c "
import re, subprocess, sys
cur = None; d = {}
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: cur = m.group(1); d[cur] = {}; continue
    m = re.search(r'remark:\s+(VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)', l)
    if m and cur: d[cur][m.group(1).split(' [')[0]] = int(m.group(2))
for k, v in d.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    print(re.sub(r'\(.*', '', name).replace('rayn_p0::', '').replace('void ', ''), v)
"
