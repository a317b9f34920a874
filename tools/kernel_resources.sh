#!/bin/bash
# Register / scratch / LDS use of every kernel in a compiled object (code-object metadata notes): no GPU needed.
# usage: bash tools/kernel_resources.sh [rayn_amd/csrc/kernels_p0.o]
OBJ=${1:-rayn_amd/csrc/kernels_p0.o}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$TMP/fb.bin $OBJ
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fb.bin --output=$TMP/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/dev.co | python3 -c "
import sys, re
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.%s:\s*(\S+)' % k, blk) or [None, '?'])[1]
    name = g('name')
    name = re.sub(r'^_ZN\d+rayn_p\d\d*', '', name)
    print(f\"{name[:60]:60s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s} spill_v {g('vgpr_spill_count'):>3s} spill_s {g('sgpr_spill_count'):>3s}\")
"
rm -rf $TMP
