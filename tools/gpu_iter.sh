# quick iteration: parity tests + c2 bench (1 step) -> gpurun_out/iter_<tag>.json
set -x
TAG=${1:-x}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 python bench.py --workload c2 --steps 3 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 | tee gpurun_out/iter_$TAG.json | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('VALUE', j['value'], 'ms', j['ms_per_step']); print(j['kernel_ms']); print(j['roofline'])"
