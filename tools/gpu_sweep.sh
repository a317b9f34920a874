cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { echo "== $*"; env "$@" timeout 300 python bench.py --workload c2 --steps 1 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms']; print('VALUE', j['value'], 'extend', k['ms_extend'], 'shadow', k['ms_shadow'], 'setup', k['ms_shade'], 'total', k['ms_total'])"; }
run RAYN_HIP_FAST_PATH=0
run RAYN_HIP_FAST_PATH=1
run RAYN_HIP_PREFETCH_EXTEND=8 RAYN_HIP_PREFETCH_SHADOW=8
run RAYN_HIP_PREFETCH_EXTEND=32 RAYN_HIP_PREFETCH_SHADOW=32
run RAYN_HIP_PREFETCH_EXTEND=48 RAYN_HIP_PREFETCH_SHADOW=48
