cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { echo "== $*"; env "$@" timeout 300 python bench.py --workload c2 --steps 1 --warmup 0 --cpu-seconds 0 2>&1 | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms']; print('VALUE', j['value'], 'extend', k['ms_extend'], 'shadow', k['ms_shadow'], 'setup', k['ms_shade'], 'resolve', k['ms_resolve'], 'total', k['ms_total'])"; }
run RAYN_HIP_BATCH_PATHS=33554432
run RAYN_HIP_BATCH_PATHS=67108864
run RAYN_HIP_BATCH_PATHS=134217728
run RAYN_HIP_BATCH_PATHS=268435456
