cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python bench.py --workload c2 --steps 1 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms']; print('VALUE', j['value'], 'extend', k['ms_extend'], 'shadow', k['ms_shadow'], 'setup', k['ms_shade'], 'finish', k['ms_finish'], 'total', k['ms_total'])"; }
run RAYN_HIP_ABLATE=0
run RAYN_HIP_ABLATE=1
run RAYN_HIP_ABLATE=2
run RAYN_HIP_ABLATE=4
run RAYN_HIP_ABLATE=8
run RAYN_HIP_ABLATE=16
run RAYN_HIP_ABLATE=31
