cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-roofline 2>&1 | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('VALUE', j['value'], 'ms', j['ms_per_step'])"; }
run RAYN_HIP_WORKERS=2
run RAYN_HIP_WORKERS=3
run RAYN_HIP_WORKERS=4
run RAYN_HIP_WORKERS=3 RAYN_HIP_PERSISTENT_BLOCKS=1024
run RAYN_HIP_WORKERS=4 RAYN_HIP_PERSISTENT_BLOCKS=1024
