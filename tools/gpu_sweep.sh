cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { echo "== $*"; timeout 600 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-roofline "$@" 2>&1 | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('VALUE', j['value'], 'ms', j['ms_per_step'])"; }
run --workload c2
run --workload c3 --steps 1
run --workload c2 --fma-policy 1
