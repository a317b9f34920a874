cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { echo "== $*"; timeout 300 python bench.py --workload c2 --steps 1 --warmup 1 --cpu-seconds 0 "$@" 2>&1 | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); k=j['kernel_ms']; print('VALUE', j['value'], 'extend', k['ms_extend'], 'shadow', k['ms_shadow'], 'setup', k['ms_shade'], 'total', k['ms_total'])"; }
run --fma-policy 0
run --fma-policy 1
