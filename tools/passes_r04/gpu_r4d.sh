# round 4: shared shadow starts + k_shadow_first: parity, then shares
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
run() { echo "== $*"; env RAYN_HIP_ENV_TUNING=1 "$@" timeout 300 python tools/share_profile.py $SH x 2>&1 | tail -2 | cut -c1-330; }
SH="3 8 c3"; run X=1
SH="3 8 c2"; run X=1
SH="0 1 shipped"; run X=1
timeout 600 python tools/fuzz_parity.py 100 21000 2>&1 | tail -2
