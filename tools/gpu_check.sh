# quick GPU check: parity suite + fuzz + one-eighth shares of c3 / c2 (per-class times)     usage: bash tools/gpu_check.sh [fuzz_n] [first_seed]
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/fuzz_parity.py ${1:-150} ${2:-13000} 2>&1 | tail -3
timeout 300 python tools/share_profile.py 3 8 c3 2>&1 | tail -1
timeout 300 python tools/share_profile.py 3 8 c2 2>&1 | tail -1
