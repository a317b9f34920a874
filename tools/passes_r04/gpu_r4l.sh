set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/share_profile.py 0 1 c2 2>&1 | tail -1 | cut -c1-300
timeout 300 python tools/share_profile.py 0 1 shipped 2>&1 | tail -1 | cut -c1-300
timeout 300 python tools/share_profile.py 3 8 c3 2>&1 | tail -1 | cut -c1-300
timeout 300 python tools/share_profile.py 0 1 bulb 2>&1 | tail -1 | cut -c1-300
timeout 600 python tools/fuzz_parity.py 150 61000 2>&1 | tail -1
