cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/fuzz_parity.py 80 41000 2>&1 | tail -2
timeout 300 python tools/share_profile.py 3 8 c3 2>&1 | tail -1
timeout 300 python tools/share_profile.py 3 8 c2 2>&1 | tail -1
timeout 300 python tools/share_profile.py 5 64 c5 2>&1 | tail -1
