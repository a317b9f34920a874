set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for SH in "3 8 c2" "0 1 shipped" "3 8 c3" "0 1 c2"; do timeout 300 python tools/share_profile.py $SH 2>&1 | tail -1 | cut -c1-300; done
timeout 600 python tools/fuzz_parity.py 100 41000 2>&1 | tail -1
