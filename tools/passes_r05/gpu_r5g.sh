# r5 pass G: emulated N-GPU balance (every rank's share rendered in turn on the one GPU, best of two) for the 8-GPU config c4, the named workload bulb3, and c3 at N = 2, 4, 8
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(for A in "8 c4" "8 bulb3" "2 c3" "4 c3" "8 c3"; do timeout 900 python tools/share_balance.py $A 2>&1 | tail -1; done) | tee gpurun_out/r05_share_balance.txt
