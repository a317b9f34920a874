cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for st in 1 2 4 8; do rm -f /tmp/ready_*; for i in 0 1 2 3 4 5 6 7; do touch /tmp/ready_$i; done; python tools/overlap_probe.py 1 $st 5 2>&1 | grep first= || python tools/overlap_probe.py 0 $st 5; done
