cd $GRAFT_REPO_ROOT
rm -f /tmp/ready_*; python tools/overlap_probe.py 0 1 3
rm -f /tmp/ready_*; (python tools/overlap_probe.py 0 2 3 & python tools/overlap_probe.py 1 2 3 & wait)
rm -f /tmp/ready_*; (RAYN_HIP_PERSISTENT_BLOCKS=1024 python tools/overlap_probe.py 0 2 3 & RAYN_HIP_PERSISTENT_BLOCKS=1024 python tools/overlap_probe.py 1 2 3 & wait)
