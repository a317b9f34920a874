cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc.sh fetch_c2 "FETCH_SIZE" --workload c2 2>&1 | tail -22
bash tools/gpu_pmc.sh write_c2 "WRITE_SIZE" --workload c2 2>&1 | tail -22
