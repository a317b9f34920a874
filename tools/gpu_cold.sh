# cold single-frame cost (tools/cold_frame.py <workload> <cold_bytes: -1 default, 0 = full-size arena at once> <warm frames>), fresh processes back to back
set -x
cd $GRAFT_REPO_ROOT
run() { timeout 300 python tools/cold_frame.py $1 $2 $3; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "first_frame or host_entry or batching" 2>&1 | tail -2
sleep 6
echo "== single-frame processes back to back (the reference's usage)"
run c3 -1 0; run c3 -1 0; run c2 -1 0; run c2 -1 0; run c4 -1 0; run c4 -1 0
echo "== right after a process that used a full-size arena (175 GB)"
run c3 0 0; run c3 -1 0; run c3 0 0; run c2 -1 0
