# cold single-frame cost, back-to-back processes (each line: a fresh process right after the previous one exited)
#   args of cold_frame.py: workload, cold_paths (-1 default, 0 = full-size arena at once = r2 behaviour), warm frames after the cold one
set -x
cd $GRAFT_REPO_ROOT
run() { timeout 300 python tools/cold_frame.py $1 $2 $3; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
sleep 6
echo "== single-frame processes back to back (the reference's usage)"
run c3 -1 0; run c3 -1 0; run c3 -1 0
run c2 -1 0; run c2 -1 0
run c4 -1 0; run c4 -1 0
echo "== r2 behaviour: full-size arena on the first frame"
run c3 0 0; run c3 0 0; run c2 0 0; run c2 0 0
echo "== after a multi-frame process (full-size arenas were in use)"
run c3 -1 1; run c3 -1 1; run c2 -1 2; run c2 -1 2
