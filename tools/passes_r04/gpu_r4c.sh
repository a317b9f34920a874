# round 4, third GPU pass: parity with worker 0 on the caller's stream + the small-share two-worker rule + scalar loads in the scatters; shares; cold
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { echo "== $*"; env RAYN_HIP_ENV_TUNING=1 "$@" timeout 300 python tools/share_profile.py $SH 2>&1 | tail -1 | cut -c1-330; }
SH="3 8 c2"; run X=1
SH="0 1 shipped"; run X=1
SH="3 8 c3"; run X=1; run RAYN_HIP_WORKERS=2 RAYN_HIP_WORKER_MIN_PATHS=0
SH="0 1 c2"; run X=1
for i in 1 2; do timeout 120 python tools/cold_breakdown.py shipped 0 2>&1 | tail -1; done
timeout 120 python tools/cold_breakdown.py c2 0 2>&1 | tail -1
timeout 300 python bench.py --workload shipped 2>&1 | tail -1 > gpurun_out/r04_bench_shipped_b.json; cut -c1-300 gpurun_out/r04_bench_shipped_b.json
