# round 4, first GPU pass: parity suite (new shipped / bulb / per-rank tests), the shipped workload's bench line, where a cold start goes, baseline shares
set -x
cd $GRAFT_REPO_ROOT  # (these scripts are run as: gpurun -- bash tools/passes_r04/<name>.sh)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --workload shipped 2>&1 | tail -1 > gpurun_out/r04_bench_shipped_a.json; cut -c1-600 gpurun_out/r04_bench_shipped_a.json
for i in 1 2; do timeout 120 python tools/cold_breakdown.py shipped 0 2>&1 | tail -1; done
for i in 1 2; do timeout 120 python tools/cold_breakdown.py shipped 1 2>&1 | tail -1; done
timeout 120 python tools/cold_breakdown.py c2 1 2>&1 | tail -1
timeout 300 python tools/share_profile.py 3 8 c3 2>&1 | tail -1
timeout 300 python tools/share_profile.py 3 8 c2 2>&1 | tail -1
timeout 300 python tools/share_profile.py 0 1 shipped 2>&1 | tail -1
