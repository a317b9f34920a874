set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 300 python bench.py --workload small --steps 2 --warmup 1 --cpu-seconds 3 2>&1 | tail -3 | tee gpurun_out/bench_small.json
timeout 900 python bench.py --workload c2 --steps 1 --warmup 0 --cpu-seconds 10 2>&1 | tail -3 | tee gpurun_out/bench_c2_first.json
