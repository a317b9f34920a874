set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --workload c5 --steps 1 --warmup 0 --no-roofline --no-cold --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r04_bench_c5_v3.json; cut -c1-260 gpurun_out/r04_bench_c5_v3.json
