set -x
TAG=${1:-v2}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/gpu_profile.sh c2_$TAG --workload c2
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_c2_$TAG.json
cat gpurun_out/bench_c2_$TAG.json | cut -c1-400
