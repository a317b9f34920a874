set -x
TAG=${1:-v4}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
# per-kernel evidence is collected single-worker (the two-worker pipeline overlaps kernels, which inflates their durations)
export RAYN_HIP_WORKERS=1
bash tools/gpu_profile.sh c2 --workload c2 > /dev/null 2>&1
bash tools/gpu_pmc.sh fetch_c2 "FETCH_SIZE" --workload c2 > /dev/null 2>&1
bash tools/gpu_pmc.sh write_c2 "WRITE_SIZE" --workload c2 > /dev/null 2>&1
unset RAYN_HIP_WORKERS
bash tools/gpu_profile.sh c2_2workers --workload c2 > /dev/null 2>&1
python tools/pmc_to_json.py c2 profiles/r01_pmc_hbm_c2.json gpurun_out/prof_c2_kernel_stats.csv > /dev/null
cp profiles/r01_pmc_hbm_c2.json gpurun_out/
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_c2_$TAG.json
timeout 900 python bench.py --fma-policy 1 --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/bench_c2_fma1_$TAG.json
python -c "
import json
for f in ('gpurun_out/bench_c2_$TAG.json','gpurun_out/bench_c2_fma1_$TAG.json'):
    j=json.load(open(f)); print('VALUE', j['value'], j['ms_per_step']); print(j['kernel_ms']); print({k:j['roofline'][k] for k in ('kernel','achieved','frac','traffic')}); print(j['cpu_baseline'])"
