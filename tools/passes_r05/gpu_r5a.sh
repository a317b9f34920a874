# r5 pass A: the whole GPU suite with the new entries (packed film, bench modes, bulb3 digests), then the metric's literally named workload
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py --workload bulb3 2>gpurun_out/r05_bench_bulb3_a.err | tail -1 > gpurun_out/r05_bench_bulb3_a.json
python - <<'PY'
import json
j=json.load(open('gpurun_out/r05_bench_bulb3_a.json'))
print('bulb3 VALUE', j['value'], j['ms_per_step'], j['kernel_ms'])
print({k:j['roofline'][k] for k in ('kernel','achieved','frac','flop_per_dist_eval','dist_evals','sdf_iterations','whole_frame')})
print(j['cpu_baseline'])
PY
tail -3 gpurun_out/r05_bench_bulb3_a.err
timeout 300 python tools/share_profile.py 3 8 c3 2>&1 | tail -1 | cut -c1-400
timeout 300 python tools/share_profile.py 3 8 bulb3 2>&1 | tail -1 | cut -c1-400
timeout 300 python bench.py --gpus 2 --single-process --share-gpu --workload c2 --steps 2 --warmup 1 --cpu-seconds 0 --check-film 2>&1 | tail -1 | cut -c1-1500
