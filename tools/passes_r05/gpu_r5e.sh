# r5 pass E: what the driver runs at round end, on a fresh box, at HEAD: smoke(), the GPU suite, `python bench.py` with no flags; then a long fuzz run
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 1200 python bench.py 2>gpurun_out/r05_bench_c3_driver_like.err | tail -1 > gpurun_out/r05_bench_c3_driver_like.json
python -c "
import json
j=json.load(open('gpurun_out/r05_bench_c3_driver_like.json')); print('VALUE', j['value'], j['ms_per_step'], j['named_workload']['value'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'], j['cold_ms'])"
timeout 1500 python tools/fuzz_parity.py 2400 90000 2>&1 | tail -3 | tee gpurun_out/r05_fuzz_2400.txt
