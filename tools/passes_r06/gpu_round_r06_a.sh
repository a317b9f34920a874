# r6 evidence pass A (one gpurun call): what the driver runs (smoke, GPU suite, `python bench.py`), then per workload (c3 = bench default, bulb3 = the metric's
# literally named workload): rocprofv3 kernel stats (1 worker, 2 workers), PMC FETCH_SIZE / WRITE_SIZE and SQ passes, bench JSON lines; c2 / shipped / c4 lines.
# usage: bash tools/passes_r06/gpu_round_r06_a.sh <tag>      outputs under gpurun_out/ (copied to profiles/r06_* by hand) and profiles/r06_pmc_hbm_*.json
set -x
TAG=${1:-v1}
R=r06
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${R}_smoke_$TAG.txt
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/${R}_pytest_gpu_$TAG.txt
export RAYN_HIP_ENV_TUNING=1   # the library reads RAYN_HIP_WORKERS / _COLD_BYTES / ... only under this opt-in (include/rayn_hip.h)
for WL in c3 bulb3; do
  # per-kernel evidence is collected single-worker (the two-worker pipeline overlaps kernels, which inflates their durations)
  export RAYN_HIP_WORKERS=1
  export RAYN_HIP_COLD_BYTES=0   # the profiled frame is the only frame of its process: full-size batches at once (bench.py --no-cold)
  bash tools/gpu_profile.sh $WL --workload $WL > /dev/null 2>&1
  bash tools/gpu_pmc.sh fetch_$WL "FETCH_SIZE" --workload $WL > /dev/null 2>&1
  bash tools/gpu_pmc.sh write_$WL "WRITE_SIZE" --workload $WL > /dev/null 2>&1
  bash tools/gpu_pmc.sh sq1_$WL "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES" --workload $WL > /dev/null 2>&1
  bash tools/gpu_pmc.sh sq2_$WL "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU" --workload $WL > /dev/null 2>&1
  python tools/pmc_join.py gpurun_out/pmc_sq1_$WL.csv gpurun_out/pmc_sq2_$WL.csv > gpurun_out/${R}_${WL}_${TAG}_pmc_sq.csv
  unset RAYN_HIP_WORKERS
  bash tools/gpu_profile.sh ${WL}_2workers --workload $WL > /dev/null 2>&1
  unset RAYN_HIP_COLD_BYTES
  python tools/pmc_to_json.py $WL profiles/${R}_pmc_hbm_$WL.json gpurun_out/prof_${WL}_kernel_stats.csv gpurun_out/${R}_${WL}_${TAG}_pmc_sq.csv > gpurun_out/pmc_hbm_$WL.txt
  cp profiles/${R}_pmc_hbm_$WL.json gpurun_out/
  cp gpurun_out/prof_${WL}_kernel_stats.csv gpurun_out/${R}_${WL}_${TAG}_kernel_stats_1worker.csv
  cp gpurun_out/prof_${WL}_2workers_kernel_stats.csv gpurun_out/${R}_${WL}_${TAG}_kernel_stats_2workers.csv
  cp gpurun_out/pmc_fetch_$WL.csv gpurun_out/${R}_${WL}_${TAG}_pmc_fetch_size.csv
  cp gpurun_out/pmc_write_$WL.csv gpurun_out/${R}_${WL}_${TAG}_pmc_write_size.csv
  cat gpurun_out/pmc_hbm_$WL.txt
done
unset RAYN_HIP_ENV_TUNING
# the driver's own command (the PMC files of this very tree are in profiles/ now: the line quotes traffic, lanes_enabled and valu_issue for both workloads)
timeout 1200 python bench.py 2>gpurun_out/${R}_bench_c3_driver_like.err | tail -1 > gpurun_out/${R}_bench_c3_driver_like.json
python -c "
import json
j=json.load(open('gpurun_out/${R}_bench_c3_driver_like.json')); rf=j['roofline']; nw=j['named_workload']
print('c3 VALUE', j['value'], j['ms_per_step'], 'frac', rf['frac'], 'traffic', rf['traffic'], 'lanes', rf.get('lanes_enabled'), 'cpu', j['cpu_baseline']['value'], 'cold', j['cold_ms'])
print('named', nw.get('value'), nw.get('roofline', {}).get('frac'), nw.get('roofline', {}).get('lanes_enabled'), nw.get('roofline', {}).get('traffic'), nw.get('error'))
print(j['kernel_ms'])
for k, v in j['roofline_hbm']['kernels'].items(): print('   ', k[:40], v['ms'], v['frac'])"
timeout 900 python bench.py --workload bulb3 2>&1 | tail -1 > gpurun_out/${R}_bench_bulb3_$TAG.json
python -c "
import json
j=json.load(open('gpurun_out/${R}_bench_bulb3_$TAG.json')); rf=j['roofline']; print('bulb3 VALUE', j['value'], j['ms_per_step']); print(j['kernel_ms']); print({k:rf.get(k) for k in ('kernel','achieved','frac','flop_per_dist_eval','traffic','lanes_enabled','valu_issue','bulb_stage_occupancy','whole_frame')}); print(j['cpu_baseline'])
for k, v in j['roofline_hbm']['kernels'].items(): print('   ', k[:40], v['ms'], v['frac'])"
timeout 900 python bench.py --workload c3 --fma-policy 1 --cpu-seconds 0 --no-named 2>&1 | tail -1 > gpurun_out/${R}_bench_c3_${TAG}_fma1.json; cut -c1-220 gpurun_out/${R}_bench_c3_${TAG}_fma1.json
timeout 600 python bench.py --workload c2 2>&1 | tail -1 > gpurun_out/${R}_bench_c2_$TAG.json; cut -c1-220 gpurun_out/${R}_bench_c2_$TAG.json
timeout 600 python bench.py --workload bulb --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/${R}_bench_bulb_$TAG.json; cut -c1-220 gpurun_out/${R}_bench_bulb_$TAG.json
timeout 600 python bench.py --workload shipped 2>&1 | tail -1 > gpurun_out/${R}_bench_shipped_$TAG.json; cut -c1-400 gpurun_out/${R}_bench_shipped_$TAG.json
timeout 600 python bench.py --workload c4 --steps 1 --warmup 1 --no-roofline --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/${R}_bench_c4_$TAG.json; cut -c1-220 gpurun_out/${R}_bench_c4_$TAG.json
