# Round evidence pass (one gpurun call): parity tests + fuzz, then per workload (c3 = bench default, c2):
#   rocprofv3 kernel stats (1 worker and 2 workers), PMC FETCH_SIZE / WRITE_SIZE passes, bench JSON lines.
# usage: bash tools/passes_r05/gpu_round_r05.sh <tag> [workloads...]      outputs under gpurun_out/ and profiles/${R}_*   (r5 copy of tools/gpu_round.sh: + bulb3, fuzz through the packed shares, bench modes)
set -x
TAG=${1:-v1}; shift
R=${ROUND:-r05}
WLS=${@:-c3 c2}
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1   # the library reads RAYN_HIP_WORKERS / _COLD_BYTES / ... only under this opt-in (include/rayn_hip.h)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/fuzz_parity.py ${FUZZ_N:-300} 7000 2>&1 | tail -4 | tee gpurun_out/${R}_fuzz_$TAG.txt
for WL in $WLS; do
  # per-kernel evidence is collected single-worker (the two-worker pipeline overlaps kernels, which inflates their durations)
  export RAYN_HIP_WORKERS=1
  export RAYN_HIP_COLD_BYTES=0   # the profiled frame is the only frame of its process: full-size batches at once (bench.py --no-cold)
  bash tools/gpu_profile.sh $WL --workload $WL > /dev/null 2>&1
  bash tools/gpu_pmc.sh fetch_$WL "FETCH_SIZE" --workload $WL > /dev/null 2>&1
  bash tools/gpu_pmc.sh write_$WL "WRITE_SIZE" --workload $WL > /dev/null 2>&1
  # VALU issue counters of the same single-worker frame (two passes of four counters)
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
  timeout 1200 python bench.py --workload $WL 2>&1 | tail -1 > gpurun_out/${R}_bench_${WL}_$TAG.json
  timeout 900 python bench.py --workload $WL --fma-policy 1 --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/${R}_bench_${WL}_${TAG}_fma1.json
  python -c "
import json
for f in ('gpurun_out/${R}_bench_${WL}_$TAG.json','gpurun_out/${R}_bench_${WL}_${TAG}_fma1.json'):
    j=json.load(open(f)); print('VALUE', j['value'], j['ms_per_step']); print(j['kernel_ms']); print({k:j['roofline'][k] for k in ('kernel','achieved','frac','traffic')}); print(j['cpu_baseline'])
    for k, v in j['roofline_hbm']['kernels'].items(): print('   ', k[:40], v['ms'], v['frac'])"
  cat gpurun_out/pmc_hbm_$WL.txt
done
# other configurations (one rank's share each) and the PCIe-inclusive rate of the host-buffer entry
timeout 600 python tools/share_profile.py 3 8 c3 2>&1 | tail -1 | sed 's/^/c3 1\/8 share: /'
timeout 600 python tools/share_profile.py 3 8 c2 2>&1 | tail -1 | sed 's/^/c2 1\/8 share: /'
timeout 600 python tools/share_profile.py 3 8 c4 2>&1 | tail -1 | sed 's/^/c4 1\/8 share: /'
timeout 600 python tools/share_profile.py 5 64 c5 2>&1 | tail -1 | sed 's/^/c5 1\/64 share: /'
timeout 600 python tools/share_profile.py 0 1 bulb 2>&1 | tail -1 | sed 's/^/bulb: /'
timeout 600 python tools/share_profile.py 3 8 bulb3 2>&1 | tail -1 | sed 's/^/bulb3 1\/8 share: /'
# the metric's literally named workload (Mandelbulb + volume at configs[2]'s size): rocprofv3 kernel stats of one frame, then its own bench line
export RAYN_HIP_WORKERS=1 RAYN_HIP_COLD_BYTES=0
bash tools/gpu_profile.sh bulb3 --workload bulb3 > /dev/null 2>&1
cp gpurun_out/prof_bulb3_kernel_stats.csv gpurun_out/${R}_bulb3_${TAG}_kernel_stats_1worker.csv
unset RAYN_HIP_WORKERS RAYN_HIP_COLD_BYTES
timeout 900 python bench.py --workload bulb3 2>&1 | tail -1 > gpurun_out/${R}_bench_bulb3_$TAG.json
python -c "
import json
j=json.load(open('gpurun_out/${R}_bench_bulb3_$TAG.json')); print('bulb3 VALUE', j['value'], j['ms_per_step']); print(j['kernel_ms']); print({k:j['roofline'][k] for k in ('kernel','achieved','frac','flop_per_dist_eval','whole_frame')}); print(j['cpu_baseline'])"
# the N>1 bench modes on the one GPU: self-launch (gloo, shared GPU), one process over two entries, the RCCL path at world 1 with the gather alone
timeout 600 python bench.py --gpus 2 --workload c2 --steps 2 --warmup 1 --backend gloo --share-gpu --check-film --cpu-seconds 0 --no-roofline 2>/dev/null | tail -1 | cut -c1-1800
timeout 600 python bench.py --gpus 2 --single-process --share-gpu --workload c2 --steps 2 --warmup 1 --cpu-seconds 0 --check-film 2>/dev/null | tail -1 | cut -c1-1800
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --workload c3 --backend nccl --force-dist --gather-only 2>/dev/null | tail -1 | tee gpurun_out/${R}_gather_only_world1_$TAG.json | cut -c1-1500
timeout 600 python tools/host_rate.py c3 2>&1 | tail -1
timeout 600 python tools/host_rate.py c2 2>&1 | tail -1
timeout 600 python bench.py --workload c4 --steps 1 --warmup 1 --no-roofline --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/${R}_bench_c4_$TAG.json; cat gpurun_out/${R}_bench_c4_$TAG.json | cut -c1-200
if [ -n "$WHOLE_C5" ]; then  # 136 G paths, ~3 GPU-minutes: only on request (r3's number stands, the march kernels did not change in r4)
timeout 900 python bench.py --workload c5 --steps 1 --warmup 0 --no-roofline --no-cold --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/${R}_bench_c5_$TAG.json; cat gpurun_out/${R}_bench_c5_$TAG.json | cut -c1-200
fi
# the reference's own workload (src/main.rs:47-82): bench line with cold_ms, whole-frame 16x16 CPU leg; what an 8-GPU launch waits for
timeout 600 python bench.py --workload shipped 2>&1 | tail -1 > gpurun_out/${R}_bench_shipped_$TAG.json; cut -c1-400 gpurun_out/${R}_bench_shipped_$TAG.json
timeout 600 python tools/share_balance.py 8 c2 2>&1 | tail -1
timeout 600 python tools/share_balance.py 8 c3 2>&1 | tail -1
for i in 1 2 3; do timeout 120 python tools/cold_breakdown.py shipped 0 2>&1 | tail -1; done
# the reference's own usage: one frame per process, back to back (tools/cold_frame.py)
sleep 6
for WL in c3 c2 c2 shipped shipped; do timeout 300 python tools/cold_frame.py $WL -1 0 2>&1 | tail -1; done
