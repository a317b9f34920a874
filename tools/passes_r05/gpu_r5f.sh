# r5 pass F: PMC passes (HBM traffic, VALU issue) of the metric's literally named workload, then its bench line quoting them
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export RAYN_HIP_ENV_TUNING=1 RAYN_HIP_WORKERS=1 RAYN_HIP_COLD_BYTES=0
WL=bulb3; R=r05; TAG=v2
bash tools/gpu_profile.sh $WL --workload $WL > /dev/null 2>&1
bash tools/gpu_pmc.sh fetch_$WL "FETCH_SIZE" --workload $WL > /dev/null 2>&1
bash tools/gpu_pmc.sh write_$WL "WRITE_SIZE" --workload $WL > /dev/null 2>&1
bash tools/gpu_pmc.sh sq1_$WL "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES" --workload $WL > /dev/null 2>&1
bash tools/gpu_pmc.sh sq2_$WL "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU" --workload $WL > /dev/null 2>&1
python tools/pmc_join.py gpurun_out/pmc_sq1_$WL.csv gpurun_out/pmc_sq2_$WL.csv > gpurun_out/${R}_${WL}_${TAG}_pmc_sq.csv
unset RAYN_HIP_WORKERS RAYN_HIP_COLD_BYTES
python tools/pmc_to_json.py $WL profiles/${R}_pmc_hbm_$WL.json gpurun_out/prof_${WL}_kernel_stats.csv gpurun_out/${R}_${WL}_${TAG}_pmc_sq.csv
cp profiles/${R}_pmc_hbm_$WL.json gpurun_out/
cp gpurun_out/prof_${WL}_kernel_stats.csv gpurun_out/${R}_${WL}_${TAG}_kernel_stats_1worker.csv
cp gpurun_out/pmc_fetch_$WL.csv gpurun_out/${R}_${WL}_${TAG}_pmc_fetch_size.csv
cp gpurun_out/pmc_write_$WL.csv gpurun_out/${R}_${WL}_${TAG}_pmc_write_size.csv
timeout 900 python bench.py --workload $WL 2>&1 | tail -1 > gpurun_out/${R}_bench_${WL}_$TAG.json
python -c "
import json
j=json.load(open('gpurun_out/${R}_bench_${WL}_$TAG.json')); print('VALUE', j['value'], j['ms_per_step']); rf=j['roofline']; print({k:rf.get(k) for k in ('kernel','frac','flop_per_dist_eval','traffic','lanes_enabled','valu_issue','traffic_note')}); print(j['cpu_baseline'])"
