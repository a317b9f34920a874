# r6: SQ counter passes (VALU issue, enabled lanes) of one workload's single-worker frame, joined per kernel
#   WL=bulb3 bash tools/passes_r06/pmc_sq.sh      -> gpurun_out/r06_${WL}_pmc_sq.csv
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1 RAYN_HIP_WORKERS=1 RAYN_HIP_COLD_BYTES=0
WL=${WL:-bulb3}
bash tools/gpu_pmc.sh sq1_$WL "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES" --workload $WL > /dev/null 2>&1
bash tools/gpu_pmc.sh sq2_$WL "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU" --workload $WL > /dev/null 2>&1
python tools/pmc_join.py gpurun_out/pmc_sq1_$WL.csv gpurun_out/pmc_sq2_$WL.csv > gpurun_out/r06_${WL}_pmc_sq.csv
cat gpurun_out/r06_${WL}_pmc_sq.csv
