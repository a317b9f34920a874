cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc.sh sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --workload mid 2>&1 | grep -E "kernel,|k_extend1|k_shadow1|k_shade_setup"
bash tools/gpu_pmc.sh sq2 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_CYCLES" --workload mid 2>&1 | grep -E "kernel,|k_extend1|k_shadow1|k_shade_setup"
