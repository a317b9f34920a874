cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "Name:\s+SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' | head -c 6000
echo
bash tools/gpu_pmc.sh sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --workload mid
bash tools/gpu_pmc.sh sq2 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_CYCLES" --workload mid
