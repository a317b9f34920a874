# r5 pass C: the round's evidence pass (tools/passes_r05/gpu_round_r05.sh: GPU suite, fuzz incl. packed shares, rocprofv3 stats + PMC passes of c3 / c2,
# bench lines of c3 / c2 / bulb3 / c4 / c5 / shipped, shares, the N > 1 bench modes on the one GPU, cold frames)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FUZZ_N=600 WHOLE_C5=1
bash tools/passes_r05/gpu_round_r05.sh v1 > gpurun_out/r05_round_v1_log.txt 2>&1
tail -120 gpurun_out/r05_round_v1_log.txt | cut -c1-600
