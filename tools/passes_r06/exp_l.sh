# r6 experiment pass L: k_shade_finish with the records of four samples loaded first (product) against the rolled per-sample loops (variant library finrolled)
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_l.txt
mkdir -p gpurun_out; : > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_config_digests.py -m gpu -x -q 2>&1 | tail -1 >> $OUT
V="RAYN_HIP_ALLOW_VARIANT=1 RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_finrolled.so"
run() { label=$1; wl=$2; shift; shift; echo "$label $(env "$@" timeout 300 python tools/share_profile.py 3 8 $wl 2>&1 | grep '^wall' | cut -c1-250)" >> $OUT; }
run c3_batched c3
run c3_rolled c3 $V
run c2_batched c2
run c2_rolled c2 $V
run bulb3_batched bulb3
run bulb3_rolled bulb3 $V
run c3_batched_b c3
run c3_rolled_b c3 $V
cat $OUT
