# r6 experiment pass H: k_extend_bulb (tools/variants/r6_k_extend_bulb.h) measured again after rcp_sqrt_rn / two steps per trip / the LDS table - an eighth of bulb3 per setting
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_h.txt
mkdir -p gpurun_out; : > $OUT
run() { label=$1; shift; echo "$label $(env "$@" timeout 300 python tools/share_profile.py 3 8 bulb3 2>&1 | grep '^wall' | cut -c1-250)" >> $OUT; }
run generic
run ext_k2_s2 RAYN_HIP_BULB_EXTEND=2
run ext_k2_s1 RAYN_HIP_BULB_EXTEND=2 RAYN_HIP_BULB_STEPS=1
run ext_k3_s2 RAYN_HIP_BULB_EXTEND=3
run generic_b
env RAYN_HIP_BULB_EXTEND=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_config_digests.py -m gpu -x -q -k "bulb" 2>&1 | tail -1 >> $OUT
cat $OUT
