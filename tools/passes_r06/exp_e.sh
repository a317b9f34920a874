# r6 experiment pass E: k_shadow_bulb's run-time parameters re-tuned after the orbit step got cheaper (rcp_sqrt_rn) - one rank's eighth of bulb3 per setting
#   bash tools/passes_r06/exp_e.sh      (one gpurun call; writes gpurun_out/r06_exp_e.txt)
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_e.txt
mkdir -p gpurun_out; : > $OUT
run() { label=$1; shift; echo "$label $(env "$@" timeout 300 python tools/share_profile.py 3 8 bulb3 2>&1 | grep '^wall' | cut -c1-330)" >> $OUT; }
run default
run rays2 RAYN_HIP_BULB_RAYS=2
run rays4 RAYN_HIP_BULB_RAYS=4
run steps2 RAYN_HIP_BULB_STEPS=2
run orbit16 RAYN_HIP_BULB_ORBIT_MIN=16
run orbit32 RAYN_HIP_BULB_ORBIT_MIN=32
run orbit40 RAYN_HIP_BULB_ORBIT_MIN=40
run prefetch16 RAYN_HIP_BULB_PREFETCH=16
run prefetch64 RAYN_HIP_BULB_PREFETCH=64
run prefetch96 RAYN_HIP_BULB_PREFETCH=96
run rays2_orbit16 RAYN_HIP_BULB_RAYS=2 RAYN_HIP_BULB_ORBIT_MIN=16
run default_b
cat $OUT
