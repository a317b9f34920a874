# r6 experiment pass E2: orbit steps per trip of k_shadow_bulb (1..4) against the other run-time parameters, one rank's eighth of bulb3 per setting
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_e2.txt
mkdir -p gpurun_out; : > $OUT
run() { label=$1; shift; echo "$label $(env "$@" timeout 300 python tools/share_profile.py 3 8 bulb3 2>&1 | grep '^wall' | cut -c1-250)" >> $OUT; }
run steps1
run steps2 RAYN_HIP_BULB_STEPS=2
run steps3 RAYN_HIP_BULB_STEPS=3
run steps4 RAYN_HIP_BULB_STEPS=4
run steps2_orbit16 RAYN_HIP_BULB_STEPS=2 RAYN_HIP_BULB_ORBIT_MIN=16
run steps2_orbit32 RAYN_HIP_BULB_STEPS=2 RAYN_HIP_BULB_ORBIT_MIN=32
run steps2_orbit8 RAYN_HIP_BULB_STEPS=2 RAYN_HIP_BULB_ORBIT_MIN=8
run steps2_rays2 RAYN_HIP_BULB_STEPS=2 RAYN_HIP_BULB_RAYS=2
run steps2_rays4 RAYN_HIP_BULB_STEPS=2 RAYN_HIP_BULB_RAYS=4
run steps2_prefetch16 RAYN_HIP_BULB_STEPS=2 RAYN_HIP_BULB_PREFETCH=16
run steps3_orbit16 RAYN_HIP_BULB_STEPS=3 RAYN_HIP_BULB_ORBIT_MIN=16
run steps2_b RAYN_HIP_BULB_STEPS=2
cat $OUT
