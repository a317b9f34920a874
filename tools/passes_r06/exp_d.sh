# r6 experiment pass D: dmf_round_safe on the bits of the binary64 result - the detmath probes + parity, then c3 / bulb3 frames against the variant library
# that keeps the r5 form ((float)(d - eps d) == (float)(d + eps d)).      bash tools/passes_r06/exp_d.sh      (one gpurun call; writes gpurun_out/r06_exp_d.txt)
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_d.txt
mkdir -p gpurun_out; : > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_config_digests.py -m gpu -x -q 2>&1 | tail -3 >> $OUT
run() { # label, workload, env...
  label=$1; wl=$2; shift; shift
  line=$(env "$@" timeout 400 python bench.py --workload $wl --steps ${STEPS:-1} --warmup 1 --no-cold --no-named --cpu-seconds 0 2>&1 | tail -1)
  echo "$label $(echo "$line" | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read()); k=j['kernel_ms']; r=j['roofline']
    print(j['value'], 'ms', j['ms_per_step'], 'extend', k['ms_extend'], 'shadow', k['ms_shadow'], 'setup', k['ms_shade'], 'finish', k['ms_finish'], 'raygen', k['ms_raygen'], 'variant', j['config']['build_variant'])
except Exception as e: print('ERR', e)
")" >> $OUT
}
V="RAYN_HIP_ALLOW_VARIANT=1 RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_roundr5.so"
run c3 c3
run c3_r5 c3 $V
run bulb3 bulb3
run bulb3_r5 bulb3 $V
run c3_b c3
run c3_r5_b c3 $V
cat $OUT
