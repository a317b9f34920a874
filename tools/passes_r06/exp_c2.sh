# r6 experiment pass C2: rcp_sqrt_rn with Markstein's exception routed to the IEEE path - exhaustive check + the detmath / SDF probes, then bulb3 / c3 frames
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_c2.txt
mkdir -p gpurun_out; : > $OUT
python tools/passes_r06/diag_rcp_sqrt.py 2>&1 | grep -v amdgpu.ids | tail -5 >> $OUT
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
run bulb3 bulb3
run c3 c3
cat $OUT
