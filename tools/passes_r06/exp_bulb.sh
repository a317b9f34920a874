# r6 experiment pass: the Mandelbulb march kernels (march_bulb.h) against the generic single-SDF kernels, on the metric's named workload (bulb3).
#   bash tools/passes_r06/exp_bulb.sh        (one gpurun call; writes gpurun_out/r06_exp_bulb.txt)
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_bulb.txt
mkdir -p gpurun_out; : > $OUT
[ -n "$SKIP_TESTS" ] || timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_config_digests.py -m gpu -x -q -k "bulb" 2>&1 | tail -3 >> $OUT
run() { # label, env...
  label=$1; shift
  line=$(env "$@" timeout 400 python bench.py --workload ${WL:-bulb3} --steps ${STEPS:-1} --warmup 1 --no-cold --cpu-seconds 0 2>&1 | tail -1)
  echo "$label $(echo "$line" | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read()); k=j['kernel_ms']; r=j['roofline']
    print(j['value'], 'ms', j['ms_per_step'], 'extend', k['ms_extend'], 'shadow', k['ms_shadow'], 'setup', k['ms_shade'], 'evals', r['all_march_kernels']['shadow']['dist_evals'], r['all_march_kernels']['extend']['dist_evals'], 'jobs', r['zero_throughput_elision']['shadow_jobs_marched'], 'oob', r['zero_throughput_elision']['samples_out_of_bounds'], 'occ', r.get('bulb_stage_occupancy'))
except Exception as e: print('ERR', e)
")" >> $OUT
}
if [ -n "$VARIANTS" ]; then eval "$VARIANTS"; else
run generic RAYN_HIP_BULB_PATH=0
run bulb_s1_e40 RAYN_HIP_BULB_STEPS=1
run bulb_s2_e40 RAYN_HIP_BULB_STEPS=2
fi
cat $OUT
