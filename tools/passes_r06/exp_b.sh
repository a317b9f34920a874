# r6 experiment pass B: (1) parity of the tree (NEE records of exact-zero samples no longer stored: VIS_ZERO; the Mandelbulb logarithm's table in LDS),
# (2) c3 and bulb3 frames, (3) bulb3 with the table read from global memory (variant library) for comparison.
#   bash tools/passes_r06/exp_b.sh        (one gpurun call; writes gpurun_out/r06_exp_b.txt)
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_b.txt
mkdir -p gpurun_out; : > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_config_digests.py -m gpu -x -q 2>&1 | tail -3 >> $OUT
run() { # label, workload, env...
  label=$1; wl=$2; shift; shift
  line=$(env "$@" timeout 400 python bench.py --workload $wl --steps ${STEPS:-1} --warmup 1 --no-cold --no-named --cpu-seconds 0 2>&1 | tail -1)
  echo "$label $(echo "$line" | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read()); k=j['kernel_ms']; r=j['roofline']
    print(j['value'], 'ms', j['ms_per_step'], 'extend', k['ms_extend'], 'shadow', k['ms_shadow'], 'setup', k['ms_shade'], 'finish', k['ms_finish'], 'bin', k['ms_bin'], 'resolve', k['ms_resolve'], 'variant', j['config']['build_variant'], 'occ', r.get('bulb_stage_occupancy'))
except Exception as e: print('ERR', e)
")" >> $OUT
}
run c3 c3
run bulb3 bulb3
run bulb3_logtab_global bulb3 RAYN_HIP_ALLOW_VARIANT=1 RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_logglobal.so
run bulb3_b bulb3
run c3_b c3
cat $OUT
