# r6 experiment pass F: the Mandelbulb step's two roots behind one wave-uniform window test (product) against the per-lane tests (variant sqrtlane); steps = 2 default
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_f.txt
mkdir -p gpurun_out; : > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_config_digests.py -m gpu -x -q -k "bulb or detmath or sqrt or probe or closest or occluded" 2>&1 | tail -3 >> $OUT
V="RAYN_HIP_ALLOW_VARIANT=1 RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_sqrtlane.so"
run() { label=$1; shift; echo "$label $(env "$@" timeout 300 python tools/share_profile.py 3 8 bulb3 2>&1 | grep '^wall' | cut -c1-250)" >> $OUT; }
run uniform
run perlane $V
run uniform_b
run perlane_b $V
runb() { # label, workload, env...
  label=$1; wl=$2; shift; shift
  line=$(env "$@" timeout 400 python bench.py --workload $wl --steps 1 --warmup 1 --no-cold --no-named --cpu-seconds 0 2>&1 | tail -1)
  echo "$label $(echo "$line" | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read()); k=j['kernel_ms']; r=j['roofline']
    print(j['value'], 'ms', j['ms_per_step'], 'extend', k['ms_extend'], 'shadow', k['ms_shadow'], 'setup', k['ms_shade'], 'finish', k['ms_finish'], 'variant', j['config']['build_variant'], 'occ', r.get('bulb_stage_occupancy'))
except Exception as e: print('ERR', e)
")" >> $OUT
}
runb bulb3 bulb3
runb bulb3_perlane bulb3 $V
cat $OUT
