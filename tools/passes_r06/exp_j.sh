# r6 experiment pass J: the march point of k_extend1 / k_shadow1 as STATE (set at promotion and at the end of a trip that goes on) against the per-trip select
# `first ? origin : origin + dir t` (variant library ptselect): an eighth of c3 / c2 / bulb3 each, then whole c3 frames
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_j.txt
mkdir -p gpurun_out; : > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_config_digests.py -m gpu -x -q 2>&1 | tail -1 >> $OUT
V="RAYN_HIP_ALLOW_VARIANT=1 RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_ptselect.so"
run() { label=$1; wl=$2; shift; shift; echo "$label $(env "$@" timeout 300 python tools/share_profile.py 3 8 $wl 2>&1 | grep '^wall' | cut -c1-250)" >> $OUT; }
run c3_state c3
run c3_select c3 $V
run c2_state c2
run c2_select c2 $V
run bulb3_state bulb3
run bulb3_select bulb3 $V
run c3_state_b c3
run c3_select_b c3 $V
runb() { label=$1; wl=$2; shift; shift
  line=$(env "$@" timeout 400 python bench.py --workload $wl --steps 1 --warmup 1 --no-cold --no-named --cpu-seconds 0 2>&1 | tail -1)
  echo "$label $(echo "$line" | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read()); k=j['kernel_ms']
    print(j['value'], 'ms', j['ms_per_step'], 'extend', k['ms_extend'], 'shadow', k['ms_shadow'], 'setup', k['ms_shade'], j['config']['build_variant'])
except Exception as e: print('ERR', e)
")" >> $OUT; }
runb c3_frame_state c3
runb c3_frame_select c3 $V
cat $OUT
