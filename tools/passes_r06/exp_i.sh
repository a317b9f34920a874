# r6 experiment pass I: k_extend1 / k_shadow1 instantiated for the MandelBox in its shipped shape (12 iterations + verified short division at compile time) against the
# per-kind instantiation (RAYN_HIP_BOX12S=0): an eighth of c3 and of c2, then whole c3 frames
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_i.txt
mkdir -p gpurun_out; : > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1 >> $OUT
run() { label=$1; wl=$2; shift; shift; echo "$label $(env "$@" timeout 300 python tools/share_profile.py 3 8 $wl 2>&1 | grep '^wall' | cut -c1-250)" >> $OUT; }
run c3_box12s c3
run c3_perkind c3 RAYN_HIP_BOX12S=0
run c2_box12s c2
run c2_perkind c2 RAYN_HIP_BOX12S=0
run c3_box12s_b c3
run c3_perkind_b c3 RAYN_HIP_BOX12S=0
runb() { label=$1; wl=$2; shift; shift
  line=$(env "$@" timeout 400 python bench.py --workload $wl --steps 1 --warmup 1 --no-cold --no-named --cpu-seconds 0 2>&1 | tail -1)
  echo "$label $(echo "$line" | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read()); k=j['kernel_ms']
    print(j['value'], 'ms', j['ms_per_step'], 'extend', k['ms_extend'], 'shadow', k['ms_shadow'], 'setup', k['ms_shade'])
except Exception as e: print('ERR', e)
")" >> $OUT; }
runb c3_frame_box12s c3
runb c3_frame_perkind c3 RAYN_HIP_BOX12S=0
cat $OUT
