# round 2, call D: parity after the queue-kernel rework, bench lines (1 and 2 workers), per-launch trace share vs full
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for WL in c3 c2; do
  RAYN_HIP_WORKERS=1 timeout 900 python bench.py --workload $WL --steps 2 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/r2d_${WL}_w1.json
  RAYN_HIP_WORKERS=2 timeout 900 python bench.py --workload $WL --steps 2 --warmup 1 --cpu-seconds 0 --no-roofline 2>&1 | tail -1 > gpurun_out/r2d_${WL}_w2.json
  python - <<PY
import json
for w in (1, 2):
    j = json.load(open('gpurun_out/r2d_${WL}_w%d.json' % w))
    print('$WL workers', w, j['value'], j['ms_per_step'])
    if j.get('roofline_hbm'):
        print(j['kernel_ms'])
        for k, v in j['roofline_hbm']['kernels'].items(): print('   ', k[:50], v)
        print(j['roofline'])
PY
done
export TMPDIR=/tmp
for SH in "3 8" "0 1"; do
  TAG=$(echo $SH | tr ' ' '_')
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG -- python $GRAFT_REPO_ROOT/tools/share_profile.py $SH c2 > $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG.log 2>&1)
  F=$(find gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1)
  python tools/trace_launches.py $F $([ "$SH" = "3 8" ] && echo 9 || echo 36)
  rm -rf gpurun_out/trace_$TAG
done
