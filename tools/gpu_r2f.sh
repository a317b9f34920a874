# round 2, call F: parity with the register resolve + kernel stats of full c3 / c2 frames (1 worker)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
export TMPDIR=/tmp
for WL in c2 c3; do
  (cd /tmp && RAYN_HIP_WORKERS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/st_$WL -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 1 --warmup 0 --no-roofline --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/st_$WL.log 2>&1)
  tail -1 gpurun_out/st_$WL.log | cut -c1-200
  F=$(find gpurun_out/st_$WL -name "*kernel_stats.csv" | head -1)
  cp $F gpurun_out/r2f_${WL}_kernel_stats_1worker.csv
  python - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/r2f_${WL}_kernel_stats_1worker.csv')):
    if 'rayn' in r['Name']: print(r['Name'][:60].ljust(62), r['Calls'].rjust(6), ('%.2f' % (float(r['TotalDurationNs'])/1e6)).rjust(10), 'ms', r['Percentage'])
PY
  rm -rf gpurun_out/st_$WL
  timeout 900 python bench.py --workload $WL --steps 2 --warmup 1 --cpu-seconds 0 --no-roofline 2>&1 | tail -1 | cut -c1-200
done
