# round 2, call I: parity + fuzz after the atomics-free job list; kernel stats of full c3 / c2 frames (1 worker)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 python tools/fuzz_parity.py 90 7000 2>&1 | tail -3
export TMPDIR=/tmp
for WL in c2 c3; do
  (cd /tmp && RAYN_HIP_WORKERS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/st_$WL -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 1 --warmup 0 --no-roofline --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/st_$WL.log 2>&1)
  F=$(find gpurun_out/st_$WL -name "*kernel_stats.csv" | head -1)
  python - <<PY
import csv
for r in csv.DictReader(open('$F')):
    if 'rayn' in r['Name'] and float(r['Percentage']) > 0.01: print(r['Name'][:60].ljust(62), r['Calls'].rjust(6), ('%.2f' % (float(r['TotalDurationNs'])/1e6)).rjust(10), 'ms', r['Percentage'])
PY
  rm -rf gpurun_out/st_$WL
  timeout 900 python bench.py --workload $WL --steps 2 --warmup 1 --cpu-seconds 0 --no-roofline 2>&1 | tail -1 | cut -c1-200
done
