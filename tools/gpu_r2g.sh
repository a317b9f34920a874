# round 2, call G: parity (register resolve v2, shadow scan), kernel stats c3 share with shadow scan on/off
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
export TMPDIR=/tmp
for SC in 1 0; do
  (cd /tmp && RAYN_HIP_SHADOW_SCAN=$SC timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/st_$SC -- python $GRAFT_REPO_ROOT/tools/share_profile.py 3 8 c3 > $GRAFT_REPO_ROOT/gpurun_out/st_$SC.log 2>&1)
  echo "== shadow_scan $SC"; tail -1 gpurun_out/st_$SC.log
  F=$(find gpurun_out/st_$SC -name "*kernel_stats.csv" | head -1)
  python - <<PY
import csv
for r in csv.DictReader(open('$F')):
    if 'rayn' in r['Name'] and float(r['Percentage']) > 0.05: print(r['Name'][:60].ljust(62), r['Calls'].rjust(6), ('%.2f' % (float(r['TotalDurationNs'])/1e6/3)).rjust(10), 'ms/frame', r['Percentage'])
PY
  rm -rf gpurun_out/st_$SC
done
RAYN_HIP_SHADOW_SCAN=1 timeout 300 python tools/share_profile.py 3 8 c2 2>&1 | tail -1
RAYN_HIP_SHADOW_SCAN=0 timeout 300 python tools/share_profile.py 3 8 c2 2>&1 | tail -1
