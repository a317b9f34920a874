# round 2, call E: NEE-record diet parity + per-kernel stats of a c3 1/8 share (default build vs 64 ids per thread in k_shadow_list)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
export TMPDIR=/tmp
for V in default scan64; do
  LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip.so; [ $V != default ] && LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_$V.so
  (cd /tmp && RAYN_HIP_LIB=$LIB timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/st_$V -- python $GRAFT_REPO_ROOT/tools/share_profile.py 3 8 c3 > $GRAFT_REPO_ROOT/gpurun_out/st_$V.log 2>&1)
  tail -1 gpurun_out/st_$V.log
  F=$(find gpurun_out/st_$V -name "*kernel_stats.csv" | head -1)
  cp $F gpurun_out/r2e_c3share_${V}_kernel_stats.csv
  python - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/r2e_c3share_${V}_kernel_stats.csv')):
    print(r['Name'][:60].ljust(62), r['Calls'].rjust(6), ('%.2f' % (float(r['TotalDurationNs'])/1e6/3)).rjust(10), 'ms/frame', r['Percentage'])
PY
  rm -rf gpurun_out/st_$V
done
