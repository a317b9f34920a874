# rocprofv3 kernel stats of one rank's share: bash tools/gpu_share_stats.sh <first> <step> <workload>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp RAYN_HIP_WORKERS=1 PYTHONPATH=$GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_share
rm -rf $OUT
(cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python tools/share_profile.py $1 $2 $3 > gpurun_out/prof_share.log 2>&1)
tail -2 gpurun_out/prof_share.log | cut -c1-200
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")
    print("%-36s calls %5s total %9.2f ms  %s%%" % (n[:36], r["Calls"], float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
find $OUT -name "*kernel_trace.csv" -delete
