# usage: bash tools/gpu_pmc.sh <tag> "<counters>" [bench args]  -> gpurun_out/pmc_<tag>.csv (per-kernel sums)
set -x
TAG=$1; CTRS=$2; shift; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-roofline --no-cold --cpu-seconds 0 "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.log 2>&1)
tail -3 gpurun_out/pmc_$TAG.log
F=$(find $OUT -name "*counter_collection.csv" | head -1)
python - "$F" > gpurun_out/pmc_$TAG.csv <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].split("<")[0][:60]  # drop template arguments (they contain commas)
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
names = sorted({c for v in agg.values() for c in v})
print("kernel," + ",".join(names))
for k, v in sorted(agg.items()):
    print(k + "," + ",".join(f"{v.get(c,0):.0f}" for c in names))
PY
cat gpurun_out/pmc_$TAG.csv
rm -rf $OUT
