# usage: bash tools/gpu_profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/ kernel stats
set -x
TAG=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-roofline --no-cold --cpu-seconds 0 "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1)
tail -2 gpurun_out/prof_$TAG.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_${TAG}_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
cat gpurun_out/prof_${TAG}_kernel_stats.csv | head -20
