set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_share
(cd /tmp && RAYN_HIP_ENV_TUNING=1 RAYN_HIP_WORKERS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/tools/share_profile.py 3 8 c3 > $GRAFT_REPO_ROOT/gpurun_out/prof_share.log 2>&1)
tail -1 gpurun_out/prof_share.log | cut -c1-300
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_share_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
head -12 gpurun_out/prof_share_kernel_stats.csv | cut -c1-160
