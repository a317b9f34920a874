# round 2, call C: exec-mask follow-up, full GPU test suite (multi-device ctx), dense-fold variants, per-launch trace share vs full
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/ubench/exec_mask2 2>&1 | tee gpurun_out/r2c_exec_mask2.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for V in dense9 dense16 dense65; do
  echo "== variant $V"
  RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_$V.so timeout 300 python tools/share_profile.py 3 8 c3 2>&1 | tail -1
  RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_$V.so timeout 300 python tools/share_profile.py 3 8 c2 2>&1 | tail -1
done
RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_dense16.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "film_parity or randomised" 2>&1 | tail -2
export TMPDIR=/tmp
for SH in "3 8" "0 1"; do
  TAG=$(echo $SH | tr ' ' '_')
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG -- python $GRAFT_REPO_ROOT/tools/share_profile.py $SH c2 > $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG.log 2>&1)
  F=$(find gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1)
  python tools/trace_launches.py $F $([ "$SH" = "3 8" ] && echo 9 || echo 36)
  rm -rf gpurun_out/trace_$TAG
done
