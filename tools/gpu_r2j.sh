# round 2, call J: fast elementary functions in the kernels: parity, then k_shade_setup time of the three builds
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 python tools/fuzz_parity.py 120 9000 2>&1 | tail -3
for V in default noinl refmath; do
  LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip.so; [ $V != default ] && LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_$V.so
  echo "== $V"
  RAYN_HIP_LIB=$LIB timeout 300 python tools/share_profile.py 3 8 c3 2>&1 | tail -1
  RAYN_HIP_LIB=$LIB timeout 300 python tools/share_profile.py 3 8 c2 2>&1 | tail -1
done
RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_noinl.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "detmath or film_parity" 2>&1 | tail -2
