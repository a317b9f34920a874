# round 2, call B: microbenchmarks (exec-mask cost, empty-block dispatch), k_shade_setup variants + ablations, forced 2 workers on a share
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/ubench/exec_mask 2>&1 | tee gpurun_out/r2b_exec_mask.txt
for ST in 0 1; do
  echo "== setup_stride $ST"
  RAYN_HIP_SETUP_STRIDE=$ST timeout 300 python tools/share_profile.py 3 8 c2 2>&1 | tail -1
  RAYN_HIP_SETUP_STRIDE=$ST timeout 300 python tools/share_profile.py 0 1 c2 2>&1 | tail -1
  RAYN_HIP_SETUP_STRIDE=$ST timeout 600 python tools/share_profile.py 3 8 c3 2>&1 | tail -1
done
echo "== 2 workers forced on the 1/8 share"
RAYN_HIP_WORKERS=2 RAYN_HIP_WORKER_MIN_PATHS=0 timeout 300 python tools/share_profile.py 3 8 c2 2>&1 | tail -1
RAYN_HIP_WORKERS=2 RAYN_HIP_WORKER_MIN_PATHS=0 timeout 300 python tools/share_profile.py 3 8 c3 2>&1 | tail -1
echo "== k_shade_setup ablations (timing only) on the c3 1/8 share: 1 normal, 2 surface NEE, 4 scatter, 8 sphere occlusion, 16 BSDF::f"
for AB in 1 2 4 8 16 31; do
  echo "ablate $AB"; RAYN_HIP_ABLATE=$AB timeout 300 python tools/share_profile.py 3 8 c3 2>&1 | tail -1
done
echo "== pinned f64 transcendentals replaced by hardware approximations (timing only)"
RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_nodetmath.so timeout 300 python tools/share_profile.py 3 8 c3 2>&1 | tail -1
RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_nodetmath.so timeout 300 python tools/share_profile.py 3 8 c2 2>&1 | tail -1
