# r5 pass D: machine-scheduler strategies of the AMDGPU backend as VARIANT libraries (never the product) on one rank's eighth of config 3 and on config 2
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export RAYN_HIP_ALLOW_VARIANT=1
for V in "" maxilp iterilp minreg; do
  if [ -n "$V" ]; then export RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_$V.so; else unset RAYN_HIP_LIB; fi
  echo "== variant '$V'"
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "closed_set or film_parity or packet_order" 2>&1 | tail -1
  timeout 300 python tools/share_profile.py 3 8 c3 2>&1 | tail -1 | cut -c1-400
  timeout 300 python tools/share_profile.py 0 1 c2 2>&1 | tail -1 | cut -c1-400
done 2>&1 | tee gpurun_out/r05_exp_sched_strategies.txt
