# round 4, second GPU pass: parity after the queue-kernel changes; do two co-resident workers hide the march-kernel tails of SMALL
# launches (1/8 share of c2, the shipped frame)?; where context creation goes; how the CPU oracle scales on the box
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
export RAYN_HIP_ENV_TUNING=1
run() { echo "== $*"; env "$@" timeout 200 python tools/share_profile.py $SH 2>&1 | tail -1 | cut -c1-330; }
for SH in "3 8 c2" "0 1 shipped"; do
  run X=1
  run RAYN_HIP_WORKERS=2 RAYN_HIP_WORKER_MIN_PATHS=0
  run RAYN_HIP_WORKERS=2 RAYN_HIP_WORKER_MIN_PATHS=0 RAYN_HIP_PERSISTENT_BLOCKS=1024
  run RAYN_HIP_WORKERS=2 RAYN_HIP_WORKER_MIN_PATHS=0 RAYN_HIP_PERSISTENT_BLOCKS=1536
  run RAYN_HIP_WORKERS=3 RAYN_HIP_WORKER_MIN_PATHS=0
  run RAYN_HIP_WORKERS=4 RAYN_HIP_WORKER_MIN_PATHS=0 RAYN_HIP_PERSISTENT_BLOCKS=1024
  run RAYN_HIP_PERSISTENT_BLOCKS=1024
done
SH="3 8 c3"; run X=1
unset RAYN_HIP_ENV_TUNING
for i in 1 2; do timeout 120 python tools/cold_breakdown.py shipped 2 2>&1 | tail -1; done
timeout 300 python tools/cpu_scaling.py 2>&1 | tail -9
