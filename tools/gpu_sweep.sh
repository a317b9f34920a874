# tuning sweep on one rank's share of a workload: usage bash tools/gpu_sweep.sh <workload> <first> <step>
cd $GRAFT_REPO_ROOT
WL=${1:-c3}; F=${2:-1}; S=${3:-8}
run() { echo "$* : $(env "$@" python tools/share_profile.py $F $S $WL 2>&1 | tail -1 | sed -E 's/.*ms_extend.: ([0-9.]+).*ms_shade.: ([0-9.]+).*ms_shadow.: ([0-9.]+).*sum ([0-9.]+).*/extend \1 setup \2 shadow \3 sum \4/')"; }
run X=0
run RAYN_HIP_PREFETCH_SHADOW=16
run RAYN_HIP_PREFETCH_SHADOW=48
run RAYN_HIP_PREFETCH_EXTEND=16
run RAYN_HIP_PREFETCH_EXTEND=48
run RAYN_HIP_PERSISTENT_BLOCKS=1536
run RAYN_HIP_PERSISTENT_BLOCKS=1024
run RAYN_HIP_PERSISTENT_BLOCKS=4096
