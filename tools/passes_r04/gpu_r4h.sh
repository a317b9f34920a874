set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "== $*"; env RAYN_HIP_ENV_TUNING=1 "$@" timeout 300 python tools/share_profile.py $SH 2>&1 | tail -1 | cut -c1-300; }
SH="3 8 c3"
run X=1
run RAYN_HIP_PREFETCH_SHADOW=16
run RAYN_HIP_PREFETCH_SHADOW=48
run RAYN_HIP_PREFETCH_SHADOW=56
run RAYN_HIP_PREFETCH_EXTEND=16
run RAYN_HIP_PREFETCH_EXTEND=48
export RAYN_HIP_ALLOW_VARIANT=1
for V in fin3072 fin4096; do RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_$V.so run V=$V; done
SH="0 1 c2"
run X=1
for V in fin3072 fin4096; do RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_$V.so run V=$V; done
