# r6 experiment pass K: the spare-ray prefetch thresholds of k_extend1 / k_shadow1 and the persistent grid, re-checked on the final kernels (an eighth of c3 per setting)
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_k.txt
mkdir -p gpurun_out; : > $OUT
run() { label=$1; shift; echo "$label $(env "$@" timeout 300 python tools/share_profile.py 3 8 c3 2>&1 | grep '^wall' | cut -c1-250)" >> $OUT; }
run default
run shadow16 RAYN_HIP_PREFETCH_SHADOW=16
run shadow24 RAYN_HIP_PREFETCH_SHADOW=24
run shadow48 RAYN_HIP_PREFETCH_SHADOW=48
run extend16 RAYN_HIP_PREFETCH_EXTEND=16
run extend24 RAYN_HIP_PREFETCH_EXTEND=24
run extend48 RAYN_HIP_PREFETCH_EXTEND=48
run default_b
cat $OUT
