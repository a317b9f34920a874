# r6 experiment pass G: k_extend1 capped at 96 SGPRs (8 waves per SIMD instead of 7, 7 scalar spills) - variant library sgpr96 against the product, an eighth of c3 and of bulb3
cd $GRAFT_REPO_ROOT
export RAYN_HIP_ENV_TUNING=1
OUT=gpurun_out/r06_exp_g.txt
mkdir -p gpurun_out; : > $OUT
V="RAYN_HIP_ALLOW_VARIANT=1 RAYN_HIP_LIB=$GRAFT_REPO_ROOT/rayn_amd/csrc/librayn_hip_sgpr96.so"
run() { label=$1; wl=$2; shift; shift; echo "$label $(env "$@" timeout 300 python tools/share_profile.py 3 8 $wl 2>&1 | grep '^wall' | cut -c1-250)" >> $OUT; }
run c3_product c3
run c3_sgpr96 c3 $V
run bulb3_product bulb3
run bulb3_sgpr96 bulb3 $V
run c2_product c2
run c2_sgpr96 c2 $V
run c3_product_b c3
run c3_sgpr96_b c3 $V
env $V timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "film_parity or closest or closed_set" 2>&1 | tail -1 >> $OUT
cat $OUT
