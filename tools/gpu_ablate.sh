# timing-only ablations of k_shade_setup (results are wrong when RAYN_HIP_ABLATE != 0): per-class ms of a 1-worker c2 frame
cd $GRAFT_REPO_ROOT
for a in 0 8 16 24 2 1 4 31; do
  echo "ABLATE $a: $(RAYN_HIP_ABLATE=$a python tools/share_profile.py 0 1 2>&1 | tail -1 | cut -c1-400)"
done
