# round 2, call A: parity after the sync-free depth loop + worker-count decision + 1/8-share emulation
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for WL in c2 c3; do
  for NW in 1 2; do
    RAYN_HIP_WORKERS=$NW timeout 600 python bench.py --workload $WL --steps 2 --warmup 1 --cpu-seconds 0 --no-roofline 2>&1 | tail -1 > gpurun_out/r2a_${WL}_w$NW.json
    python -c "
import json; j=json.load(open('gpurun_out/r2a_${WL}_w$NW.json')); print('$WL workers $NW', j['value'], j['ms_per_step'])"
  done
done
timeout 300 python tools/share_profile.py 3 8 c2 2>&1 | tail -1
timeout 300 python tools/share_profile.py 0 1 c2 2>&1 | tail -1
timeout 600 python tools/share_profile.py 3 8 c3 2>&1 | tail -1
timeout 600 python tools/share_profile.py 0 1 c3 2>&1 | tail -1
