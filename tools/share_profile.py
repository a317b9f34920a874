"""Per-kernel-class times of ONE rank's share of a bench workload.
usage: share_profile.py <tile_first> <tile_step> [workload=c2]"""
import os, sys, time
import os as _os; _os.environ.setdefault("RAYN_HIP_ENV_TUNING", "1")  # the library reads RAYN_HIP_* tuning only under this opt-in (include/rayn_hip.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rayn_amd
from rayn_amd import setup as S
from bench import WORKLOADS
first, step = int(sys.argv[1]), int(sys.argv[2])
scene, W, H, samples, bounces, desc = WORKLOADS[sys.argv[3] if len(sys.argv) > 3 else "c2"]
cam, w = S.SCENES[scene]((W, H))
p = rayn_amd.frame_params(W, H, samples, bounces, tile_first=first, tile_step=step)
tabs = rayn_amd.build_tables(4 * samples, bounces, p.volume_marches, p.frame, W, H)
ctx = rayn_amd.Context(0); ctx.upload_world(w.to_desc(cam))
d = [torch.from_numpy(t).cuda() for t in tabs]
film = rayn_amd.film.alloc_device_film(W, H, "cuda:0")
for _ in range(2):  # the first frame of a context runs in small batches (cold-start policy), the second grows the arenas
    ctx.render_device(p, d, film); torch.cuda.synchronize()
walls = []
for _ in range(2):
    t = time.perf_counter(); ctx.render_device(p, d, film); torch.cuda.synchronize(); walls.append((time.perf_counter() - t) * 1e3)
wall = min(walls)
ctx.set_workers(1); ctx.set_profiling(True, False)
ctx.render_device(p, d, film); torch.cuda.synchronize()
st = ctx.stats()
ks = {k: round(v, 2) for k, v in st.items() if k.startswith("ms_")}
print("wall", round(wall, 1), ks, "sum", round(sum(v for k, v in ks.items() if k != "ms_total"), 1), "launches", st["launches_extend"], "batches", st["batches"],
      "segments", st["segments"], "paths", st["paths"])
if len(sys.argv) > 4:  # any 4th argument: one more pass with the SDF-evaluation counters on
    ctx.set_profiling(True, True); ctx.render_device(p, d, film); torch.cuda.synchronize()
    print("evals", ctx.eval_counts())
