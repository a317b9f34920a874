"""Per-kernel-class times of ONE rank's share of config 2 (usage: share_profile.py <tile_first> <tile_step>)."""
import sys, time
sys.path.insert(0, '.')
import torch, rayn_amd
from rayn_amd import setup as S
first, step = int(sys.argv[1]), int(sys.argv[2])
W, H, samples, bounces = 1920, 1080, 64, 8
cam, w = S.setup_s1((W, H))
p = rayn_amd.frame_params(W, H, samples, bounces, tile_first=first, tile_step=step)
tabs = rayn_amd.build_tables(4 * samples, bounces, 2, 1, W, H)
ctx = rayn_amd.Context(0); ctx.upload_world(w.to_desc(cam))
d = [torch.from_numpy(t).cuda() for t in tabs]
film = rayn_amd.film.alloc_device_film(W, H, "cuda:0")
ctx.render_device(p, d, film); torch.cuda.synchronize()
t = time.perf_counter(); ctx.render_device(p, d, film); torch.cuda.synchronize(); wall = (time.perf_counter() - t) * 1e3
ctx.set_workers(1); ctx.set_profiling(True, False)
ctx.render_device(p, d, film); torch.cuda.synchronize()
st = ctx.stats()
ks = {k: round(v, 2) for k, v in st.items() if k.startswith("ms_")}
print("wall", round(wall, 1), ks, "sum", round(sum(v for k, v in ks.items() if k != "ms_total"), 1), "launches", st["launches_extend"], "batches", st["batches"])
