"""Does co-running two independent half-frames on one GPU beat one full frame?  (decides whether a
two-stream batch pipeline is worth building)   usage: overlap_probe.py <tile_first> <tile_step> <steps> [blocks]"""
import sys, time
sys.path.insert(0, '.')
import torch, rayn_amd
from rayn_amd import setup as S
first, step, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
W, H, samples, bounces = 1920, 1080, 64, 8
cam, w = S.setup_s1((W, H))
p = rayn_amd.frame_params(W, H, samples, bounces, tile_first=first, tile_step=step)
tabs = rayn_amd.build_tables(4 * samples, bounces, 2, 1, W, H)
ctx = rayn_amd.Context(0); ctx.upload_world(w.to_desc(cam))
d = [torch.from_numpy(t).cuda() for t in tabs]
film = rayn_amd.film.alloc_device_film(W, H, "cuda:0")
ctx.render_device(p, d, film); torch.cuda.synchronize()
open(f"/tmp/ready_{first}", "w").write("x")
import os
while not all(os.path.exists(f"/tmp/ready_{i}") for i in range(step)): time.sleep(0.01)
t = time.perf_counter()
for _ in range(steps): ctx.render_device(p, d, film)
torch.cuda.synchronize()
print(f"first={first} step={step}: {(time.perf_counter()-t)/steps*1e3:.1f} ms per (partial) frame", flush=True)
