"""Load balance of the tile partition: the wall time of EVERY rank's share of a workload on one GPU (one context, shares rendered
one after the other), i.e. what an N-GPU launch of bench.py would wait for (max over ranks) before the gather.
usage: share_balance.py <N> [workload=c3]"""
import os, sys, time
import os as _os; _os.environ.setdefault("RAYN_HIP_ENV_TUNING", "1")  # the library reads RAYN_HIP_* tuning only under this opt-in (include/rayn_hip.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rayn_amd
from rayn_amd import setup as S
from bench import WORKLOADS
N = int(sys.argv[1]); wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
scene, W, H, samples, bounces, desc = WORKLOADS[wl]
cam, w = S.SCENES[scene]((W, H))
tabs = rayn_amd.build_tables(4 * samples, bounces, 2, 1, W, H)
ctx = rayn_amd.Context(0); ctx.upload_world(w.to_desc(cam))
d = [torch.from_numpy(t).cuda() for t in tabs]
film = rayn_amd.film.alloc_device_film(W, H, "cuda:0")
def run(first, step):
    p = rayn_amd.frame_params(W, H, samples, bounces, tile_first=first, tile_step=step)
    t = time.perf_counter(); ctx.render_device(p, d, film); torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3, ctx.stats()
for _ in range(2): run(0, N)          # first frame: small batches; second: arenas grown
full = min(run(0, 1)[0] for _ in range(2))
ms, segs = [], []
for r in range(N):
    t, st = min((run(r, N) for _ in range(2)), key=lambda x: x[0])
    ms.append(t); segs.append(st["segments"])
print(f"{wl}: whole frame {full:.1f} ms; {N} shares: " + " ".join(f"{x:.1f}" for x in ms) + f" ms (max {max(ms):.1f}, mean {sum(ms)/N:.1f}) -> {full/max(ms):.2f}x of {N} before the gather; "
      f"segments per share {min(segs)/1e6:.1f}-{max(segs)/1e6:.1f} M")
