"""PCIe-inclusive rate: rayn_hip_render_frame with HOST buffers (tables H2D + film D2H inside the call) vs the
device-buffer entry.  usage: host_rate.py [workload=c2]"""
import sys, time
import os as _os; _os.environ.setdefault("RAYN_HIP_ENV_TUNING", "1")  # the library reads RAYN_HIP_* tuning only under this opt-in (include/rayn_hip.h)
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, rayn_amd
from rayn_amd import setup as S
from bench import WORKLOADS
scene, W, H, samples, bounces, desc = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
cam, w = S.SCENES[scene]((W, H))
p = rayn_amd.frame_params(W, H, samples, bounces)
tabs = rayn_amd.build_tables(4 * samples, bounces, p.volume_marches, p.frame, W, H)
ctx = rayn_amd.Context(0); ctx.upload_world(w.to_desc(cam))
d = [torch.from_numpy(t).cuda() for t in tabs]
film = rayn_amd.film.alloc_device_film(W, H, "cuda:0")
for _ in range(2):  # frame 1 runs in small batches (first-frame policy), frame 2 grows the arenas: neither belongs in the rate
    ctx.render_device(p, d, film); torch.cuda.synchronize()
t = time.perf_counter(); ctx.render_device(p, d, film); torch.cuda.synchronize(); t_dev = time.perf_counter() - t
out = ctx.render_host(p, tabs)
t = time.perf_counter(); out = ctx.render_host(p, tabs); t_host = time.perf_counter() - t
paths = W * H * 4 * samples
print(f"{desc}: device buffers {t_dev*1e3:.1f} ms ({paths/t_dev/1e6:.1f} Mpath/s), host buffers {t_host*1e3:.1f} ms ({paths/t_host/1e6:.1f} Mpath/s), "
      f"tables {sum(x.nbytes for x in tabs)/1e6:.1f} MB in, film {sum(v.nbytes for v in out.values())/1e6:.1f} MB out")
