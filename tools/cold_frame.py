"""Single-frame cost, the reference's own usage (src/main.rs:47-96 renders frame_range 1..2 and exits): a fresh process, a fresh
context, ONE frame through the host-buffer entry (rayn_hip_render_frame).  Prints the cold time (context creation -> film on
the host), the batches it took, and the following warm frames for comparison.  No torch: this is what a C / Rust host pays.
usage: cold_frame.py [workload=c2] [cold_bytes] [warm_frames=2]      (cold_bytes 0 = full-size arena up front, the r2 behaviour; -1 = default)
RAYN_HIP_BATCH_PATHS / RAYN_HIP_WORKERS in the environment select the arena footprint (this script opts in to the library's
environment tuning itself: RAYN_HIP_ENV_TUNING=1, include/rayn_hip.h - without it the library ignores those names)."""
import os, sys, time
os.environ.setdefault("RAYN_HIP_ENV_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayn_amd
from rayn_amd import setup as S
from bench import WORKLOADS
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
scene, W, H, samples, bounces, desc = WORKLOADS[wl]
cam, w = S.SCENES[scene]((W, H))
p = rayn_amd.frame_params(W, H, samples, bounces)
tabs = rayn_amd.build_tables(4 * samples, bounces, p.volume_marches, p.frame, W, H)
wd = w.to_desc(cam)
t0 = time.perf_counter()
ctx = rayn_amd.Context(0)
if len(sys.argv) > 2 and int(sys.argv[2]) >= 0:
    ctx.set_cold_bytes(int(sys.argv[2]))
ctx.upload_world(wd)
t1 = time.perf_counter()
out = ctx.render_host(p, tabs)
t2 = time.perf_counter()
st = ctx.stats()
warm = []
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    t = time.perf_counter(); ctx.render_host(p, tabs); warm.append((time.perf_counter() - t) * 1e3)
st2 = ctx.stats()
print(f"{wl} batch {os.environ.get('RAYN_HIP_BATCH_PATHS', 'default')} cold_bytes {sys.argv[2] if len(sys.argv) > 2 else 'default'}: cold {1e3*(t2-t0):.1f} ms (context {1e3*(t1-t0):.1f} + frame {1e3*(t2-t1):.1f}; device {st['ms_total']:.1f} ms, "
      f"{st['batches']} batches) | warm {' / '.join(f'{x:.1f}' for x in warm)} ms ({st2['batches']} batches, device {st2['ms_total']:.1f} ms)")
ctx.close()
