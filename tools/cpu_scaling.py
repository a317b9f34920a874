"""How the CPU oracle (bench.py's cpu_baseline leg) scales with threads on this host: the same spread tile sample of the shipped
workload at 8 .. os.cpu_count() threads.  Also prints what the container is allowed to use (cgroup cpu.max, affinity)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayn_amd import setup as S, params as P
from oracle import oracle_py as O
W, H = 1280, 720
cam, world = S.setup((W, H)); wd = world.to_desc(cam); p = P.frame_params(W, H, 2, 3)
tabs = O.build_tables(8, 3, p.volume_marches, p.frame, W, H)
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("cpu.max n/a", e)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
sub = np.arange(0, 3600, 3, dtype=np.uint32)
for th in (8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    t = time.perf_counter(); _, ctr = O.render(wd, p, tabs, threads=th, tile_subset=sub); dt = time.perf_counter() - t
    print(f"threads {th:4d}: {ctr.paths / dt / 1e6:.4f} Mpath/s ({dt:.2f} s for {ctr.paths} paths)", flush=True)
