import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import rayn_amd
from rayn_amd._lib import lib
from common import case
from oracle import oracle_py as O
ctx = rayn_amd.Context(0)
wd, p = case("bulb", 64, 64, 1, 3)
ctx.upload_world(wd)
pts = np.random.default_rng(17).uniform(-1.6, 1.6, 3 * 200000).astype(np.float32).reshape(-1, 3)
pts[:100, 0] = 0.0; pts[:100, 2] = 0.0
out = np.zeros(len(pts), np.float32)
fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
lib().rayn_hip_probe_sdf_dist(ctx.h, C.byref(p), 1, fp(pts), fp(out), len(pts))
ref = O.sdf_dist(wd.hitables[1], pts)
bad = np.nonzero(out.view(np.uint32) != ref.view(np.uint32))[0]
print(len(bad), 'mismatches of', len(pts))
for i in bad[:12]:
    print(i, pts[i], out[i], ref[i], hex(out[i:i+1].view(np.uint32)[0]), hex(ref[i:i+1].view(np.uint32)[0]))
