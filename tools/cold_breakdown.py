"""Where the first frame of a fresh process goes: HIP runtime start-up vs rayn_hip_create vs the first render
(arena + code-object load + frame) vs later renders, and what the first use of the OTHER mul_add policy's code object costs.
No torch.  usage: cold_breakdown.py [workload=shipped] [preinit=0|1]
  preinit=1 calls hipInit / hipGetDeviceCount / hipSetDevice / hipFree(0) through libamdhip64 first and times them separately,
  so that what remains in rayn_hip_create is the library's own share."""
import ctypes as C
import os as _os; _os.environ.setdefault("RAYN_HIP_ENV_TUNING", "1")  # the library reads RAYN_HIP_* tuning only under this opt-in (include/rayn_hip.h)
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = sys.argv[1] if len(sys.argv) > 1 else "shipped"
preinit = len(sys.argv) > 2 and int(sys.argv[2]) != 0

import rayn_amd  # noqa: E402  (loads librayn_hip.so; dlopen of libamdhip64 happens here, no device is touched)
from rayn_amd import setup as S  # noqa: E402
from bench import WORKLOADS  # noqa: E402

scene, W, H, samples, bounces, desc = WORKLOADS[wl]
cam, w = S.SCENES[scene]((W, H))
p = rayn_amd.frame_params(W, H, samples, bounces)
tabs = rayn_amd.build_tables(4 * samples, bounces, p.volume_marches, p.frame, W, H)
wd = w.to_desc(cam)

ms = lambda a, b: round((b - a) * 1e3, 2)
rec = {}
if preinit:
    hip = C.CDLL("libamdhip64.so")
    n = C.c_int(0)
    t = time.perf_counter(); hip.hipInit(0); t1 = time.perf_counter(); rec["hipInit"] = ms(t, t1)
    t = time.perf_counter(); hip.hipGetDeviceCount(C.byref(n)); t1 = time.perf_counter(); rec["hipGetDeviceCount"] = ms(t, t1)
    t = time.perf_counter(); hip.hipSetDevice(0); t1 = time.perf_counter(); rec["hipSetDevice"] = ms(t, t1)
    t = time.perf_counter(); hip.hipFree(None); t1 = time.perf_counter(); rec["hipFree(0) [primary context]"] = ms(t, t1)
    if int(sys.argv[2]) >= 2:  # the first REAL device objects of the process: what the runtime charges whoever creates them first
        st, pp, ev = C.c_void_p(), C.c_void_p(), C.c_void_p()
        t = time.perf_counter(); hip.hipStreamCreate(C.byref(st)); t1 = time.perf_counter(); rec["first hipStreamCreate"] = ms(t, t1)
        t = time.perf_counter(); hip.hipMalloc(C.byref(pp), C.c_size_t(4096)); t1 = time.perf_counter(); rec["first hipMalloc"] = ms(t, t1)
        t = time.perf_counter(); hip.hipEventCreate(C.byref(ev)); t1 = time.perf_counter(); rec["first hipEventCreate"] = ms(t, t1)
        t = time.perf_counter(); hip.hipStreamCreate(C.byref(st)); t1 = time.perf_counter(); rec["second hipStreamCreate"] = ms(t, t1)
t0 = time.perf_counter()
ctx = rayn_amd.Context(0)
t1 = time.perf_counter()
ctx.upload_world(wd)
t2 = time.perf_counter()
ctx.render_host(p, tabs)
t3 = time.perf_counter()
dev1 = ctx.stats()["ms_total"]
ctx.render_host(p, tabs)
t4 = time.perf_counter()
dev2 = ctx.stats()["ms_total"]
ctx.render_host(p, tabs)
t5 = time.perf_counter()
ctx.set_fma_policy(1)
ctx.render_host(p, tabs)
t6 = time.perf_counter()
ctx.render_host(p, tabs)
t7 = time.perf_counter()
ctx.close()
t8 = time.perf_counter()
rec.update({"rayn_hip_create": ms(t0, t1), "upload_world": ms(t1, t2), "frame1 (host entry)": ms(t2, t3), "frame1 device": round(dev1, 2),
            "frame2": ms(t3, t4), "frame2 device": round(dev2, 2), "frame3": ms(t4, t5), "first fused-policy frame": ms(t5, t6), "second fused-policy frame": ms(t6, t7), "destroy": ms(t7, t8),
            "cold (create -> frame1 on the host)": ms(t0, t3)})
print(wl, "preinit" if preinit else "plain", rec)
