#!/usr/bin/env python3
"""Which floats does rcp_sqrt_rn (device_core.h) get wrong?  Sweeps every non-negative float through probe op 15 (per-64K-block mismatch counts), then evaluates
the mismatching blocks through op 16 and prints the inputs, the device result, the IEEE result and the mantissa patterns involved."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rayn_amd
from rayn_amd._lib import lib

ctx = rayn_amd.Context(0)
fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
bases = (np.arange(0, 0x7F80_0000 + 65536, 65536, dtype=np.uint64)).astype(np.uint32)
out = np.zeros(bases.size, np.float32)
dummy = np.zeros(bases.size, np.float32)
assert lib().rayn_hip_probe_detmath(ctx.h, 15, fp(bases.view(np.float32)), fp(dummy), fp(out), bases.size) == 0
bad_blocks = np.nonzero(out > 0)[0]
print("mismatching inputs:", int(out.sum()), "in", bad_blocks.size, "of", bases.size, "blocks")
shown = 0
expo_hist = {}
mant_s = []
for b in bad_blocks[:400]:
    xb = (np.uint32(bases[b]) + np.arange(65536, dtype=np.uint32)).astype(np.uint32)
    x = xb.view(np.float32)
    got = np.zeros_like(x)
    assert lib().rayn_hip_probe_detmath(ctx.h, 16, fp(x), fp(x), fp(got), x.size) == 0
    with np.errstate(all="ignore"):
        s = np.sqrt(x)
        want = np.float32(1.0) / s
    bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want)))[0]
    for i in bad:
        e = (int(xb[i]) >> 23) & 0xFF
        expo_hist[e & 1] = expo_hist.get(e & 1, 0) + 1
        mant_s.append(int(s[i:i + 1].view(np.uint32)[0]) & 0x7FFFFF)
        if shown < 24:
            shown += 1
            print(f"x=0x{int(xb[i]):08x} ({x[i]:.9g}) s=0x{int(s[i:i+1].view(np.uint32)[0]):08x} got=0x{int(got[i:i+1].view(np.uint32)[0]):08x} want=0x{int(want[i:i+1].view(np.uint32)[0]):08x}"
                  f" diff_ulps={int(got[i:i+1].view(np.int32)[0]) - int(want[i:i+1].view(np.int32)[0])}")
print("exponent parity of x among the mismatches (first 400 blocks):", expo_hist)
ms = np.array(mant_s, dtype=np.int64)
if ms.size:
    print("mantissa of s among mismatches: min 0x%06x max 0x%06x; count with mantissa >= 0x7ffff0: %d; == 0x7fffff: %d; == 0: %d" % (ms.min(), ms.max(), (ms >= 0x7ffff0).sum(), (ms == 0x7fffff).sum(), (ms == 0).sum()))
ctx.close()
