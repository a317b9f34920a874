#!/usr/bin/env python3
"""Dump the shadow segments (TracedSDF::occluded calls) of one tile through the oracle's diagnostic sink.
usage: dump_shadow_segments.py <c2|c3> <tile_index> <samples> <out.bin>     records: 8 float32 {depth, sample, start.xyz, end.xyz}"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import case  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

wl, tile, samples, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
scene = {"c2": "s1", "c3": "s2"}[wl]
wd, p = case(scene, 1920, 1080, samples, 8)
tabs = O.build_tables(4 * samples, 8, p.volume_marches, p.frame, 1920, 1080)
L = O.lib()
L.oracle_take_shadow_sink.restype = C.c_uint64
L.oracle_take_shadow_sink.argtypes = [C.POINTER(C.c_float), C.c_uint64]
L.oracle_set_shadow_sink(1)
film, ctr = O.render(wd, p, tabs, threads=1, tile_subset=[tile])
n = L.oracle_take_shadow_sink(None, 0)
buf = np.zeros(n, np.float32)
L.oracle_take_shadow_sink(buf.ctypes.data_as(C.POINTER(C.c_float)), n)
L.oracle_set_shadow_sink(0)
buf.tofile(out)
print("tile", tile, "paths", ctr.paths, "segments", ctr.segments, "shadow lanes", n // 8)
