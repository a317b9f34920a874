"""ctypes binding of the CPU oracle (oracle/librayn_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (rayn_amd/) never imports this module."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class Counters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("packets", C.c_uint64),
                ("dist_evals", C.c_uint64), ("tiles", C.c_uint64)]


def build(force=False):
    """Compile the oracle with g++ (oracle/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE, "-j2"], stdout=subprocess.DEVNULL)


_libs = {}


VARIANTS = ("minmax_swapped", "minmax_ieee", "libm", "normalize_div", "dot_plain", "normals_central", "normals_order", "lerp_alt")


def build_variants():
    """The alternative-reading libraries of oracle/Makefile `variants` (oracle/SENSITIVITY.md); never used by the parity tests."""
    subprocess.check_call(["make", "-C", _HERE, "-j4", "variants"], stdout=subprocess.DEVNULL)


def lib(fma=False, variant=None):
    """The oracle library: default (unfused mul_add), fma=True (fused), or variant=<name of VARIANTS> = ONE assumption of the
    header's list read the other way (sensitivity analysis / naming a wrong assumption when pinning against rayn)."""
    if variant is not None:
        assert variant in VARIANTS and not fma, variant
    name = f"librayn_oracle_{variant}.so" if variant else ("librayn_oracle_fma.so" if fma else "librayn_oracle.so")
    if name not in _libs:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build_variants() if variant else build()
        L = C.CDLL(path)
        fp = C.POINTER(C.c_float)
        up = C.POINTER(C.c_uint32)
        L.oracle_sets_1d.restype = C.c_uint32
        L.oracle_sets_2d.restype = C.c_uint32
        L.oracle_tile_count.restype = C.c_uint32
        L.oracle_render_frame.restype = C.c_int
        L.oracle_render_frame.argtypes = [C.c_void_p, C.c_void_p, fp, fp, fp, fp, fp, fp, fp, fp, C.c_int, up,
                                          C.c_uint32, C.POINTER(Counters)]
        L.oracle_trace_tile.restype = C.c_int64
        L.oracle_trace_tile.argtypes = [C.c_void_p, C.c_void_p, fp, fp, fp, fp, C.c_uint32, C.c_uint64,
                                        up, up, up, up, up, up]
        L.oracle_build_rd_tables.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, fp, fp]
        L.oracle_build_scramble.argtypes = [C.c_uint32, C.c_uint32, fp]
        L.oracle_build_fis_table.argtypes = [C.c_uint32, C.c_float, fp]
        L.oracle_build_fis_table_ex.argtypes = [C.c_uint32, C.c_float, C.c_float, C.c_float, fp]
        _libs[name] = L
    return _libs[name]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def build_tables(spp, max_bounces, volume_marches, frame, width, height, filter_kind=0, filter_radius=1.5, fma=False, filter_params=(0.0, 0.0), variant=None):
    """Samples::new_rd + per-pixel scramble + FilterImportanceSampler::new, oracle versions."""
    L = lib(fma, variant)
    n1 = L.oracle_sets_1d(max_bounces, volume_marches)
    n2 = L.oracle_sets_2d(max_bounces, volume_marches)
    s1 = np.zeros(spp * n1, np.float32)
    s2 = np.zeros(spp * 2 * n2, np.float32)
    L.oracle_build_rd_tables(spp, n1, n2, frame, _fp(s1), _fp(s2))
    scr = np.zeros(width * height, np.float32)
    L.oracle_build_scramble(width, height, _fp(scr))
    fis = np.zeros(512, np.float32)
    L.oracle_build_fis_table_ex(filter_kind, filter_radius, filter_params[0], filter_params[1], _fp(fis))
    return s1, s2, scr, fis


def render(world_desc, params, tables, threads=None, tile_subset=None, fma=False, variant=None):
    """Film::render_frame_into on the CPU.  Returns (film dict, Counters)."""
    L = lib(fma, variant)
    s1, s2, scr, fis = tables
    n = params.width * params.height
    color = np.zeros((n, 3), np.float32)
    alpha = np.zeros(n, np.float32)
    bg = np.zeros((n, 3), np.float32)
    normal = np.zeros((n, 3), np.float32)
    ctr = Counters()
    if threads is None:
        threads = os.cpu_count() or 1
    sub, nsub = None, 0
    if tile_subset is not None:
        arr = np.ascontiguousarray(tile_subset, dtype=np.uint32)
        sub, nsub = _up(arr), len(arr)
    rc = L.oracle_render_frame(C.byref(world_desc), C.byref(params), _fp(s1), _fp(s2), _fp(scr), _fp(fis),
                               _fp(color), _fp(alpha), _fp(bg), _fp(normal), threads, sub, nsub, C.byref(ctr))
    if rc != 0:
        raise RuntimeError(f"oracle_render_frame failed: {rc}")
    h, w = params.height, params.width
    return {"color": color.reshape(h, w, 3), "alpha": alpha.reshape(h, w), "background": bg.reshape(h, w, 3),
            "normal": normal.reshape(h, w, 3)}, ctr


def trace_tile(world_desc, params, tables, tile_index, fma=False, variant=None):
    """Per-depth packet lanes of one tile in process_hits order: dict of uint32 arrays."""
    L = lib(fma, variant)
    s1, s2, scr, fis = tables
    cap = (params.tile_w * params.tile_h * params.samples * 4 + 64) * (params.max_bounces + 2)
    arrs = [np.zeros(cap, np.uint32) for _ in range(6)]
    n = L.oracle_trace_tile(C.byref(world_desc), C.byref(params), _fp(s1), _fp(s2), _fp(scr), _fp(fis), tile_index,
                            cap, *[_up(a) for a in arrs])
    if n < 0 or n > cap:
        raise RuntimeError(f"oracle_trace_tile failed: {n}")
    keys = ["depth", "obj", "px", "py", "sample", "valid"]
    return {k: a[:n].copy() for k, a in zip(keys, arrs)}


def sdf_dist(hitable, pts, fma=False):
    L = lib(fma)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
    out = np.zeros(len(pts), np.float32)
    L.oracle_sdf_dist(C.byref(hitable), _fp(pts), _fp(out), C.c_uint64(len(pts)))
    return out


def closest_hit(world_desc, params, depth, org, dirs, fma=False):
    L = lib(fma)
    org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
    dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    t = np.zeros(len(org), np.float32)
    obj = np.zeros(len(org), np.uint32)
    L.oracle_closest_hit(C.byref(world_desc), C.byref(params), C.c_uint32(depth), _fp(org), _fp(dirs), _fp(t), _up(obj),
                         C.c_uint64(len(org)))
    return t, obj


def test_occluded(world_desc, params, start, end, fma=False):
    L = lib(fma)
    start = np.ascontiguousarray(start, np.float32).reshape(-1, 3)
    end = np.ascontiguousarray(end, np.float32).reshape(-1, 3)
    out = np.zeros(len(start), np.float32)
    L.oracle_test_occluded(C.byref(world_desc), C.byref(params), _fp(start), _fp(end), _fp(out), C.c_uint64(len(start)))
    return out


def detmath(op, a, b=None, fma=False):
    L = lib(fma)
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(a if b is None else b, np.float32)
    out = np.zeros_like(a)
    L.oracle_detmath(C.c_uint32(op), _fp(a), _fp(b), _fp(out), C.c_uint64(a.size))
    return out
