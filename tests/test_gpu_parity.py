"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bar: per-pixel L2 < 1e-4 (BASELINE.json north_star); the design target is bit-exact."""
import numpy as np
import pytest

from common import bits_equal, case, film_equal_bits, film_l2

pytestmark = pytest.mark.gpu
L2_TOL = 1e-4  # north_star: per-pixel L2 < 1e-4 (float)


def _tables(oracle, p):
    return oracle.build_tables(4 * p.samples, p.max_bounces, p.volume_marches, p.frame, p.width, p.height)


def _rand(n, lo, hi, seed):
    return np.random.default_rng(seed).uniform(lo, hi, n).astype(np.float32)


@pytest.mark.parametrize("op,lo,hi", [(0, -30, 5), (1, -8, 8), (2, -8, 8), (3, -1.5, 1.5), (4, -6, 6), (5, 0, 1), (0, -110, 95), (1, -2e4, 2e4), (5, 0, 40)])
def test_detmath_bit_exact(gpu_ctx, oracle, op, lo, hi):
    """The kernels' elementary functions against the oracle's evaluation of include/rayn_detmath.h: same bits on 2 M arguments per
    case, incl. huge / tiny results and special values.  The kernels evaluate them through include/rayn_detmath_fast.h (shorter
    polynomials + rounding-safety test + fallback): this is the device-side proof that the shortcut returns rayn_detmath.h's bits."""
    import ctypes as C
    from rayn_amd._lib import lib
    n = 2_000_000
    a = _rand(n, lo, hi, 1 + op)
    b = _rand(n, 0.05, 310.0 if hi > 1 else 12.0, 50 + op)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40, 3e38, -3e38, 88.0, -87.0, 1e4, -1e4], np.float32)
    a[:special.size] = special
    b[special.size:2 * special.size] = special
    a[special.size:2 * special.size] = _rand(special.size, lo, hi, 99)
    out = np.zeros_like(a)
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    assert lib().rayn_hip_probe_detmath(gpu_ctx.h, op, fp(a), fp(b), fp(out), a.size) == 0
    ref = oracle.detmath(op, a, b)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_fast_division_is_ieee(gpu_ctx):
    """div_nr (the fold's division) == IEEE division bit for bit over the operand range the host enables it for."""
    import ctypes as C
    from rayn_amd._lib import lib
    rng = np.random.default_rng(5)
    n = 4_000_000
    expo = rng.integers(-60, 61, n)
    den = (rng.uniform(1.0, 2.0, n) * np.exp2(expo.astype(np.float64))).astype(np.float32)
    num = np.full(n, np.float32(1.9 * 1.9), np.float32)
    num[n // 2:] = (rng.uniform(1.0, 2.0, n - n // 2) * np.exp2(rng.integers(-30, 31, n - n // 2).astype(np.float64))).astype(np.float32)
    keep = np.abs(np.log2(num.astype(np.float64) / den)) < 100  # quotient stays a normal float
    num, den = num[keep], den[keep]
    out = np.zeros_like(num)
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    assert lib().rayn_hip_probe_detmath(gpu_ctx.h, 6, fp(num), fp(den), fp(out), num.size) == 0
    assert np.array_equal(out.view(np.uint32), (num / den).view(np.uint32))


def test_fast_sqrt_is_ieee_exhaustive(gpu_ctx):
    """sqrt_rn (rsq + one residual correction inside [2^-60, 2^60), hipcc's IEEE sqrt elsewhere) == IEEE sqrt for EVERY
    non-negative float bit pattern, plus negative / NaN samples."""
    import ctypes as C
    from rayn_amd._lib import lib
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    bases = (np.arange(0, 0x7F80_0000 + 65536, 65536, dtype=np.uint64)).astype(np.uint32)  # covers 0 .. +inf and the first NaNs
    out = np.zeros(bases.size, np.float32)
    dummy = np.zeros(bases.size, np.float32)
    assert lib().rayn_hip_probe_detmath(gpu_ctx.h, 13, fp(bases.view(np.float32)), fp(dummy), fp(out), bases.size) == 0
    assert out.sum() == 0, f"{int(out.sum())} mismatching inputs, first block base 0x{int(bases[np.argmax(out > 0)]):08x}"
    x = np.concatenate([_rand(100000, -1e6, 1e6, 3), np.array([-0.0, np.nan, -np.inf, np.inf, 0.0], np.float32)])
    got = np.zeros_like(x)
    assert lib().rayn_hip_probe_detmath(gpu_ctx.h, 8, fp(x), fp(x), fp(got), x.size) == 0
    with np.errstate(invalid="ignore"):
        assert bits_equal(got, np.sqrt(x))


def test_fast_rcp_sqrt_is_ieee_exhaustive(gpu_ctx):
    """rcp_sqrt_rn (1 / |v| of normalized(), the Mandelbulb step's 1 / sqrt(k3^7): sqrt and reciprocal from ONE v_rsq_f32 inside [2^-60, 2^60), the IEEE
    sequences elsewhere) == fl(1 / fl(sqrt(x))) for EVERY non-negative float bit pattern, plus negative / NaN samples."""
    import ctypes as C
    from rayn_amd._lib import lib
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    bases = (np.arange(0, 0x7F80_0000 + 65536, 65536, dtype=np.uint64)).astype(np.uint32)
    out = np.zeros(bases.size, np.float32)
    dummy = np.zeros(bases.size, np.float32)
    assert lib().rayn_hip_probe_detmath(gpu_ctx.h, 15, fp(bases.view(np.float32)), fp(dummy), fp(out), bases.size) == 0
    assert out.sum() == 0, f"{int(out.sum())} mismatching inputs, first block base 0x{int(bases[np.argmax(out > 0)]):08x}"
    x = np.concatenate([_rand(100000, -1e6, 1e6, 4), np.array([-0.0, np.nan, -np.inf, np.inf, 0.0, 1e-45, 3e38], np.float32)])
    got = np.zeros_like(x)
    assert lib().rayn_hip_probe_detmath(gpu_ctx.h, 16, fp(x), fp(x), fp(got), x.size) == 0
    with np.errstate(invalid="ignore", divide="ignore"):
        assert bits_equal(got, np.float32(1.0) / np.sqrt(x))


def test_fast_normalise_is_ieee(gpu_ctx):
    """v / |v| through div_by_mag (shared refined reciprocal inside the exponent window, IEEE divisions outside) ==
    three IEEE divisions, for ordinary vectors, vectors with zero / tiny / huge components and extreme ratios."""
    import ctypes as C
    from rayn_amd._lib import lib
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    rng = np.random.default_rng(11)
    n = 1_000_000
    v = rng.uniform(-1, 1, (n, 3))
    v *= np.exp2(rng.integers(-70, 71, (n, 1)).astype(np.float64))          # whole-vector scale across and beyond the window
    v[: n // 4] *= np.exp2(rng.integers(-40, 1, (n // 4, 3)).astype(np.float64))  # very uneven components
    v = v.astype(np.float32)
    v[:1000, 0] = 0.0; v[1000:2000, 1] = -0.0; v[2000:2100] = 0.0; v[2100:2200, 2] = np.float32(1e-42)
    v[2200:2300, 0] = np.inf; v[2300:2400, 1] = np.nan
    dummy = np.zeros(n, np.float32)
    res = []
    for op in (9, 10, 11, 12):
        out = np.zeros(n, np.float32)
        assert lib().rayn_hip_probe_detmath(gpu_ctx.h, op, fp(v), fp(dummy), fp(out), n) == 0
        res.append(out)
    with np.errstate(all="ignore"):
        x, y, z = v[:, 0], v[:, 1], v[:, 2]
        m = np.sqrt(x * x + (y * y + z * z))  # dot(): muladd(x, x, muladd(y, y, z * z)), unfused
        assert bits_equal(res[3], m)
        for c in range(3):
            assert bits_equal(res[c], v[:, c] / m), c


def _probe_setup(gpu_ctx, name):
    wd, p = case(name, 64, 64, 1, 3)
    gpu_ctx.upload_world(wd)
    return wd, p


def test_mandelbulb_dist_bit_exact(gpu_ctx, oracle):
    import ctypes as C
    from rayn_amd._lib import lib
    wd, p = _probe_setup(gpu_ctx, "bulb")
    pts = _rand(3 * 200000, -1.6, 1.6, 17).reshape(-1, 3)
    pts[:100, 0] = 0.0; pts[:100, 2] = 0.0  # the degenerate axis x = z = 0 (k3 = 0 -> inf/NaN), must agree too
    out = np.zeros(len(pts), np.float32)
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    assert lib().rayn_hip_probe_sdf_dist(gpu_ctx.h, C.byref(p), 1, fp(pts), fp(out), len(pts)) == 0
    ref = oracle.sdf_dist(wd.hitables[1], pts)
    assert bits_equal(out, ref)
    assert np.isnan(ref[:100]).all() and np.isfinite(ref[100:]).mean() > 0.99


def test_mandelbox_dist_bit_exact(gpu_ctx, oracle):
    import ctypes as C
    from rayn_amd._lib import lib
    wd, p = _probe_setup(gpu_ctx, "s1")
    pts = _rand(3 * 300000, -3.5, 3.5, 7).reshape(-1, 3)
    pts[:1000] *= 30.0  # far field
    out = np.zeros(len(pts), np.float32)
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    assert lib().rayn_hip_probe_sdf_dist(gpu_ctx.h, C.byref(p), 1, fp(pts), fp(out), len(pts)) == 0
    ref = oracle.sdf_dist(wd.hitables[1], pts)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def _probe_world(name, sdf_only=False):
    """(world_desc, frame_params) of a probe scene: a SCENES tag, or one of _custom_world's kinds; sdf_only drops the analytic spheres (rayn_hip_probe_shadow
    marches the TracedSDF factors of HitableStore::test_occluded - the spheres are the shading kernel's part)."""
    import rayn_amd as R
    from rayn_amd import params as P
    from rayn_amd import setup as S
    if name in S.SCENES:
        cam, world = S.SCENES[name]((64, 64))
    else:
        cam, world = _custom_scene(name, (64, 64))
    if sdf_only:
        world.hitables[:] = [h for h in world.hitables if isinstance(h, R.TracedSDF)]
    return world.to_desc(cam), P.frame_params(64, 64, 1, 3)


# the PRODUCT extend kernels (k_extend1 for a single-SDF scene - MandelBox, sphere SDF, Mandelbulb - and the generic k_extend of a multi-SDF scene) on a synthetic queue
@pytest.mark.parametrize("name,depth", [("s1", 0), ("s1", 2), ("s0", 0), ("bulb", 0), ("bulb", 2), ("two_sdfs", 0), ("two_sdfs", 2)])
def test_closest_hit_bit_exact(gpu_ctx, oracle, name, depth):
    import ctypes as C
    from rayn_amd._lib import lib
    wd, p = _probe_world(name)
    gpu_ctx.upload_world(wd)
    n = 40000 - 13  # not a multiple of 64: the queue's padding entries
    org = _rand(3 * n, -3.0, 3.0, 11).reshape(-1, 3)
    org[: n // 2] = np.array([-1.0125, 0.45, 4.5], np.float32)  # the shipped camera position
    d = _rand(3 * n, -1.0, 1.0, 12).reshape(-1, 3)
    d[: n // 2] = -org[: n // 2] + _rand(3 * (n // 2), -1.5, 1.5, 13).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    t = np.zeros(n, np.float32)
    obj = np.zeros(n, np.uint32)
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    assert lib().rayn_hip_probe_extend(gpu_ctx.h, C.byref(p), depth, fp(org), fp(d), fp(t), obj.ctypes.data_as(C.POINTER(C.c_uint32)), n) == 0, gpu_ctx.last_error()
    rt, robj = oracle.closest_hit(wd, p, depth, org, d)
    assert np.array_equal(obj, robj)
    assert np.array_equal(t.view(np.uint32), rt.view(np.uint32))
    assert len(np.unique(robj)) >= 3  # sky, lights, the SDF(s)


# the PRODUCT shadow-march kernels: k_shadow1 (MandelBox, sphere SDF), k_shadow_bulb (Mandelbulb), the generic k_shadow (two SDFs)
@pytest.mark.parametrize("name", ["s1", "s0", "bulb", "two_sdfs"])
def test_occluded_bit_exact(gpu_ctx, oracle, name):
    import ctypes as C
    from rayn_amd._lib import lib
    wd, p = _probe_world(name, sdf_only=True)
    gpu_ctx.upload_world(wd)
    n = 40000 + 21
    a = _rand(3 * n, -2.0, 2.0, 21).reshape(-1, 3)
    b = _rand(3 * n, -2.0, 2.0, 22).reshape(-1, 3)
    out = np.zeros(n, np.float32)
    fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    assert lib().rayn_hip_probe_shadow(gpu_ctx.h, C.byref(p), fp(a), fp(b), fp(out), n) == 0, gpu_ctx.last_error()
    ref = oracle.test_occluded(wd, p, a, b)
    assert np.array_equal(out, ref)
    assert 0.02 < ref.mean() < 0.98  # both outcomes exercised


FILM_CASES = [
    ("s0", 64, 64, 4, 4, {}),                         # BASELINE config 1 shape, scaled
    ("s1", 48, 32, 2, 3, {}),                         # shipped scene, volumes off
    ("s1", 40, 24, 4, 8, {}),                         # config 2 depth (8 bounces, roulette active)
    ("s2", 40, 24, 2, 3, {}),                         # shipped scene incl. homogeneous volume (config 3 path)
    ("s1", 50, 37, 1, 2, {}),                         # ragged: partial tiles on both axes
    ("s1", 20, 16, 1, 2, {"tile_size": (16, 16)}),    # width 20: reference tile-count quirk (under-coverage)
    ("s2", 32, 32, 1, 2, {"volume_marches": 3}),      # VM = 3: samples_1d[3] doubles as fresnel sample
    ("bulb", 48, 32, 2, 4, {}),                       # EXTENSION: Mandelbulb DE (not in the reference)
    ("s1", 16, 16, 1, 0, {}),                         # max_bounces 0: everything terminates at depth 0
    ("s1", 1, 1, 3, 2, {}),                           # 1x1 film, spp = 12 (not a power of two: resolve sort padding)
    ("s2", 3, 5, 5, 3, {}),                           # smaller than one tile, spp = 20
    ("s1", 40, 24, 1, 3, {"tile_size": (8, 4)}),      # small tiles: packet grouping is per tile, so the result differs from 16x16
    ("s1", 64, 32, 1, 2, {"tile_size": (32, 32)}),    # 1024-pixel tiles (the supported maximum)
    ("s1", 32, 32, 1, 3, {"frame": 7, "time_range": (0.5, 0.75)}),  # other frame seed + shutter interval
    ("s1", 32, 16, 1, 2, {"max_marches": 12, "max_vis_marches": 5}),  # march budgets exhausted: 't' is returned as a hit (src/sdf.rs:82)
    ("s1", 32, 16, 1, 2, {"sdf_detail_scale": 2.0, "world_radius": 20.0}),
    ("s3", 6, 4, 1024, 16, {"tile_size": (2, 2)}),   # config 5's regime: 4096 spp (the four-wave resolve's maximum), 16 bounces, moving camera (motion blur)
    ("s1", 3, 2, 400, 5, {"tile_size": (2, 2)}),     # 1600 spp: k_resolve_blk<256, 8> (1025..2048 spp), not a power of two
    ("s2", 2, 2, 512, 4, {"tile_size": (2, 2)}),     # 2048 spp exactly, volume
    ("s1", 2, 2, 700, 6, {"tile_size": (2, 1)}),     # 2800 spp: k_resolve_blk<512, 8> with padding
    ("s1", 4, 3, 150, 6, {"tile_size": (2, 2)}),     # 600 spp: k_resolve_blk<128, 8> with padding
    ("s1", 2, 1, 1100, 3, {"tile_size": (2, 1)}),    # 4400 spp: just above 4096, not a power of two (sixteen-wave resolve, mostly padding)
    ("s3", 4, 2, 2048, 6, {"tile_size": (2, 2)}),    # 8192 spp, moving camera
    ("s2", 2, 2, 4096, 4, {"tile_size": (2, 2)}),    # 16384 spp (the supported maximum), volume
    ("s1", 2, 2, 2048, 0, {"tile_size": (2, 2)}),    # 8192 spp, depth 0 only: the keys arrive sorted (the resolve's no-sort path)
    ("s1", 24, 16, 256, 0, {"tile_size": (4, 4)}),   # 1024 spp, depth 0 only: 32 pixels see the fractal (Color) AND an emissive proxy (Background, a later object) -> the
                                                      # Background samples are NOT a prefix of the add order: k_resolve_blk's flag-per-term fallback
    ("s1", 8, 8, 256, 12, {"tile_size": (4, 4)}),    # config 4's regime: 1024 spp, 12 bounces
    ("s2", 16, 8, 256, 8, {"tile_size": (4, 4)}),    # config 3's regime: 1024 spp (resolve sorts 1024 keys), 8 bounces, volume; 4x4 tiles keep the oracle fast
]


@pytest.mark.parametrize("name,w,h,samples,bounces,kw", FILM_CASES)
def test_film_parity(gpu_ctx, oracle, name, w, h, samples, bounces, kw):
    wd, p = case(name, w, h, samples, bounces, **kw)
    tabs = _tables(oracle, p)
    ref, ctr = oracle.render(wd, p, tabs)
    gpu_ctx.upload_world(wd)
    out = gpu_ctx.render_host(p, tabs)
    st = gpu_ctx.stats()
    assert st["paths"] == ctr.paths
    assert st["segments"] == ctr.segments
    assert st["shaded_slots"] >= 4 * ctr.packets  # GPU pads tile tails to 64
    l2 = film_l2(out, ref)
    assert l2 < L2_TOL, f"per-pixel L2 {l2}"
    assert film_equal_bits(out, ref), f"not bit-exact (L2 {l2})"


@pytest.mark.parametrize("name,tile", [("s1", 1), ("s2", 4), ("s1", 5)])
def test_packet_order_matches_oracle(gpu_ctx, oracle, name, tile):
    """the GPU's binned queue of a tile, per depth, IS the oracle's packet list (object-major,
    insertion order, x4 padding) lane for lane — not just the same film."""
    wd, p = case(name, 48, 48, 2, 4)
    tabs = _tables(oracle, p)
    ref = oracle.trace_tile(wd, p, tabs, tile)
    gpu_ctx.upload_world(wd)
    gpu_ctx.set_trace_tile(tile)
    try:
        gpu_ctx.render_host(p, tabs)
        got = gpu_ctx.trace()
    finally:
        gpu_ctx.set_trace_tile(-1)
    assert len(ref["depth"]) > 1000 and ref["depth"].max() >= 3
    for k in ("depth", "obj", "px", "py", "sample", "valid"):
        assert np.array_equal(got[k], ref[k]), k


def test_batching_is_invisible(gpu_ctx, oracle):
    """Splitting the frame into many small batches must not change a single bit."""
    wd, p = case("s1", 64, 48, 2, 3)
    tabs = _tables(oracle, p)
    gpu_ctx.upload_world(wd)
    a = gpu_ctx.render_host(p, tabs)
    gpu_ctx.set_batch_paths(4096)
    b = gpu_ctx.render_host(p, tabs)
    assert gpu_ctx.stats()["batches"] > 1
    gpu_ctx.set_batch_paths(1 << 28)
    assert film_equal_bits(a, b)


def test_first_frame_batching_is_invisible(oracle):
    """The first frame of a context runs in batches sized for a small arena (rayn_hip_set_cold_bytes), later frames in
    full-size batches out of a re-allocated arena: same bits as the oracle on both, with one and two workers, and with the policy off."""
    import rayn_amd
    wd, p = case("s2", 96, 64, 2, 3)  # 24 tiles of 2048 paths
    tabs = _tables(oracle, p)
    ref, ctr = oracle.render(wd, p, tabs)
    for workers, cold in ((1, 4096 * 700), (2, 4096 * 700), (1, 0)):  # 667 B per path with the volume path: 4298-path batches
        ctx = rayn_amd.Context(0)
        try:
            ctx.upload_world(wd)
            ctx.set_workers(workers, 0)
            ctx.set_cold_bytes(cold)
            first = ctx.render_host(p, tabs)
            st = ctx.stats()
            assert st["paths"] == ctr.paths and st["segments"] == ctr.segments
            assert st["batches"] == (12 if cold else workers), st["batches"]  # two 2048-path tiles per batch
            assert film_equal_bits(first, ref)
            second = ctx.render_host(p, tabs)  # full-size batches: one per worker, arena grown
            assert ctx.stats()["batches"] == workers
            assert film_equal_bits(second, ref)
        finally:
            ctx.close()


def test_host_entry_preserves_unowned_pixels_only_when_there_are_any(gpu_ctx, oracle):
    """rayn_hip_render_frame uploads the caller's film only when the call leaves pixels alone (a tile share): those keep the
    caller's values; a whole-frame call overwrites every pixel without reading the caller's buffer."""
    wd, p = case("s1", 64, 48, 1, 2)
    tabs = _tables(oracle, p)
    ref, _ = oracle.render(wd, p, tabs)
    gpu_ctx.upload_world(wd)
    n = p.width * p.height
    mark = lambda: {"color": np.full((n, 3), 7.0, np.float32), "alpha": np.full(n, 7.0, np.float32), "background": np.full((n, 3), 7.0, np.float32),
                    "normal": np.full((n, 3), 7.0, np.float32)}
    whole = gpu_ctx.render_host(p, tabs, out=mark())
    assert film_equal_bits(whole, ref)
    _, half = case("s1", 64, 48, 1, 2, tile_first=1, tile_step=2)
    part = gpu_ctx.render_host(half, tabs, out=mark())
    owned = np.zeros((p.height, p.width), bool)
    from rayn_amd.distributed import owned_pixels
    owned.reshape(-1)[owned_pixels(p.width, p.height, p.tile_w, p.tile_h, 1, 2)] = True
    assert owned.any() and not owned.all()
    assert np.array_equal(part["alpha"][owned].view(np.uint32), ref["alpha"][owned].view(np.uint32))
    assert np.array_equal(part["color"][owned].view(np.uint32), ref["color"][owned].view(np.uint32))
    assert (part["alpha"][~owned] == 7.0).all() and (part["color"][~owned] == 7.0).all()


def test_two_workers_are_invisible(gpu_ctx, oracle):
    """The two-worker pipeline (two host threads / streams) on a small frame, with several batches per worker."""
    wd, p = case("s2", 64, 48, 2, 3)
    tabs = _tables(oracle, p)
    ref, ctr = oracle.render(wd, p, tabs)
    gpu_ctx.upload_world(wd)
    gpu_ctx.set_workers(2, 0)
    gpu_ctx.set_batch_paths(8192)
    try:
        out = gpu_ctx.render_host(p, tabs)
        st = gpu_ctx.stats()
    finally:
        gpu_ctx.set_workers(2)
        gpu_ctx.set_batch_paths(1 << 28)
    assert st["batches"] >= 4 and st["paths"] == ctr.paths and st["segments"] == ctr.segments
    assert film_equal_bits(out, ref)
    gpu_ctx.set_workers(1)
    try:
        one = gpu_ctx.render_host(p, tabs)
    finally:
        gpu_ctx.set_workers(2)
    assert film_equal_bits(one, ref)


def test_tuning_variants_are_invisible(oracle, monkeypatch):
    """The code paths behind the tuning switches (generic multi-SDF march kernels instead of the single-SDF fast path, other
    refill / prefetch thresholds of the persistent waves, one worker) give the same film bit for bit."""
    import rayn_amd
    wd, p = case("s2", 48, 32, 2, 3)
    tabs = _tables(oracle, p)
    ref, ctr = oracle.render(wd, p, tabs)
    monkeypatch.setenv("RAYN_HIP_ENV_TUNING", "1")  # the library reads its tuning variables only under this opt-in (rayn_hip.h)
    for env in ({"RAYN_HIP_FAST_PATH": "0"}, {"RAYN_HIP_PREFETCH_SHADOW": "8", "RAYN_HIP_PREFETCH_EXTEND": "60"}, {"RAYN_HIP_FAST_PATH": "0", "RAYN_HIP_REFILL_SHADOW": "1", "RAYN_HIP_WORKERS": "1"},
                {"RAYN_HIP_SDF_TEMPLATES": "0"}, {"RAYN_HIP_BOX12S": "0"}):  # r6: the single-SDF kernels that read the SDF kind from the object instead of their per-kind instantiations
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = rayn_amd.Context(0)
        try:
            ctx.upload_world(wd)
            out = ctx.render_host(p, tabs)
            st = ctx.stats()
        finally:
            ctx.close()
            for k in env:
                monkeypatch.delenv(k)
        assert st["segments"] == ctr.segments and st["shadow_jobs"] > 0
        assert film_equal_bits(out, ref), env


def test_bulb_march_kernel_variants_are_invisible(oracle, monkeypatch):
    """r6: a single-Mandelbulb scene marches its shadow segments with k_shadow_bulb (rayn_amd/csrc/march_bulb.h: K rays per lane, rounds of refill / orbits /
    epilogues, orbits pulled from a per-wave job list in LDS and carried across rounds).  Every shape of it - 2 / 3 / 4 rays per lane, one or two orbit steps
    per trip, drain-everything rounds (ORBIT_MIN 0), eager and lazy refill - and the generic k_shadow1 give the oracle's film bit for bit, with and without the
    volume, with a moving bulb (packet times) and with exhausted march budgets."""
    import rayn_amd
    from rayn_amd import params as P
    cases = []
    for name, kw in (("bulbv", {}), ("bulb", {"max_marches": 12, "max_vis_marches": 5}), ("bulbm", {})):
        cam, world = rayn_amd.setup.SCENES[name]((40, 24))
        wd = world.to_desc(cam)
        p = P.frame_params(40, 24, 2, 4, **kw)
        tabs = _tables(oracle, p)
        cases.append((wd, p, tabs, oracle.render(wd, p, tabs)))
    monkeypatch.setenv("RAYN_HIP_ENV_TUNING", "1")
    envs = ({"RAYN_HIP_BULB_PATH": "0"}, {}, {"RAYN_HIP_BULB_PATH": "0", "RAYN_HIP_SDF_TEMPLATES": "0"}, {"RAYN_HIP_BULB_RAYS": "2", "RAYN_HIP_BULB_STEPS": "2"}, {"RAYN_HIP_BULB_RAYS": "4", "RAYN_HIP_BULB_ORBIT_MIN": "0", "RAYN_HIP_BULB_STEPS": "1"},
            {"RAYN_HIP_BULB_RAYS": "3", "RAYN_HIP_BULB_ORBIT_MIN": "63", "RAYN_HIP_BULB_PREFETCH": "1", "RAYN_HIP_BULB_STEPS": "1"}, {"RAYN_HIP_BULB_RAYS": "2", "RAYN_HIP_BULB_STEPS": "1"}, {"RAYN_HIP_BULB_RAYS": "4", "RAYN_HIP_BULB_STEPS": "2", "RAYN_HIP_BULB_PREFETCH": "200"})
    for env in envs:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = rayn_amd.Context(0)
        try:
            for wd, p, tabs, (ref, ctr) in cases:
                ctx.upload_world(wd)
                out = ctx.render_host(p, tabs)
                st = ctx.stats()
                assert st["segments"] == ctr.segments and st["shadow_jobs"] > 0
                assert film_equal_bits(out, ref), env
            if not env:  # the instrumented variant of the shipped shape: same film, and the stage accounting bench.py quotes is consistent
                wd, p, tabs, (ref, ctr) = cases[0]
                ctx.upload_world(wd)
                ctx.set_profiling(False, True)
                out = ctx.render_host(p, tabs)
                ev, it, slots = ctx.eval_counts(), ctx.sdf_iterations(), ctx.stage_slots()
                ctx.set_profiling(False, False)
                assert film_equal_bits(out, ref)
                assert 0 < it["shadow"] <= slots["shadow_orbit"] and 0 < ev["shadow"] <= slots["shadow_epilogue"], (ev, it, slots)
        finally:
            ctx.close()
            for k in env:
                monkeypatch.delenv(k)


def test_env_tuning_is_opt_in(oracle, monkeypatch):
    """A host that merely exports RAYN_HIP_BATCH_PATHS / RAYN_HIP_PROFILE does not change how its frames run: the library reads its
    tuning variables only when RAYN_HIP_ENV_TUNING=1 is set at context creation (include/rayn_hip.h)."""
    import rayn_amd
    wd, p = case("s1", 64, 48, 2, 2)  # 24 576 paths in 12 tiles
    tabs = _tables(oracle, p)
    monkeypatch.setenv("RAYN_HIP_BATCH_PATHS", "4096")
    monkeypatch.setenv("RAYN_HIP_PROFILE", "1")
    monkeypatch.delenv("RAYN_HIP_ENV_TUNING", raising=False)
    got = {}
    for opt_in in (False, True):
        if opt_in:
            monkeypatch.setenv("RAYN_HIP_ENV_TUNING", "1")
        ctx = rayn_amd.Context(0)
        try:
            ctx.upload_world(wd)
            got[opt_in] = (ctx.render_host(p, tabs), ctx.stats())
        finally:
            ctx.close()
    assert got[False][1]["batches"] == 1 and got[False][1]["ms_extend"] == 0.0   # ignored: one batch, no per-launch events
    assert got[True][1]["batches"] >= 6 and got[True][1]["ms_extend"] > 0.0      # honoured under the opt-in
    assert film_equal_bits(got[False][0], got[True][0])


def test_tile_partition_union(gpu_ctx, oracle):
    """tile_first/tile_step (the multi-GPU film partition): the union of the strided renders == the full frame."""
    wd, p = case("s1", 64, 48, 1, 2)
    tabs = _tables(oracle, p)
    gpu_ctx.upload_world(wd)
    full = gpu_ctx.render_host(p, tabs)
    n = p.width * p.height
    out = {"color": np.zeros((n, 3), np.float32), "alpha": np.zeros(n, np.float32), "background": np.zeros((n, 3), np.float32),
           "normal": np.zeros((n, 3), np.float32)}
    for r in range(3):
        wd2, p2 = case("s1", 64, 48, 1, 2, tile_first=r, tile_step=3)
        res = gpu_ctx.render_host(p2, tabs, out)
    assert film_equal_bits(res, full)


def test_instrumented_variant_matches(gpu_ctx, oracle):
    """The eval-counting kernel variants (roofline accounting) produce the same film and a plausible count."""
    wd, p = case("s1", 48, 32, 1, 2)
    tabs = _tables(oracle, p)
    ref, ctr = oracle.render(wd, p, tabs)
    gpu_ctx.upload_world(wd)
    gpu_ctx.set_profiling(True, True)
    out = gpu_ctx.render_host(p, tabs)
    ev = gpu_ctx.eval_counts()
    st = gpu_ctx.stats()
    gpu_ctx.set_profiling(False, False)
    assert film_equal_bits(out, ref)
    assert all(v > 0 for v in ev.values())
    assert sum(ev.values()) <= ctr.dist_evals  # oracle counts all 4 lanes of every packet call
    assert st["ms_extend"] > 0 and st["ms_shade"] > 0 and st["ms_shadow"] > 0


def test_errors_instead_of_panics(gpu_ctx):
    import rayn_amd
    wd, p = case("s1", 32, 32, 1, 2)
    gpu_ctx.upload_world(wd)
    p.volume_marches = 1
    with pytest.raises(rayn_amd.film.RaynHipError):
        gpu_ctx.render_host(p, [np.zeros(8, np.float32)] * 4)
    # closed-set limits are errors too: more than 16384 spp / more than 120 bounces
    for samples, bounces in ((4097, 2), (1, 121)):
        wd, p = case("s1", 2, 2, samples, bounces, tile_size=(2, 2))
        with pytest.raises(rayn_amd.film.RaynHipError):
            gpu_ctx.render_host(p, [np.zeros(8, np.float32)] * 4)


# ---- the rest of the closed set (SURVEY.md section 8 f/N4): cameras, Lambertian, Box filter, odd scenes ----
def _custom_scene(kind, res):
    """(camera handle, World) of one of the closed-set test scenes"""
    import rayn_amd as R
    from rayn_amd import setup as S
    from rayn_amd.scene import _mul
    cam_h, world = S.setup(res, volumes=(kind == "thinlens_volume"), sdf="mandelbox")
    resf = (float(res[0]), float(res[1]))
    origin = _mul(R.vec3(-0.45, 0.2, 2.0), 2.25)
    if kind in ("thinlens", "thinlens_volume"):
        world.cameras = R.CameraStore()
        cam_h = world.cameras.add_camera(R.ThinLensCamera(resf, 50.0, 0.08, origin, R.vec3(0, 0, 0), R.vec3(0, 1, 0), R.vec3(0.2, 0.1, 0.9)))
    elif kind == "ortho":
        world.cameras = R.CameraStore()
        cam_h = world.cameras.add_camera(R.OrthographicCamera(resf, 11.0 / 4.0, R.vec3(9.5, -3.5, 9.5), R.vec3(0.0, 0.8, 0.0), R.vec3(0, 1, 0)))
    elif kind == "anim_pinhole":  # camera motion blur: origin and up are closures |t| base + vel*t (lane-0 time quirk)
        world.cameras = R.CameraStore()
        cam_h = world.cameras.add_camera(R.PinholeCamera(resf, 60.0, R.Linear(origin, R.vec3(6.0, -2.0, 1.0)), R.vec3(0, 0, 0),
                                                         R.Linear(R.vec3(0, 1, 0), R.vec3(2.0, 0.0, 0.0))))
    elif kind == "anim_thinlens":
        world.cameras = R.CameraStore()
        cam_h = world.cameras.add_camera(R.ThinLensCamera(resf, 50.0, 0.05, origin, R.Linear(R.vec3(0, 0, 0), R.vec3(1.0, 2.0, 0.0)), R.vec3(0, 1, 0),
                                                          R.Linear(R.vec3(0.2, 0.1, 0.9), R.vec3(0.0, 0.0, -4.0))))
    elif kind == "anim_spheres":  # closure transform_seq on Spheres: motion-blurred light proxies + a moving diffuse ball
        for i in (2, 3, 4):
            s = world.hitables[i]
            s.transform_seq = R.Linear(s.transform_seq, R.vec3(3.0, -2.0, 1.5))
        ball = world.materials.add_material(R.Lambertian(R.Srgb(0.7, 0.6, 0.5)))
        world.hitables.push(R.Sphere(R.Linear(R.vec3(-1.6, -0.4, 1.9), R.vec3(9.0, 3.0, 0.0)), 0.3, ball))
    elif kind == "lambertian":
        world.materials[1] = R.Lambertian(R.Srgb(0.6, 0.3, 0.2))
    elif kind == "no_lights":
        world.lights = []
    elif kind == "spheres_only":
        del world.hitables[1]
    elif kind == "two_sdfs":
        # a second TracedSDF (sphere SDF) after the MandelBox + one before it: exercises the multi-SDF march paths
        world.hitables.insert(1, R.TracedSDF(R.SphereSDF(0.35), 1))
        world.hitables.push(R.TracedSDF(R.MandelBox(6, R.BoxFold(1.0), R.SphereFold(0.5, 1.0), -2.0), 1))
    elif kind == "offset_sdf":  # EXTENSION: TracedSDF with a constant origin (the SDF's frame is translated)
        world.hitables[1].transform_seq = R.vec3(0.35, -0.2, 0.15)
    elif kind == "moving_sdf":  # EXTENSION: TracedSDF origin as the closure |t| base + vel*t (motion-blurred fractal), volume on
        cam_h, world = S.setup(res, volumes=True, sdf="mandelbox")
        world.hitables[1].transform_seq = R.Linear(R.vec3(0.1, 0.0, -0.05), R.vec3(-4.0, 3.0, 2.0))
    elif kind == "moving_two_sdfs":  # both SDFs of a multi-SDF scene move differently (generic march kernels)
        world.hitables.insert(1, R.TracedSDF(R.SphereSDF(0.35), 1, R.Linear(R.vec3(1.4, 0.9, 0.6), R.vec3(0.0, -6.0, 0.0))))
        world.hitables[2].transform_seq = R.Linear(R.vec3(0.0, 0.0, 0.0), R.vec3(3.0, 0.0, -2.0))
    elif kind == "moving_bulb":
        cam_h, world = S.setup_bulb(res)
        world.hitables[1].transform_seq = R.Linear(R.vec3(0.0, 0.05, 0.0), R.vec3(2.0, -1.0, 0.5))
    elif kind == "morphing_box":  # EXTENSION: MandelBox scale as the closure |t| scale + scale_vel*t (single-SDF fast-path kernels), volume on
        cam_h, world = S.setup(res, volumes=True, sdf="mandelbox")
        world.hitables[1].sdf.scale_vel = 4.0
    elif kind == "morphing_two_sdfs":  # ... in a multi-SDF scene (generic march kernels), together with a moving origin
        world.hitables.push(R.TracedSDF(R.MandelBox(7, R.BoxFold(1.0), R.SphereFold(0.5, 1.0), -2.0, scale_vel=-3.0), 1,
                                        R.Linear(R.vec3(0.3, 0.0, 0.0), R.vec3(0.0, 2.0, 0.0))))
        world.hitables[1].sdf.scale_vel = 2.5
    elif kind == "full_house":  # RAYN_MAX_HITABLES = 16 objects: every class of the bin histogram / scan / scatter is populated (ids 0..15)
        mats = [world.materials.add_material(R.Lambertian(R.Srgb(0.8, 0.3, 0.2))), world.materials.add_material(R.Dielectric.new_remap(R.Srgb(0.3, 0.6, 0.3), 0.4)),
                world.materials.add_material(R.Emissive.new_splat(R.Srgb(2.0, 1.5, 1.0)))]
        k = 0
        while len(world.hitables) < 16:
            a = 0.7 * k
            world.hitables.push(R.Sphere(R.vec3(1.9 * np.cos(a), -1.1 + 0.25 * k, 1.9 * np.sin(a)), 0.22 + 0.02 * k, mats[k % 3]))
            k += 1
    elif kind == "lambert_sdf_sphere":
        world.hitables[1] = R.TracedSDF(R.SphereSDF(1.0), world.materials.add_material(R.Lambertian(R.Srgb(0.5, 0.5, 0.5))))
    elif kind == "huge_lights":
        # r6 (zero-throughput NEE elision, k_shade_setup): one light of absurd power and size next to an over-bright dielectric, volume on.  Close to that light
        # pdf < 1 and x = Le * f * tr lies within a factor 1/pdf of FLT_MAX, so (x * occluded) / pdf is +inf for a VISIBLE sample and 0 for an occluded one;
        # on a segment whose throughput is exactly 0 the reference then adds inf * 0 = NaN (src/integrator.rs:91-92) or nothing - the march outcome reaches
        # the film, the sample fails the elision's bounds (|x| <= 2^60) and must be tested and marched like any other.
        cam_h, world = S.setup(res, volumes=True, sdf="mandelbox")
        world.lights[1].emission = R.Srgb(3.3e38, 1e38, 3e37)
        world.lights[1].rad = 0.9
        world.materials[1] = R.Dielectric.new_remap(R.Srgb(3.0, 3.0, 3.0), 0.6)
    else:
        raise ValueError(kind)
    return cam_h, world


def _custom_world(kind, res):
    cam_h, world = _custom_scene(kind, res)
    return world.to_desc(cam_h)


@pytest.mark.parametrize("kind", ["thinlens", "thinlens_volume", "ortho", "anim_pinhole", "anim_thinlens", "anim_spheres", "lambertian", "no_lights", "spheres_only", "two_sdfs", "lambert_sdf_sphere",
                                  "offset_sdf", "moving_sdf", "moving_two_sdfs", "moving_bulb", "morphing_box", "morphing_two_sdfs", "full_house"])
def test_closed_set_parity(gpu_ctx, oracle, kind):
    from rayn_amd import params as P
    w, h, samples, bounces = 40, 32, 2, 4
    wd = _custom_world(kind, (w, h))
    p = P.frame_params(w, h, samples, bounces)
    tabs = _tables(oracle, p)
    ref, ctr = oracle.render(wd, p, tabs)
    gpu_ctx.upload_world(wd)
    out = gpu_ctx.render_host(p, tabs)
    st = gpu_ctx.stats()
    assert st["paths"] == ctr.paths and st["segments"] == ctr.segments
    assert film_l2(out, ref) < L2_TOL
    assert film_equal_bits(out, ref)
    assert np.abs(ref["color"]).sum() + np.abs(ref["background"]).sum() > 0


def test_zero_throughput_elision_counts_and_fallback(gpu_ctx, oracle):
    """r6: k_shade_setup parks no shadow segment for a NEE sample whose weight is an exact zero (throughput (0, 0, 0), DielectricBSDF::scatter's zeroed
    specular lobe, src/material.rs:240-243; every NEE term is multiplied by it, src/integrator.rs:91-92,128-129).  (1) On the shipped scene the elision is
    active (counted by the instrumented kernels), never needs its fallback, and the film keeps every bit.  (2) "huge_lights" forces the fallback: samples
    whose contribution is not provably finite are marched as ever, because inf * 0 = NaN must reach the film exactly where the reference produces it."""
    from rayn_amd import params as P
    wd, p = case("s2", 40, 24, 4, 8)
    tabs = _tables(oracle, p)
    ref, ctr = oracle.render(wd, p, tabs)
    gpu_ctx.upload_world(wd)
    gpu_ctx.set_profiling(False, True)
    try:
        out = gpu_ctx.render_host(p, tabs)
        el, st = gpu_ctx.elision_counts(), gpu_ctx.stats()
        assert film_equal_bits(out, ref) and st["segments"] == ctr.segments
        assert 0 < el["zero_throughput_slots"] < st["segments"] and el["elided_shadow_jobs"] > 0 and el["samples_out_of_bounds"] == 0, el
        gpu_ctx.set_profiling(False, False)
        jobs_counted = st["shadow_jobs"]
        out = gpu_ctx.render_host(p, tabs)  # the product kernels: same film, same job count as the instrumented ones
        assert film_equal_bits(out, ref) and gpu_ctx.stats()["shadow_jobs"] == jobs_counted
        w, h = 48, 32
        wd = _custom_world("huge_lights", (w, h))
        p = P.frame_params(w, h, 4, 6)
        tabs = _tables(oracle, p)
        ref, ctr = oracle.render(wd, p, tabs)
        nan_px = int(np.isnan(ref["color"]).any(-1).sum())
        assert nan_px > 0 and np.isfinite(ref["color"]).all(-1).sum() > 100 and np.isinf(ref["color"]).any(-1).sum() > nan_px
        gpu_ctx.upload_world(wd)
        gpu_ctx.set_profiling(False, True)
        out = gpu_ctx.render_host(p, tabs)
        el, st = gpu_ctx.elision_counts(), gpu_ctx.stats()
        assert st["segments"] == ctr.segments
        assert film_equal_bits(out, ref)  # NaNs compare equal whatever their payload (common.bits_equal), everything else bit for bit
        assert np.array_equal(np.isnan(out["color"]), np.isnan(ref["color"]))
        assert el["samples_out_of_bounds"] > 0 and el["zero_throughput_slots"] > 0, el
    finally:
        gpu_ctx.set_profiling(False, False)


def test_box_filter_tables_parity(gpu_ctx, oracle):
    import rayn_amd
    wd, p = case("s0", 32, 32, 2, 2)
    tabs = oracle.build_tables(8, 2, 2, 1, 32, 32, filter_kind=1, filter_radius=0.5)
    mine = rayn_amd.build_tables(8, 2, 2, 1, 32, 32, rayn_amd.BoxFilter(0.5))
    assert all(np.array_equal(a, b) for a, b in zip(tabs, mine))
    ref, _ = oracle.render(wd, p, tabs)
    gpu_ctx.upload_world(wd)
    assert film_equal_bits(gpu_ctx.render_host(p, tabs), ref)


def test_mitchell_filter_film_parity(gpu_ctx, oracle):
    """A parameterised filter end to end: B-spline Mitchell-Netravali (b = 1, c = 0; no negative lobe) pixel jitter."""
    import rayn_amd
    wd, p = case("s1", 32, 32, 2, 2)
    filt = rayn_amd.MitchellNetravaliFilter(2.0, 1.0, 0.0)
    tabs = oracle.build_tables(8, 2, 2, 1, 32, 32, filter_kind=2, filter_radius=2.0, filter_params=(1.0, 0.0))
    mine = rayn_amd.build_tables(8, 2, 2, 1, 32, 32, filt)
    assert all(np.array_equal(a, b) for a, b in zip(tabs, mine))
    ref, _ = oracle.render(wd, p, tabs)
    gpu_ctx.upload_world(wd)
    assert film_equal_bits(gpu_ctx.render_host(p, mine), ref)


def test_film_class_mirror(oracle):
    """The host mirror with the reference's names: Film::new + render_frame_into + channel access."""
    import rayn_amd as R
    from rayn_amd import setup as S
    W, H = 48, 32
    cam, world = S.setup((W, H), volumes=True)
    film = R.Film([R.ChannelKind.Color, R.ChannelKind.Alpha, R.ChannelKind.Background, R.ChannelKind.WorldNormal], (W, H))
    integ = R.PathTracingIntegrator(max_bounces=3, volume_marches=2)
    st = film.render_frame_into(world, cam, integ, R.BlackmanHarrisFilter(1.5), (16, 16), 1, None, 2)
    p = R.frame_params(W, H, 2, 3)
    ref, ctr = oracle.render(world.to_desc(cam), p, _tables(oracle, p))
    assert st["paths"] == ctr.paths
    got = {"color": film.channel(R.ChannelKind.Color), "alpha": film.channel(R.ChannelKind.Alpha),
           "background": film.channel(R.ChannelKind.Background), "normal": film.channel(R.ChannelKind.WorldNormal)}
    assert film_equal_bits(got, ref)
    with pytest.raises(ValueError):
        R.Film([R.ChannelKind.Color, R.ChannelKind.Color], (W, H))


@pytest.mark.parametrize("name,w,h,samples,bounces", [("s1", 48, 32, 2, 4), ("s2", 32, 24, 1, 3), ("s0", 48, 48, 2, 3)])
def test_fma_policy_variant(gpu_ctx, oracle, name, w, h, samples, bounces):
    """mul_add policy 1 (rayn built with +fma): the fused kernels against the fused oracle build, and they
    must differ from the default build (otherwise the switch does nothing)."""
    wd, p = case(name, w, h, samples, bounces)
    tabs = _tables(oracle, p)
    ref1, _ = oracle.render(wd, p, tabs, fma=True)
    ref0, _ = oracle.render(wd, p, tabs, fma=False)
    gpu_ctx.upload_world(wd)
    gpu_ctx.set_fma_policy(1)
    try:
        out1 = gpu_ctx.render_host(p, tabs)
    finally:
        gpu_ctx.set_fma_policy(0)
    assert film_equal_bits(out1, ref1)
    assert not film_equal_bits(ref1, ref0)


def test_full_size_config2(gpu_ctx, oracle):
    """BASELINE configs[1] at FULL size (1920x1080, 256 spp, 8 bounces, 531 M paths): size-independent properties
    (path/segment conservation, alpha + background complementarity, no NaN) and bit-exact agreement with the CPU
    oracle on a spread sample of whole tiles (tiles are independent, so a tile subset is a valid oracle run)."""
    import torch
    import rayn_amd
    wd, p = case("s1", 1920, 1080, 64, 8)
    tabs = rayn_amd.build_tables(256, 8, 2, 1, 1920, 1080)
    gpu_ctx.upload_world(wd)
    d_tabs = [torch.from_numpy(t).cuda() for t in tabs]
    film = rayn_amd.film.alloc_device_film(1920, 1080, "cuda:0")
    gpu_ctx.render_device(p, d_tabs, film)
    torch.cuda.synchronize()
    st = gpu_ctx.stats()
    assert st["paths"] == 1920 * 1080 * 256 and st["tiles"] == 8160
    assert st["paths"] <= st["segments"] <= 9 * st["paths"]
    out = {"color": film["color"].cpu().numpy().reshape(1080, 1920, 3), "alpha": film["alpha"].cpu().numpy().reshape(1080, 1920),
           "background": film["background"].cpu().numpy().reshape(1080, 1920, 3), "normal": film["normal"].cpu().numpy().reshape(1080, 1920, 3)}
    assert not np.isnan(out["color"]).any() and not np.isnan(out["background"]).any()
    assert out["alpha"].min() >= 0.0 and out["alpha"].max() <= 1.0
    # every camera path ends in exactly one of Background (sky at depth 0) or Alpha=1: alpha + P(background) == 1 per pixel
    n_tiles = 8160
    subset = np.unique(np.linspace(0, n_tiles - 1, 24).astype(np.uint32))
    ref, ctr = oracle.render(wd, p, tabs, tile_subset=subset)
    assert ctr.paths == len(subset) * 65536 or ctr.paths < len(subset) * 65536  # bottom tile row is half height
    for k in subset:
        tx, ty = int(k) // 68, int(k) % 68
        sl = (slice(ty * 16, min(ty * 16 + 16, 1080)), slice(tx * 16, tx * 16 + 16))
        for ch in ("color", "alpha", "background", "normal"):
            assert bits_equal(out[ch][sl], ref[ch][sl]), (k, ch)
    # run-to-run determinism of the whole frame
    film2 = rayn_amd.film.alloc_device_film(1920, 1080, "cuda:0")
    gpu_ctx.render_device(p, d_tabs, film2)
    torch.cuda.synchronize()
    for ch in ("color", "alpha", "background", "normal"):
        assert torch.equal(film[ch].view(torch.int32), film2[ch].view(torch.int32)), ch


def test_full_size_config3(gpu_ctx):
    """BASELINE configs[2] (the bench default) at FULL size: 1920x1080, 1024 spp, 8 bounces, volume = 2.12 G paths in 16
    tile batches.  Oracle parity: the whole 16x16 tiles of tests/golden/config_digests.json (CPU-oracle SHA-256 per tile
    and channel, incl. fractal-heavy, sky and half-height tiles) are hashed out of the full frame and must match.  Plus the
    size-independent properties: path conservation, finite film, and partition independence - one rank's 1/8 share of the
    tiles (different batching, one worker) must reproduce the full frame's pixels bit for bit."""
    import json
    import os
    import sys
    import torch
    import rayn_amd
    from rayn_amd.distributed import owned_pixels
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_config_digests as G
    W, H = 1920, 1080
    wd, p = case("s2", W, H, 256, 8)
    tabs = rayn_amd.build_tables(1024, 8, p.volume_marches, p.frame, W, H)
    gpu_ctx.upload_world(wd)
    d_tabs = [torch.from_numpy(t).cuda() for t in tabs]
    film = rayn_amd.film.alloc_device_film(W, H, "cuda:0")
    gpu_ctx.render_device(p, d_tabs, film)
    torch.cuda.synchronize()
    st = gpu_ctx.stats()
    assert st["paths"] == W * H * 1024 and st["tiles"] == 8160 and st["batches"] >= 2
    assert st["paths"] <= st["segments"] <= 9 * st["paths"]
    for ch in ("color", "background", "normal", "alpha"):
        assert bool(torch.isfinite(film[ch]).all()), ch
    assert float(film["alpha"].min()) >= 0.0 and float(film["alpha"].max()) <= 1.0
    host = {"color": film["color"].cpu().numpy().reshape(H, W, 3), "alpha": film["alpha"].cpu().numpy().reshape(H, W),
            "background": film["background"].cpu().numpy().reshape(H, W, 3), "normal": film["normal"].cpu().numpy().reshape(H, W, 3)}
    cfg = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_digests.json")))["c3"]
    for t in cfg["tiles"]:
        assert G.tile_digests(host, p, t["tile"]) == t["sha256"], t["tile"]
    wd2, p8 = case("s2", W, H, 256, 8, tile_first=3, tile_step=8)
    share = rayn_amd.film.alloc_device_film(W, H, "cuda:0")
    gpu_ctx.render_device(p8, d_tabs, share)
    torch.cuda.synchronize()
    idx = torch.from_numpy(owned_pixels(W, H, p.tile_w, p.tile_h, 3, 8)).cuda()
    assert 0.11 < idx.numel() / (W * H) < 0.14
    for ch in ("color", "alpha", "background", "normal"):
        assert torch.equal(film[ch][idx].view(torch.int32), share[ch][idx].view(torch.int32)), ch


@pytest.mark.parametrize("seed", list(range(12)))
def test_randomised_scene_parity(gpu_ctx, oracle, seed):
    """Seeded random variations of the shipped scene: camera, MandelBox parameters (scale, box side, fold radii, iteration
    count - every combination re-runs the device check behind the 4-instruction fold division), light placement, volume
    coefficients, frame seed, shutter interval, spp and bounce count.  Film bit-identical to the oracle each time."""
    import rayn_amd as R
    from rayn_amd import setup as S
    from rayn_amd import params as P
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(17, 49)), int(rng.integers(9, 33))
    volumes = bool(rng.integers(0, 2))
    cam_h, world = S.setup((w, h), volumes=volumes, sdf="mandelbox")
    box = world.hitables[1].sdf
    box.iterations = int(rng.integers(5, 15))
    box.scale = float(np.float32(rng.uniform(-2.6, -1.7) if rng.integers(0, 2) else rng.uniform(1.8, 2.8)))
    box.box_fold.side_length = float(np.float32(rng.uniform(0.7, 1.3)))
    box.sphere_fold.min_radius = float(np.float32(rng.uniform(0.005, 0.6)))
    box.sphere_fold.fixed_radius = float(np.float32(rng.uniform(0.8, 2.2)))
    if volumes:
        world.volume_params = R.VolumeParams(float(np.float32(rng.uniform(0.05, 0.6))), float(np.float32(rng.uniform(0.01, 0.2))))
    for L in world.lights:
        L.pos = L.pos + rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    cam = world.cameras.get(cam_h)
    cam.origin = (cam.origin * np.float32(rng.uniform(0.6, 1.4)) + rng.uniform(-0.5, 0.5, 3).astype(np.float32)).astype(np.float32)
    if rng.integers(0, 3) == 0:
        world.hitables[1].transform_seq = R.Linear(rng.uniform(-0.2, 0.2, 3).astype(np.float32), rng.uniform(-3, 3, 3).astype(np.float32))
    if rng.integers(0, 3) == 0:
        box.scale_vel = float(np.float32(rng.uniform(-3, 3)))
    wd = world.to_desc(cam_h)
    samples, bounces = int(rng.integers(1, 4)), int(rng.integers(1, 7))
    t0 = float(np.float32(rng.uniform(0.0, 2.0)))
    p = P.frame_params(w, h, samples, bounces, frame=int(rng.integers(1, 50)), time_range=(t0, float(np.float32(t0 + rng.uniform(0.01, 0.2)))),
                       tile_size=(int(rng.choice([4, 8, 16])), int(rng.choice([4, 8, 16]))))
    tabs = _tables(oracle, p)
    ref, ctr = oracle.render(wd, p, tabs)
    gpu_ctx.upload_world(wd)
    out = gpu_ctx.render_host(p, tabs)
    st = gpu_ctx.stats()
    assert st["paths"] == ctr.paths and st["segments"] == ctr.segments
    assert film_equal_bits(out, ref), f"seed {seed}: L2 {film_l2(out, ref)}"
