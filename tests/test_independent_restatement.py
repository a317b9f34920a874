"""A SECOND, independent restatement of the innermost reference functions - numpy binary32, written from the Rust text alone
(no code shared with oracle/rayn_oracle.cpp, include/rayn_detmath.h or the kernels) - compared bit for bit with the C++ oracle.

The oracle is unpinned against rayn itself (no Rust toolchain).  What this file adds: oracle and kernels share
two headers, so a transcription slip in a shared piece could move both sides together; these functions share nothing, so the
operand order of MandelBox::dist, BoxFold / SphereFold, TracedSDF::{hit, occluded}, Sphere::{hit, occluded}, the depth-dependent
hit threshold and the closest-hit fold of HitableStore::add_hits are now stated twice and must agree in every bit.  Third-party
semantics (oracle assumptions A1, A2, A4: unfused mul_add, SSE max/min, ultraviolet's dot / mag forms) are taken as the oracle
states them - this checks the transcription, not the assumptions.  unfused policy only (numpy has no fused multiply-add)."""
import math

import numpy as np
import pytest

from common import case

f32 = np.float32


def sse_max(a, b):  # wide f32x4::max = maxps: a > b ? a : b (b when unordered), oracle assumption A2
    return np.where(a > b, a, b).astype(f32)


def sse_min(a, b):  # minps: a < b ? a : b
    return np.where(a < b, a, b).astype(f32)


def mag_sq(v):  # ultraviolet (A4): x.mul_add(x, y.mul_add(y, z * z)), unfused
    return (v[0] * v[0] + (v[1] * v[1] + v[2] * v[2])).astype(f32)


def dot(a, b):  # x.mul_add(ox, y.mul_add(oy, z * oz))
    return (a[0] * b[0] + (a[1] * b[1] + a[2] * b[2])).astype(f32)


def mandelbox_dist(p0, iterations, side, min_radius, fixed_radius, scale):
    """MandelBox::dist (src/sdf.rs:125-140) with BoxFold::box_fold (:159-162) and SphereFold::sphere_fold (:181-187).  p0: [3][n]"""
    l, mrs, frs, s = f32(side), f32(f32(min_radius) * f32(min_radius)), f32(f32(fixed_radius) * f32(fixed_radius)), f32(scale)
    p = [c.copy() for c in p0]
    dr = np.ones_like(p0[0])
    for _ in range(iterations):
        for c in range(3):  # point.clamped(neg_l, l).mul_add(two, -point): clamped = max(min).min(max) per component
            cl = sse_min(sse_max(p[c], -l), l)
            p[c] = (cl * f32(2.0) + (-p[c])).astype(f32)
        r2 = mag_sq(p)
        mul = sse_max(np.ones_like(r2), (frs / sse_max(np.full_like(r2, mrs), r2)).astype(f32))  # ONE.max(fixed / min_rad_sq.max(r2))
        for c in range(3):
            p[c] = (p[c] * mul).astype(f32)
        dr = (dr * mul).astype(f32)
        for c in range(3):  # p.mul_add(scale_vec, offset)
            p[c] = (p[c] * s + p0[c]).astype(f32)
        dr = ((-dr) * s + f32(1.0)).astype(f32)  # (-dr).mul_add(scale, one)
    return (np.sqrt(mag_sq(p)) / np.abs(dr)).astype(f32)


def box_params(h):
    return dict(iterations=int(h.iterations), side=h.box_side, min_radius=h.min_radius, fixed_radius=h.fixed_radius, scale=h.scale)


def rand(n, lo, hi, seed):
    return np.random.default_rng(seed).uniform(lo, hi, n).astype(f32)


def test_mandelbox_dist_second_restatement(oracle):
    wd, _ = case("s1", 64, 64, 1, 1)
    pts = rand(3 * 200000, -3.5, 3.5, 71).reshape(-1, 3)
    pts[:2000] *= f32(30.0)           # far field
    pts[2000:4000] *= f32(1e-3)       # deep inside the minimum-radius ball of the sphere fold
    pts[4000] = 0.0
    ref = oracle.sdf_dist(wd.hitables[1], pts)
    with np.errstate(all="ignore"):
        got = mandelbox_dist([pts[:, 0].copy(), pts[:, 1].copy(), pts[:, 2].copy()], **box_params(wd.hitables[1]))
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), (np.flatnonzero(~same)[:5], got[~same][:5], ref[~same][:5])
    # other fold parameters / iteration counts (the fuzz generator's ranges)
    rng = np.random.default_rng(5)
    for _ in range(6):
        h = type(wd.hitables[1]).from_buffer_copy(wd.hitables[1])
        h.iterations = int(rng.integers(4, 16)); h.scale = float(f32(rng.uniform(-2.8, 3.0))); h.box_side = float(f32(rng.uniform(0.6, 1.5)))
        h.min_radius = float(f32(rng.uniform(0.002, 0.7))); h.fixed_radius = float(f32(rng.uniform(0.75, 2.5)))
        sub = pts[:30000]
        with np.errstate(all="ignore"):
            got = mandelbox_dist([sub[:, 0].copy(), sub[:, 1].copy(), sub[:, 2].copy()], **box_params(h))
        ref = oracle.sdf_dist(h, sub)
        assert ((got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))).all()


# ---- HitableStore::add_hits (src/hitable.rs:170-210) over Sphere::hit (src/sphere.rs:48-72) and TracedSDF::hit (src/sdf.rs:59-83) ----
def sphere_hit(center, radius, o, d, t_max):
    oc = [o[c] - f32(center[c]) for c in range(3)]
    b = dot(oc, d)
    cq = (mag_sq(oc) - f32(f32(radius) * f32(radius))).astype(f32)
    descrim = (b * b - cq).astype(f32)
    pos = descrim > 0
    with np.errstate(invalid="ignore"):
        ds = np.sqrt(descrim).astype(f32)
    t1, t2 = ((-b) - ds).astype(f32), ((-b) + ds).astype(f32)
    t1v = (t1 > f32(0.0001)) & (t1 <= t_max) & pos
    t2v = (t2 > f32(0.0001)) & (t2 <= t_max) & pos
    t = np.where((t1 < t2) & t1v, t1, t2)
    return np.where(t1v | t2v, t, np.finfo(f32).max).astype(f32)


def traced_sdf_hit(h, o, d, t_max, thr_at, detail_scale, max_marches=256):
    def dist(p):
        with np.errstate(all="ignore"):
            return mandelbox_dist(p, **box_params(h))
    t = dist(o)
    nan = np.isnan(t)
    done = np.zeros(len(t), bool)
    for _ in range(max_marches):
        pt = [(d[c] * t + o[c]).astype(f32) for c in range(3)]  # Ray::point_at: dir.mul_add(t, origin)
        di = dist(pt)
        hit = np.abs(di) < sse_max(np.full_like(t, f32(0.00005) * f32(detail_scale)), (f32(f32(0.05) * f32(detail_scale)) * thr_at(t)).astype(f32))
        stop = hit | nan | (t > t_max)
        # the reference re-evaluates stopped lanes at the same t and gets the same mask: freezing them is result-identical
        t = np.where(stop | done, t, (t + di).astype(f32))
        done |= stop
        if done.all():
            break
    return t


def scene_closest_hit(wd, p, depth, o, d):
    cam = wd.camera
    theta = f32(f32(cam.vfov_or_size) * f32(math.pi) / f32(180.0))
    half_pixel = f32(f32(math.tan(float(f32(theta / f32(2.0))))) / f32(cam.res_h))  # PinholeCamera::new, src/camera.rs:60-65
    if depth == 0:
        thr_at = lambda t: (half_pixel * t).astype(f32)                      # camera.half_pixel_size_at, src/film.rs:539-543
    else:
        k = f32(f32(f32(0.0001) * f32(2.0)) * f32(depth))                    # 0.0001 * 2.0 * depth as f32, src/film.rs:545-549
        thr_at = lambda t: (k * t).astype(f32)
    closest = np.full(len(o[0]), f32(p.world_radius * 2.0), f32)
    ids = np.full(len(o[0]), 0xFFFFFFFF, np.uint32)
    for i in range(wd.n_hitables):
        h = wd.hitables[i]
        if h.kind == 0:
            t = sphere_hit((h.center.x, h.center.y, h.center.z), h.radius, o, d, closest)
        else:
            t = traced_sdf_hit(h, o, d, closest, thr_at, p.sdf_detail_scale)
        win = t < closest
        closest = np.where(win, t, closest)
        ids = np.where(win, np.uint32(i), ids)
    return closest, ids


@pytest.mark.parametrize("depth", [0, 3])
def test_closest_hit_second_restatement(oracle, depth):
    wd, p = case("s1", 96, 64, 1, 4)
    n = 6000
    org = rand(3 * n, -3.0, 3.0, 11).reshape(-1, 3)
    org[: n // 2] = np.array([-1.0125, 0.45, 4.5], f32)  # the shipped camera position, src/setup.rs:132
    dirs = rand(3 * n, -1.0, 1.0, 12).reshape(-1, 3)
    dirs[: n // 2] = -org[: n // 2] + rand(3 * (n // 2), -1.5, 1.5, 13).reshape(-1, 3)
    dirs = (dirs / np.linalg.norm(dirs.astype(np.float64), axis=1, keepdims=True)).astype(f32)
    t_ref, obj_ref = oracle.closest_hit(wd, p, depth, org, dirs)
    t, ids = scene_closest_hit(wd, p, depth, [org[:, c].copy() for c in range(3)], [dirs[:, c].copy() for c in range(3)])
    hit = ids != 0xFFFFFFFF
    assert hit.sum() > n // 2 and (ids[hit] == 1).sum() > 500  # plenty of rays end on the fractal
    assert np.array_equal(ids[hit], obj_ref[hit]) and (obj_ref[~hit] >= wd.n_hitables).all()
    assert np.array_equal(t[hit].view(np.uint32), t_ref[hit].view(np.uint32))


# ---- HitableStore::test_occluded (src/hitable.rs:164-168) over Sphere::occluded (src/sphere.rs:24-46) and TracedSDF::occluded (src/sdf.rs:25-57) ----
def sphere_occluded(center, radius, a, b):
    dirv = [(b[c] - a[c]).astype(f32) for c in range(3)]
    dist = np.sqrt(mag_sq(dirv)).astype(f32)
    dirv = [(dirv[c] / dist).astype(f32) for c in range(3)]
    oc = [a[c] - f32(center[c]) for c in range(3)]
    bq = dot(oc, dirv)
    cq = (mag_sq(oc) - f32(f32(radius) * f32(radius))).astype(f32)
    descrim = (bq * bq - cq).astype(f32)
    with np.errstate(invalid="ignore"):
        ds = np.sqrt(descrim).astype(f32)
    t1, t2 = ((-bq) - ds).astype(f32), ((-bq) + ds).astype(f32)
    valid = (sse_min(t1, t2) > f32(0.001)) & (t1 <= dist) & (descrim > 0)
    return np.where(valid, f32(0.0), f32(1.0)).astype(f32)


def traced_sdf_occluded(h, a, b, detail_scale, max_vis=100):
    def dist_fn(p):
        with np.errstate(all="ignore"):
            return mandelbox_dist(p, **box_params(h))
    dirv = [(b[c] - a[c]).astype(f32) for c in range(3)]
    max_dist = np.sqrt(mag_sq(dirv)).astype(f32)
    dirv = [(dirv[c] / max_dist).astype(f32) for c in range(3)]
    dist = dist_fn(a)
    nan = np.isnan(dist)
    gt_nan = (dist > max_dist) | nan
    hit = dist < f32(0.0001)
    t = dist.copy()
    frozen = np.zeros(len(t), bool)
    for _ in range(max_vis):
        gt_nan_new = (t > max_dist) | nan
        gt_nan = np.where(frozen, gt_nan, gt_nan_new)
        if (gt_nan | frozen).all():
            break
        pt = [(dirv[c] * t + a[c]).astype(f32) for c in range(3)]
        di = dist_fn(pt)
        hit_new = np.abs(di) < sse_max(np.full_like(t, f32(0.0001) * f32(detail_scale)), (f32(f32(0.00001) * f32(detail_scale)) * t).astype(f32))
        hit = np.where(frozen, hit, hit_new)
        stop = hit | gt_nan
        t = np.where(stop | frozen, t, (t + di).astype(f32))
        frozen |= stop  # a stopped lane re-evaluates at the same t in the reference: same masks every further trip
        if frozen.all():
            break
    return np.where(hit & ~gt_nan, f32(0.0), f32(1.0)).astype(f32)


def test_occluded_second_restatement(oracle):
    wd, p = case("s1", 96, 64, 1, 4)
    n = 5000
    start = rand(3 * n, -2.2, 2.2, 21).reshape(-1, 3)
    lights = np.array([[wd.lights[i].pos.x, wd.lights[i].pos.y, wd.lights[i].pos.z] for i in range(wd.n_lights)], f32)
    end = (lights[np.random.default_rng(22).integers(0, wd.n_lights, n)] + rand(3 * n, -0.15, 0.15, 23).reshape(-1, 3)).astype(f32)
    start[: n // 2] = (end[: n // 2] + rand(3 * (n // 2), -0.7, 0.7, 24).reshape(-1, 3)).astype(f32)  # short segments next to a light: mostly visible
    ref = oracle.test_occluded(wd, p, start, end)
    a, b = [start[:, c].copy() for c in range(3)], [end[:, c].copy() for c in range(3)]
    vis = np.ones(n, f32)
    for i in range(wd.n_hitables):  # fold(ONE, acc * hitable.occluded(..)): every factor is exactly 0 or 1
        h = wd.hitables[i]
        occ = sphere_occluded((h.center.x, h.center.y, h.center.z), h.radius, a, b) if h.kind == 0 else traced_sdf_occluded(h, a, b, p.sdf_detail_scale)
        vis = (vis * occ).astype(f32)
    assert 0.05 < ref.mean() < 0.95  # both outcomes are exercised
    assert np.array_equal(vis.view(np.uint32), ref.view(np.uint32))


# ---- the whole path: ray-gen -> closest hit -> bins / packets -> shading info -> NEE (surface + volume) -> scatter -> roulette ->
# ---- film add order -> normalise, restated a second time in tests/restatement_np.py ---------------------------------------------------
@pytest.mark.parametrize("name,W,H,samples,bounces,kw", [
    ("s2", 32, 16, 2, 4, {}),                       # MandelBox + homogeneous volume, two 16x16 tiles, roulette depth reached
    ("s1", 24, 20, 2, 5, {"tile_size": (8, 8)}),   # volumes off, 3 x 3 tiles of 8 x 8 incl. clipped ones ((20 + 20 % 8) / 8 = 3 rows)
    ("s0", 20, 16, 2, 3, {"tile_size": (16, 16)}),  # BASELINE config 1's scene; 20 columns at tile 16: the reference's grid leaves 4 uncovered
])
def test_whole_film_second_restatement(oracle, name, W, H, samples, bounces, kw):
    """tests/restatement_np.py renders the film with code written from the Rust text alone (numpy binary32; transcendentals = binary64
    numpy functions rounded once): every float of Color / Alpha / Background / WorldNormal and the path / segment counts must equal
    the C++ oracle's.  This pins the oracle's transcription of packet grouping (F7), per-bounce sample indexing, the NEE and BSDF
    operand orders and the film add order against a second reading of the reference; it cannot pin the third-party semantics
    both restatements assume (A1-A9)."""
    import restatement_np as RN
    wd, p = case(name, W, H, samples, bounces, **kw)
    tabs = oracle.build_tables(4 * samples, bounces, p.volume_marches, p.frame, W, H)
    ref, ctr = oracle.render(wd, p, tabs)
    got, c2 = RN.render(wd, p, tabs)
    assert c2["paths"] == ctr.paths and c2["segments"] == ctr.segments
    assert ctr.segments > ctr.paths  # paths do bounce
    for k in ("color", "alpha", "background", "normal"):
        a = got[k].reshape(-1).view(np.uint32)
        b = np.ascontiguousarray(ref[k], np.float32).reshape(-1).view(np.uint32)
        assert np.array_equal(a, b), (k, int((a != b).sum()))
    assert float(ref["color"].max()) > 0 and float(ref["alpha"].max()) > 0


@pytest.mark.parametrize("kind", ["thinlens", "thinlens_volume", "ortho", "anim_pinhole", "anim_thinlens", "anim_spheres", "lambertian", "no_lights", "spheres_only",
                                  "two_sdfs", "lambert_sdf_sphere", "offset_sdf", "moving_sdf", "moving_two_sdfs", "morphing_box", "morphing_two_sdfs"])
def test_closed_set_second_restatement(oracle, kind):
    """The rest of the reference's closed set through the second restatement: ThinLens / Orthographic cameras (src/camera.rs:120-285),
    closure-sequenced camera parameters and sphere centres with their lane-0-time semantics (src/animation.rs:62-68), Lambertian,
    scenes without lights / without an SDF / with two SDFs - same scenes as tests/test_gpu_parity.py::test_closed_set_parity renders on
    the GPU, so kernels, oracle and this restatement all produce the same bits.  The last five are this repo's EXTENSIONS (a TracedSDF with a
    constant / closure origin, the MandelBox scale as a closure: include/rayn_hip.h) - there the second restatement follows the header's
    description, not the reference."""
    import restatement_np as RN
    from rayn_amd import params as P
    from test_gpu_parity import _custom_world
    W, H, samples, bounces = 16, 16, 1, 4
    wd = _custom_world(kind, (W, H))
    p = P.frame_params(W, H, samples, bounces, time_range=(0.1, 0.3))
    tabs = oracle.build_tables(4 * samples, bounces, p.volume_marches, p.frame, W, H)
    ref, ctr = oracle.render(wd, p, tabs)
    got, c2 = RN.render(wd, p, tabs)
    assert c2["paths"] == ctr.paths and c2["segments"] == ctr.segments
    for k in ("color", "alpha", "background", "normal"):
        a = got[k].reshape(-1).view(np.uint32)
        b = np.ascontiguousarray(ref[k], np.float32).reshape(-1).view(np.uint32)
        assert np.array_equal(a, b), (k, int((a != b).sum()))


@pytest.mark.parametrize("kind,radius,params", [(0, 1.5, (0.0, 0.0)), (1, 0.5, (0.0, 0.0)), (2, 2.0, (1.0 / 3.0, 1.0 / 3.0)), (2, 2.0, (1.0, 0.0)), (3, 3.0, (3.0, 0.0))])
def test_filter_importance_sampler_second_restatement(oracle, kind, radius, params):
    """FilterImportanceSampler::new + CDF over BlackmanHarris / Box / MitchellNetravali / LanczosSinc restated a second time (scalar
    binary32, cos / sin from binary64): the 512-entry inverse CDF equals the oracle's and the product's host builder bit for bit."""
    import rayn_amd
    import restatement_np as RN
    import ctypes as C
    from rayn_amd._lib import lib
    mine = RN.fis_table(kind, radius, *params)
    ref = oracle.build_tables(4, 0, 2, 1, 4, 4, filter_kind=kind, filter_radius=radius, filter_params=params)[3]
    assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32)), int((mine.view(np.uint32) != ref.view(np.uint32)).sum())
    prod = np.zeros(512, np.float32)
    assert lib().rayn_build_fis_table_ex(kind, C.c_float(radius), C.c_float(params[0]), C.c_float(params[1]), prod.ctypes.data_as(C.POINTER(C.c_float))) == 0
    assert np.array_equal(mine.view(np.uint32), prod.view(np.uint32))
