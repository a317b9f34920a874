"""A second, independent restatement of rayn's per-sample integrator hot path for the shipped scene types - numpy binary32, tile by tile,
vectorised over the lanes of a depth - used by tests/test_independent_restatement.py to check the C++ oracle's TRANSCRIPTION of the
reference (operand order, packet grouping, sample indexing, film add order) with code that shares nothing with it.

Written from the Rust text (citations are to /root/reference/src).  Third-party semantics are the oracle's documented assumptions
A1-A8 (oracle/rayn_oracle.cpp header): unfused mul_add, SSE max/min, ultraviolet's dot / normalized / cross / reflected / Mat3*Vec3
forms, sdfu's tetrahedral normals_fast and lerp, powi(5) = (x^2)^2 * x, correctly rounded transcendentals (here: binary64 numpy
functions rounded once to binary32 - identical except on near-ties, ~1e-8 per call).  Closed set: PinholeCamera with constant
parameters, Sphere with a constant centre, TracedSDF<MandelBox>, Dielectric / Lambertian / Sky / Emissive, SphereLight, optional
homogeneous volume, unfused policy.  TEST INFRASTRUCTURE: nothing under rayn_amd/ imports this."""
import math

import numpy as np

f32 = np.float32
PI = f32(math.pi)
TWO_PI = f32(2.0 * math.pi)
FRAC_PI_2 = f32(math.pi / 2.0)
FRAC_PI_4 = f32(math.pi / 4.0)
F32_MAX = np.finfo(f32).max
F32_EPS = np.finfo(f32).eps


# ---- f32x4 / Wec3 primitives (assumptions A1, A2, A4, A5, A8) ------------------------------------------------------------
def smax(a, b):  # a.max(b) = maxps(a, b): a > b ? a : b
    return np.where(a > b, a, b).astype(f32)


def smin(a, b):
    return np.where(a < b, a, b).astype(f32)


def mul_add(a, b, c):  # A1: unfused
    return (a * b + c).astype(f32)


def dot(a, b):
    return mul_add(a[0], b[0], mul_add(a[1], b[1], (a[2] * b[2]).astype(f32)))


def mag_sq(a):
    return dot(a, a)


def mag(a):
    return np.sqrt(mag_sq(a)).astype(f32)


def vsub(a, b):
    return [(a[c] - b[c]).astype(f32) for c in range(3)]


def vadd(a, b):
    return [(a[c] + b[c]).astype(f32) for c in range(3)]


def vscale(a, s):
    return [(a[c] * s).astype(f32) for c in range(3)]


def vmul(a, b):
    return [(a[c] * b[c]).astype(f32) for c in range(3)]


def vdiv(a, s):
    return [(a[c] / s).astype(f32) for c in range(3)]


def vneg(a):
    return [(-a[c]).astype(f32) for c in range(3)]


def normalized(a):
    r = (f32(1.0) / mag(a)).astype(f32)
    return vscale(a, r)


def cross(a, b):
    return [mul_add(a[1], b[2], (-(a[2] * b[1])).astype(f32)), mul_add(a[2], b[0], (-(a[0] * b[2])).astype(f32)), mul_add(a[0], b[1], (-(a[1] * b[0])).astype(f32))]


def signum(x):  # A8: +-1 for +-0, NaN stays NaN
    return np.where(np.isnan(x), x, np.copysign(f32(1.0), x)).astype(f32)


def splat(v, n):
    return [np.full(n, f32(v[c]), f32) for c in range(3)]


def vwhere(m, a, b):
    return [np.where(m, a[c], b[c]).astype(f32) for c in range(3)]


def expf(x):
    return np.exp(x.astype(np.float64)).astype(f32)


def sin_cos(x):
    x64 = x.astype(np.float64)
    return np.sin(x64).astype(f32), np.cos(x64).astype(f32)


def tanf(x):
    return np.tan(x.astype(np.float64)).astype(f32)


def atan2f(y, x):
    return np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(f32)


def powf(a, b):
    return np.power(a.astype(np.float64), np.asarray(b, np.float64)).astype(f32)


def powi5(x):
    x2 = (x * x).astype(f32)
    return ((x2 * x2).astype(f32) * x).astype(f32)


def lerp(a, b, t):  # sdfu Lerp (A5)
    return (a * (f32(1.0) - t) + b * t).astype(f32)


def fract(x):  # f32::fract = x - trunc(x)
    return (x - np.trunc(x)).astype(f32)


def onb(nor):  # get_orthonormal_basis, src/math.rs:49-59 -> columns (uu, vv, nor)
    ks = signum(nor[2])
    ka = (f32(1.0) / (f32(1.0) + np.abs(nor[2]))).astype(f32)
    kb = ((((-ks) * nor[0]).astype(f32) * nor[1]).astype(f32) * ka).astype(f32)
    uu = [(f32(1.0) - ((nor[0] * nor[0]).astype(f32) * ka).astype(f32)).astype(f32), (ks * kb).astype(f32), ((-ks) * nor[0]).astype(f32)]
    vv = [kb, (ks - (((nor[1] * nor[1]).astype(f32) * ka).astype(f32) * ks).astype(f32)).astype(f32), (-nor[1]).astype(f32)]
    return [uu, vv, nor]


def mat_vec(m, v):  # Wat3 * Wec3 = c0 * x + c1 * y + c2 * z (A4)
    return vadd(vadd(vscale(m[0], v[0]), vscale(m[1], v[1])), vscale(m[2], v[2]))


# ---- MandelBox (src/sdf.rs:104-188) --------------------------------------------------------------------------------------
def mandelbox_dist(p0, h, t0=None):
    """t0 (lane 0's time of the calling packet) only matters for the EXTENSION `scale = |t| scale + scale_vel * t` (include/rayn_hip.h)"""
    if h.sdf_kind == 0:  # sdfu::Sphere (A5): |p| - r
        with np.errstate(all="ignore"):
            return (mag(p0) - f32(h.sdf_radius)).astype(f32)
    assert h.sdf_kind == 1, "the Mandelbulb extension is not restated here"
    l, mrs, frs, s = f32(h.box_side), f32(f32(h.min_radius) * f32(h.min_radius)), f32(f32(h.fixed_radius) * f32(h.fixed_radius)), f32(h.scale)
    if h.scale_vel != 0.0:
        s = (f32(h.scale) + (f32(h.scale_vel) * t0).astype(f32)).astype(f32)
    p = [c.copy() for c in p0]
    dr = np.ones_like(p0[0])
    with np.errstate(all="ignore"):
        for _ in range(int(h.iterations)):
            for c in range(3):
                cl = smin(smax(p[c], -l), l)
                p[c] = mul_add(cl, f32(2.0), (-p[c]).astype(f32))
            r2 = mag_sq(p)
            m = smax(np.ones_like(r2), (frs / smax(np.full_like(r2, mrs), r2)).astype(f32))
            p = vscale(p, m)
            dr = (dr * m).astype(f32)
            p = [mul_add(p[c], s, p0[c]) for c in range(3)]
            dr = mul_add((-dr).astype(f32), s, f32(1.0))
        return (mag(p) / np.abs(dr)).astype(f32)


# ---- hitables ------------------------------------------------------------------------------------------------------------
def sphere_center(h, t0):
    """WSequenced::sample_at(&self.transform_seq, time): a constant Vec3 clones itself (src/animation.rs:27-51); the closure
    `|t| base + vel * t` is evaluated at LANE 0's time for all four lanes (src/animation.rs:62-68) - t0 = that time per lane."""
    n = len(t0)
    if not h.animated:
        return splat((h.center.x, h.center.y, h.center.z), n)
    return [(f32((h.center.x, h.center.y, h.center.z)[c]) + (f32((h.center_vel.x, h.center_vel.y, h.center_vel.z)[c]) * t0).astype(f32)).astype(f32) for c in range(3)]


def sphere_hit(h, o, d, t_max, t0):  # src/sphere.rs:48-72
    oc = vsub(o, sphere_center(h, t0))
    b = dot(oc, d)
    cq = (mag_sq(oc) - f32(f32(h.radius) * f32(h.radius))).astype(f32)
    descrim = ((b * b).astype(f32) - cq).astype(f32)
    pos = descrim > 0
    with np.errstate(invalid="ignore"):
        ds = np.sqrt(descrim).astype(f32)
    t1, t2 = ((-b) - ds).astype(f32), ((-b) + ds).astype(f32)
    t1v = (t1 > f32(0.0001)) & (t1 <= t_max) & pos
    t2v = (t2 > f32(0.0001)) & (t2 <= t_max) & pos
    t = np.where((t1 < t2) & t1v, t1, t2)
    return np.where(t1v | t2v, t, F32_MAX).astype(f32)


def sphere_occluded(h, a, b, t0):  # src/sphere.rs:24-46: 0 occluded, 1 not
    dirv = vsub(b, a)
    dist = mag(dirv)
    dirv = vdiv(dirv, dist)
    oc = vsub(a, sphere_center(h, t0))
    bq = dot(oc, dirv)
    cq = (mag_sq(oc) - f32(f32(h.radius) * f32(h.radius))).astype(f32)
    descrim = ((bq * bq).astype(f32) - cq).astype(f32)
    with np.errstate(invalid="ignore"):
        ds = np.sqrt(descrim).astype(f32)
    t1, t2 = ((-bq) - ds).astype(f32), ((-bq) + ds).astype(f32)
    valid = (smin(t1, t2) > f32(0.001)) & (t1 <= dist) & (descrim > 0)
    return np.where(valid, f32(0.0), f32(1.0)).astype(f32)


def sdf_hit(h, o, d, t_max, thr_at, ds_, max_marches, t0):  # src/sdf.rs:59-83 (+ EXTENSION: the SDF's frame is translated by its origin)
    o = vsub(o, sphere_center(h, t0))
    t = mandelbox_dist(o, h, t0)
    nan = np.isnan(t)
    done = np.zeros(len(t), bool)
    for _ in range(max_marches):
        pt = [mul_add(d[c], t, o[c]) for c in range(3)]
        di = mandelbox_dist(pt, h, t0)
        hit = np.abs(di) < smax(np.full_like(t, f32(0.00005) * f32(ds_)), (f32(f32(0.05) * f32(ds_)) * thr_at(t)).astype(f32))
        stop = hit | nan | (t > t_max)
        t = np.where(stop | done, t, (t + di).astype(f32))  # a stopped lane is idempotent in the packet loop
        done |= stop
        if done.all():
            break
    return t


def sdf_occluded(h, a, b, ds_, max_vis, t0):  # src/sdf.rs:25-57 (+ EXTENSION: both ends in the SDF's frame)
    org = sphere_center(h, t0)
    a, b = vsub(a, org), vsub(b, org)
    dirv = vsub(b, a)
    max_dist = mag(dirv)
    dirv = vdiv(dirv, max_dist)
    dist = mandelbox_dist(a, h, t0)
    nan = np.isnan(dist)
    gt_nan = (dist > max_dist) | nan
    hit = dist < f32(0.0001)
    t = dist.copy()
    frozen = np.zeros(len(t), bool)
    for _ in range(max_vis):
        gt_nan = np.where(frozen, gt_nan, (t > max_dist) | nan)
        if (gt_nan | frozen).all():
            break
        pt = [mul_add(dirv[c], t, a[c]) for c in range(3)]
        di = mandelbox_dist(pt, h, t0)
        hit_new = np.abs(di) < smax(np.full_like(t, f32(0.0001) * f32(ds_)), (f32(f32(0.00001) * f32(ds_)) * t).astype(f32))
        hit = np.where(frozen | gt_nan, hit, hit_new)
        stop = hit | gt_nan
        t = np.where(stop | frozen, t, (t + di).astype(f32))
        frozen |= stop
        if frozen.all():
            break
    return np.where(hit & ~gt_nan, f32(0.0), f32(1.0)).astype(f32)


def test_occluded(wd, p, a, b, t0):  # HitableStore::test_occluded, src/hitable.rs:164-168; t0 = lane 0's time of the calling packet
    vis = np.ones(len(a[0]), f32)
    with np.errstate(all="ignore"):
        for i in range(wd.n_hitables):
            h = wd.hitables[i]
            occ = sphere_occluded(h, a, b, t0) if h.kind == 0 else sdf_occluded(h, a, b, p.sdf_detail_scale, int(p.max_vis_marches), t0)
            vis = (vis * occ).astype(f32)
    return vis


# ---- lights (src/light.rs) -----------------------------------------------------------------------------------------------
def light_sample(L, u0, u1, pnt):  # SphereLight::sample :38-72 -> (point, pdf); emission is the light's constant
    n = len(u0)
    pos = splat((L.pos.x, L.pos.y, L.pos.z), n)
    rad = f32(L.rad)
    dtl = vsub(pos, pnt)
    d2 = mag_sq(dtl)
    dist = np.sqrt(d2).astype(f32)
    dtl = vdiv(dtl, dist)
    basis = onb(vneg(dtl))
    r2 = f32(rad * rad)
    sin_max2 = (r2 / d2).astype(f32)
    cos_max = np.sqrt(smax(np.zeros(n, f32), (f32(1.0) - sin_max2).astype(f32))).astype(f32)
    cos_t = ((f32(1.0) - u0).astype(f32) + (u0 * cos_max).astype(f32)).astype(f32)
    sin_t = np.sqrt(smax(np.zeros(n, f32), (f32(1.0) - (cos_t * cos_t).astype(f32)).astype(f32))).astype(f32)
    phi = (u1 * TWO_PI).astype(f32)
    ds = ((dist * cos_t).astype(f32) - np.sqrt(smax(np.zeros(n, f32), (r2 - ((d2 * sin_t).astype(f32) * sin_t).astype(f32)).astype(f32))).astype(f32)).astype(f32)
    cos_a = ((((d2 + r2).astype(f32) - (ds * ds).astype(f32)).astype(f32)) / ((f32(2.0) * dist).astype(f32) * rad).astype(f32)).astype(f32)
    sin_a = np.sqrt(smax(np.zeros(n, f32), (f32(1.0) - (cos_a * cos_a).astype(f32)).astype(f32))).astype(f32)
    sin_p, cos_p = sin_cos(phi)
    off = vadd(vadd(vscale(vscale(basis[0], sin_a), cos_p), vscale(vscale(basis[1], sin_a), sin_p)), vscale(basis[2], cos_a))
    point = vadd(pos, vscale(off, rad))
    pdf = (f32(1.0) / (TWO_PI * (f32(1.0) - cos_max).astype(f32)).astype(f32)).astype(f32)  # uniform_cone_pdf :105-107
    return point, pdf


def light_sample_volume(L, sample, ro, rd, max_distance):  # :75-102 -> (sample_dist, pdf)
    n = len(sample)
    pos = splat((L.pos.x, L.pos.y, L.pos.z), n)
    delta = dot(vsub(pos, ro), rd)
    closest = vadd(ro, vscale(rd, delta))  # ray_o + delta * ray_d
    d = mag(vsub(closest, pos))
    theta_a = atan2f((-delta).astype(f32), d)
    theta_b = atan2f((max_distance - delta).astype(f32), d)
    t = (d * tanf(lerp(theta_a, theta_b, sample))).astype(f32)
    sample_dist = (delta + t).astype(f32)
    pdf = (d / ((theta_b - theta_a).astype(f32) * mul_add(d, d, (t * t).astype(f32))).astype(f32)).astype(f32)
    return sample_dist, pdf


# ---- BSDFs (src/material.rs) ---------------------------------------------------------------------------------------------
MAT_LAMBERT, MAT_DIELECTRIC, MAT_SKY, MAT_EMISSIVE = 0, 1, 2, 3


def f_schlick(cos, f0):
    return (f0 + ((f32(1.0) - f0) * powi5((f32(1.0) - cos).astype(f32))).astype(f32)).astype(f32)


def concentric_circle_map(u0, u1):  # src/math.rs:201-219
    a = mul_add(u0, f32(2.0), f32(-1.0))
    b = mul_add(u1, f32(2.0), f32(-1.0))
    zero = (a == 0) & (b == 0)
    b = np.where(zero, f32(0.0001), b).astype(f32)
    with np.errstate(all="ignore"):
        phi1 = ((FRAC_PI_4 * b).astype(f32) / a).astype(f32)
        phi2 = mul_add(((-FRAC_PI_4) / b).astype(f32), a, FRAC_PI_2)
    mask = (a * a).astype(f32) > (b * b).astype(f32)
    r = np.where(mask, a, b).astype(f32)
    phi = np.where(mask, phi1, phi2).astype(f32)
    s, c = sin_cos(phi)
    return (r * c).astype(f32), (r * s).astype(f32)


def cosine_weighted_in_hemisphere(u0, u1):  # :99-103
    x, y = concentric_circle_map(u0, u1)
    m2 = mul_add(x, x, (y * y).astype(f32))  # Wec2::mag_sq (A4 form for two components)
    z = np.sqrt((f32(1.0) - smin(m2, np.ones_like(m2))).astype(f32)).astype(f32)
    return [x, y, z]


def cosine_power_weighted(u0, u1, power):  # :106-113 - the azimuth is 2 * u radians
    a = powf(u0, (f32(1.0) / (power + f32(1.0))).astype(f32))
    a2 = (a * a).astype(f32)
    b = np.sqrt((f32(1.0) - a2).astype(f32)).astype(f32)
    s, c = sin_cos((f32(2.0) * u1).astype(f32))
    return [(b * c).astype(f32), (b * s).astype(f32), a]


def bsdf_le(m, wo, n):
    if m.kind == MAT_SKY:  # :441-447
        t = (f32(0.5) * (wo[1] + f32(1.0)).astype(f32)).astype(f32)
        top, bot = (m.a.x, m.a.y, m.a.z), (m.b.x, m.b.y, m.b.z)
        return [((f32(top[c]) * (f32(1.0) - t).astype(f32)).astype(f32) + (f32(bot[c]) * t).astype(f32)).astype(f32) for c in range(3)]
    if m.kind == MAT_EMISSIVE:
        return splat((m.a.x, m.a.y, m.a.z), n)
    return splat((0.0, 0.0, 0.0), n)


def bsdf_f(m, wi, wo, nrm):
    """BSDF::f(wi, wo, n) with the argument ORDER of the trait (src/material.rs:25); the integrator calls bsdf.f(wo, wi, normal),
    i.e. the parameter named wi receives wo (src/integrator.rs:229)."""
    n = len(wi[0])
    alb = (m.a.x, m.a.y, m.a.z)
    if m.kind == MAT_LAMBERT:
        return [np.full(n, f32(f32(alb[c]) / PI), f32) for c in range(3)]
    d = smax(np.zeros(n, f32), dot(wi, nrm))  # :195-205
    fres = f_schlick(d, f32(0.04))
    half = normalized(vadd(wo, wi))
    rough = f32(m.exponent)
    cos_alpha = powf(smax(np.zeros(n, f32), dot(half, nrm)), rough)
    spec_factor = ((cos_alpha * (rough + f32(2.0))).astype(f32) / (f32(2.0) * PI)).astype(f32)
    spec = [((f32(1.0) * spec_factor).astype(f32) * fres).astype(f32) for _ in range(3)]
    diff = [((f32(alb[c]) / PI) * (f32(1.0) - fres).astype(f32)).astype(f32) for c in range(3)]
    return vadd(spec, diff)


def bsdf_scatter(m, wo, nrm, basis, s1d, s2d):
    """-> (wi, f, pdf); s2d = the four 2-D components 8+8*VM .. +3"""
    n = len(s1d)
    alb = (m.a.x, m.a.y, m.a.z)
    dsample = cosine_weighted_in_hemisphere(s2d[0], s2d[1])
    dbounce = normalized(mat_vec(basis, dsample))
    if m.kind == MAT_LAMBERT:  # :118-137
        return dbounce, [np.full(n, f32(f32(alb[c]) / PI), f32) for c in range(3)], (dsample[2] / PI).astype(f32)
    rough = f32(m.exponent)  # Dielectric :207-256
    cos = np.abs(dot(nrm, wo))
    dpdf = smax(np.full(n, f32(0.00001), f32), (dsample[2] / PI).astype(f32))
    df = [np.full(n, f32(f32(alb[c]) / PI), f32) for c in range(3)]
    ssample = cosine_power_weighted(s2d[2], s2d[3], rough)
    refl = vsub(wo, vscale(nrm, (f32(2.0) * dot(wo, nrm)).astype(f32)))  # wo.reflected(norm): v - (2 * dot(v, n)) * n
    sbounce = normalized(mat_vec(onb(refl), ssample))
    cap = smax(powf(ssample[2], rough), np.full(n, F32_EPS, f32))
    spdf = (((rough + f32(1.0)) / TWO_PI) * cap).astype(f32)
    scoeff = (((rough + f32(2.0)) / TWO_PI) * cap).astype(f32)
    scoeff = np.where(dot(nrm, sbounce) < 0, f32(0.0), scoeff).astype(f32)
    sf = [(f32(1.0) * scoeff).astype(f32) for _ in range(3)]
    fres = f_schlick(cos, f32(0.04))
    mask = s1d < fres
    pdf = ((fres * spdf).astype(f32) + ((f32(1.0) - fres).astype(f32) * dpdf).astype(f32)).astype(f32)
    return vwhere(mask, sbounce, dbounce), vwhere(mask, sf, df), pdf


# ---- Film::render_frame_into (src/film.rs:382-691) -----------------------------------------------------------------------
def fis_sample(inv, u):  # FilterImportanceSampler::sample, src/filter.rs:222-235 (scalar f32)
    u = f32(f32(2.0) * f32(u - f32(0.5)))
    mult = f32(-1.0) if u < 0 else f32(1.0)
    u = min(max(abs(u), f32(0.0)), f32(0.99999))
    idx_full = f32(u * f32(511.0))
    idx = int(math.floor(float(idx_full)))
    t = f32(idx_full - f32(math.trunc(float(idx_full))))
    return f32(mult * f32(f32(inv[idx] * f32(f32(1.0) - t)) + f32(inv[idx + 1] * t)))


def render(wd, p, tabs):
    """-> (film dict of float32 arrays like the oracle's, counters dict)"""
    s1d, s2d, scramble, fis = [np.asarray(t, f32) for t in tabs]
    W, H, samples, tw, th = int(p.width), int(p.height), int(p.samples), int(p.tile_w), int(p.tile_h)
    nspp, B, VM = 4 * samples, int(p.max_bounces), int(p.volume_marches)
    nl = int(wd.n_lights)
    cam = wd.camera
    res_w, res_h = f32(cam.res_w), f32(cam.res_h)
    if cam.kind in (0, 1):  # PinholeCamera::new / ThinLensCamera::new, src/camera.rs:53-72,134-157
        theta = f32(f32(f32(cam.vfov_or_size) * PI) / f32(180.0))
        half_h = f32(math.tan(float(f32(theta / f32(2.0)))))
        half_w = f32(f32(res_w / res_h) * half_h)
        half_pixel = f32(half_h / res_h)
    else:  # OrthographicCamera::new, src/camera.rs:228-240
        vsz = f32(cam.vfov_or_size)
        full_w, full_h = f32(vsz * f32(res_w / res_h)), vsz
        half_w, half_h = f32(full_w / f32(2.0)), f32(full_h / f32(2.0))
        half_pixel = f32(f32(vsz / res_h) / f32(2.0))

    def cam_param(base, vel, bit, t0):  # WSequenced for a camera parameter: constant, or the closure |t| base + vel * t at lane 0's time
        n = len(t0)
        if not (cam.animated & bit):
            return splat((base.x, base.y, base.z), n)
        return [(f32((base.x, base.y, base.z)[c]) + (f32((vel.x, vel.y, vel.z)[c]) * t0).astype(f32)).astype(f32) for c in range(3)]

    film = {"color": np.zeros((H, W, 3), f32), "alpha": np.zeros((H, W), f32), "background": np.zeros((H, W, 3), f32), "normal": np.zeros((H, W, 3), f32)}
    ctr = {"paths": 0, "segments": 0}
    ndc = (f32(f32(1.0) / f32(W)), f32(f32(1.0) / f32(H)))
    rho_t = f32(wd.coeff_extinction) if wd.has_extinction else None
    rho_s = f32(wd.coeff_scattering) if wd.has_scattering else None
    n1, n2 = 3 + VM, 12 + 8 * VM

    def samp1(set_, sample, scr):
        return fract((s1d[sample + nspp * set_] + scr).astype(f32))

    def samp2(dim, set_, sample, scr):
        return fract((s2d[dim + sample * 2 + nspp * 2 * set_] + scr).astype(f32))

    nx, ny = (W + W % tw) // tw, (H + H % th) // th  # the reference's tile grid, src/film.rs:399-427
    for tx in range(nx):
        for ty in range(ny):
            x0, y0, x1, y1 = tx * tw, ty * th, min(tx * tw + tw, W), min(ty * th + th, H)
            if x1 <= x0 or y1 <= y0:
                continue
            ew, eh = x1 - x0, y1 - y0
            # ---- ray-gen, src/film.rs:456-529 (x outer, y inner, samp inner-most; 4 lanes = samples 4*samp..+3)
            xs, ys, sn = np.meshgrid(np.arange(x0, x1), np.arange(y0, y1), np.arange(nspp), indexing="ij")
            xs, ys, sn = xs.reshape(-1), ys.reshape(-1), sn.reshape(-1)
            n = len(xs)
            ctr["paths"] += n
            scr = scramble[xs + ys * W].astype(f32)
            u0, u1 = samp2(0, 0, sn, scr), samp2(1, 0, sn, scr)
            fx = np.array([fis_sample(fis, v) for v in u0], f32)
            fy = np.array([fis_sample(fis, v) for v in u1], f32)
            uvx = (ndc[0] * ((xs.astype(f32) + f32(0.5)).astype(f32) + fx).astype(f32)).astype(f32)
            uvy = (ndc[1] * ((ys.astype(f32) + f32(0.5)).astype(f32) + fy).astype(f32)).astype(f32)
            time = (f32(p.time_start) + (f32(f32(p.time_end) - f32(p.time_start)) * samp1(0, sn, scr)).astype(f32)).astype(f32)
            t0 = time[(np.arange(n) // 4) * 4]  # lane 0 of the ray-gen packet (4 consecutive samples of one pixel)
            org = cam_param(cam.origin, cam.origin_vel, 1, t0)
            at, up = cam_param(cam.at, cam.at_vel, 2, t0), cam_param(cam.up, cam.up_vel, 4, t0)
            if cam.kind == 0:  # PinholeCamera::get_rays, src/camera.rs:81-114
                bw = normalized(vsub(org, at))
                bu = normalized(cross(up, bw))
                bv = cross(bw, bu)
                ll = vsub(vsub(vsub(org, vscale(bu, half_w)), vscale(bv, half_h)), bw)
                horiz = vscale(vscale(vscale(bu, half_w), f32(2.0)), uvx)
                verti = vscale(vscale(vscale(bv, half_h), f32(2.0)), uvy)
                ray_o, ray_d = org, normalized(vsub(vadd(vadd(ll, horiz), verti), org))
            elif cam.kind == 1:  # ThinLensCamera::get_rays, src/camera.rs:168-208; lens sample = 2-D set 1 (src/film.rs:520-523)
                focus = cam_param(cam.focus, cam.focus_vel, 8, t0)
                fd = mag(vsub(focus, org))
                bw = normalized(vsub(org, at))
                bu = normalized(cross(up, bw))
                bv = cross(bw, bu)
                ll = vsub(vsub(vsub(org, vscale(vscale(bu, half_w), fd)), vscale(vscale(bv, half_h), fd)), vscale(bw, fd))
                horiz = vscale(vscale(vscale(vscale(bu, half_w), fd), f32(2.0)), uvx)
                verti = vscale(vscale(vscale(vscale(bv, half_h), fd), f32(2.0)), uvy)
                rdx, rdy = concentric_circle_map(samp2(0, 1, sn, scr), samp2(1, 1, sn, scr))
                rdx, rdy = (rdx * f32(cam.aperture)).astype(f32), (rdy * f32(cam.aperture)).astype(f32)
                ray_o = vadd(org, vadd(vscale(bu, rdx), vscale(bv, rdy)))
                ray_d = normalized(vsub(vadd(vadd(ll, horiz), verti), ray_o))
            else:  # OrthographicCamera::get_rays, src/camera.rs:249-280
                bw = normalized(vsub(at, org))
                bu = normalized(cross(bw, up))
                bv = cross(bu, bw)
                ll = vsub(vsub(org, vscale(bu, half_w)), vscale(bv, half_h))
                ray_o = vadd(ll, vadd(vscale(vscale(bu, uvx), full_w), vscale(vscale(bv, uvy), full_h)))
                ray_d = bw
            rays = {"time": time, "o": ray_o, "d": ray_d, "rad": splat((0, 0, 0), n), "thr": splat((1, 1, 1), n),
                    "tx": (xs - x0).astype(np.int64), "ty": (ys - y0).astype(np.int64), "valid": np.ones(n, bool), "scr": scr, "samp": sn.astype(np.int64)}
            sums = {"color": np.zeros((ew, eh, 3), f32), "alpha": np.zeros((ew, eh), f32), "background": np.zeros((ew, eh, 3), f32), "normal": np.zeros((ew, eh, 3), f32)}
            for depth in range(10 ** 6):
                n = len(rays["time"])
                if n == 0:
                    break
                if depth == 0:  # camera.half_pixel_size_at (Orthographic: the constant, src/camera.rs:282-284)
                    thr_at = (lambda t: np.full(len(t), half_pixel, f32)) if cam.kind == 2 else (lambda t: (half_pixel * t).astype(f32))
                else:
                    k = f32(f32(f32(0.0001) * f32(2.0)) * f32(depth))
                    thr_at = lambda t, k=k: (k * t).astype(f32)
                # ---- add_hits (src/hitable.rs:170-210): invalid lanes are never binned, so only valid lanes are traced
                v = np.flatnonzero(rays["valid"])
                o, d = [rays["o"][c][v] for c in range(3)], [rays["d"][c][v] for c in range(3)]
                closest = np.full(len(v), f32(f32(p.world_radius) * f32(2.0)), f32)
                ids = np.full(len(v), -1, np.int64)
                with np.errstate(all="ignore"):
                    for i in range(wd.n_hitables):
                        h = wd.hitables[i]
                        t = sphere_hit(h, o, d, closest, rays["time"][(v // 4) * 4]) if h.kind == 0 else sdf_hit(h, o, d, closest, thr_at, p.sdf_detail_scale, int(p.max_marches), rays["time"][(v // 4) * 4])
                        win = t < closest
                        closest = np.where(win, t, closest).astype(f32)
                        ids = np.where(win, i, ids)
                ctr["segments"] += len(v)
                # ---- HitStore bins -> packets, object-major, insertion order, padded to x4 (src/hitable.rs:94-134)
                lane_src, lane_t, lane_obj = [], [], []
                for i in range(wd.n_hitables):
                    sel = np.flatnonzero(ids == i)
                    if len(sel) == 0:
                        continue
                    pad = (-len(sel)) % 4
                    lane_src.append(np.concatenate([v[sel], np.full(pad, -1, np.int64)]))
                    lane_t.append(np.concatenate([closest[sel], np.zeros(pad, f32)]))
                    lane_obj.append(np.full(len(sel) + pad, i, np.int64))
                if not lane_src:
                    break
                src, ht, hobj = np.concatenate(lane_src), np.concatenate(lane_t).astype(f32), np.concatenate(lane_obj)
                m = len(src)
                real = src >= 0
                g = np.where(real, src, 0)
                nanv = np.full(m, np.nan, f32)
                R = {"time": np.where(real, rays["time"][g], nanv).astype(f32),
                     "o": [np.where(real, rays["o"][c][g], nanv).astype(f32) for c in range(3)], "d": [np.where(real, rays["d"][c][g], nanv).astype(f32) for c in range(3)],
                     "rad": [np.where(real, rays["rad"][c][g], f32(0)).astype(f32) for c in range(3)], "thr": [np.where(real, rays["thr"][c][g], f32(0)).astype(f32) for c in range(3)],
                     "tx": np.where(real, rays["tx"][g], 0), "ty": np.where(real, rays["ty"][g], 0), "valid": real.copy(),
                     "scr": np.where(real, rays["scr"][g], f32(0)).astype(f32), "samp": np.where(real, rays["samp"][g], 0)}
                with np.errstate(all="ignore"):
                    point = [mul_add(R["d"][c], ht, R["o"][c]) for c in range(3)]  # WHit::point
                    normal = [np.zeros(m, f32) for _ in range(3)]
                    offset_by = np.zeros(m, f32)
                    for i in range(wd.n_hitables):
                        sel = np.flatnonzero(hobj == i)
                        if len(sel) == 0:
                            continue
                        h = wd.hitables[i]
                        ps = [point[c][sel] for c in range(3)]
                        if h.kind == 0:  # src/sphere.rs:74-86
                            nn = normalized(vsub(ps, sphere_center(h, R["time"][(sel // 4) * 4])))
                            ob = np.zeros(len(sel), f32)
                        else:  # src/sdf.rs:85-101 + sdfu normals_fast (A5)
                            ob = smax(np.full(len(sel), f32(0.0001), f32), (f32(p.sdf_detail_scale) * thr_at(ht[sel])).astype(f32))
                            tp = R["time"][(sel // 4) * 4]
                            ps = vsub(ps, sphere_center(h, tp))  # EXTENSION: the normal is estimated in the SDF's frame (zero origin in the reference)
                            gsum = None
                            for kx, ky, kz in ((1, -1, -1), (-1, -1, 1), (-1, 1, -1), (1, 1, 1)):
                                kv = [np.full(len(sel), f32(kx), f32), np.full(len(sel), f32(ky), f32), np.full(len(sel), f32(kz), f32)]
                                term = vscale(kv, mandelbox_dist(vadd(ps, vscale(kv, ob)), h, tp))
                                gsum = term if gsum is None else vadd(gsum, term)
                            nn = normalized(gsum)
                        for c in range(3):
                            normal[c][sel] = nn[c]
                        offset_by[sel] = ob
                    basis = onb(normal)
                    # ---- per-bounce samples (src/film.rs:564-589), per lane by ITS (sample, scramble)
                    S1 = [samp1(1 + k_ + depth * n1, R["samp"], R["scr"]) for k_ in range(n1)]
                    S2 = [samp2(i % 2, 2 + i // 2 + depth * n2 // 2, R["samp"], R["scr"]) for i in range(n2)]
                    # ---- PathTracingIntegrator::integrate (src/integrator.rs:47-204)
                    wo = vneg(R["d"])
                    mat_idx = np.array([wd.hitables[int(i)].material for i in hobj])
                    vol_T = expf((-rho_t * ht).astype(f32)) if rho_t is not None else np.ones(m, f32)
                    le = [np.zeros(m, f32) for _ in range(3)]
                    recv = np.zeros(m, bool)
                    for mi in np.unique(mat_idx):
                        sel = np.flatnonzero(mat_idx == mi)
                        mt = wd.materials[int(mi)]
                        l_ = bsdf_le(mt, [wo[c][sel] for c in range(3)], len(sel))
                        for c in range(3):
                            le[c][sel] = l_[c]
                        recv[sel] = mt.kind in (MAT_LAMBERT, MAT_DIELECTRIC)
                    rad = vadd(R["rad"], vscale(vmul(le, R["thr"]), vol_T))
                    pk = (np.arange(m) // 4) * 4  # first lane of each lane's packet
                    if nl > 0:
                        picks0 = np.nan_to_num(np.floor((S1[0] * f32(nl)).astype(f32)), nan=0.0).astype(np.int64).clip(0, nl - 1)  # `as usize` (A8)
                        corr = f32(f32(nl) / f32(4.0))
                        for i in range(4):  # every lane is lit by the light lane i of ITS packet picked
                            li_idx = picks0[pk + i]
                            rsel = np.flatnonzero(recv)
                            if len(rsel) == 0:
                                break
                            pt_, nn_ = [point[c][rsel] for c in range(3)], [normal[c][rsel] for c in range(3)]
                            endp, pdf, em = [np.zeros(len(rsel), f32) for _ in range(3)], np.ones(len(rsel), f32), [np.zeros(len(rsel), f32) for _ in range(3)]
                            for li in np.unique(li_idx[rsel]):  # surface_sample_one_light, src/integrator.rs:207-240
                                s2_ = np.flatnonzero(li_idx[rsel] == li)
                                L = wd.lights[int(li)]
                                e_, p_ = light_sample(L, S2[2 * i][rsel][s2_], S2[2 * i + 1][rsel][s2_], [pt_[c][s2_] for c in range(3)])
                                for c in range(3):
                                    endp[c][s2_] = e_[c]
                                    em[c][s2_] = f32((L.emission.x, L.emission.y, L.emission.z)[c])
                                pdf[s2_] = p_
                            wi = vsub(endp, pt_)
                            dist = mag(wi)
                            wi = vdiv(wi, dist)
                            ndw = dot(nn_, wi)
                            occp = vadd(pt_, vscale(vscale(nn_, signum(ndw)), offset_by[rsel]))
                            occ = test_occluded(wd, p, occp, endp, R["time"][pk][rsel])
                            fval = [np.zeros(len(rsel), f32) for _ in range(3)]
                            for mi in np.unique(mat_idx[rsel]):
                                s2_ = np.flatnonzero(mat_idx[rsel] == mi)
                                ff = bsdf_f(wd.materials[int(mi)], [wo[c][rsel][s2_] for c in range(3)], [wi[c][s2_] for c in range(3)], [nn_[c][s2_] for c in range(3)])
                                for c in range(3):
                                    fval[c][s2_] = ff[c]
                            fv = vscale(fval, smax(ndw, np.zeros(len(rsel), f32)))
                            tr = expf((-rho_t * dist).astype(f32)) if rho_t is not None else np.ones(len(rsel), f32)
                            li_v = vdiv(vscale(vscale(vmul(em, fv), tr), occ), pdf)
                            contrib = [np.zeros(m, f32) for _ in range(3)]
                            for c in range(3):
                                contrib[c][rsel] = li_v[c]
                            add = vscale(vscale(vmul(contrib, R["thr"]), corr), vol_T)
                            rad = vwhere(recv, vadd(rad, add), rad)
                    if rho_s is not None and nl > 0:
                        corr = f32(f32(f32(nl) / f32(4.0)) / f32(VM))
                        for march in range(VM):
                            picks = np.nan_to_num(np.floor((S1[march + 1] * f32(nl)).astype(f32)), nan=0.0).astype(np.int64).clip(0, nl - 1)
                            for i in range(4):
                                li_idx = picks[pk + i]
                                vd, vpdf, lpdf = np.zeros(m, f32), np.ones(m, f32), np.ones(m, f32)
                                sp, endp, em = [np.zeros(m, f32) for _ in range(3)], [np.zeros(m, f32) for _ in range(3)], [np.zeros(m, f32) for _ in range(3)]
                                for li in np.unique(li_idx):  # volume_sample_one_light, src/integrator.rs:242-281
                                    sel = np.flatnonzero(li_idx == li)
                                    L = wd.lights[int(li)]
                                    ro, rd_ = [R["o"][c][sel] for c in range(3)], [R["d"][c][sel] for c in range(3)]
                                    vd_, vp_ = light_sample_volume(L, S1[1][sel], ro, rd_, ht[sel])
                                    sp_ = vadd(ro, vscale(rd_, vd_))
                                    e_, lp_ = light_sample(L, S2[8 + 8 * march + 2 * i][sel], S2[8 + 8 * march + 2 * i + 1][sel], sp_)
                                    for c in range(3):
                                        sp[c][sel] = sp_[c]
                                        endp[c][sel] = e_[c]
                                        em[c][sel] = f32((L.emission.x, L.emission.y, L.emission.z)[c])
                                    vd[sel], vpdf[sel], lpdf[sel] = vd_, vp_, lp_
                                dl = mag(vsub(endp, sp))
                                occ = test_occluded(wd, p, sp, endp, R["time"][pk])
                                fph = f32(f32(1.0) / f32(f32(4.0) * PI))
                                tr = expf((-rho_t * dl).astype(f32)) if rho_t is not None else np.ones(m, f32)
                                contrib = vdiv(vscale(vscale(vscale(em, fph), tr), occ), (vpdf * lpdf).astype(f32))
                                trn = expf((-rho_t * vd).astype(f32)) if rho_t is not None else np.ones(m, f32)
                                rad = vadd(rad, vscale(vscale(vscale(vmul(contrib, R["thr"]), corr), rho_s), trn))
                # ---- scatter / roulette / outputs (src/integrator.rs:134-203)
                with np.errstate(all="ignore"):
                    wi_n = [np.zeros(m, f32) for _ in range(3)]
                    f_n = [np.zeros(m, f32) for _ in range(3)]
                    pdf_n = np.ones(m, f32)
                    for mi in np.unique(mat_idx[recv]) if recv.any() else []:
                        sel = np.flatnonzero(recv & (mat_idx == mi))
                        bsel = [[basis[a][c][sel] for c in range(3)] for a in range(3)]
                        w_, f_, p_ = bsdf_scatter(wd.materials[int(mi)], [wo[c][sel] for c in range(3)], [normal[c][sel] for c in range(3)], bsel, S1[3][sel],
                                                  [S2[8 + 8 * VM + q][sel] for q in range(4)])
                        for c in range(3):
                            wi_n[c][sel] = w_[c]
                            f_n[c][sel] = f_[c]
                        pdf_n[sel] = p_
                    ndl = np.abs(dot(wi_n, normal))
                    new_thr = vdiv(vscale(vmul(vscale(R["thr"], vol_T), f_n), ndl), pdf_n)
                    rr = np.zeros(m, f32)
                    if depth > 2:
                        rr = smax((f32(1.0) - smax(smax(R["thr"][0], R["thr"][1]), R["thr"][2])).astype(f32), np.full(m, f32(0.05), f32))
                        new_thr = vdiv(new_thr, (f32(1.0) - rr).astype(f32))
                    new_o = vadd(point, vscale(vscale(normal, signum(dot(normal, wi_n))), offset_by))  # create_rays, src/hitable.rs:42-47
                # emission order: packets in order; inside a packet (receiving): depth-0 AOVs of its valid lanes, then per lane Color or spawn;
                # (non-receiving): Background (depth 0) / Color per valid lane.  All adds of one pixel happen in this order (src/film.rs:54-61,604-606).
                spawn = []
                for j in range(m):
                    if not R["valid"][j]:
                        continue
                    px, py = int(R["tx"][j]), int(R["ty"][j])
                    if recv[j]:
                        if depth == 0:
                            sums["alpha"][px, py] = f32(sums["alpha"][px, py] + f32(1.0))
                            for c in range(3):
                                sums["normal"][px, py, c] = f32(sums["normal"][px, py, c] + normal[c][j])
                        if depth >= B or S1[4][j] < rr[j]:
                            for c in range(3):
                                sums["color"][px, py, c] = f32(sums["color"][px, py, c] + rad[c][j])
                        else:
                            spawn.append(j)
                    else:
                        ch = "background" if depth == 0 else "color"
                        for c in range(3):
                            sums[ch][px, py, c] = f32(sums[ch][px, py, c] + rad[c][j])
                # Note on the AOV order: the reference pushes Alpha / WorldNormal of ALL valid lanes of a packet before the packet's Color
                # samples; different channels never interact, so per-channel order is what matters and it is lane order either way.
                sp_ = np.array(spawn, np.int64)
                k = len(sp_)
                pad = (-k) % 4
                nanp = np.full(pad, np.nan, f32)
                zp = np.zeros(pad, f32)
                nan_thr = np.isnan(new_thr[0][sp_]) | np.isnan(new_thr[1][sp_]) | np.isnan(new_thr[2][sp_]) if k else np.zeros(0, bool)
                rays = {"time": np.concatenate([R["time"][sp_], nanp]).astype(f32),
                        "o": [np.concatenate([new_o[c][sp_], nanp]).astype(f32) for c in range(3)], "d": [np.concatenate([wi_n[c][sp_], nanp]).astype(f32) for c in range(3)],
                        "rad": [np.concatenate([rad[c][sp_], zp]).astype(f32) for c in range(3)],
                        "thr": [np.concatenate([np.where(nan_thr, R["thr"][c][sp_], new_thr[c][sp_]), zp]).astype(f32) for c in range(3)],
                        "tx": np.concatenate([R["tx"][sp_], np.zeros(pad, np.int64)]), "ty": np.concatenate([R["ty"][sp_], np.zeros(pad, np.int64)]),
                        "valid": np.concatenate([np.ones(k, bool), np.zeros(pad, bool)]), "scr": np.concatenate([R["scr"][sp_], zp]).astype(f32),
                        "samp": np.concatenate([R["samp"][sp_], np.zeros(pad, np.int64)])}
            # ---- tile_finished: film = tile_sum / n, src/film.rs:82-98 (y = 0 is the bottom row)
            nn_ = f32(nspp)
            for lx in range(ew):
                for ly in range(eh):
                    film["color"][y0 + ly, x0 + lx] = sums["color"][lx, ly] / nn_
                    film["background"][y0 + ly, x0 + lx] = sums["background"][lx, ly] / nn_
                    film["normal"][y0 + ly, x0 + lx] = sums["normal"][lx, ly] / nn_
                    film["alpha"][y0 + ly, x0 + lx] = sums["alpha"][lx, ly] / nn_
    return film, ctr


# ---- FilterImportanceSampler::new over the four Filter impls (src/filter.rs:12-220) + CDF (src/math.rs:136-191), scalar binary32 ----
def _cosf(x):
    return f32(math.cos(float(x)))


def _sinf(x):
    return f32(math.sin(float(x)))


def filter_evaluate(kind, radius, b, c, p):
    """kind: 0 BlackmanHarris, 1 Box, 2 MitchellNetravali(b, c), 3 LanczosSinc(tau = b) - include/rayn_hip.h's numbering"""
    radius, p = f32(radius), f32(p)
    if kind == 0:  # :42-48
        if abs(p) > radius:
            return f32(0.0)
        x = f32(f32(abs(f32(p / radius)) * f32(0.5)) + f32(0.5))
        a0, a1, a2, a3 = f32(0.35875), f32(0.48829), f32(0.14128), f32(0.01168)
        twopi, fourpi, sixpi = f32(PI * f32(2.0)), f32(PI * f32(4.0)), f32(PI * f32(6.0))
        return f32(f32(f32(a0 - f32(a1 * _cosf(f32(twopi * x)))) + f32(a2 * _cosf(f32(fourpi * x)))) + f32(a3 * _cosf(f32(sixpi * x))))
    if kind == 1:  # :132-139
        return f32(0.0) if abs(p) > radius else f32(1.0)
    if kind == 2:  # :75-93
        b, c = f32(b), f32(c)
        x = f32(abs(f32(f32(f32(2.0) * p) / radius)))
        if x >= f32(2.0):
            return f32(0.0)
        sixth = f32(f32(1.0) / f32(6.0))
        if x > f32(1.0):
            t3 = f32(f32(f32(f32(f32(-b) - f32(f32(6.0) * c)) * x) * x) * x)
            t2 = f32(f32(f32(f32(f32(6.0) * b) + f32(f32(30.0) * c)) * x) * x)
            t1 = f32(f32(f32(f32(-12.0) * b) - f32(f32(48.0) * c)) * x)
            t0 = f32(f32(f32(8.0) * b) + f32(f32(24.0) * c))
            return f32(f32(f32(f32(t3 + t2) + t1) + t0) * sixth)
        t3 = f32(f32(f32(f32(f32(f32(12.0) - f32(f32(9.0) * b)) - f32(f32(6.0) * c)) * x) * x) * x)
        t2 = f32(f32(f32(f32(f32(-18.0) + f32(f32(12.0) * b)) + f32(f32(6.0) * c)) * x) * x)
        t0 = f32(f32(6.0) - f32(f32(2.0) * b))
        return f32(f32(f32(t3 + t2) + t0) * sixth)
    tau = f32(b)  # LanczosSinc :163-185

    def sinc(x):
        x = f32(abs(x))
        if x <= f32(0.00001):
            return f32(1.0)
        pix = f32(PI * x)
        return f32(_sinf(pix) / pix)
    x = f32(abs(p))
    if x > radius:
        return f32(0.0)
    lanczos = sinc(f32(x / tau))
    return f32(sinc(x) * lanczos)


def fis_table(kind, radius, b=0.0, c=0.0):
    n = 512
    radius = f32(radius)
    items, weight_sum = [], f32(0.0)
    for k in range(n):  # src/filter.rs:198-204
        t = f32(f32(k) / f32(n - 1))
        d = f32(f32(f32(0.0) * f32(f32(1.0) - t)) + f32(radius * t))  # 0.0.lerp(f_rad, t)
        w = filter_evaluate(kind, radius, b, c, d)
        items.append([d, w])
        weight_sum = f32(weight_sum + w)
    for it in items:  # CDF::prepare, src/math.rs:157-178
        it[1] = f32(it[1] / weight_sum)
    dens, cum = [], f32(0.0)
    for _, w in items:
        cum = f32(cum + w)
        dens.append(cum)
    for i in range(n - 1, -1, -1):
        dens[i] = f32(1.0)
        if items[i][1] > 0:
            break
    out = np.zeros(n, f32)
    for k in range(n):  # src/filter.rs:210-214 + CDF::sample
        u = f32(f32(k) / f32(n - 1))
        for (d, _), dn in zip(items, dens):
            if dn >= u:
                out[k] = d
                break
    return out
