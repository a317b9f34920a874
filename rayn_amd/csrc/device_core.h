// device_core.h — per-lane (scalar) device restatement of rayn's arithmetic for the wavefront
// kernels.  One GPU lane = one lane of a reference f32x4 packet; cross-lane effects of the packet
// code (the 4 light picks of a packet, src/integrator.rs:76-93,100-131) are handled in the shade
// kernel with wave shuffles.  Operation order follows the reference line by line so results are
// bit-identical with the CPU oracle; transcendentals and the mul_add policy come from
// include/rayn_detmath.h.  Compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rayn_detmath.h"
#include "../../include/rayn_detmath_fast.h"
#include "device_scene.h"
#include "kernels.h"

#ifndef RAYN_KNS
#define RAYN_KNS rayn_p0
#endif

// The pinned elementary functions are evaluated through include/rayn_detmath_fast.h (dmf_*): shorter fused polynomials on the
// bit-identical reduced argument + a rounding-safety test + the rayn_detmath.h evaluation as the fallback (~3e-5 of the calls) -
// the same result bits as rayn_detmath.h (tests/test_detmath.py on the CPU, the C-ABI probe on the device), about half the
// binary64 operations.

namespace RAYN_KNS {
using namespace rayn;

#define RD __device__ __forceinline__

// ---- f32 helpers with wide/SSE semantics -------------------------------------------------------
RD float fmaxs(float a, float b) { return a > b ? a : b; } // a.max(b): maxps
RD float fmins(float a, float b) { return a < b ? a : b; } // a.min(b): minps
RD float signum(float x) { return x != x ? x : __builtin_copysignf(1.0f, x); }
RD float muladd(float a, float b, float c) { return rayn_muladd(a, b, c); }
RD float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; } // sdfu Lerp

constexpr float PI_F = 3.14159265358979323846f;
constexpr float TWO_PI_F = 6.28318530717958647692f;
constexpr float FRAC_PI_2_F = 1.57079632679489661923f;
constexpr float FRAC_PI_4_F = 0.78539816339744830962f;
constexpr float EPSILON_F = 1.1920929e-7f;

// ---- Vec3 (ultraviolet 0.4.6 operation order) --------------------------------------------------
RD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
RD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
RD f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
RD f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
RD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
RD float dot(f3 a, f3 b) { return muladd(a.x, b.x, muladd(a.y, b.y, a.z * b.z)); }
RD float mag_sq(f3 a) { return dot(a, a); }
// Correctly rounded sqrt.  For x in [2^-60, 2^60) one v_rsq_f32 plus a residual correction gives the IEEE
// result (verified EXHAUSTIVELY over that window on the device, tests/test_gpu_parity.py); every other input
// (0, denormals, huge, inf, NaN, negative) takes hipcc's IEEE sequence.  16+6 VALU cycles instead of 46.
RD float sqrt_rn(float x) {
    if (__float_as_uint(x) - 0x21800000u < 0x3C000000u) {
        const float y = __builtin_amdgcn_rsqf(x);
        const float g = x * y;
        const float h = 0.5f * y;
        const float d = __builtin_fmaf(-g, g, x);
        return __builtin_fmaf(d, h, g);
    }
    return __builtin_sqrtf(x);
}
RD float mag(f3 a) { return sqrt_rn(mag_sq(a)); }
// fl(1 / fl(sqrt(x))) - the reference's `1.0 / v.mag()` (two roundings) - with the reciprocal taken from the SAME v_rsq_f32: y = rsq(x) is within ~2 ulp of
// 1 / s for s = sqrt_rn(x), one Newton-Raphson step r1 = y + (1 - s y) y brings it within half an ulp (+ 2^-45), and the residual correction
// r = r1 + (1 - s r1) r1 (fma: the residual is exact) rounds it correctly.  Verified EXHAUSTIVELY on the device against hipcc's IEEE sqrt and '/'
// over every float of the window (tests/test_gpu_parity.py, probe op 15); outside the window (0, denormals, huge, inf, NaN, negative) the IEEE sequences run.
// 9 VALU instructions (one transcendental) instead of 5 + 10 (two: v_rsq_f32, v_rcp_f32).
#ifndef RAYN_RCP_SQRT_IEEE
RD float rcp_sqrt_rn(float x) {
    const uint32_t xb = __float_as_uint(x);
    // (the second test: when the significand of s is all ones, 1 / s lies just above a rounding tie and the correction rounds the wrong way - Markstein's exception.
    //  That happens for exactly the two largest significands of x in every other binade, 120 floats in all (tools/passes_r06/diag_rcp_sqrt.py); they take the IEEE path.)
    if (xb - 0x21800000u < 0x3C000000u && (xb & 0x7FFFFEu) != 0x7FFFFEu) {
        const float y = __builtin_amdgcn_rsqf(x);
        const float g = x * y;
        const float h = 0.5f * y;
        const float d = __builtin_fmaf(-g, g, x);
        const float s = __builtin_fmaf(d, h, g); // sqrt_rn(x)
        float r = __builtin_fmaf(__builtin_fmaf(-s, y, 1.0f), y, y);
        r = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
        return r;
    }
    return 1.0f / __builtin_sqrtf(x);
}
#else
RD float rcp_sqrt_rn(float x) { return 1.0f / sqrt_rn(x); } // the r5 form (variant builds: tools/variants/README.md)
#endif
RD f3 normalized(f3 a) { float r = rcp_sqrt_rn(mag_sq(a)); return f3{a.x * r, a.y * r, a.z * r}; }
// a / m (three IEEE divisions in the reference) for m = mag(a): m >= |a_i| always (sqrt(fl(x*x + ..)) >= |x|),
// so min|a_i| >= 2^-60 and m <= 2^60 put every operand and quotient inside the window where the Newton-Raphson
// steps of an IEEE '/' need neither v_div_scale nor v_div_fixup (see div_nr); the reciprocal is refined once
// and shared.  Anything else (zero components, NaN, extreme exponents) takes the IEEE divisions.
RD f3 div_by_mag(f3 a, float m) {
    const float mn = __builtin_fminf(__builtin_fminf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    if (mn >= 8.6736174e-19f && m <= 1.1529215e18f) {
        float r = __builtin_amdgcn_rcpf(m);
        const float e0 = __builtin_fmaf(-m, r, 1.0f);
        r = __builtin_fmaf(e0, r, r);
        f3 q = f3{a.x * r, a.y * r, a.z * r};
        q.x = __builtin_fmaf(__builtin_fmaf(-m, q.x, a.x), r, q.x);
        q.y = __builtin_fmaf(__builtin_fmaf(-m, q.y, a.y), r, q.y);
        q.z = __builtin_fmaf(__builtin_fmaf(-m, q.z, a.z), r, q.z);
        q.x = __builtin_fmaf(__builtin_fmaf(-m, q.x, a.x), r, q.x);
        q.y = __builtin_fmaf(__builtin_fmaf(-m, q.y, a.y), r, q.y);
        q.z = __builtin_fmaf(__builtin_fmaf(-m, q.z, a.z), r, q.z);
        return q;
    }
    return f3{a.x / m, a.y / m, a.z / m};
}
RD f3 cross(f3 a, f3 b) {
    return f3{muladd(a.y, b.z, -a.z * b.y), muladd(a.z, b.x, -a.x * b.z), muladd(a.x, b.y, -a.y * b.x)};
}
RD f3 muladd3(f3 a, float m, f3 c) { return f3{muladd(a.x, m, c.x), muladd(a.y, m, c.y), muladd(a.z, m, c.z)}; }
RD f3 reflected(f3 v, f3 n) { return v - (2.0f * dot(v, n)) * n; }
RD float component_max(f3 a) { return fmaxs(fmaxs(a.x, a.y), a.z); }
RD bool any_nan(f3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }

struct Basis { f3 c0, c1, c2; };
RD f3 mul(const Basis& m, f3 v) { return m.c0 * v.x + m.c1 * v.y + m.c2 * v.z; }
// src/math.rs:49-59
RD Basis orthonormal_basis(f3 nor) {
    float ks = signum(nor.z);
    float ka = 1.0f / (1.0f + __builtin_fabsf(nor.z));
    float kb = -ks * nor.x * nor.y * ka;
    Basis b;
    b.c0 = f3{1.0f - nor.x * nor.x * ka, ks * kb, -ks * nor.x};
    b.c1 = f3{kb, ks - nor.y * nor.y * ka * ks, -nor.y};
    b.c2 = nor;
    return b;
}

// ---- sampling maps (src/math.rs) -----------------------------------------------------------------
RD void concentric_circle_map(float u0, float u1, float* ox, float* oy) { // :201-219
    float a = muladd(u0, 2.0f, -1.0f);
    float b = muladd(u1, 2.0f, -1.0f);
    if (a == 0.0f && b == 0.0f) b = 0.0001f;
    float phi1 = FRAC_PI_4_F * b / a;
    float phi2 = muladd(-FRAC_PI_4_F / b, a, FRAC_PI_2_F);
    bool mask = (a * a) > (b * b);
    float r = mask ? a : b;
    float phi = mask ? phi1 : phi2;
    float s, c;
    dmf_sincosf(phi, &s, &c);
    *ox = r * c;
    *oy = r * s;
}
RD f3 cosine_weighted_in_hemisphere(float u0, float u1) { // :99-103
    float x, y;
    concentric_circle_map(u0, u1, &x, &y);
    float m2 = muladd(x, x, y * y);
    float z = sqrt_rn(1.0f - fmins(m2, 1.0f));
    return f3{x, y, z};
}
RD f3 cosine_power_weighted(float u0, float u1, float power) { // :106-113
    float a = dmf_powf(u0, 1.0f / (power + 1.0f));
    float a2 = a * a;
    float b = sqrt_rn(1.0f - a2);
    float s, c;
    dmf_sincosf(2.0f * u1, &s, &c);
    return f3{b * c, b * s, a};
}
RD float f_schlick(float cosv, float f0) { // :122-124
    float x = 1.0f - cosv;
    float x2 = x * x;
    return f0 + (1.0f - f0) * ((x2 * x2) * x);
}

// ---- Samples (src/sampler.rs:62-64,92-94) ----------------------------------------------------------

RD float sample_1d(const Tables& t, uint32_t n, uint32_t sample, float scramble, uint32_t set) {
    return dm_fractf(t.s1d[sample + n * set] + scramble);
}
RD float sample_2d(const Tables& t, uint32_t n, uint32_t dim, uint32_t sample, float scramble, uint32_t set) {
    return dm_fractf(t.s2d[dim + sample * 2 + n * 2 * set] + scramble);
}
// FilterImportanceSampler::sample, src/filter.rs:222-235
RD float fis_sample(const float* __restrict__ inverse_cdf, float u) {
    u = 2.0f * (u - 0.5f);
    float mult = u < 0.0f ? -1.0f : 1.0f;
    u = __builtin_fabsf(u);
    u = fmaxs(u, 0.0f);
    u = fmins(u, 0.99999f);
    float idx_full = u * (float)(RAYN_FIS_TABLE_SIZE - 1);
    uint32_t idx = (uint32_t)__builtin_floorf(idx_full);
    float t = dm_fractf(idx_full);
    return mult * lerpf(inverse_cdf[idx], inverse_cdf[idx + 1], t);
}

// Counters of the instrumented (COUNT = true) kernel variants - roofline accounting outside the timed region: SDF distance evaluations and
// the fold / orbit ITERATIONS they ran (MandelBox: always `iterations`; Mandelbulb: until bailout), from which bench.py prices an evaluation
// of the scene's own SDF.  Dead code in the product kernels (COUNT = false).
struct EvalCtr { uint32_t n = 0, it = 0; };

// ---- SDFs (src/sdf.rs:104-188; sdfu::Sphere) ---------------------------------------------------------
// EXTENSION (not in the reference): power-8 Mandelbulb distance estimator, trigonometry-free polynomial form.
// Plain IEEE f32 operations in exactly this order (the oracle restates them lane by lane).
// The logarithm is kept OUT of line: inlined, its twelve binary64 coefficients are hoisted into 24 scalar registers for the
// whole march kernel, which then sits at the 102-SGPR ceiling and marches MandelBox scenes 4 % slower (k_shadow1, c3).
__device__ __attribute__((noinline)) static float bulb_logf(float m) { return dmf_logf(m); } // r6: table + degree-7 log1p + rounding-safety test, dm_logf as the fallback (rayn_detmath_fast.h)
// The estimator in three pieces - orbit state at the point, ONE orbit step, the distance from the final (|w|^2, dz) - so that the march kernels
// written for this SDF (march_bulb.h: one loop trip = one orbit STEP) run exactly the operations of the whole-evaluation form below.
struct BulbOrbit { f3 w; float m, dz; };
RD BulbOrbit bulb_begin(f3 p) { return BulbOrbit{p, p.x * p.x + p.y * p.y + p.z * p.z, 1.0f}; }
RD void bulb_step(BulbOrbit& o, f3 p) {
    const f3 w = o.w;
    const float m = o.m;
    const float m2 = m * m, m4 = m2 * m2;
    o.dz = 8.0f * sqrt_rn(m4 * m2 * m) * o.dz + 1.0f;
    const float x = w.x, x2 = x * x, x4 = x2 * x2;
    const float y = w.y, y2 = y * y, y4 = y2 * y2;
    const float z = w.z, z2 = z * z, z4 = z2 * z2;
    const float k3 = x2 + z2;
    const float k2 = rcp_sqrt_rn(k3 * k3 * k3 * k3 * k3 * k3 * k3); // 1.0f / sqrt(..)
    const float k1 = x4 + y4 + z4 - 6.0f * y2 * z2 - 6.0f * x2 * y2 + 2.0f * z2 * x2;
    const float k4 = x2 - y2 + z2;
    o.w.x = p.x + 64.0f * x * y * z * (x2 - z2) * k4 * (x4 - 6.0f * x2 * z2 + z4) * k1 * k2;
    o.w.y = p.y + -16.0f * y2 * k3 * k4 * k4 + k1 * k1;
    o.w.z = p.z + -8.0f * y * k4 * (x4 * x4 - 28.0f * x4 * x2 * z2 + 70.0f * x4 * z4 - 28.0f * x2 * z2 * z4 + z4 * z4) * k1 * k2;
    o.m = o.w.x * o.w.x + o.w.y * o.w.y + o.w.z * o.w.z;
}
constexpr float BULB_BAILOUT = 256.0f;
RD float bulb_finish(float m, float dz) { return 0.25f * bulb_logf(m) * sqrt_rn(m) / dz; }
RD float bulb_finish_inl(float m, float dz, const double* __restrict__ logtab) { return 0.25f * dmf_logf_t(m, logtab) * sqrt_rn(m) / dz; } // the same with the logarithm's fast path inline, its table wherever the caller keeps it
template <bool COUNT>
RD float mandelbulb_dist(f3 p, uint32_t iterations, EvalCtr& evals) {
    BulbOrbit o = bulb_begin(p);
    for (uint32_t i = 0; i < iterations; i++) {
        bulb_step(o, p);
        if (o.m > BULB_BAILOUT) { if (COUNT) evals.it -= iterations - 1u - i; break; } // orbit steps NOT run (roofline accounting only)
    }
    return bulb_finish(o.m, o.dz);
}

// max(a, b) for non-NaN operands as a single instruction
RD float vmax_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Correctly rounded n/d for finite normal n, d whose quotient is a normal number: the same Newton-Raphson
// + residual-correction steps hipcc emits for an IEEE '/', minus v_div_scale / v_div_fixup (which only act
// on extreme exponents and specials).  22 VALU cycles instead of 36.  The host enables it per object only
// when the fold constants keep every operand in [2^-60, 2^60] (DHitable::fast_div); tests check it
// against IEEE division over that whole range.
RD float div_nr(float n, float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    float q = n * r;
    const float e1 = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e1, r, q);
    const float e2 = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e2, r, q);
}

// n/d as v_rcp_f32, one multiply and ONE residual correction (4 instructions, 14 VALU cycles).  Not correctly rounded for
// arbitrary operands - the host enables it per object (DHitable::fast_div == 2) only after k_verify_short_div has compared
// it with the IEEE quotient for the object's numerator (fixed_radius^2) and EVERY float denominator the sphere fold can
// produce, [min_radius^2, fixed_radius^2): 1.3e8 values for the shipped MandelBox, all exact on gfx950.
RD float div_short(float n, float d) {
    const float rc = __builtin_amdgcn_rcpf(d);
    const float q = n * rc;
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), rc, q);
}

// MandelBox scale at the packet time t0 (EXTENSION, rayn_hip.h: the closure |t| scale + scale_vel * t); constants ignore t0
RD float sdf_scale(const DHitable& h, float t0) { return h.scale_vel != 0.0f ? h.scale + h.scale_vel * t0 : h.scale; }

// 'scale' = sdf_scale(h, t0) of the calling packet (h.scale itself in the reference's time-independent case)
//
// MandelBox::dist (src/sdf.rs:125-188), one fold iteration = box_fold, sphere_fold, scale + offset:
//   box_fold     p = clamped(-l, l).mul_add(2, -p)
//   sphere_fold  mul = max(1, R2 / max(r2min, r2));  p *= mul;  dr *= mul
//                For r2 >= R2 (and for NaN) the quotient is <= 1, mul is exactly 1 and the multiplies are identities, so the
//                block only matters for lanes with r2 < R2, where the quotient is >= 1: max(1, q) == q.  It runs under the
//                folding lanes' exec mask (a plain per-lane 'if': v_cmp + s_and_saveexec + s_cbranch_execz, the branch taken
//                only when NO lane of the wave folds).  A lane folds in 2.4-2.9 of the 12 iterations, SOME lane of a wave in
//                9-11.7 of them; every other layout that was measured (wave-uniform branch, branch-free select, dense below
//                a lane count, packed binary32) is slower - DESIGN.md section 4, tools/variants/README.md.
//   scale+offset p = p.mul_add(s, p0);  dr = (-dr).mul_add(s, 1)
// DIV is the quotient R2 / max(r2min, r2) (IEEE '/', div_nr or div_short, chosen by the host per object: DHitable::fast_div),
// BOX the 2c - p of the box fold (mul_add per policy, or one fma where the product 2c is exact).
#define RAYN_FOLD_ITER(DIV, BOX)                                          \
    {                                                                     \
        p.x = BOX(__builtin_amdgcn_fmed3f(p.x, nl, l), p.x);              \
        p.y = BOX(__builtin_amdgcn_fmed3f(p.y, nl, l), p.y);              \
        p.z = BOX(__builtin_amdgcn_fmed3f(p.z, nl, l), p.z);              \
        const float r2 = mag_sq(p);                                       \
        if (r2 < frs_eff) { /* not NaN on a folding lane: ONE raw v_max_f32 */ \
            const float m = DIV(frs, vmax_raw(r2, mrs_v));                \
            p.x *= m; p.y *= m; p.z *= m;                                 \
            dr *= m;                                                      \
        }                                                                 \
        p.x = muladd(p.x, s, offset.x);                                   \
        p.y = muladd(p.y, s, offset.y);                                   \
        p.z = muladd(p.z, s, offset.z);                                   \
        dr = muladd(-dr, s, 1.0f);                                        \
    }
#define RAYN_DIV_IEEE(a, b) ((a) / (b))
#define RAYN_BOX_REF(c, q) muladd(c, 2.0f, -(q))
// |c| <= |l| <= 2^60: the product 2c is exact, so ONE rounding (fma) == the reference's two (mul, add)
#define RAYN_BOX_FMA(c, q) __builtin_fmaf(c, 2.0f, -(q))
#define RAYN_FOLD_X4(DIV, BOX) RAYN_FOLD_ITER(DIV, BOX) RAYN_FOLD_ITER(DIV, BOX) RAYN_FOLD_ITER(DIV, BOX) RAYN_FOLD_ITER(DIV, BOX)
// SDFK >= 0: the SDF kind is known at compile time (the single-SDF march kernels are instantiated per kind: no per-evaluation branch, and the registers
// / scalar constants of the other SDFs are not carried through the march loop); -1: the kind is read from the object
// SDFK == SDFK_MANDELBOX_12S: the MandelBox in the shape the reference ships (12 fold iterations, src/setup.rs:44) with the short division verified for its fold
// constants (DHitable::fast_div == 2) - iteration count and division form are compile-time facts too: the march loop carries no mode dispatch at all.
constexpr int SDFK_MANDELBOX_12S = 100;
template <bool COUNT, int SDFK = -1>
RD float sdf_dist(const DHitable& h, f3 p, EvalCtr& evals, float scale) {
    constexpr bool BOX12S = SDFK == SDFK_MANDELBOX_12S;
    const uint32_t kind = BOX12S ? (uint32_t)RAYN_SDF_MANDELBOX : (SDFK >= 0 ? (uint32_t)SDFK : h.sdf_kind);
    if (COUNT) { evals.n++; evals.it += kind == RAYN_SDF_SPHERE ? 0u : h.iterations; } // the Mandelbulb takes its early exits off again (mandelbulb_dist)
    if (kind == RAYN_SDF_MANDELBOX) {
        const f3 offset = p;
        float dr = 1.0f;
        const float l = h.box_l, nl = -h.box_l, s = scale;
        const float mrs = h.min_rad_sq, frs = h.fixed_rad_sq;
        // if min_rad_sq > fixed_rad_sq the quotient is < 1 for every r2: never enter the block
        const float frs_eff = mrs <= frs ? frs : -1.0f;
        // min_radius^2 once in a vector register: vmax_raw's operands are "v", and left alone the compiler re-materialises the
        // copy from the scalar register inside every fold block (one VALU instruction in a block of eleven)
        float mrs_v = mrs;
        asm("" : "+v"(mrs_v));
        // NaN-free inputs stay NaN-free here and a NaN point yields NaN through '-p', so the hardware
        // med3/max (IEEE maxNum) forms are bit-identical to the reference's SSE max/min semantics.
        if (BOX12S) {
            RAYN_FOLD_X4(div_short, RAYN_BOX_FMA) RAYN_FOLD_X4(div_short, RAYN_BOX_FMA) RAYN_FOLD_X4(div_short, RAYN_BOX_FMA)
        } else if (h.fast_div == 2) {
            uint32_t i = 0;
            if (h.iterations == 12) { // the shipped iteration count (src/setup.rs:44), fully unrolled: 2 % over the rolled loop
                RAYN_FOLD_X4(div_short, RAYN_BOX_FMA) RAYN_FOLD_X4(div_short, RAYN_BOX_FMA) RAYN_FOLD_X4(div_short, RAYN_BOX_FMA)
                i = 12;
            }
            for (; i + 4 <= h.iterations; i += 4) { RAYN_FOLD_X4(div_short, RAYN_BOX_FMA) }
            for (; i < h.iterations; i++) RAYN_FOLD_ITER(div_short, RAYN_BOX_FMA)
        } else if (h.fast_div) {
            for (uint32_t i = 0; i < h.iterations; i++) RAYN_FOLD_ITER(div_nr, RAYN_BOX_FMA)
        } else {
            for (uint32_t i = 0; i < h.iterations; i++) RAYN_FOLD_ITER(RAYN_DIV_IEEE, RAYN_BOX_REF)
        }
        return mag(p) / __builtin_fabsf(dr);
    }
    if (kind == RAYN_SDF_MANDELBULB) return mandelbulb_dist<COUNT>(p, h.iterations, evals);
    return mag(p) - h.sdf_radius;
}
#undef RAYN_FOLD_X4
#undef RAYN_FOLD_ITER
#undef RAYN_DIV_IEEE
#undef RAYN_BOX_REF
#undef RAYN_BOX_FMA

// hit threshold closure, src/film.rs:540-551 (Camera::half_pixel_size_at src/camera.rs:116,210,282)
struct Thr { float k; bool constant; };
RD Thr make_thr(const DScene& sc, uint32_t depth) {
    Thr t;
    if (depth == 0) { t.k = sc.cam.half_pixel_size; t.constant = sc.cam.kind == RAYN_CAM_ORTHOGRAPHIC; }
    else { t.k = 0.0001f * 2.0f * (float)depth; t.constant = false; }
    return t;
}
RD float thr_at(const Thr& th, float t) { return th.constant ? th.k : th.k * t; }

// Hitable origin at the packet's lane-0 time t0 (closure transform_seq, src/animation.rs:62-68); constants ignore t0.
// Sphere::transform_seq in the reference (src/sphere.rs:8-21).  EXTENSION: a TracedSDF honours the same field - the SDF
// is evaluated in the frame translated by this origin (hit: ray origin - origin; occluded: both segment ends - origin;
// get_shading_info: normal at point - origin, the shading point stays in world space).  A zero constant origin, the
// only case the reference has, is an exact no-op (x - 0 == x).
RD f3 sphere_center(const DHitable& h, float t0) { return h.animated ? h.center + h.center_vel * t0 : h.center; }

// sdfu normals_fast (tetrahedron), called at src/sdf.rs:94-96
template <bool COUNT>
RD f3 sdf_normal(const DHitable& h, f3 p, float eps, EvalCtr& evals, float sv) {
    float d1 = sdf_dist<COUNT>(h, f3{p.x + eps, p.y + -eps, p.z + -eps}, evals, sv);
    float d2 = sdf_dist<COUNT>(h, f3{p.x + -eps, p.y + -eps, p.z + eps}, evals, sv);
    float d3 = sdf_dist<COUNT>(h, f3{p.x + -eps, p.y + eps, p.z + -eps}, evals, sv);
    float d4 = sdf_dist<COUNT>(h, f3{p.x + eps, p.y + eps, p.z + eps}, evals, sv);
    f3 g = f3{1.0f * d1, -1.0f * d1, -1.0f * d1} + f3{-1.0f * d2, -1.0f * d2, 1.0f * d2} +
           f3{-1.0f * d3, 1.0f * d3, -1.0f * d3} + f3{1.0f * d4, 1.0f * d4, 1.0f * d4};
    return normalized(g);
}

// ---- Sphere (src/sphere.rs:23-71) ------------------------------------------------------------------

RD float sphere_hit(const DHitable& h, f3 o, f3 d, float t_max, float t0) {
    f3 oc = o - sphere_center(h, t0);
    float b = dot(oc, d);
    float c = mag_sq(oc) - h.radius_sq;
    float descrim = b * b - c;
    bool desc_pos = descrim > 0.0f;
    if (!desc_pos) return 3.40282347e+38f; // both roots are invalid whatever sqrt returns: skip it (exact)
    float desc_sqrt = sqrt_rn(descrim);
    float t1 = -b - desc_sqrt;
    bool t1_valid = (t1 > 0.0001f) && (t1 <= t_max) && desc_pos;
    float t2 = -b + desc_sqrt;
    bool t2_valid = (t2 > 0.0001f) && (t2 <= t_max) && desc_pos;
    bool take_t1 = (t1 < t2) && t1_valid;
    float t = take_t1 ? t1 : t2;
    return (t1_valid || t2_valid) ? t : 3.40282347e+38f;
}
// Sphere::occluded (src/sphere.rs:49-71) with the segment direction and length computed once by the caller
// (they do not depend on the sphere).  Two exact early-outs, both "this sphere cannot occlude":
//  * b >= 0 (sphere centre behind the start): t1 = -b - sqrt(..) is a sum of two non-positive numbers, so
//    min(t1, t2) <= 0 and the reference's 'min > 0.001' test fails;
//  * c < -1e-30 (start inside the sphere, e.g. the sky dome): descrim = fl(fl(b*b) - c) >= fl(b*b), sqrt is
//    monotonic and sqrt(fl(b*b)) == |b| (or descrim >= 1e-30 > b*b when b*b underflows), so again t1 <= 0.
// NaN operands fail both tests and take the literal path.
RD float sphere_occluded_dir(const DHitable& h, f3 start, f3 dir, float dist, float t0) {
    f3 oc = start - sphere_center(h, t0);
    float b = dot(oc, dir);
    if (b >= 0.0f) return 1.0f;
    float c = mag_sq(oc) - h.radius_sq;
    if (c < -1e-30f) return 1.0f;
    float descrim = b * b - c;
    bool desc_pos = descrim > 0.0f;
    if (!desc_pos) return 1.0f; // 'valid' needs desc_pos: the segment cannot be occluded by this sphere (exact)
    float desc_sqrt = sqrt_rn(descrim);
    float t1 = -b - desc_sqrt;
    float t2 = -b + desc_sqrt;
    float mn = fmins(t1, t2);
    bool valid = (mn > 0.001f) && (t1 <= dist) && desc_pos;
    return valid ? 0.0f : 1.0f;
}

// ---- SphereLight (src/light.rs:38-107) ---------------------------------------------------------------
RD void light_sample(const DLight& L, float u0, float u1, f3 p, f3* out_point, float* out_pdf) {
    f3 dir_to_light = L.pos - p;
    float dist_sq = mag_sq(dir_to_light);
    float dist = sqrt_rn(dist_sq);
    dir_to_light = div_by_mag(dir_to_light, dist);
    Basis basis = orthonormal_basis(-dir_to_light);
    float r2 = L.rad * L.rad;
    float sin_theta_max_2 = r2 / dist_sq;
    float cos_theta_max = sqrt_rn(fmaxs(0.0f, 1.0f - sin_theta_max_2));
    float cos_theta = (1.0f - u0) + u0 * cos_theta_max;
    float sin_theta = sqrt_rn(fmaxs(0.0f, 1.0f - cos_theta * cos_theta));
    float phi = u1 * TWO_PI_F;
    float ds = dist * cos_theta - sqrt_rn(fmaxs(0.0f, r2 - dist_sq * sin_theta * sin_theta));
    float cos_alpha = (dist_sq + r2 - ds * ds) / (2.0f * dist * L.rad);
    float sin_alpha = sqrt_rn(fmaxs(0.0f, 1.0f - cos_alpha * cos_alpha));
    float sin_phi, cos_phi;
    dmf_sincosf(phi, &sin_phi, &cos_phi);
    f3 offset = basis.c0 * sin_alpha * cos_phi + basis.c1 * sin_alpha * sin_phi + basis.c2 * cos_alpha;
    *out_point = L.pos + offset * L.rad;
    *out_pdf = 1.0f / (TWO_PI_F * (1.0f - cos_theta_max)); // uniform_cone_pdf
}
RD void light_sample_volume(const DLight& L, float sample, f3 ray_o, f3 ray_d, float max_distance, float* out_dist, float* out_pdf) {
    float delta = dot(L.pos - ray_o, ray_d);
    f3 closest_point = ray_o + delta * ray_d;
    float d = mag(closest_point - L.pos);
    float theta_a = dmf_atan2f(-delta, d);
    float theta_b = dmf_atan2f(max_distance - delta, d);
    float t = d * dmf_tanf(lerpf(theta_a, theta_b, sample));
    *out_dist = delta + t;
    *out_pdf = d / ((theta_b - theta_a) * muladd(d, d, t * t));
}

// ---- BSDFs (src/material.rs) -----------------------------------------------------------------------------
// BSDF::f as CALLED: f(wo, wi, n).  DielectricBSDF declares the parameters as (wi, wo, n)
// (src/material.rs:195), so its 'wi' is the caller's wo — kept literally.
RD f3 bsdf_f(const DMaterial& m, f3 arg0, f3 arg1, f3 n) {
    if (m.kind == RAYN_MAT_DIELECTRIC) {
        float dt = fmaxs(0.0f, dot(arg0, n));
        float fresnel = f_schlick(dt, 0.04f);
        f3 half = normalized(arg1 + arg0);
        float cos_alpha = dmf_powf(fmaxs(0.0f, dot(half, n)), m.exponent);
        float spec_factor = cos_alpha * (m.exponent + 2.0f) / (2.0f * PI_F);
        f3 spec_f = f3{1.0f, 1.0f, 1.0f} * spec_factor * fresnel;
        f3 diffuse_f = m.a / PI_F * (1.0f - fresnel);
        return spec_f + diffuse_f;
    }
    if (m.kind == RAYN_MAT_LAMBERTIAN) return m.a / PI_F;
    return f3{0.0f, 0.0f, 0.0f}; // Emissive (:503-505); Sky::f panics in the reference and is never reached
}
RD f3 bsdf_le(const DMaterial& m, f3 wo) {
    if (m.kind == RAYN_MAT_SKY) { // :444-448
        float t = 0.5f * (wo.y + 1.0f);
        return m.a * (1.0f - t) + m.b * t;
    }
    if (m.kind == RAYN_MAT_EMISSIVE) return m.a;
    return f3{0.0f, 0.0f, 0.0f};
}
struct Scatter { f3 wi, f; float pdf; };
RD Scatter bsdf_scatter(const DMaterial& m, f3 wo, f3 normal, const Basis& basis, float s1, float u0, float u1, float u2, float u3) {
    Scatter se;
    if (m.kind == RAYN_MAT_DIELECTRIC) { // :207-256
        float cosv = __builtin_fabsf(dot(normal, wo));
        f3 diffuse_sample = cosine_weighted_in_hemisphere(u0, u1);
        f3 diffuse_bounce = normalized(mul(basis, diffuse_sample));
        float diffuse_pdf = fmaxs(0.00001f, diffuse_sample.z / PI_F);
        f3 diffuse_f = m.a / PI_F;
        f3 spec_sample = cosine_power_weighted(u2, u3, m.exponent);
        f3 reflection = reflected(wo, normal);
        Basis rb = orthonormal_basis(reflection);
        f3 spec_bounce = normalized(mul(rb, spec_sample));
        float cos_alpha_pow = fmaxs(dmf_powf(spec_sample.z, m.exponent), EPSILON_F);
        float spec_pdf = (m.exponent + 1.0f) / TWO_PI_F * cos_alpha_pow;
        float spec_coeff = (m.exponent + 2.0f) / TWO_PI_F * cos_alpha_pow;
        if (dot(normal, spec_bounce) < 0.0f) spec_coeff = 0.0f;
        f3 spec_f = f3{1.0f, 1.0f, 1.0f} * spec_coeff;
        float fresnel = f_schlick(cosv, 0.04f);
        bool pick_spec = s1 < fresnel;
        se.wi = pick_spec ? spec_bounce : diffuse_bounce;
        se.f = pick_spec ? spec_f : diffuse_f;
        se.pdf = fresnel * spec_pdf + (1.0f - fresnel) * diffuse_pdf;
    } else { // Lambertian :118-137
        f3 diffuse_sample = cosine_weighted_in_hemisphere(u0, u1);
        se.wi = normalized(mul(basis, diffuse_sample));
        se.pdf = diffuse_sample.z / PI_F;
        se.f = m.a / PI_F;
    }
    return se;
}

// ---- cameras (src/camera.rs) ---------------------------------------------------------------------------------
// t0 = the ray time of lane 0 of the ray-gen packet: closure-sequenced parameters are evaluated there for all
// four lanes (WSequenced<Wec3> for F: Fn(f32) -> Vec3, src/animation.rs:62-68)
RD void camera_ray(const DCamera& c, float uvx, float uvy, float lens0, float lens1, float t0, f3* out_o, f3* out_d) {
    f3 o = c.origin, a = c.at, u = c.up, fo = c.focus;
    if (c.animated) {
        if (c.animated & 1u) o = c.origin + c.origin_vel * t0;
        if (c.animated & 2u) a = c.at + c.at_vel * t0;
        if (c.animated & 4u) u = c.up + c.up_vel * t0;
        if (c.animated & 8u) fo = c.focus + c.focus_vel * t0;
    }
    if (c.kind == RAYN_CAM_PINHOLE) { // :81-114
        f3 basis_w = normalized(o - a);
        f3 basis_u = normalized(cross(u, basis_w));
        f3 basis_v = cross(basis_w, basis_u);
        f3 lower_left = o - basis_u * c.half_w - basis_v * c.half_h - basis_w;
        f3 horiz = basis_u * c.half_w * 2.0f * uvx;
        f3 verti = basis_v * c.half_h * 2.0f * uvy;
        *out_o = o;
        *out_d = normalized(lower_left + horiz + verti - o);
    } else if (c.kind == RAYN_CAM_THIN_LENS) { // :168-208
        float focus_dist = mag(fo - o);
        f3 basis_w = normalized(o - a);
        f3 basis_u = normalized(cross(u, basis_w));
        f3 basis_v = cross(basis_w, basis_u);
        f3 lower_left = o - basis_u * c.half_w * focus_dist - basis_v * c.half_h * focus_dist - basis_w * focus_dist;
        f3 horiz = basis_u * c.half_w * focus_dist * 2.0f * uvx;
        f3 verti = basis_v * c.half_h * focus_dist * 2.0f * uvy;
        float rx, ry;
        concentric_circle_map(lens0, lens1, &rx, &ry);
        rx = rx * c.aperture; ry = ry * c.aperture;
        f3 offset = basis_u * rx + basis_v * ry;
        f3 o2 = o + offset;
        *out_o = o2;
        *out_d = normalized(lower_left + horiz + verti - o2);
    } else { // orthographic :249-280
        f3 basis_w = normalized(a - o);
        f3 basis_u = normalized(cross(basis_w, u));
        f3 basis_v = cross(basis_u, basis_w);
        f3 lower_left = o - basis_u * c.half_w - basis_v * c.half_h;
        f3 offset = basis_u * uvx * c.full_w + basis_v * uvy * c.full_h;
        *out_o = lower_left + offset;
        *out_d = basis_w;
    }
}

// `(x * n_lights).floor() as usize`, clamped where the reference would index out of bounds.
RD uint32_t light_index(float s, uint32_t nl) {
    float f = __builtin_floorf(s * (float)nl);
    if (!(f > 0.0f)) return 0;
    uint32_t i = f >= 4.0e9f ? 0xFFFFFFFFu : (uint32_t)f;
    return i < nl ? i : nl - 1;
}

} // namespace RAYN_KNS
