/* rayn_oracle.cpp — CPU restatement of fu5ha/rayn's per-sample integrator hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  Nothing under rayn_amd/ links, imports or calls it.
 *
 * What it is: a packet-of-4 ("f32x4"), tile-serial, trait-object-shaped C++ restatement of
 *   Film::render_frame_into            src/film.rs:382-691
 *   PathTracingIntegrator::integrate   src/integrator.rs:38-281
 *   TracedSDF / MandelBox              src/sdf.rs:9-188
 *   HitStore / HitableStore            src/hitable.rs:20-211
 *   Sphere                             src/sphere.rs:23-87
 *   Lambertian/Dielectric/Sky/Emissive src/material.rs:11-257,394-520
 *   SphereLight                        src/light.rs:19-107
 *   Pinhole/ThinLens/Ortho cameras     src/camera.rs:41-285
 *   Samples                            src/sampler.rs:17-126
 *   BlackmanHarris + FIS + CDF         src/filter.rs:29-49,187-236, src/math.rs:136-191
 *   ONB, sampling maps, schlick        src/math.rs:45-124,201-219
 * written in the same order of floating-point operations as the Rust, one function per reference
 * function, each citing the lines it follows.  It deliberately shares NO code with the HIP path
 * except include/rayn_hip.h (the POD scene description both consume) and include/rayn_detmath.h
 * (the pinned exp/sin/cos/tan/atan2/pow and the FMA policy — see that header for why).
 *
 * PARITY UNPINNED.  The reference has no tests, golden vectors or fixtures and
 * cannot be built here (no Rust toolchain), so this oracle could not be checked
 * against rayn itself.  Semantics that live in un-vendored crates are restated from their
 * published behaviour and are assumptions, listed here so a maintainer with a Rust toolchain can
 * check each one:
 *   A1 wide 0.4.6 f32x4::mul_add is an unfused a*b+c on the default x86-64 target (no RUSTFLAGS
 *      in the repo).  RAYN_FMA_POLICY=1 switches both sides to fused.
 *   A2 wide f32x4::max/min are SSE maxps/minps: a.max(b) = a > b ? a : b (b when unordered).
 *   A3 wide transcendentals = correctly rounded libm (here: rayn_detmath.h).
 *   A4 ultraviolet 0.4.6: dot = x.mul_add(ox, y.mul_add(oy, z*oz)); mag = sqrt(mag_sq);
 *      normalize multiplies by r_mag = 1/mag; cross = (y.mul_add(oz, -z*oy), ...);
 *      reflected(n) = v - (2*dot(v,n))*n; clamped = max(min).min(max) per component;
 *      Mat3*Vec3 = c0*x + c1*y + c2*z; component_max = x.max(y).max(z).
 *   A5 sdfu 0.3.0: normals_fast = tetrahedral estimator; Lerp::lerp(a,b,t) = a*(1-t) + b*t;
 *      Sphere::dist = |p| - r.
 *   A6 quasi-rd @ce117035: Roberts R_d, x_k = frac(1/2 + alpha*(offset+1+k)) in exact modular
 *      arithmetic, alpha = phi_d^-j, top 24 bits -> f32.  (Tables are INPUTS to the GPU path.)
 *   A7 rand 0.7.2 SmallRng = Pcg64Mcg; seed_from_u64 = PCG32 expansion; gen::<f32>() =
 *      (next_u32() >> 8) * 2^-24.  (Scrambles are INPUTS to the GPU path.)  The two generator
 *      cores ARE pinned to external known-answer vectors (tests/test_tables.py: rand_pcg's
 *      Mcg128Xsl64::new(42) outputs, the PCG demo's pcg32_srandom(42, 54) outputs); the seed
 *      expansion's word order and the float conversion remain assumptions.
 *   A8 f32::signum(+-0) = +-1, NaN -> NaN; `as usize` saturates (NaN -> 0); light indices are
 *      clamped to n_lights-1 where the reference would panic on an out-of-range index.
 *   A9 SCALAR arithmetic in scene construction: Rust's scalar f32::mul_add is always a fused multiply-add (std uses libm's fmaf
 *      when the target has no FMA unit), independent of A1's policy for the wide types.  If ultraviolet's scalar Vec3 shares
 *      the mul_add forms of A4 (unverified), Vec3::mag_sq / dot / normalized are fused on the host under BOTH policies.  The
 *      only scalar Vec3 call sites on the shipped scene's path are the two Srgb::normalized() of src/setup.rs:100-101 (the
 *      commented-out sun at src/setup.rs:88-97 would add a Vec3::normalized); everything else in src/setup.rs:46-170 is
 *      f32 * / + - and tan().  This file never sees them: it consumes the flattened POD.  The host mirrors that build the POD
 *      (rayn_amd/scene.py normalized(), include/rayn_host.hpp Vec3::normalized) evaluate the fused form unconditionally; for
 *      the shipped constants every product is exact, so the POD is the same either way (tests/test_oracle.py).
 *
 * ALTERNATIVE READINGS (round 4; oracle/SENSITIVITY.md).  Each assumption above that can change a pixel has a build switch
 * that compiles the OTHER plausible reading into a variant library (oracle/Makefile `variants`; never the default, never used
 * by the parity tests): whoever has a Rust toolchain runs tools/pin_against_rayn.sh, and if the default disagrees with rayn,
 * the variant that agrees names the assumption that was wrong.  oracle/sensitivity.py measures how far each variant moves
 * whole frames, i.e. which assumptions exceed north_star's per-pixel 1e-4 and must be checked first.
 *   RAYN_ORACLE_ALT_MINMAX=1   A2: a.max(b) = maxps(b, a) (operands swapped: a when unordered / both zero)
 *   RAYN_ORACLE_ALT_MINMAX=2   A2: lane-wise f32::max / f32::min (IEEE maxNum / minNum: the non-NaN operand)
 *   RAYN_ORACLE_LIBM=1         A3: the host libm's expf / sinf / cosf / tanf / atan2f / powf (what lane-wise f32::exp etc. call on Linux)
 *   RAYN_ORACLE_ALT_NORMALIZE=1 A4: normalized() divides the three components by mag instead of multiplying by 1 / mag
 *   RAYN_ORACLE_ALT_DOT=1      A4: dot = (x*ox + y*oy) + z*oz, plain left-to-right products and sums (no mul_add nesting)
 *   RAYN_ORACLE_ALT_NORMALS=1  A5: central differences (sdfu `normals`, 6 evaluations) instead of the tetrahedral `normals_fast`
 *   RAYN_ORACLE_ALT_NORMALS=2  A5: the tetrahedral estimator with the four terms summed in the order xxx, xyy, yxy, yyx
 *   RAYN_ORACLE_ALT_LERP=1     A5: Lerp::lerp(a, b, t) = a + (b - a) * t
 */
#include "../include/rayn_detmath.h"
#define DMF_COUNT_FALLBACKS /* host check of the kernels' fast elementary functions (oracle_detmath_fast); the oracle's path does not use them */
#include "../include/rayn_detmath_fast.h"
#include "../include/rayn_hip.h"

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

namespace {

/* ------------------------------------------------------------------ f32x4 (wide 0.4.6) ---- */
/* f32x4 as a real SSE vector (GCC vector extension): packed IEEE operations give each lane exactly the scalar result, and the
 * CPU baseline this file also serves as is then SIMD like the reference's `wide` f32x4 rather than four scalar lanes. */
typedef float f4v __attribute__((vector_size(16)));
typedef int i4v __attribute__((vector_size(16)));
struct F4 {
    union { f4v q; float v[4]; };
    F4() : q{0, 0, 0, 0} {}
    F4(float s) : q{s, s, s, s} {}
    F4(float a, float b, float c, float d) : q{a, b, c, d} {}
    explicit F4(f4v x) : q(x) {}
    float& operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
};
struct M4 { /* lane mask (comparison result): all-ones / all-zeros per lane */
    union { i4v q; int v[4]; };
    M4() : q{0, 0, 0, 0} {}
    explicit M4(i4v x) : q(x) {}
    int move_mask() const { return __builtin_ia32_movmskps((f4v)q); }
};
inline F4 operator+(F4 a, F4 b) { return F4(a.q + b.q); }
inline F4 operator-(F4 a, F4 b) { return F4(a.q - b.q); }
inline F4 operator*(F4 a, F4 b) { return F4(a.q * b.q); }
inline F4 operator/(F4 a, F4 b) { return F4(a.q / b.q); }
inline F4 operator-(F4 a) { return F4((f4v)((i4v)a.q ^ (i4v){(int)0x80000000, (int)0x80000000, (int)0x80000000, (int)0x80000000})); } /* sign flip, also of zero / NaN */
inline F4& operator+=(F4& a, F4 b) { a = a + b; return a; }
inline F4& operator*=(F4& a, F4 b) { a = a * b; return a; }
inline F4& operator/=(F4& a, F4 b) { a = a / b; return a; }
inline M4 cmp_lt(F4 a, F4 b) { return M4(a.q < b.q); }
inline M4 cmp_le(F4 a, F4 b) { return M4(a.q <= b.q); }
inline M4 cmp_gt(F4 a, F4 b) { return M4(a.q > b.q); }
inline M4 cmp_eq(F4 a, F4 b) { return M4(a.q == b.q); }
inline M4 cmp_nan(F4 a, F4 b) { return M4((a.q != a.q) | (b.q != b.q)); }
inline M4 operator|(M4 a, M4 b) { return M4(a.q | b.q); }
inline M4 operator&(M4 a, M4 b) { return M4(a.q & b.q); }
inline M4 operator!(M4 a) { return M4(~a.q); }
inline F4 merge(M4 m, F4 t, F4 f) { return F4((f4v)(((i4v)t.q & m.q) | ((i4v)f.q & ~m.q))); }
/* A2: SSE semantics: maxps(a, b) = a > b ? a : b (b when unordered or equal), minps likewise */
#if !defined(RAYN_ORACLE_ALT_MINMAX) || RAYN_ORACLE_ALT_MINMAX == 0
inline F4 fmax4(F4 a, F4 b) { return F4(__builtin_ia32_maxps(a.q, b.q)); }
inline F4 fmin4(F4 a, F4 b) { return F4(__builtin_ia32_minps(a.q, b.q)); }
#elif RAYN_ORACLE_ALT_MINMAX == 1 /* alternative reading: the intrinsic's operands the other way round */
inline F4 fmax4(F4 a, F4 b) { return F4(__builtin_ia32_maxps(b.q, a.q)); }
inline F4 fmin4(F4 a, F4 b) { return F4(__builtin_ia32_minps(b.q, a.q)); }
#else /* alternative reading: lane-wise f32::max / f32::min = IEEE maxNum / minNum */
inline F4 fmax4(F4 a, F4 b) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = __builtin_fmaxf(a.v[i], b.v[i]); return r; }
inline F4 fmin4(F4 a, F4 b) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = __builtin_fminf(a.v[i], b.v[i]); return r; }
#endif
inline F4 abs4(F4 a) { return F4((f4v)((i4v)a.q & (i4v){0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff})); }
inline F4 sqrt4(F4 a) { return F4(__builtin_ia32_sqrtps(a.q)); }
inline F4 floor4(F4 a) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = __builtin_floorf(a.v[i]); return r; }
#if RAYN_FMA_POLICY
inline F4 mul_add(F4 a, F4 b, F4 c) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = rayn_muladd(a.v[i], b.v[i], c.v[i]); return r; } /* A1, fused build */
#else
inline F4 mul_add(F4 a, F4 b, F4 c) { return F4(a.q * b.q + c.q); } /* A1: unfused (-ffp-contract=off keeps it two roundings) */
#endif
inline float signum1(float x) { return x != x ? x : __builtin_copysignf(1.0f, x); } /* A8 */
inline F4 signum4(F4 a) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = signum1(a.v[i]); return r; }
#if defined(RAYN_ORACLE_LIBM) && RAYN_ORACLE_LIBM /* alternative reading of A3: the platform libm, lane by lane */
inline F4 exp4(F4 a) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = ::expf(a.v[i]); return r; }
inline F4 tan4(F4 a) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = ::tanf(a.v[i]); return r; }
inline F4 atan2_4(F4 y, F4 x) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = ::atan2f(y.v[i], x.v[i]); return r; }
inline F4 powf4(F4 a, F4 b) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = ::powf(a.v[i], b.v[i]); return r; }
inline void sin_cos4(F4 a, F4* s, F4* c) { for (int i = 0; i < 4; i++) { s->v[i] = ::sinf(a.v[i]); c->v[i] = ::cosf(a.v[i]); } }
#else
inline F4 exp4(F4 a) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = dm_expf(a.v[i]); return r; }
inline F4 tan4(F4 a) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = dm_tanf(a.v[i]); return r; }
inline F4 atan2_4(F4 y, F4 x) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = dm_atan2f(y.v[i], x.v[i]); return r; }
inline F4 powf4(F4 a, F4 b) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = dm_powf(a.v[i], b.v[i]); return r; }
inline void sin_cos4(F4 a, F4* s, F4* c) { for (int i = 0; i < 4; i++) dm_sincosf(a.v[i], &s->v[i], &c->v[i]); }
#endif
inline F4 powi5(F4 a) { F4 r; for (int i = 0; i < 4; i++) { float x = a.v[i], x2 = x * x; r.v[i] = (x2 * x2) * x; } return r; } /* powi(5) */
#if defined(RAYN_ORACLE_ALT_LERP) && RAYN_ORACLE_ALT_LERP /* alternative reading of A5 */
inline F4 lerp4(F4 a, F4 b, F4 t) { return a + (b - a) * t; }
inline float lerp1(float a, float b, float t) { return a + (b - a) * t; }
#else
inline F4 lerp4(F4 a, F4 b, F4 t) { return a * (F4(1.0f) - t) + b * t; } /* A5 */
inline float lerp1(float a, float b, float t) { return a * (1.0f - t) + b * t; }
#endif

const float PI_F = 3.14159265358979323846f;
const float TWO_PI_F = 6.28318530717958647692f;
const float FRAC_PI_2_F = 1.57079632679489661923f;
const float FRAC_PI_4_F = 0.78539816339744830962f;
const float EPSILON_F = 1.1920929e-7f;

/* ------------------------------------------------------- Vec3 / Wec3 (ultraviolet 0.4.6, A4) */
struct V3 { float x, y, z; };
struct W3 {
    F4 x, y, z;
    W3() {}
    W3(F4 a, F4 b, F4 c) : x(a), y(b), z(c) {}
    static W3 splat(V3 v) { return W3(F4(v.x), F4(v.y), F4(v.z)); }
    static W3 broadcast(F4 s) { return W3(s, s, s); }
};
inline W3 operator+(W3 a, W3 b) { return W3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline W3 operator-(W3 a, W3 b) { return W3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline W3 operator*(W3 a, W3 b) { return W3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline W3 operator*(W3 a, F4 s) { return W3(a.x * s, a.y * s, a.z * s); }
inline W3 operator*(F4 s, W3 a) { return W3(s * a.x, s * a.y, s * a.z); }
inline W3 operator/(W3 a, F4 s) { return W3(a.x / s, a.y / s, a.z / s); }
inline W3 operator-(W3 a) { return W3(-a.x, -a.y, -a.z); }
inline W3& operator+=(W3& a, W3 b) { a = a + b; return a; }
inline W3& operator*=(W3& a, F4 s) { a = a * s; return a; }
inline W3& operator/=(W3& a, F4 s) { a = a / s; return a; }
#if defined(RAYN_ORACLE_ALT_DOT) && RAYN_ORACLE_ALT_DOT /* alternative reading of A4: plain products summed left to right */
inline F4 dot(W3 a, W3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
#else
inline F4 dot(W3 a, W3 b) { return mul_add(a.x, b.x, mul_add(a.y, b.y, a.z * b.z)); }
#endif
inline F4 mag_sq(W3 a) { return dot(a, a); }
inline F4 mag(W3 a) { return sqrt4(mag_sq(a)); }
#if defined(RAYN_ORACLE_ALT_NORMALIZE) && RAYN_ORACLE_ALT_NORMALIZE /* alternative reading of A4: three divisions */
inline W3 normalized(W3 a) { F4 m = mag(a); return W3(a.x / m, a.y / m, a.z / m); }
#else
inline W3 normalized(W3 a) { F4 r_mag = F4(1.0f) / mag(a); return W3(a.x * r_mag, a.y * r_mag, a.z * r_mag); }
#endif
inline W3 cross(W3 a, W3 b) {
    return W3(mul_add(a.y, b.z, -a.z * b.y), mul_add(a.z, b.x, -a.x * b.z), mul_add(a.x, b.y, -a.y * b.x));
}
inline W3 mul_add(W3 a, W3 m, W3 c) { return W3(mul_add(a.x, m.x, c.x), mul_add(a.y, m.y, c.y), mul_add(a.z, m.z, c.z)); }
inline W3 clamped(W3 a, W3 lo, W3 hi) { return W3(fmin4(fmax4(a.x, lo.x), hi.x), fmin4(fmax4(a.y, lo.y), hi.y), fmin4(fmax4(a.z, lo.z), hi.z)); }
inline F4 component_max(W3 a) { return fmax4(fmax4(a.x, a.y), a.z); }
inline W3 reflected(W3 v, W3 n) { return v - (F4(2.0f) * dot(v, n)) * n; }
inline W3 merge(M4 m, W3 t, W3 f) { return W3(merge(m, t.x, f.x), merge(m, t.y, f.y), merge(m, t.z, f.z)); }
inline V3 lane(W3 a, int i) { return V3{a.x.v[i], a.y.v[i], a.z.v[i]}; }
inline void set_lane(W3& a, int i, V3 v) { a.x.v[i] = v.x; a.y.v[i] = v.y; a.z.v[i] = v.z; }
struct Wat3 { W3 cols[3]; };
inline W3 operator*(const Wat3& m, W3 v) { return m.cols[0] * v.x + m.cols[1] * v.y + m.cols[2] * v.z; }
struct W2 { F4 x, y; };
inline F4 mag_sq(W2 a) { return mul_add(a.x, a.x, a.y * a.y); }
typedef W3 WSrgb; /* src/spectrum.rs: newtype over Wec3 */
typedef V3 Srgb;

/* src/math.rs:49-59 */
inline Wat3 get_orthonormal_basis(W3 nor) {
    F4 ks = signum4(nor.z);
    F4 ka = F4(1.0f) / (F4(1.0f) + abs4(nor.z));
    F4 kb = -ks * nor.x * nor.y * ka;
    W3 uu(F4(1.0f) - nor.x * nor.x * ka, ks * kb, -ks * nor.x);
    W3 vv(kb, ks - nor.y * nor.y * ka * ks, -nor.y);
    Wat3 m; m.cols[0] = uu; m.cols[1] = vv; m.cols[2] = nor; return m;
}
/* src/math.rs:201-219 */
inline W2 concentric_circle_map(F4 u0, F4 u1) {
    F4 two(2.0f);
    F4 a = mul_add(u0, two, -F4(1.0f));
    F4 b = mul_add(u1, two, -F4(1.0f));
    M4 zero_mask = cmp_eq(a, F4(0.0f)) & cmp_eq(b, F4(0.0f));
    b = merge(zero_mask, F4(0.0001f), b);
    F4 phi1 = F4(FRAC_PI_4_F) * b / a;
    F4 phi2 = mul_add(-F4(FRAC_PI_4_F) / b, a, F4(FRAC_PI_2_F));
    M4 mask = cmp_gt(a * a, b * b);
    F4 r = merge(mask, a, b);
    F4 phi = merge(mask, phi1, phi2);
    F4 s, c; sin_cos4(phi, &s, &c);
    return W2{r * c, r * s};
}
/* src/math.rs:99-103 */
inline W3 cosine_weighted_in_hemisphere(F4 u0, F4 u1) {
    W2 xy = concentric_circle_map(u0, u1); /* rand_in_unit_disk, src/math.rs:68-74 */
    F4 z = sqrt4(F4(1.0f) - fmin4(mag_sq(xy), F4(1.0f)));
    return W3(xy.x, xy.y, z);
}
/* src/math.rs:106-113 (azimuth really is 2*u radians) */
inline W3 cosine_power_weighted(F4 u0, F4 u1, F4 power) {
    F4 two(2.0f);
    F4 a = powf4(u0, F4(1.0f) / (power + F4(1.0f)));
    F4 a2 = a * a;
    F4 b = sqrt4(F4(1.0f) - a2);
    F4 s, c; sin_cos4(two * u1, &s, &c);
    return W3(b * c, b * s, a);
}
/* src/math.rs:122-124 */
inline F4 f_schlick(F4 cos, F4 f0) { return f0 + (F4(1.0f) - f0) * powi5(F4(1.0f) - cos); }

/* ------------------------------------------------------------------ rays (src/ray.rs:4-90) */
struct Ray {
    float time; V3 origin, dir; Srgb radiance, throughput;
    uint32_t tcx, tcy; bool valid; float scramble; size_t sample;
};
struct WRay {
    F4 time; W3 origin, dir; WSrgb radiance, throughput;
    uint32_t tcx[4], tcy[4]; bool valid[4]; float scramble[4]; size_t sample[4];
    W3 point_at(F4 t) const { return mul_add(dir, W3(t, t, t), origin); } /* src/ray.rs:21-23 */
};
inline Ray ray_new_invalid() { /* src/ray.rs:54-66 */
    float n = dm_nanf();
    return Ray{n, {n, n, n}, {n, n, n}, {0, 0, 0}, {0, 0, 0}, 0, 0, false, 0.0f, 0};
}
inline WRay wray_from(const Ray r[4]) { /* src/ray.rs:114-160 */
    WRay w;
    for (int i = 0; i < 4; i++) {
        w.time.v[i] = r[i].time; set_lane(w.origin, i, r[i].origin); set_lane(w.dir, i, r[i].dir);
        set_lane(w.radiance, i, r[i].radiance); set_lane(w.throughput, i, r[i].throughput);
        w.tcx[i] = r[i].tcx; w.tcy[i] = r[i].tcy; w.valid[i] = r[i].valid;
        w.scramble[i] = r[i].scramble; w.sample[i] = r[i].sample;
    }
    return w;
}
inline void wray_into(const WRay& w, Ray r[4]) { /* src/ray.rs:162-214 */
    for (int i = 0; i < 4; i++)
        r[i] = Ray{w.time.v[i], lane(w.origin, i), lane(w.dir, i), lane(w.radiance, i), lane(w.throughput, i),
                   w.tcx[i], w.tcy[i], w.valid[i], w.scramble[i], w.sample[i]};
}

typedef std::function<F4(F4)> ThresholdFn; /* &dyn Fn(f32x4) -> f32x4 */

struct Hit { Ray ray; float t; };   /* src/hitable.rs:50-54 */
struct WHit { WRay ray; F4 t; W3 point() const { return ray.point_at(t); } }; /* :56-67 */

/* src/hitable.rs:20-48 */
struct WShadingPoint {
    WRay ray; F4 t; W3 point; F4 offset_by; W3 normal; Wat3 basis;
    static WShadingPoint make(const WHit& hit, W3 point, F4 offset_by, W3 normal) {
        WShadingPoint s; s.ray = hit.ray; s.t = hit.t; s.point = point; s.offset_by = offset_by;
        s.normal = normal; s.basis = get_orthonormal_basis(normal); return s;
    }
    WRay create_rays(W3 dir) const {
        WRay r = ray;
        r.origin = point + normal * signum4(dot(normal, dir)) * offset_by;
        r.dir = dir;
        return r;
    }
};

struct Counters { uint64_t paths = 0, segments = 0, packets = 0; }; /* one per worker thread */
thread_local uint64_t tl_dist_evals = 0;

/* ------------------------------------------------------------------ SDFs (sdfu::SDF) ------ */
struct SDF {
    virtual ~SDF() {}
    virtual F4 dist(W3 p) const = 0;
    /* EXTENSION hook: an SDF with closure-sequenced parameters evaluates them at t0 = lane 0's time of the calling packet
     * (src/animation.rs:62-68); the reference's SDFs ignore time, which is the default here. */
    virtual F4 dist_at(W3 p, float /*t0*/) const { return dist(p); }
};
struct SphereSDF : SDF { /* sdfu::Sphere (A5) */
    F4 radius;
    F4 dist(W3 p) const override { tl_dist_evals++; return mag(p) - radius; }
};
struct MandelBox : SDF { /* src/sdf.rs:104-188 */
    size_t iterations; F4 scale; W3 scale_vec;
    float scale_base = 0.0f, scale_vel = 0.0f; /* EXTENSION: scale(t) = scale_base + scale_vel * t when scale_vel != 0 */
    W3 l, neg_l, two;          /* BoxFold, src/sdf.rs:143-158 */
    F4 min_rad_sq, fixed_rad_sq; /* SphereFold, src/sdf.rs:165-179 */
    MandelBox(size_t it, float side, float min_radius, float fixed_radius, float sc) {
        iterations = it; scale = F4(sc); scale_vec = W3::broadcast(F4(sc)); scale_base = sc;
        l = W3::broadcast(F4(side)); neg_l = -l; two = W3::broadcast(F4(2.0f));
        min_rad_sq = F4(min_radius * min_radius); fixed_rad_sq = F4(fixed_radius * fixed_radius);
    }
    void box_fold(W3& p) const { p = mul_add(clamped(p, neg_l, l), two, -p); } /* :160-162 */
    void sphere_fold(W3& p, F4& dr) const {                                    /* :181-187 */
        F4 r2 = mag_sq(p);
        F4 mul = fmax4(F4(1.0f), fixed_rad_sq / fmax4(min_rad_sq, r2));
        p *= mul; dr *= mul;
    }
    F4 dist_with(W3 p, F4 sc) const { /* :126-140 with the scale as an argument */
        tl_dist_evals++;
        W3 offset = p; F4 one(1.0f); F4 dr = one;
        const W3 sv = W3::broadcast(sc);
        for (size_t i = 0; i < iterations; i++) {
            box_fold(p);
            sphere_fold(p, dr);
            p = mul_add(p, sv, offset);
            dr = mul_add(-dr, sc, one);
        }
        return mag(p) / abs4(dr);
    }
    F4 dist(W3 p) const override { return dist_with(p, scale); }
    F4 dist_at(W3 p, float t0) const override { return scale_vel != 0.0f ? dist_with(p, F4(scale_base + scale_vel * t0)) : dist_with(p, scale); }
};

/* EXTENSION, not in the reference: power-8 Mandelbulb DE, polynomial form (I. Quilez, "Mandelbulb"),
 * bailout |w|^2 > 256 per lane, d = 0.25 ln(m) sqrt(m) / dz.  f32, this exact operation order. */
struct Mandelbulb : SDF {
    size_t iterations;
    F4 dist(W3 pp) const override {
        tl_dist_evals++;
        F4 out;
        for (int l = 0; l < 4; l++) {
            const float px = pp.x.v[l], py = pp.y.v[l], pz = pp.z.v[l];
            float wx = px, wy = py, wz = pz;
            float m = wx * wx + wy * wy + wz * wz, dz = 1.0f;
            for (size_t i = 0; i < iterations; i++) {
                float m2 = m * m, m4 = m2 * m2;
                dz = 8.0f * __builtin_sqrtf(m4 * m2 * m) * dz + 1.0f;
                float x = wx, x2 = x * x, x4 = x2 * x2;
                float y = wy, y2 = y * y, y4 = y2 * y2;
                float z = wz, z2 = z * z, z4 = z2 * z2;
                float k3 = x2 + z2;
                float k2 = 1.0f / __builtin_sqrtf(k3 * k3 * k3 * k3 * k3 * k3 * k3);
                float k1 = x4 + y4 + z4 - 6.0f * y2 * z2 - 6.0f * x2 * y2 + 2.0f * z2 * x2;
                float k4 = x2 - y2 + z2;
                wx = px + 64.0f * x * y * z * (x2 - z2) * k4 * (x4 - 6.0f * x2 * z2 + z4) * k1 * k2;
                wy = py + -16.0f * y2 * k3 * k4 * k4 + k1 * k1;
                wz = pz + -8.0f * y * k4 * (x4 * x4 - 28.0f * x4 * x2 * z2 + 70.0f * x4 * z4 - 28.0f * x2 * z2 * z4 + z4 * z4) * k1 * k2;
                m = wx * wx + wy * wy + wz * wz;
                if (m > 256.0f) break;
            }
            out.v[l] = 0.25f * dm_logf(m) * __builtin_sqrtf(m) / dz;
        }
        return out;
    }
};

/* ------------------------------------------------------------------ Hitable trait --------- */
struct ShadingInfo { size_t material; WShadingPoint sp; };
struct Config { uint32_t max_marches, max_vis_marches; float detail_scale; };
struct Hitable {
    virtual ~Hitable() {}
    virtual F4 hit(const WRay& rays, F4 t_max, const ThresholdFn& thr) const = 0;
    virtual F4 occluded(W3 start, W3 end, F4 time) const = 0;
    virtual ShadingInfo get_shading_info(const WHit& hit, const ThresholdFn& hps) const = 0;
};

/* DIAGNOSTICS (tools/coherence_sim.cpp only; never set by the tests): when a sink is installed through
 * oracle_set_shadow_sink, every lane of every TracedSDF::occluded call appends {depth, NEE sample index, start.xyz, end.xyz}.
 * Single-threaded use (render one tile with threads = 1). */
static std::vector<float>* g_shadow_sink = nullptr;
static float g_sink_depth = 0.0f, g_sink_sample = 0.0f;

/* src/sdf.rs:12-102 */
struct TracedSDF : Hitable {
    std::unique_ptr<SDF> sdf; size_t material; Config cfg;
    /* EXTENSION (not in the reference, whose TracedSDF has no transform): a translation of the SDF's frame with
     * Sphere's transform_seq semantics (src/sphere.rs:8-21, src/animation.rs:62-68): constant, or the closure
     * |t| center + vel * t sampled at lane 0's time.  Zero constant = the reference's behaviour (x - 0 == x). */
    V3 center{0, 0, 0}; bool animated = false; V3 center_vel{0, 0, 0};
    W3 origin_at(F4 time) const {
        if (!animated) return W3::splat(center);
        float t = time.v[0];
        return W3::splat(V3{center.x + center_vel.x * t, center.y + center_vel.y * t, center.z + center_vel.z * t});
    }
    F4 occluded(W3 start, W3 end, F4 time) const override { /* :25-57 */
        W3 frame = origin_at(time);
        start = start - frame;
        end = end - frame;
        if (g_shadow_sink)
            for (int i = 0; i < 4; i++) {
                const float rec[8] = {g_sink_depth, g_sink_sample, start.x.v[i], start.y.v[i], start.z.v[i], end.x.v[i], end.y.v[i], end.z.v[i]};
                g_shadow_sink->insert(g_shadow_sink->end(), rec, rec + 8);
            }
        W3 dir = end - start;
        F4 max_dist = mag(dir);
        dir = dir / max_dist;
        const float t0 = time.v[0];
        F4 dist = sdf->dist_at(start, t0);
        M4 nan_mask = cmp_nan(dist, dist);
        M4 gt_mask = cmp_gt(dist, max_dist);
        M4 gt_nan_mask = gt_mask | nan_mask;
        M4 hit_mask = cmp_lt(dist, F4(0.0001f));
        F4 t = dist;
        for (uint32_t m = 0; m < cfg.max_vis_marches; m++) {
            M4 gt = cmp_gt(t, max_dist);
            gt_nan_mask = gt | nan_mask;
            if (gt_nan_mask.move_mask() == 0xF) break;
            W3 point = mul_add(dir, W3::broadcast(t), start);
            F4 d = sdf->dist_at(point, t0);
            hit_mask = cmp_lt(abs4(d), fmax4(F4(0.0001f * cfg.detail_scale), F4(0.00001f * cfg.detail_scale) * t));
            M4 hit_gt_nan = hit_mask | gt_nan_mask;
            if (hit_gt_nan.move_mask() == 0xF) break;
            t = merge(hit_gt_nan, t, t + d);
        }
        return merge(hit_mask & !gt_nan_mask, F4(0.0f), F4(1.0f));
    }
    F4 hit(const WRay& ray, F4 t_max, const ThresholdFn& thr) const override { /* :59-83 */
        W3 local_origin = ray.origin - origin_at(ray.time);
        const float t0 = ray.time.v[0];
        F4 dist = sdf->dist_at(local_origin, t0);
        F4 t = dist;
        M4 nan_mask = cmp_nan(t, t);
        for (uint32_t m = 0; m < cfg.max_marches; m++) {
            W3 point = mul_add(ray.dir, W3::broadcast(t), local_origin); /* Ray::point_at in the SDF's frame */
            F4 d = sdf->dist_at(point, t0);
            M4 hit_mask = cmp_lt(abs4(d), fmax4(F4(0.00005f * cfg.detail_scale), F4(0.05f * cfg.detail_scale) * thr(t)));
            M4 gt_mask = cmp_gt(t, t_max);
            M4 stop = hit_mask | nan_mask | gt_mask;
            t = merge(stop, t, t + d);
            if (stop.move_mask() == 0xF) break;
        }
        return t;
    }
    W3 normal_at(W3 p, F4 eps, float t0) const { /* sdfu normals_fast (A5), called at src/sdf.rs:94-96 */
        F4 o(1.0f), n(-1.0f);
#if defined(RAYN_ORACLE_ALT_NORMALS) && RAYN_ORACLE_ALT_NORMALS == 1 /* alternative reading of A5: central differences (sdfu `normals`) */
        F4 z(0.0f);
        (void)n;
        W3 ex(eps * o, z, z), ey(z, eps * o, z), ez(z, z, eps * o);
        W3 g(sdf->dist_at(p + ex, t0) - sdf->dist_at(p - ex, t0), sdf->dist_at(p + ey, t0) - sdf->dist_at(p - ey, t0),
             sdf->dist_at(p + ez, t0) - sdf->dist_at(p - ez, t0));
        return normalized(g);
#elif defined(RAYN_ORACLE_ALT_NORMALS) && RAYN_ORACLE_ALT_NORMALS == 2 /* alternative reading of A5: same estimator, other summation order */
        W3 xyy(o, n, n), yyx(n, n, o), yxy(n, o, n), xxx(o, o, o);
        W3 g = xxx * sdf->dist_at(p + xxx * eps, t0) + xyy * sdf->dist_at(p + xyy * eps, t0) +
               yxy * sdf->dist_at(p + yxy * eps, t0) + yyx * sdf->dist_at(p + yyx * eps, t0);
        return normalized(g);
#else
        W3 xyy(o, n, n), yyx(n, n, o), yxy(n, o, n), xxx(o, o, o);
        W3 g = xyy * sdf->dist_at(p + xyy * eps, t0) + yyx * sdf->dist_at(p + yyx * eps, t0) +
               yxy * sdf->dist_at(p + yxy * eps, t0) + xxx * sdf->dist_at(p + xxx * eps, t0);
        return normalized(g);
#endif
    }
    ShadingInfo get_shading_info(const WHit& hit, const ThresholdFn& hps) const override { /* :85-101 */
        W3 point = hit.point();
        F4 half_pixel_size = fmax4(F4(0.0001f), F4(cfg.detail_scale) * hps(hit.t));
        W3 normal = normal_at(point - origin_at(hit.ray.time), half_pixel_size, hit.ray.time.v[0]);
        return ShadingInfo{material, WShadingPoint::make(hit, point, half_pixel_size, normal)};
    }
};

/* src/sphere.rs:7-87 (TR = constant Vec3: WSequenced impl returns self, src/animation.rs:27-36,50) */
struct Sphere : Hitable {
    V3 center; float radius; size_t material;
    bool animated = false; V3 center_vel{0, 0, 0};
    /* WSequenced::sample_at(&self.transform_seq, time): a constant clones itself; the closure |t| center + vel*t is
     * evaluated at lane 0's time for all four lanes (src/animation.rs:62-68) */
    W3 sample_at(F4 time) const {
        if (!animated) return W3::splat(center);
        float t = time.v[0];
        return W3::splat(V3{center.x + center_vel.x * t, center.y + center_vel.y * t, center.z + center_vel.z * t});
    }
    F4 occluded(W3 start, W3 end, F4 time) const override { /* :24-46 */
        W3 dir = end - start;
        F4 dist = mag(dir);
        dir = dir / dist;
        W3 origin = sample_at(time);
        W3 oc = start - origin;
        F4 b = dot(oc, dir);
        F4 c = mag_sq(oc) - F4(radius * radius);
        F4 descrim = b * b - c;
        M4 desc_pos = cmp_gt(descrim, F4(0.0f));
        F4 desc_sqrt = sqrt4(descrim);
        F4 t1 = -b - desc_sqrt;
        F4 t2 = -b + desc_sqrt;
        F4 mn = fmin4(t1, t2);
        M4 valid = cmp_gt(mn, F4(0.001f)) & cmp_le(t1, dist) & desc_pos;
        return merge(valid, F4(0.0f), F4(1.0f));
    }
    F4 hit(const WRay& ray, F4 t_max, const ThresholdFn&) const override { /* :48-71 */
        W3 origin = sample_at(ray.time);
        W3 oc = ray.origin - origin;
        F4 b = dot(oc, ray.dir);
        F4 c = mag_sq(oc) - F4(radius * radius);
        F4 descrim = b * b - c;
        M4 desc_pos = cmp_gt(descrim, F4(0.0f));
        F4 miss(3.40282347e+38f);
        F4 desc_sqrt = sqrt4(descrim);
        F4 t1 = -b - desc_sqrt;
        M4 t1_valid = cmp_gt(t1, F4(0.0001f)) & cmp_le(t1, t_max) & desc_pos;
        F4 t2 = -b + desc_sqrt;
        M4 t2_valid = cmp_gt(t2, F4(0.0001f)) & cmp_le(t2, t_max) & desc_pos;
        M4 take_t1 = cmp_lt(t1, t2) & t1_valid;
        F4 t = merge(take_t1, t1, t2);
        return merge(t1_valid | t2_valid, t, miss);
    }
    ShadingInfo get_shading_info(const WHit& hit, const ThresholdFn&) const override { /* :73-86 */
        W3 point = hit.point();
        W3 origin = sample_at(hit.ray.time);
        W3 normal = normalized(point - origin);
        return ShadingInfo{material, WShadingPoint::make(hit, point, F4(0.0f), normal)};
    }
};

/* ------------------------------------------------------------------ BSDFs (src/material.rs) */
struct WScatteringEvent { W3 wi; WSrgb f; F4 pdf; };
struct BSDF {
    virtual ~BSDF() {}
    virtual bool receives_light() const { return true; }
    virtual WScatteringEvent scatter(W3 wo, const WShadingPoint& isect, F4 s1d, const F4* s2d) const = 0;
    virtual WSrgb f(W3 wo, W3 wi, W3 n) const = 0;
    virtual WSrgb le(W3, const WShadingPoint&) const { return WSrgb(F4(0.0f), F4(0.0f), F4(0.0f)); }
};
struct LambertianBSDF : BSDF { /* :117-142 */
    WSrgb albedo;
    WScatteringEvent scatter(W3, const WShadingPoint& isect, F4, const F4* s2d) const override {
        W3 diffuse_sample = cosine_weighted_in_hemisphere(s2d[0], s2d[1]);
        W3 diffuse_bounce = normalized(isect.basis * diffuse_sample);
        F4 diffuse_pdf = diffuse_sample.z / F4(PI_F);
        WSrgb diffuse_f = albedo / F4(PI_F);
        return WScatteringEvent{diffuse_bounce, diffuse_f, diffuse_pdf};
    }
    WSrgb f(W3, W3, W3) const override { return albedo / F4(PI_F); }
};
struct DielectricBSDF : BSDF { /* :194-257 */
    WSrgb albedo; F4 roughness;
    /* NOTE the reference declares f(&self, wi, wo, n) but is called as f(wo, wi, n)
     * (src/material.rs:195 vs src/integrator.rs:230): the names below are the DECLARED ones. */
    WSrgb f(W3 wi, W3 wo, W3 n) const override {
        F4 d = fmax4(F4(0.0f), dot(wi, n));
        F4 fresnel = f_schlick(d, F4(0.04f));
        W3 half = normalized(wo + wi);
        F4 cos_alpha = powf4(fmax4(F4(0.0f), dot(half, n)), roughness);
        F4 two(2.0f);
        F4 spec_factor = cos_alpha * (roughness + two) / (two * F4(PI_F));
        WSrgb spec_f = WSrgb(F4(1.0f), F4(1.0f), F4(1.0f)) * spec_factor * fresnel;
        WSrgb diffuse_f = albedo / F4(PI_F) * (F4(1.0f) - fresnel);
        return spec_f + diffuse_f;
    }
    WScatteringEvent scatter(W3 wo, const WShadingPoint& isect, F4 s1d, const F4* s2d) const override {
        W3 norm = isect.normal;
        F4 cos = abs4(dot(norm, wo));
        F4 two(2.0f);
        W3 diffuse_sample = cosine_weighted_in_hemisphere(s2d[0], s2d[1]);
        W3 diffuse_bounce = normalized(isect.basis * diffuse_sample);
        F4 diffuse_pdf = fmax4(F4(0.00001f), diffuse_sample.z / F4(PI_F));
        WSrgb diffuse_f = albedo / F4(PI_F);
        W3 spec_sample = cosine_power_weighted(s2d[2], s2d[3], roughness);
        W3 reflection = reflected(wo, norm);
        Wat3 basis = get_orthonormal_basis(reflection);
        W3 spec_bounce = normalized(basis * spec_sample);
        F4 cos_alpha_pow = fmax4(powf4(spec_sample.z, roughness), F4(EPSILON_F));
        F4 spec_pdf = (roughness + F4(1.0f)) / F4(TWO_PI_F) * cos_alpha_pow;
        F4 spec_coeff = (roughness + two) / F4(TWO_PI_F) * cos_alpha_pow;
        M4 below_horizon = cmp_lt(dot(norm, spec_bounce), F4(0.0f));
        spec_coeff = merge(below_horizon, F4(0.0f), spec_coeff);
        WSrgb spec_f = WSrgb(F4(1.0f), F4(1.0f), F4(1.0f)) * spec_coeff;
        F4 fresnel = f_schlick(cos, F4(0.04f));
        M4 fresnel_mask = cmp_lt(s1d, fresnel);
        WScatteringEvent se;
        se.wi = merge(fresnel_mask, spec_bounce, diffuse_bounce);
        se.f = merge(fresnel_mask, spec_f, diffuse_f);
        se.pdf = fresnel * spec_pdf + (F4(1.0f) - fresnel) * diffuse_pdf;
        return se;
    }
};
struct SkyBSDF : BSDF { /* :419-449 */
    WSrgb top, bottom;
    bool receives_light() const override { return false; }
    WSrgb f(W3, W3, W3) const override { return WSrgb(F4(dm_nanf()), F4(dm_nanf()), F4(dm_nanf())); } /* panic!() */
    WScatteringEvent scatter(W3, const WShadingPoint&, F4, const F4*) const override {
        return WScatteringEvent{W3(F4(0.0f), F4(0.0f), F4(0.0f)), WSrgb(F4(0.0f), F4(0.0f), F4(0.0f)), F4(0.0f)};
    }
    WSrgb le(W3 wo, const WShadingPoint&) const override {
        F4 t = F4(0.5f) * (wo.y + F4(1.0f));
        return top * (F4(1.0f) - t) + bottom * t;
    }
};
struct EmissiveBSDF : BSDF { /* :489-520 */
    WSrgb emission; LambertianBSDF inner;
    bool receives_light() const override { return false; }
    WSrgb f(W3, W3, W3) const override { return WSrgb(F4(0.0f), F4(0.0f), F4(0.0f)); }
    WScatteringEvent scatter(W3 wo, const WShadingPoint& i, F4 a, const F4* b) const override { return inner.scatter(wo, i, a, b); }
    WSrgb le(W3, const WShadingPoint&) const override { return emission; }
};
/* Material::get_bsdf_at (src/material.rs:31-38): parameters are constants in every shipped
 * generator (blanket impl :79-83), so one BSDF per material is built up front. */
std::unique_ptr<BSDF> make_bsdf(const rayn_material& m) {
    auto splat = [](rayn_vec3 v) { return WSrgb(F4(v.x), F4(v.y), F4(v.z)); };
    switch (m.kind) {
    case RAYN_MAT_LAMBERTIAN: { auto b = std::make_unique<LambertianBSDF>(); b->albedo = splat(m.a); return b; }
    case RAYN_MAT_DIELECTRIC: { auto b = std::make_unique<DielectricBSDF>(); b->albedo = splat(m.a); b->roughness = F4(m.exponent); return b; }
    case RAYN_MAT_SKY: { auto b = std::make_unique<SkyBSDF>(); b->top = splat(m.a); b->bottom = splat(m.b); return b; }
    default: { auto b = std::make_unique<EmissiveBSDF>(); b->emission = splat(m.a); b->inner.albedo = WSrgb(F4(0.5f), F4(0.5f), F4(0.5f)); return b; }
    }
}

/* ------------------------------------------------------------------ SphereLight (src/light.rs) */
inline F4 uniform_cone_pdf(F4 cos_theta_max) { return F4(1.0f) / (F4(TWO_PI_F) * (F4(1.0f) - cos_theta_max)); } /* :105-107 */
struct SphereLight {
    W3 pos; WSrgb emission; F4 rad;
    void sample(const F4* samples, W3 p, W3* out_point, WSrgb* out_li, F4* out_pdf) const { /* :38-72 */
        W3 dir_to_light = pos - p;
        F4 dist_to_light_sq = mag_sq(dir_to_light);
        F4 dist_to_light = sqrt4(dist_to_light_sq);
        dir_to_light = dir_to_light / dist_to_light;
        Wat3 basis = get_orthonormal_basis(-dir_to_light);
        F4 r2 = rad * rad;
        F4 sin_theta_max_2 = r2 / dist_to_light_sq;
        F4 cos_theta_max = sqrt4(fmax4(F4(0.0f), F4(1.0f) - sin_theta_max_2));
        F4 cos_theta = (F4(1.0f) - samples[0]) + samples[0] * cos_theta_max;
        F4 sin_theta = sqrt4(fmax4(F4(0.0f), F4(1.0f) - cos_theta * cos_theta));
        F4 phi = samples[1] * F4(TWO_PI_F);
        F4 ds = dist_to_light * cos_theta - sqrt4(fmax4(F4(0.0f), r2 - dist_to_light_sq * sin_theta * sin_theta));
        F4 cos_alpha = (dist_to_light_sq + r2 - ds * ds) / (F4(2.0f) * dist_to_light * rad);
        F4 sin_alpha = sqrt4(fmax4(F4(0.0f), F4(1.0f) - cos_alpha * cos_alpha));
        F4 sin_phi, cos_phi; sin_cos4(phi, &sin_phi, &cos_phi);
        W3 offset = basis.cols[0] * sin_alpha * cos_phi + basis.cols[1] * sin_alpha * sin_phi + basis.cols[2] * cos_alpha;
        *out_point = pos + offset * rad;
        *out_pdf = uniform_cone_pdf(cos_theta_max);
        *out_li = emission;
    }
    void sample_volume_scattering(F4 sample, W3 ray_o, W3 ray_d, F4 max_distance, F4* dist, F4* pdf) const { /* :75-102 */
        F4 delta = dot(pos - ray_o, ray_d);
        W3 closest_point = ray_o + delta * ray_d;
        F4 d = mag(closest_point - pos);
        F4 theta_a = atan2_4(-delta, d);
        F4 theta_b = atan2_4(max_distance - delta, d);
        F4 t = d * tan4(lerp4(theta_a, theta_b, sample));
        *dist = delta + t;
        *pdf = d / ((theta_b - theta_a) * mul_add(d, d, t * t));
    }
};

/* ------------------------------------------------------------------ cameras (src/camera.rs) */
struct Camera {
    uint32_t kind; W2 half_size; W2 full_size; F4 half_pixel_size;
    V3 origin, at, up, focus; float aperture;
    uint32_t animated; V3 origin_vel, at_vel, up_vel, focus_vel;
    /* WSequenced::sample_at: constants clone themselves (src/animation.rs:27-36); the closure `|t| base + vel*t`
     * is evaluated at lane 0's time for all four lanes (src/animation.rs:62-68) */
    W3 sample_at(V3 base, V3 vel, bool anim, F4 time) const {
        if (!anim) return W3::splat(base);
        float t = time.v[0];
        return W3::splat(V3{base.x + vel.x * t, base.y + vel.y * t, base.z + vel.z * t});
    }
    explicit Camera(const rayn_camera& c) {
        kind = c.kind; animated = c.animated;
        origin_vel = V3{c.origin_vel.x, c.origin_vel.y, c.origin_vel.z}; at_vel = V3{c.at_vel.x, c.at_vel.y, c.at_vel.z};
        up_vel = V3{c.up_vel.x, c.up_vel.y, c.up_vel.z}; focus_vel = V3{c.focus_vel.x, c.focus_vel.y, c.focus_vel.z};
        origin = V3{c.origin.x, c.origin.y, c.origin.z}; at = V3{c.at.x, c.at.y, c.at.z};
        up = V3{c.up.x, c.up.y, c.up.z}; focus = V3{c.focus.x, c.focus.y, c.focus.z}; aperture = c.aperture;
        if (kind == RAYN_CAM_ORTHOGRAPHIC) { /* :228-240 */
            float aspect = c.res_w / c.res_h;
            float sx = c.vfov_or_size * aspect, sy = c.vfov_or_size;
            float pixel_size = c.vfov_or_size / c.res_h;
            half_size = W2{F4(sx / 2.0f), F4(sy / 2.0f)}; full_size = W2{F4(sx), F4(sy)};
            half_pixel_size = F4(pixel_size / 2.0f);
        } else { /* :53-72, :134-157 */
            float theta = c.vfov_or_size * PI_F / 180.0f;
            float half_height = dm_tanf(theta / 2.0f);
            float aspect = c.res_w / c.res_h;
            float half_width = aspect * half_height;
            half_pixel_size = F4(half_height / c.res_h);
            half_size = W2{F4(half_width), F4(half_height)}; full_size = half_size;
        }
    }
    WRay make(W3 o, W3 d, F4 time, uint32_t tcx, uint32_t tcy, float scramble, const size_t nums[4]) const {
        WRay r; r.origin = o; r.dir = d; r.time = time; /* WRay::new, src/ray.rs:68-90 */
        r.radiance = WSrgb(F4(0.0f), F4(0.0f), F4(0.0f)); r.throughput = WSrgb(F4(1.0f), F4(1.0f), F4(1.0f));
        for (int i = 0; i < 4; i++) { r.tcx[i] = tcx; r.tcy[i] = tcy; r.valid[i] = true; r.scramble[i] = scramble; r.sample[i] = nums[i]; }
        return r;
    }
    WRay get_rays(float scramble, const size_t nums[4], uint32_t tcx, uint32_t tcy, W2 uv, F4 time, const F4* samples) const {
        W3 o = sample_at(origin, origin_vel, animated & 1u, time), a = sample_at(at, at_vel, animated & 2u, time), u = sample_at(up, up_vel, animated & 4u, time);
        if (kind == RAYN_CAM_PINHOLE) { /* :81-114 */
            W3 basis_w = normalized(o - a);
            W3 basis_u = normalized(cross(u, basis_w));
            W3 basis_v = cross(basis_w, basis_u);
            W3 lower_left = o - basis_u * half_size.x - basis_v * half_size.y - basis_w;
            W3 horiz = basis_u * half_size.x * F4(2.0f) * uv.x;
            W3 verti = basis_v * half_size.y * F4(2.0f) * uv.y;
            return make(o, normalized(lower_left + horiz + verti - o), time, tcx, tcy, scramble, nums);
        } else if (kind == RAYN_CAM_THIN_LENS) { /* :168-208 */
            W3 f = sample_at(focus, focus_vel, animated & 8u, time);
            F4 focus_dist = mag(f - o);
            F4 ap(aperture);
            W3 basis_w = normalized(o - a);
            W3 basis_u = normalized(cross(u, basis_w));
            W3 basis_v = cross(basis_w, basis_u);
            W3 lower_left = o - basis_u * half_size.x * focus_dist - basis_v * half_size.y * focus_dist - basis_w * focus_dist;
            W3 horiz = basis_u * half_size.x * focus_dist * F4(2.0f) * uv.x;
            W3 verti = basis_v * half_size.y * focus_dist * F4(2.0f) * uv.y;
            W2 rd = concentric_circle_map(samples[0], samples[1]);
            rd.x = rd.x * ap; rd.y = rd.y * ap;
            W3 offset = basis_u * rd.x + basis_v * rd.y;
            W3 o2 = o + offset;
            return make(o2, normalized(lower_left + horiz + verti - o2), time, tcx, tcy, scramble, nums);
        } else { /* :249-280 */
            W3 basis_w = normalized(a - o);
            W3 basis_u = normalized(cross(basis_w, u));
            W3 basis_v = cross(basis_u, basis_w);
            W3 lower_left = o - basis_u * half_size.x - basis_v * half_size.y;
            W3 offset = basis_u * uv.x * full_size.x + basis_v * uv.y * full_size.y;
            return make(lower_left + offset, basis_w, time, tcx, tcy, scramble, nums);
        }
    }
    F4 half_pixel_size_at(F4 t) const { /* :116-118, :210-212, :282-284 */
        return kind == RAYN_CAM_ORTHOGRAPHIC ? half_pixel_size : half_pixel_size * t;
    }
};

/* ------------------------------------------------------------------ Samples (src/sampler.rs) */
struct Samples {
    size_t samples; const float* samples_1d; const float* samples_2d;
    float sample_1d(size_t sample, float scramble, size_t set) const { return dm_fractf(samples_1d[sample + samples * set] + scramble); } /* :62-64 */
    F4 wide_sample_1d(size_t start, float scramble, size_t set) const { /* :67-74 */
        return F4(sample_1d(start, scramble, set), sample_1d(start + 1, scramble, set), sample_1d(start + 2, scramble, set), sample_1d(start + 3, scramble, set));
    }
    F4 wide_sample_1d_array(const size_t s[4], const float sc[4], size_t set) const { /* :77-89 */
        return F4(sample_1d(s[0], sc[0], set), sample_1d(s[1], sc[1], set), sample_1d(s[2], sc[2], set), sample_1d(s[3], sc[3], set));
    }
    float sample_2d(size_t dim, size_t sample, float scramble, size_t set) const { return dm_fractf(samples_2d[dim + sample * 2 + samples * 2 * set] + scramble); } /* :92-94 */
    F4 wide_sample_2d(size_t dim, size_t start, float scramble, size_t set) const { /* :97-110 */
        return F4(sample_2d(dim, start, scramble, set), sample_2d(dim, start + 1, scramble, set), sample_2d(dim, start + 2, scramble, set), sample_2d(dim, start + 3, scramble, set));
    }
    F4 wide_sample_2d_array(size_t dim, const size_t s[4], const float sc[4], size_t set) const { /* :113-126 */
        return F4(sample_2d(dim, s[0], sc[0], set), sample_2d(dim, s[1], sc[1], set), sample_2d(dim, s[2], sc[2], set), sample_2d(dim, s[3], sc[3], set));
    }
};

/* FilterImportanceSampler::sample, src/filter.rs:222-235 */
inline float fis_sample(const float* inverse_cdf, float u) {
    u = 2.0f * (u - 0.5f);
    float mult = u < 0.0f ? -1.0f : 1.0f;
    u = __builtin_fabsf(u);
    u = u > 0.0f ? u : 0.0f; /* .max(0.0) */
    u = u < 0.99999f ? u : 0.99999f; /* .min(0.99999) */
    float idx_full = u * (float)(RAYN_FIS_TABLE_SIZE - 1);
    size_t idx = (size_t)__builtin_floorf(idx_full);
    float t = dm_fractf(idx_full);
    return mult * lerp1(inverse_cdf[idx], inverse_cdf[idx + 1], t);
}

/* ------------------------------------------------------------------ World (src/world.rs) -- */
struct World {
    std::vector<std::unique_ptr<Hitable>> hitables;
    std::vector<SphereLight> lights;
    std::vector<std::unique_ptr<BSDF>> materials;
    bool has_scatter, has_extinct; float rho_s, rho_t;
    World(const rayn_world_desc& d, const Config& cfg) {
        for (uint32_t i = 0; i < d.n_hitables; i++) {
            const rayn_hitable& h = d.hitables[i];
            if (h.kind == RAYN_HITABLE_SPHERE) {
                auto s = std::make_unique<Sphere>();
                s->center = V3{h.center.x, h.center.y, h.center.z}; s->radius = h.radius; s->material = h.material;
                s->animated = h.animated != 0; s->center_vel = V3{h.center_vel.x, h.center_vel.y, h.center_vel.z};
                hitables.push_back(std::move(s));
            } else {
                auto t = std::make_unique<TracedSDF>();
                if (h.sdf_kind == RAYN_SDF_MANDELBOX) {
                    auto mb = std::make_unique<MandelBox>(h.iterations, h.box_side, h.min_radius, h.fixed_radius, h.scale);
                    mb->scale_vel = h.scale_vel;
                    t->sdf = std::move(mb);
                }
                else if (h.sdf_kind == RAYN_SDF_MANDELBULB) { auto s = std::make_unique<Mandelbulb>(); s->iterations = h.iterations; t->sdf = std::move(s); }
                else { auto s = std::make_unique<SphereSDF>(); s->radius = F4(h.sdf_radius); t->sdf = std::move(s); }
                t->material = h.material; t->cfg = cfg;
                t->center = V3{h.center.x, h.center.y, h.center.z}; t->animated = h.animated != 0;
                t->center_vel = V3{h.center_vel.x, h.center_vel.y, h.center_vel.z};
                hitables.push_back(std::move(t));
            }
        }
        for (uint32_t i = 0; i < d.n_materials; i++) materials.push_back(make_bsdf(d.materials[i]));
        for (uint32_t i = 0; i < d.n_lights; i++) {
            SphereLight l; /* SphereLight::new, src/light.rs:26-34 */
            l.pos = W3(F4(d.lights[i].pos.x), F4(d.lights[i].pos.y), F4(d.lights[i].pos.z));
            l.emission = WSrgb(F4(d.lights[i].emission.x), F4(d.lights[i].emission.y), F4(d.lights[i].emission.z));
            l.rad = F4(d.lights[i].rad);
            lights.push_back(l);
        }
        has_scatter = d.has_scattering != 0; has_extinct = d.has_extinction != 0;
        rho_s = d.coeff_scattering; rho_t = d.coeff_extinction;
    }
    /* HitableStore::test_occluded, src/hitable.rs:164-168 */
    F4 test_occluded(W3 start, W3 end, F4 time) const {
        F4 acc(1.0f);
        for (auto& h : hitables) acc = acc * h->occluded(start, end, time);
        return acc;
    }
};

/* HitStore, src/hitable.rs:77-141 */
struct HitStore {
    std::vector<std::vector<Hit>> hits;
    explicit HitStore(size_t n) : hits(n) {}
    void add_hit(size_t obj, const Hit& h) { hits[obj].push_back(h); }
    void reset() { for (auto& h : hits) h.clear(); }
};
/* HitableStore::add_hits, src/hitable.rs:170-210 */
void add_hits(const World& w, const WRay& ray, F4 t_max, HitStore& store, const ThresholdFn& thr, Counters* ctr) {
    size_t ids[4] = {SIZE_MAX, SIZE_MAX, SIZE_MAX, SIZE_MAX};
    F4 closest = t_max;
    for (size_t id = 0; id < w.hitables.size(); id++) {
        F4 t = w.hitables[id]->hit(ray, closest, thr);
        for (int i = 0; i < 4; i++)
            if (t.v[i] < closest.v[i]) { closest.v[i] = t.v[i]; ids[i] = id; }
    }
    Ray rays[4]; wray_into(ray, rays);
    for (int i = 0; i < 4; i++)
        if (ids[i] < SIZE_MAX && rays[i].valid) { store.add_hit(ids[i], Hit{rays[i], closest.v[i]}); if (ctr) ctr->segments++; }
}

enum SampleKind { S_COLOR, S_ALPHA, S_BACKGROUND, S_NORMAL };
struct ChannelSample { uint32_t tcx, tcy; SampleKind kind; V3 v; };

/* surface_sample_one_light, src/integrator.rs:207-240 */
WSrgb surface_sample_one_light(const World& w, size_t light_idx, const F4* samples, const WShadingPoint& isect, const BSDF* bsdf) {
    W3 end_point; WSrgb li; F4 pdf;
    w.lights[light_idx].sample(samples, isect.point, &end_point, &li, &pdf);
    W3 wo = -isect.ray.dir;
    W3 wi = end_point - isect.point;
    F4 dist = mag(wi);
    wi = wi / dist;
    W3 occlude_point = isect.point + isect.normal * signum4(dot(isect.normal, wi)) * isect.offset_by;
    F4 occluded = w.test_occluded(occlude_point, end_point, isect.ray.time);
    WSrgb f = bsdf->f(wo, wi, isect.normal) * fmax4(dot(isect.normal, wi), F4(0.0f));
    F4 transmission = w.has_extinct ? exp4(F4(-w.rho_t) * dist) : F4(1.0f);
    return li * f * transmission * occluded / pdf;
}
/* volume_sample_one_light, src/integrator.rs:242-281 */
WSrgb volume_sample_one_light(const World& w, size_t light_idx, const F4* light_samples, F4 volume_sample,
                              W3 ray_o, W3 ray_d, F4 max_distance, F4 time, F4* out_t) {
    const SphereLight& light = w.lights[light_idx];
    F4 vol_sample_dist, vol_sample_pdf;
    light.sample_volume_scattering(volume_sample, ray_o, ray_d, max_distance, &vol_sample_dist, &vol_sample_pdf);
    W3 sampled_point = ray_o + ray_d * vol_sample_dist;
    W3 end_point; WSrgb li; F4 light_pdf;
    light.sample(light_samples, sampled_point, &end_point, &li, &light_pdf);
    W3 wi = end_point - sampled_point;
    F4 dist_point_to_light = mag(wi);
    F4 occluded = w.test_occluded(sampled_point, end_point, time);
    F4 f = F4(1.0f) / (F4(4.0f) * F4(PI_F));
    F4 transmission = w.has_extinct ? exp4(F4(-w.rho_t) * dist_point_to_light) : F4(1.0f);
    *out_t = vol_sample_dist;
    return li * f * transmission * occluded / (vol_sample_pdf * light_pdf);
}
inline size_t light_index(float s, size_t n) { /* `(x).floor() as usize` saturating, clamped (A8) */
    float f = __builtin_floorf(s);
    if (!(f > 0.0f)) return 0;
    size_t i = f >= 1.8e19f ? SIZE_MAX : (size_t)f;
    return i < n ? i : n - 1;
}

/* PathTracingIntegrator::integrate, src/integrator.rs:47-204.  VM = volume_marches. */
struct Integrator {
    size_t max_bounces, volume_marches;
    void integrate(const World& world, const F4* samples_1d, const F4* samples_2d, size_t depth, size_t material,
                   WShadingPoint isect, std::vector<Ray>& spawned_rays, std::vector<ChannelSample>& out) const {
        W3 wo = -isect.ray.dir;
        const BSDF* bsdf = world.materials[material].get();
        F4 volume_transmission = world.has_extinct ? exp4(F4(-world.rho_t) * isect.t) : F4(1.0f);
        isect.ray.radiance += bsdf->le(wo, isect) * isect.ray.throughput * volume_transmission;
        const size_t nl = world.lights.size();
        if (bsdf->receives_light() && nl > 0) {
            F4 lts = floor4(samples_1d[0] * F4((float)nl));
            F4 correction_factor((float)nl / 4.0f);
            for (size_t i = 0; i < 4; i++) {
                size_t light_idx = light_index(lts.v[i], nl);
                g_sink_depth = (float)depth; g_sink_sample = (float)i;
                WSrgb li = surface_sample_one_light(world, light_idx, &samples_2d[0 + i * 2], isect, bsdf);
                isect.ray.radiance += li * isect.ray.throughput * correction_factor * volume_transmission;
            }
        }
        if (world.has_scatter) {
            F4 rho_s(world.rho_s);
            for (size_t march = 0; march < volume_marches; march++) {
                F4 lts = floor4(samples_1d[march + 1] * F4((float)nl));
                F4 correction_factor((float)nl / 4.0f / (float)volume_marches);
                for (size_t i = 0; i < 4; i++) {
                    size_t light_idx = light_index(lts.v[i], nl);
                    F4 t;
                    g_sink_depth = (float)depth; g_sink_sample = (float)(4 + 4 * march + i);
                    WSrgb li = volume_sample_one_light(world, light_idx, &samples_2d[8 + 8 * march + i * 2], samples_1d[1],
                                                       isect.ray.origin, isect.ray.dir, isect.t, isect.ray.time, &t);
                    F4 transmission = world.has_extinct ? exp4(F4(-world.rho_t) * t) : F4(1.0f);
                    isect.ray.radiance += li * isect.ray.throughput * correction_factor * rho_s * transmission;
                }
            }
        }
        if (bsdf->receives_light()) {
            WScatteringEvent se = bsdf->scatter(wo, isect, samples_1d[3], &samples_2d[8 + 8 * volume_marches]);
            F4 ndl = abs4(dot(se.wi, isect.normal));
            WSrgb new_throughput = isect.ray.throughput * volume_transmission * se.f * ndl / se.pdf;
            F4 roulette_factor(0.0f);
            if (depth > 2) {
                roulette_factor = fmax4(F4(1.0f) - component_max(isect.ray.throughput), F4(0.05f));
                new_throughput /= F4(1.0f) - roulette_factor;
            }
            Ray new_rays[4]; wray_into(isect.create_rays(se.wi), new_rays);
            if (depth == 0) {
                for (int i = 0; i < 4; i++)
                    if (new_rays[i].valid) {
                        out.push_back(ChannelSample{new_rays[i].tcx, new_rays[i].tcy, S_ALPHA, V3{1.0f, 0, 0}});
                        out.push_back(ChannelSample{new_rays[i].tcx, new_rays[i].tcy, S_NORMAL, lane(isect.normal, i)});
                    }
            }
            for (int i = 0; i < 4; i++) {
                Ray& ray = new_rays[i];
                if (ray.valid) {
                    if (depth >= max_bounces || samples_1d[4].v[i] < roulette_factor.v[i]) {
                        out.push_back(ChannelSample{ray.tcx, ray.tcy, S_COLOR, ray.radiance});
                    } else {
                        V3 nt = lane(new_throughput, i);
                        if (!(nt.x != nt.x || nt.y != nt.y || nt.z != nt.z)) ray.throughput = nt;
                        spawned_rays.push_back(ray);
                    }
                }
            }
        } else {
            Ray final_rays[4]; wray_into(isect.ray, final_rays);
            for (int i = 0; i < 4; i++)
                if (final_rays[i].valid)
                    out.push_back(ChannelSample{final_rays[i].tcx, final_rays[i].tcy, depth == 0 ? S_BACKGROUND : S_COLOR, final_rays[i].radiance});
        }
    }
};

/* ------------------------------------------------------------------ tiles (src/film.rs) --- */
struct TileBounds { uint32_t x0, y0, x1, y1; };
std::vector<TileBounds> build_tiles(uint32_t W, uint32_t H, uint32_t tw, uint32_t th) { /* src/film.rs:399-427 */
    std::vector<TileBounds> tiles;
    uint32_t remx = W % tw, remy = H % th;
    for (uint32_t tx = 0; tx < (W + remx) / tw; tx++)
        for (uint32_t ty = 0; ty < (H + remy) / th; ty++) {
            uint32_t sx = tx * tw, sy = ty * th;
            tiles.push_back(TileBounds{sx, sy, std::min(sx + tw, W), std::min(sy + th, H)});
        }
    return tiles;
}

struct TraceSink { /* optional per-depth packet dump for the packet-order tests */
    std::vector<uint32_t> depth, obj, px, py, sample, valid;
};

struct Tile {
    TileBounds b; uint32_t ew, eh;
    std::vector<V3> color, background, normal; std::vector<float> alpha; /* ChannelTileStorage, src/film.rs:42-61 */
    explicit Tile(TileBounds tb) : b(tb), ew(tb.x1 - tb.x0), eh(tb.y1 - tb.y0),
        color(ew * eh, V3{0, 0, 0}), background(ew * eh, V3{0, 0, 0}), normal(ew * eh, V3{0, 0, 0}), alpha(ew * eh, 0.0f) {}
    void add_sample(const ChannelSample& s) { /* src/film.rs:54-61,167-172 */
        size_t idx = s.tcx + s.tcy * ew;
        auto add = [](V3& a, V3 v) { a.x += v.x; a.y += v.y; a.z += v.z; };
        switch (s.kind) {
        case S_COLOR: add(color[idx], s.v); break;
        case S_ALPHA: alpha[idx] += s.v.x; break;
        case S_BACKGROUND: add(background[idx], s.v); break;
        case S_NORMAL: add(normal[idx], s.v); break;
        }
    }
};

/* the tile closure of render_frame_into, src/film.rs:439-627 */
void integrate_tile(Tile& tile, const World& world, const Camera& camera, const Integrator& integrator,
                    const Samples& sample_sets, const float* fis, const float* scramble_buf,
                    const rayn_frame_params& p, Counters* ctr, TraceSink* trace) {
    const size_t VM = p.volume_marches;
    const size_t samples = p.samples;
    const uint32_t width = p.width;
    std::vector<Ray> spawned_rays; std::vector<WRay> spawned_wrays;
    std::vector<ShadingInfo> wintersections; std::vector<ChannelSample> new_samples;
    HitStore hit_store(world.hitables.size());
    F4 time_range_range(p.time_end - p.time_start);
    float ndc_x = 1.0f / (float)p.width, ndc_y = 1.0f / (float)p.height; /* Tile::new, src/film.rs:152 */

    for (uint32_t x = tile.b.x0; x < tile.b.x1; x++)
        for (uint32_t y = tile.b.y0; y < tile.b.y1; y++) {
            uint32_t tcx = x - tile.b.x0, tcy = y - tile.b.y0;
            float scramble = scramble_buf[x + y * width]; /* src/film.rs:460-461, precomputed */
            for (size_t samp = 0; samp < samples; samp++) {
                size_t nums[4] = {4 * samp, 4 * samp + 1, 4 * samp + 2, 4 * samp + 3};
                W2 ndcs;
                for (int i = 0; i < 4; i++) { /* sample_uv, src/film.rs:695-709 */
                    float u0 = sample_sets.sample_2d(0, nums[i], scramble, 0);
                    float u1 = sample_sets.sample_2d(1, nums[i], scramble, 0);
                    float fx = fis_sample(fis, u0), fy = fis_sample(fis, u1);
                    float scx = ((float)x + 0.5f) + fx, scy = ((float)y + 0.5f) + fy;
                    ndcs.x.v[i] = ndc_x * scx; ndcs.y.v[i] = ndc_y * scy;
                }
                F4 times = F4(p.time_start) + time_range_range * sample_sets.wide_sample_1d(nums[0], scramble, 0);
                F4 lens[2] = {sample_sets.wide_sample_2d(0, nums[0], scramble, 1), sample_sets.wide_sample_2d(1, nums[0], scramble, 1)};
                spawned_wrays.push_back(camera.get_rays(scramble, nums, tcx, tcy, ndcs, times, lens));
                if (ctr) ctr->paths += 4;
            }
        }

    for (size_t depth = 0;; depth++) {
        if (spawned_wrays.empty()) break;
        hit_store.reset();
        ThresholdFn half_pixel_size_at;
        if (depth == 0) half_pixel_size_at = [&camera](F4 t) { return camera.half_pixel_size_at(t); };
        else { float k = 0.0001f * 2.0f * (float)depth; half_pixel_size_at = [k](F4 t) { return F4(k) * t; }; }

        for (const WRay& wray : spawned_wrays)
            add_hits(world, wray, F4(p.world_radius * 2.0f), hit_store, half_pixel_size_at, ctr);
        spawned_wrays.clear();

        /* HitStore::process_hits, src/hitable.rs:94-134 */
        for (auto& hits : hit_store.hits)
            while (hits.size() % 4 != 0) hits.push_back(Hit{ray_new_invalid(), 0.0f});
        for (size_t obj_id = 0; obj_id < hit_store.hits.size(); obj_id++) {
            auto& hits = hit_store.hits[obj_id];
            for (size_t k = 0; k + 4 <= hits.size(); k += 4) {
                WHit wh; Ray rr[4] = {hits[k].ray, hits[k + 1].ray, hits[k + 2].ray, hits[k + 3].ray};
                wh.ray = wray_from(rr); wh.t = F4(hits[k].t, hits[k + 1].t, hits[k + 2].t, hits[k + 3].t);
                wintersections.push_back(world.hitables[obj_id]->get_shading_info(wh, half_pixel_size_at));
                if (trace)
                    for (int i = 0; i < 4; i++) {
                        trace->depth.push_back((uint32_t)depth); trace->obj.push_back((uint32_t)obj_id);
                        trace->px.push_back(rr[i].tcx); trace->py.push_back(rr[i].tcy);
                        trace->sample.push_back((uint32_t)rr[i].sample); trace->valid.push_back(rr[i].valid ? 1 : 0);
                    }
            }
        }

        for (const ShadingInfo& si : wintersections) {
            F4 s1[3 + 4], s2[12 + 8 * 4]; /* [f32x4; 3+VM], [f32x4; 12+8*VM], VM <= 4 */
            const size_t n1 = 3 + VM, n2 = 12 + 8 * VM;
            for (size_t set = 0; set < n1; set++)
                s1[set] = sample_sets.wide_sample_1d_array(si.sp.ray.sample, si.sp.ray.scramble, 1 + set + depth * n1);
            for (size_t i = 0; i < n2; i++) {
                size_t dim = i % 2, set = i / 2;
                s2[i] = sample_sets.wide_sample_2d_array(dim, si.sp.ray.sample, si.sp.ray.scramble, 2 + set + depth * n2 / 2);
            }
            integrator.integrate(world, s1, s2, depth, si.material, si.sp, spawned_rays, new_samples);
            if (ctr) ctr->packets++;
        }
        wintersections.clear();

        for (const ChannelSample& s : new_samples) tile.add_sample(s);
        new_samples.clear();

        while (spawned_rays.size() % 4 != 0) spawned_rays.push_back(ray_new_invalid());
        for (size_t k = 0; k + 4 <= spawned_rays.size(); k += 4) spawned_wrays.push_back(wray_from(&spawned_rays[k]));
        spawned_rays.clear();
    }
}

struct FilmOut { float *color, *alpha, *background, *normal; };
/* tile_finished + ChannelStorage::copy_from_tile, src/film.rs:82-98,660-691 */
void tile_finished(const Tile& t, const FilmOut& f, uint32_t W, size_t samples) {
    float n = (float)samples;
    for (uint32_t x = 0; x < t.ew; x++)
        for (uint32_t y = 0; y < t.eh; y++) {
            size_t ti = x + y * t.ew, fi = (t.b.x0 + x) + (size_t)(t.b.y0 + y) * W;
            f.color[3 * fi] = t.color[ti].x / n; f.color[3 * fi + 1] = t.color[ti].y / n; f.color[3 * fi + 2] = t.color[ti].z / n;
            f.alpha[fi] = t.alpha[ti] / n;
            f.background[3 * fi] = t.background[ti].x / n; f.background[3 * fi + 1] = t.background[ti].y / n; f.background[3 * fi + 2] = t.background[ti].z / n;
            f.normal[3 * fi] = t.normal[ti].x / n; f.normal[3 * fi + 1] = t.normal[ti].y / n; f.normal[3 * fi + 2] = t.normal[ti].z / n;
        }
}

/* ------------------------------------------------------------------ host tables ----------- */
typedef unsigned __int128 u128;
/* A6: Samples::new_rd, src/sampler.rs:18-37 */
void fill_rd(float* out, size_t n, int dim, uint64_t offset) {
    const u128 A1 = ((u128)0x9e3779b97f4a7c15ULL << 64) | 0xf39cc0605cedc834ULL;  /* 1/phi   (tools/gen_rd_constants.py) */
    const u128 A2X = ((u128)0xc13fa9a902a6328fULL << 64) | 0x434ff71b2d97724bULL; /* 1/rho   */
    const u128 A2Y = ((u128)0x91e10da5c79e7b1cULL << 64) | 0xd438a0a8e6c9c0fcULL; /* 1/rho^2 */
    const u128 HALF = (u128)1 << 127;
    for (size_t k = 0; k < n; k++) {
        u128 idx = (u128)offset + 1 + k;
        if (dim == 1) out[k] = (float)(uint32_t)((HALF + A1 * idx) >> 104) * (1.0f / 16777216.0f);
        else {
            out[2 * k] = (float)(uint32_t)((HALF + A2X * idx) >> 104) * (1.0f / 16777216.0f);
            out[2 * k + 1] = (float)(uint32_t)((HALF + A2Y * idx) >> 104) * (1.0f / 16777216.0f);
        }
    }
}
/* A7: SmallRng::seed_from_u64(seed).gen::<f32>(), src/film.rs:460-461 */
float pcg_scramble(uint64_t state) {
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    uint32_t words[4];
    for (int i = 0; i < 4; i++) {
        state = state * MUL + INC;
        uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        words[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    u128 s = ((u128)words[3] << 96) | ((u128)words[2] << 64) | ((u128)words[1] << 32) | (u128)words[0];
    s |= 1; /* Mcg128Xsl64::new */
    const u128 MULT = ((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
    s = s * MULT;
    uint32_t rot = (uint32_t)(s >> 122);
    uint64_t xsl = (uint64_t)(s >> 64) ^ (uint64_t)s;
    uint64_t r = (xsl >> rot) | (xsl << ((64 - rot) & 63));
    uint32_t v = (uint32_t)r; /* next_u32 */
    return (float)(v >> 8) * (1.0f / 16777216.0f);
}
/* Filter::evaluate.  kind 0 BlackmanHarrisFilter src/filter.rs:37-49; 1 BoxFilter :127-140;
 * 2 MitchellNetravaliFilter(b = a, c = b2) :74-108; 3 LanczosSincFilter(tau = a) :159-185 */
struct FilterDesc { uint32_t kind; float radius, a, b2; };
float sinc_abs(float x) { /* LanczosSincFilter::sinc */
    x = __builtin_fabsf(x);
    if (x <= 0.00001f) return 1.0f;
    float pix = PI_F * x;
    float s = dm_sinf(pix);
    return s / pix;
}
float filter_eval(const FilterDesc& f, float p) {
    switch (f.kind) {
    case 1: return __builtin_fabsf(p) > f.radius ? 0.0f : 1.0f;
    case 2: {
        const float B = f.a, C = f.b2;
        float x = __builtin_fabsf(2.0f * p / f.radius);
        if (x >= 2.0f) return 0.0f;
        if (x > 1.0f) {
            float poly = (-B - 6.0f * C) * x * x * x + (6.0f * B + 30.0f * C) * x * x + (-12.0f * B - 48.0f * C) * x + (8.0f * B + 24.0f * C);
            return poly * (1.0f / 6.0f);
        }
        float poly = (12.0f - 9.0f * B - 6.0f * C) * x * x * x + (-18.0f + 12.0f * B + 6.0f * C) * x * x + (6.0f - 2.0f * B);
        return poly * (1.0f / 6.0f);
    }
    case 3: {
        float x = __builtin_fabsf(p);
        if (x > f.radius) return 0.0f;
        float lanczos = sinc_abs(x / f.a);
        return sinc_abs(x) * lanczos;
    }
    default: {
        const float A0 = 0.35875f, A1 = 0.48829f, A2 = 0.14128f, A3 = 0.01168f;
        const float TWOPI = PI_F * 2.0f, FOURPI = PI_F * 4.0f, SIXPI = PI_F * 6.0f;
        if (__builtin_fabsf(p) > f.radius) return 0.0f;
        float x = __builtin_fabsf(p / f.radius) * 0.5f + 0.5f;
        return A0 - A1 * dm_cosf(TWOPI * x) + A2 * dm_cosf(FOURPI * x) + A3 * dm_cosf(SIXPI * x);
    }
    }
}
/* FilterImportanceSampler::new + CDF, src/filter.rs:196-220, src/math.rs:143-190 */
void build_fis(const FilterDesc& filt, float* inverse_cdf) {
    const float f_rad = filt.radius;
    const size_t N = RAYN_FIS_TABLE_SIZE;
    std::vector<float> item(N), weight(N), density;
    float weight_sum = 0.0f;
    for (size_t n = 0; n < N; n++) {
        float t = (float)n / (float)(N - 1);
        float d = lerp1(0.0f, f_rad, t);
        item[n] = d; weight[n] = filter_eval(filt, d); weight_sum += weight[n];
    }
    for (size_t n = 0; n < N; n++) weight[n] /= weight_sum;
    float cum = 0.0f;
    for (size_t n = 0; n < N; n++) { cum += weight[n]; density.push_back(cum); }
    for (size_t n = N; n-- > 0;) { density[n] = 1.0f; if (weight[n] > 0.0f) break; }
    for (size_t n = 0; n < N; n++) {
        float u = (float)n / (float)(N - 1);
        float r = 0.0f;
        for (size_t k = 0; k < N; k++) if (density[k] >= u) { r = item[k]; break; }
        inverse_cdf[n] = r;
    }
}

Config cfg_of(const rayn_frame_params& p) { return Config{p.max_marches, p.max_vis_marches, p.sdf_detail_scale}; }

} // namespace

extern "C" {

struct oracle_counters { uint64_t paths, segments, packets, dist_evals, tiles; };

uint32_t oracle_sets_1d(uint32_t B, uint32_t VM) { return 1 + (B + 1) * (3 + VM); }  /* src/film.rs:431, src/integrator.rs:39-41 */
uint32_t oracle_sets_2d(uint32_t B, uint32_t VM) { return 2 + (B + 1) * (12 + 8 * VM); } /* src/film.rs:432, src/integrator.rs:43-45 */

void oracle_build_rd_tables(uint32_t spp, uint32_t sets_1d, uint32_t sets_2d, uint64_t frame, float* s1d, float* s2d) {
    for (uint32_t i = 0; i < sets_1d; i++) fill_rd(s1d + (size_t)spp * i, spp, 1, (frame + i) << 32);
    for (uint32_t i = 0; i < sets_2d; i++) fill_rd(s2d + (size_t)spp * 2 * i, spp, 2, (frame + sets_1d + i) << 32);
}
void oracle_build_scramble(uint32_t w, uint32_t h, float* out) {
    for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) out[x + (size_t)y * w] = pcg_scramble((uint64_t)(x + y * w));
}
void oracle_build_fis_table(uint32_t kind, float radius, float* out) { build_fis(FilterDesc{kind, radius, 0.0f, 0.0f}, out); }
void oracle_build_fis_table_ex(uint32_t kind, float radius, float a, float b, float* out) { build_fis(FilterDesc{kind, radius, a, b}, out); }
uint32_t oracle_tile_count(uint32_t w, uint32_t h, uint32_t tw, uint32_t th) { return (uint32_t)build_tiles(w, h, tw, th).size(); }

/* Film::render_frame_into.  tile_subset == NULL renders every tile k with (k + k / tile_step) % tile_step == tile_first
 * (all tiles for 0/1); otherwise exactly the listed tile indices.  Untouched pixels keep their value. */
int oracle_render_frame(const rayn_world_desc* wd, const rayn_frame_params* p, const float* s1d, const float* s2d,
                        const float* scramble, const float* fis, float* out_color, float* out_alpha,
                        float* out_background, float* out_normal, int threads, const uint32_t* tile_subset,
                        uint32_t n_subset, oracle_counters* counters) {
    if (!wd || !p || p->volume_marches < 2 || p->volume_marches > 4 || p->samples == 0) return -1;
    World world(*wd, cfg_of(*p));
    Camera camera(wd->camera);
    Integrator integ{p->max_bounces, p->volume_marches};
    Samples sets{(size_t)p->samples * 4, s1d, s2d};
    std::vector<TileBounds> tiles = build_tiles(p->width, p->height, p->tile_w, p->tile_h);
    std::vector<uint32_t> todo;
    if (tile_subset) todo.assign(tile_subset, tile_subset + n_subset);
    else { uint32_t step = p->tile_step ? p->tile_step : 1; for (uint32_t k = 0; k < tiles.size(); k++) if ((k + k / step) % step == p->tile_first) todo.push_back(k); }
    FilmOut film{out_color, out_alpha, out_background, out_normal};
    std::atomic<size_t> next{0}; std::atomic<uint64_t> evals{0}, n_paths{0}, n_segments{0}, n_packets{0};
    auto worker = [&]() {
        Counters ctr;
        tl_dist_evals = 0;
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= todo.size()) break;
            if (todo[i] >= tiles.size()) continue;
            Tile tile(tiles[todo[i]]);
            integrate_tile(tile, world, camera, integ, sets, fis, scramble, *p, &ctr, nullptr);
            tile_finished(tile, film, p->width, (size_t)p->samples * 4); /* disjoint pixels: no mutex needed */
        }
        evals += tl_dist_evals; n_paths += ctr.paths; n_segments += ctr.segments; n_packets += ctr.packets;
    };
    if (threads <= 1) worker();
    else { std::vector<std::thread> th; for (int i = 0; i < threads; i++) th.emplace_back(worker); for (auto& t : th) t.join(); }
    if (counters) { counters->paths = n_paths; counters->segments = n_segments; counters->packets = n_packets; counters->dist_evals = evals * 4; counters->tiles = todo.size(); }
    return 0;
}

/* Per-depth packet dump of ONE tile, in HitStore::process_hits order (object-major, insertion
 * order, padded to x4).  Returns the number of lanes written (or needed if > cap). */
int64_t oracle_trace_tile(const rayn_world_desc* wd, const rayn_frame_params* p, const float* s1d, const float* s2d,
                          const float* scramble, const float* fis, uint32_t tile_index, uint64_t cap,
                          uint32_t* depth, uint32_t* obj, uint32_t* px, uint32_t* py, uint32_t* sample, uint32_t* valid) {
    World world(*wd, cfg_of(*p));
    Camera camera(wd->camera);
    Integrator integ{p->max_bounces, p->volume_marches};
    Samples sets{(size_t)p->samples * 4, s1d, s2d};
    std::vector<TileBounds> tiles = build_tiles(p->width, p->height, p->tile_w, p->tile_h);
    if (tile_index >= tiles.size()) return -1;
    Tile tile(tiles[tile_index]); TraceSink sink;
    integrate_tile(tile, world, camera, integ, sets, fis, scramble, *p, nullptr, &sink);
    uint64_t n = sink.depth.size();
    for (uint64_t i = 0; i < n && i < cap; i++) {
        depth[i] = sink.depth[i]; obj[i] = sink.obj[i]; px[i] = sink.px[i]; py[i] = sink.py[i]; sample[i] = sink.sample[i]; valid[i] = sink.valid[i];
    }
    return (int64_t)n;
}

/* ---- known-answer helpers (single lane = lane 0 of a splatted packet) ---- */
void oracle_sdf_dist(const rayn_hitable* h, const float* pts, float* out, uint64_t n) {
    rayn_world_desc wd; memset(&wd, 0, sizeof wd); wd.n_hitables = 1; wd.hitables[0] = *h;
    World w(wd, Config{256, 100, 0.5f});
    const TracedSDF* t = dynamic_cast<const TracedSDF*>(w.hitables[0].get());
    for (uint64_t i = 0; i < n; i++) {
        if (!t) { out[i] = dm_nanf(); continue; }
        out[i] = t->sdf->dist(W3(F4(pts[3 * i]), F4(pts[3 * i + 1]), F4(pts[3 * i + 2]))).v[0];
    }
}
/* closest hit of the whole world for n rays at 'depth' (threshold closure of src/film.rs:540-551):
 * out_t, out_obj (0xFFFFFFFF = none) — HitableStore::add_hits without the bins. */
void oracle_closest_hit(const rayn_world_desc* wd, const rayn_frame_params* p, uint32_t depth, const float* org,
                        const float* dir, float* out_t, uint32_t* out_obj, uint64_t n) {
    World w(*wd, cfg_of(*p)); Camera cam(wd->camera);
    ThresholdFn thr;
    if (depth == 0) thr = [&cam](F4 t) { return cam.half_pixel_size_at(t); };
    else { float k = 0.0001f * 2.0f * (float)depth; thr = [k](F4 t) { return F4(k) * t; }; }
    for (uint64_t i = 0; i < n; i++) {
        WRay r; r.origin = W3(F4(org[3 * i]), F4(org[3 * i + 1]), F4(org[3 * i + 2]));
        r.dir = W3(F4(dir[3 * i]), F4(dir[3 * i + 1]), F4(dir[3 * i + 2])); r.time = F4(0.0f);
        F4 closest(p->world_radius * 2.0f); uint32_t id = 0xFFFFFFFFu;
        for (size_t k = 0; k < w.hitables.size(); k++) {
            F4 t = w.hitables[k]->hit(r, closest, thr);
            if (t.v[0] < closest.v[0]) { closest = F4(t.v[0]); id = (uint32_t)k; }
        }
        out_t[i] = closest.v[0]; out_obj[i] = id;
    }
}
/* HitableStore::test_occluded for n segments */
void oracle_test_occluded(const rayn_world_desc* wd, const rayn_frame_params* p, const float* start, const float* end, float* out, uint64_t n) {
    World w(*wd, cfg_of(*p));
    for (uint64_t i = 0; i < n; i++)
        out[i] = w.test_occluded(W3(F4(start[3 * i]), F4(start[3 * i + 1]), F4(start[3 * i + 2])),
                                 W3(F4(end[3 * i]), F4(end[3 * i + 1]), F4(end[3 * i + 2])), F4(0.0f)).v[0];
}
/* rayn_detmath probes so tests can compare host and device bit patterns: op 0 exp,1 sin,2 cos,3 tan,4 atan2(a,b),5 pow(a,b) */
void oracle_detmath(uint32_t op, const float* a, const float* b, float* out, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) {
        switch (op) {
        case 0: out[i] = dm_expf(a[i]); break;
        case 1: out[i] = dm_sinf(a[i]); break;
        case 2: out[i] = dm_cosf(a[i]); break;
        case 3: out[i] = dm_tanf(a[i]); break;
        case 4: out[i] = dm_atan2f(a[i], b[i]); break;
        case 6: out[i] = dm_logf(a[i]); break; /* the Mandelbulb extension's logarithm */
        default: out[i] = dm_powf(a[i], b[i]); break;
        }
    }
}
int oracle_fma_policy() { return RAYN_FMA_POLICY; }
/* Host check of include/rayn_detmath_fast.h (what the KERNELS evaluate; the oracle itself keeps using rayn_detmath.h): out = the fast
 * wrapper's result for op 0 exp, 1 sin, 2 cos, 3 tan, 4 atan2(a,b), 5 pow(a,b), 6 log; stats[0] = number of calls that fell back to the
 * reference evaluation, stats[1] = largest |d_fast - d_ref| / |d_ref| between the two binary64 evaluations over the arguments inside
 * the fast domain (must stay well below the EPS the rounding test assumes). */
void oracle_detmath_fast(uint32_t op, const float* a, const float* b, float* out, uint64_t n, double* stats) {
    dmf_fallbacks = 0;
    double worst = 0.0;
    auto dev = [&](double f, double r) { if (r != 0.0 && r == r && f == f) { double d = (f - r) / r; if (d < 0) d = -d; if (d > worst) worst = d; } };
    for (uint64_t i = 0; i < n; i++) {
        const float x = a[i], y = b[i];
        switch (op) {
        case 0: out[i] = dmf_expf(x); if (x > -87.0f && x < 88.0f) dev(dmf_exp_core((double)x), dm_exp_core((double)x)); break;
        case 1: case 2: {
            float s, c; dmf_sincosf(x, &s, &c); out[i] = op == 1 ? s : c;
            if (x > -1.0e4f && x < 1.0e4f) { double fs, fc, rs, rc; dmf_sincos_core((double)x, &fs, &fc); dm_sincos_core((double)x, &rs, &rc); dev(fs, rs); dev(fc, rc); }
            break;
        }
        case 3: out[i] = dmf_tanf(x); if (x > -1.0e4f && x < 1.0e4f) { double fs, fc, rs, rc; dmf_sincos_core((double)x, &fs, &fc); dm_sincos_core((double)x, &rs, &rc); dev(fs / fc, rs / rc); } break;
        case 4: {
            out[i] = dmf_atan2f(x, y);
            const float ax = y < 0 ? -y : y, ay = x < 0 ? -x : x;
            if (ax > 1.0e-30f && ax < 1.0e30f && ay > 1.0e-30f && ay < 1.0e30f) {
                const double dx = ax, dy = ay;
                dev(dy > dx ? dmf_atan_core(dx / dy) : dmf_atan_core(dy / dx), dy > dx ? dm_atan_core(dx / dy) : dm_atan_core(dy / dx));
            }
            break;
        }
        case 6: out[i] = dmf_logf(x); if (x > 1.0e-30f && x < 1.0e30f) dev(dmf_log_tab_core((double)x), dm_log_core((double)x)); break;
        default: {
            out[i] = dmf_powf(x, y);
            if (x > 1.0e-30f && x < 1.0e30f && x != 1.0f && y != 0.0f && y > -1.0e4f && y < 1.0e4f) {
                const double fa = (double)y * dmf_log_core((double)x), ra = (double)y * dm_log_core((double)x);
                if (fa > -87.0 && fa < 88.0 && ra > -87.0 && ra < 88.0) dev(dmf_exp_core(fa), dm_exp_core(ra));
            }
            break;
        }
        }
    }
    stats[0] = (double)dmf_fallbacks;
    stats[1] = worst;
}

/* Film::save_to's per-pixel post-process, src/film.rs:205-378 (N3), restated arm by arm.  kind: 0 Color, 1 Alpha,
 * 2 Background, 3 WorldNormal (ChannelKind, src/film.rs:103-120).  have_*: which channels the film holds.  Writes the 8-bit
 * image rows top-down (idx = x + (h-1-y)*w, :236) into out (1, 3 or 4 bytes per pixel) and returns the bytes per pixel, or
 * -1 where the reference returns Err ("insufficient channels" :283-287 / "didn't exist" :292-296,316-320,341-345).
 * Srgb::saturated / gamma_corrected: src/spectrum.rs:30-40; powf = the pinned dm_powf.  Quantisation
 * `(v * 255.0).min(255.0).max(0.0) as u8` with Rust's f32::min/max (a NaN operand yields the OTHER operand) and the
 * saturating float -> u8 cast. */
static inline float rust_min(float a, float b) { return a != a ? b : (b != b ? a : (a < b ? a : b)); }
static inline float rust_max(float a, float b) { return a != a ? b : (b != b ? a : (a > b ? a : b)); }
static inline uint8_t quant8(float v) { float q = rust_max(rust_min(v * 255.0f, 255.0f), 0.0f); return (uint8_t)q; }
static inline float saturate1(float x) { return rust_min(rust_max(x, 0.0f), 1.0f); }
static inline float gamma1(float x, float gamma) { return dm_powf(x, 1.0f / gamma); }
int oracle_save_to_pixels(uint32_t kind, int have_color, int have_alpha, int have_background, int have_normal, int transparent_background,
                          uint32_t w, uint32_t h, const float* color, const float* alpha, const float* background, const float* normal, uint8_t* out) {
    int bpp = -1;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const size_t idx = (size_t)x + (size_t)(h - 1 - y) * w, o = (size_t)x + (size_t)y * w;
            if (kind == 0) {
                if (have_color && have_alpha && transparent_background) { /* :231-254 */
                    bpp = 4;
                    for (int c = 0; c < 3; c++) out[o * 4 + c] = quant8(gamma1(saturate1(color[3 * idx + c]), 2.2f));
                    out[o * 4 + 3] = quant8(alpha[idx]);
                } else if (have_color && have_background && !transparent_background) { /* :255-277 */
                    bpp = 3;
                    for (int c = 0; c < 3; c++) out[o * 3 + c] = quant8(gamma1(saturate1(color[3 * idx + c] + background[3 * idx + c]), 2.2f));
                } else if (have_color && !have_background && !transparent_background) { /* :278-ish: gamma WITHOUT saturate */
                    bpp = 3;
                    for (int c = 0; c < 3; c++) out[o * 3 + c] = quant8(gamma1(color[3 * idx + c], 2.2f));
                } else return -1;
            } else if (kind == 2) {
                if (!have_background) return -1;
                bpp = 3;
                for (int c = 0; c < 3; c++) out[o * 3 + c] = quant8(gamma1(saturate1(background[3 * idx + c]), 2.2f));
            } else if (kind == 3) {
                if (!have_normal) return -1;
                bpp = 3;
                for (int c = 0; c < 3; c++) out[o * 3 + c] = quant8(normal[3 * idx + c] * 0.5f + 0.5f);
            } else {
                if (!have_alpha) return -1;
                bpp = 1;
                out[o] = quant8(alpha[idx]);
            }
        }
    return bpp;
}
/* DIAGNOSTICS: install (on != 0) / remove the shadow-segment sink; oracle_take_shadow_sink copies up to cap floats (records of 8)
 * and returns the float count.  Used by tools/coherence_sim.py to study wave coherence of the shadow marches on the CPU. */
void oracle_set_shadow_sink(int on) {
    delete g_shadow_sink;
    g_shadow_sink = on ? new std::vector<float>() : nullptr;
}
uint64_t oracle_take_shadow_sink(float* out, uint64_t cap) {
    if (!g_shadow_sink) return 0;
    const uint64_t n = g_shadow_sink->size();
    if (out) memcpy(out, g_shadow_sink->data(), (size_t)std::min(n, cap) * 4);
    return n;
}

} // extern "C"
