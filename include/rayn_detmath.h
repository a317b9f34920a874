/* rayn_detmath.h — pinned elementary functions + the FMA policy for the rayn hot path.
 *
 * WHY THIS EXISTS
 * rayn's arithmetic runs through `wide` 0.4.6 / `ultraviolet` 0.4.6 (Cargo.lock:641-669), whose
 * transcendentals (exp, sin_cos, tan, atan2, powf; call sites: src/integrator.rs:65,123,234,272,
 * src/math.rs:89,108,111,217, src/light.rs:61,93-97, src/material.rs:199,236) fall through to the
 * platform libm lane by lane.  A path tracer is chaotic: one ulp of difference in a transcendental
 * can flip a hit/miss three bounces later.  To make "GPU == CPU oracle" a testable statement, both
 * sides evaluate these functions with THIS header: only IEEE-754 correctly-rounded +,-,*,/,sqrt,
 * fusedMultiplyAdd, floor and int<->float conversions in binary64, no implicit contraction, no libm.
 * The polynomial (Horner) steps are EXPLICIT fused multiply-adds (__builtin_fma: one rounding, defined by
 * IEEE 754, the same result from an x86-64 FMA unit, from glibc's software fma and from gfx950's
 * v_fma_f64) - half the binary64 instructions of the mul + add form r1 used, which matters on the GPU
 * (k_shade_setup spends about a third of its cycles here).  The same source
 * compiled by g++ (oracle) and by hipcc for gfx950 (kernels) therefore yields bit-identical floats.
 * Internally everything is evaluated in double with truncation error < 1e-15, so the float result
 * is the correctly rounded one except in ~1e-8 of cases — i.e. it agrees with a good libm (glibc's
 * sinf/expf/powf are themselves double-evaluated) essentially everywhere.
 *
 * Both compilers MUST be run with -ffp-contract=off (hipcc defaults to 'fast').
 *
 * FMA POLICY: the reference ships no RUSTFLAGS, so `f32x4::mul_add` in wide 0.4.6
 * compiles to an unfused multiply then add on the default x86-64 target.  RAYN_MULADD is therefore
 * unfused unless RAYN_FMA_POLICY=1 is defined (equivalent to building rayn with +fma).
 */
#ifndef RAYN_DETMATH_H
#define RAYN_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RAYN_HD static __host__ __device__ inline
#else
#define RAYN_HD inline
#endif
#if defined(__HIPCC__) && defined(RAYN_DM_NOINLINE)
#define RAYN_HD_CORE static __host__ __device__ __attribute__((noinline))
#else
#define RAYN_HD_CORE RAYN_HD
#endif

#ifndef RAYN_FMA_POLICY
#define RAYN_FMA_POLICY 0
#endif

/* a*b + c with the policy's rounding: two roundings (policy 0) or one (policy 1). */
RAYN_HD float rayn_muladd(float a, float b, float c) {
#if RAYN_FMA_POLICY
    return __builtin_fmaf(a, b, c);
#else
    return a * b + c;
#endif
}

RAYN_HD double dm_u2d(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
RAYN_HD uint64_t dm_d2u(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
RAYN_HD float dm_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
RAYN_HD uint32_t dm_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

RAYN_HD float dm_nanf() { return dm_u2f(0x7fc00000u); }
RAYN_HD float dm_inff() { return dm_u2f(0x7f800000u); }
RAYN_HD bool dm_isnan(float x) { return x != x; }
RAYN_HD bool dm_isnand(double x) { return x != x; }

/* 2^k as a double, k in [-1022, 1023]. */
/* A polynomial coefficient.  On the GPU it is pinned to a scalar register pair right where it is used: left alone, the
 * compiler keeps all ~60 coefficients of the inlined functions live in vector registers across the whole kernel
 * (k_shade_setup then needs 161 VGPRs instead of 80).  The value is the same on both sides. */
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ inline double dm_k(double c) { asm volatile("" : "+s"(c)); return c; }
#define DM_K(c) dm_k(c)
#else
#define DM_K(c) (c)
#endif

RAYN_HD double dm_pow2i(int k) { return dm_u2d((uint64_t)(k + 1023) << 52); }

/* e^a for a double argument, returned as double; caller guarantees -745 < a < 709. */
RAYN_HD_CORE double dm_exp_core(double a) {
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01; /* 33 significant bits: k*LN2_HI exact */
    const double LN2_LO = 1.90821492927058770002e-10;
    double kf = __builtin_floor(a * LOG2E + 0.5);
    double r = (a - kf * LN2_HI) - kf * LN2_LO; /* |r| <= ~0.347 */
    /* Taylor through r^14/14!  (0.347^14/14! ~ 4e-18) */
    double p = 1.0 / 87178291200.0;
    p = __builtin_fma(p, r, DM_K(1.0 / 6227020800.0));
    p = __builtin_fma(p, r, DM_K(1.0 / 479001600.0));
    p = __builtin_fma(p, r, DM_K(1.0 / 39916800.0));
    p = __builtin_fma(p, r, DM_K(1.0 / 3628800.0));
    p = __builtin_fma(p, r, DM_K(1.0 / 362880.0));
    p = __builtin_fma(p, r, DM_K(1.0 / 40320.0));
    p = __builtin_fma(p, r, DM_K(1.0 / 5040.0));
    p = __builtin_fma(p, r, DM_K(1.0 / 720.0));
    p = __builtin_fma(p, r, DM_K(1.0 / 120.0));
    p = __builtin_fma(p, r, DM_K(1.0 / 24.0));
    p = __builtin_fma(p, r, DM_K(1.0 / 6.0));
    p = __builtin_fma(p, r, DM_K(0.5));
    p = __builtin_fma(p, r, DM_K(1.0));
    p = __builtin_fma(p, r, DM_K(1.0));
    int k = (int)kf;
    /* split the scale so that 2^k never leaves the normal double range */
    int k1 = k / 2, k2 = k - k1;
    return (p * dm_pow2i(k1)) * dm_pow2i(k2);
}

/* expf: wide f32x4::exp stand-in. */
RAYN_HD float dm_expf(float xf) {
    if (dm_isnan(xf)) return xf;
    double x = (double)xf;
    if (x > 89.0) return dm_inff();
    if (x < -104.0) return 0.0f;
    return (float)dm_exp_core(x);
}

/* natural log of a positive, finite double that came from a float (always a normal double). */
RAYN_HD_CORE double dm_log_core(double x) {
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    uint64_t u = dm_d2u(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = dm_u2d((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1,2) */
    if (m > 1.41421356237309514547) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0); /* |s| <= 0.1716 */
    double z = s * s;
    /* 2*atanh(s) = 2s(1 + z/3 + z^2/5 + ... + z^12/25) */
    double p = 1.0 / 25.0;
    p = __builtin_fma(p, z, DM_K(1.0 / 23.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 21.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 19.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 17.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 15.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 13.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 11.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 9.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 7.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 5.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 3.0));
    p = __builtin_fma(p, z, DM_K(1.0));
    double logm = 2.0 * s * p;
    double ef = (double)e;
    return ef * LN2_HI + (ef * LN2_LO + logm);
}

/* natural log of a float (used by the Mandelbulb distance estimator, an extension outside the reference). */
RAYN_HD float dm_logf(float xf) {
    if (dm_isnan(xf) || xf < 0.0f) return dm_nanf();
    if (xf == 0.0f) return -dm_inff();
    if (xf == dm_inff()) return xf;
    return (float)dm_log_core((double)xf);
}

/* powf for the domain rayn uses (x >= 0 or NaN; src/material.rs:199,236, src/math.rs:108).
 * libm conventions kept: pow(x,0)=1, pow(1,y)=1, pow(0,y>0)=0, negative base -> NaN. */
RAYN_HD float dm_powf(float xf, float yf) {
    if (yf == 0.0f) return 1.0f;
    if (xf == 1.0f) return 1.0f;
    if (dm_isnan(xf) || dm_isnan(yf)) return dm_nanf();
    if (xf < 0.0f) return dm_nanf();
    if (xf == 0.0f) return yf > 0.0f ? 0.0f : dm_inff();
    if (xf == dm_inff()) return yf > 0.0f ? dm_inff() : 0.0f;
    double a = (double)yf * dm_log_core((double)xf);
    if (dm_isnand(a)) return dm_nanf();
    if (a > 89.0) return dm_inff();
    if (a < -104.0) return 0.0f;
    return (float)dm_exp_core(a);
}

/* sin and cos of a double; accurate for |x| up to ~1e6 (rayn's arguments are < 7). */
RAYN_HD_CORE void dm_sincos_core(double x, double* sn, double* cs) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double PIO2_1T = 6.07710050650619224932e-11; /* pi/2 - PIO2_1 */
    double kf = __builtin_floor(x * TWO_OVER_PI + 0.5);
    double r = (x - kf * PIO2_1) - kf * PIO2_1T; /* |r| <= ~pi/4 */
    double z = r * r;
    /* sin r = r + r z (s1 + z(s2 + ...)), through r^17/17! */
    double ps = 1.0 / 355687428096000.0;
    ps = __builtin_fma(ps, z, DM_K(-1.0 / 1307674368000.0));
    ps = __builtin_fma(ps, z, DM_K(1.0 / 6227020800.0));
    ps = __builtin_fma(ps, z, DM_K(-1.0 / 39916800.0));
    ps = __builtin_fma(ps, z, DM_K(1.0 / 362880.0));
    ps = __builtin_fma(ps, z, DM_K(-1.0 / 5040.0));
    ps = __builtin_fma(ps, z, DM_K(1.0 / 120.0));
    ps = __builtin_fma(ps, z, DM_K(-1.0 / 6.0));
    double S = r + r * (z * ps);
    /* cos r = 1 + z (c1 + z(c2 + ...)), through r^18/18! */
    double pc = -1.0 / 6402373705728000.0;
    pc = __builtin_fma(pc, z, DM_K(1.0 / 20922789888000.0));
    pc = __builtin_fma(pc, z, DM_K(-1.0 / 87178291200.0));
    pc = __builtin_fma(pc, z, DM_K(1.0 / 479001600.0));
    pc = __builtin_fma(pc, z, DM_K(-1.0 / 3628800.0));
    pc = __builtin_fma(pc, z, DM_K(1.0 / 40320.0));
    pc = __builtin_fma(pc, z, DM_K(-1.0 / 720.0));
    pc = __builtin_fma(pc, z, DM_K(1.0 / 24.0));
    pc = __builtin_fma(pc, z, DM_K(-0.5));
    double C = 1.0 + z * pc;
    long long k = (long long)kf;
    int q = (int)(k & 3);
    if (q == 0) { *sn = S; *cs = C; }
    else if (q == 1) { *sn = C; *cs = -S; }
    else if (q == 2) { *sn = -S; *cs = -C; }
    else { *sn = -C; *cs = S; }
}

/* f32x4::sin_cos stand-in. */
RAYN_HD void dm_sincosf(float xf, float* sn, float* cs) {
    float ax = xf < 0.0f ? -xf : xf;
    if (dm_isnan(xf) || ax == dm_inff()) { *sn = dm_nanf(); *cs = dm_nanf(); return; }
    double s, c;
    dm_sincos_core((double)xf, &s, &c);
    *sn = (float)s;
    *cs = (float)c;
}

RAYN_HD float dm_cosf(float xf) { float s, c; dm_sincosf(xf, &s, &c); return c; }
RAYN_HD float dm_sinf(float xf) { float s, c; dm_sincosf(xf, &s, &c); return s; }

/* f32x4::tan stand-in (src/light.rs:97; host: src/camera.rs:61). */
RAYN_HD float dm_tanf(float xf) {
    float ax = xf < 0.0f ? -xf : xf;
    if (dm_isnan(xf) || ax == dm_inff()) return dm_nanf();
    double s, c;
    dm_sincos_core((double)xf, &s, &c);
    return (float)(s / c);
}

/* atan of t in [0,1]. */
RAYN_HD_CORE double dm_atan_core(double t) {
    const double PI_4 = 7.85398163397448278999e-01;
    double base = 0.0, u = t;
    if (t > 0.41421356237309503) { u = (t - 1.0) / (t + 1.0); base = PI_4; }
    double z = u * u; /* z <= 0.1716 */
    /* u(1 - z/3 + z^2/5 - ... ) through z^23/47 */
    double p = -1.0 / 47.0;
    p = __builtin_fma(p, z, DM_K(1.0 / 45.0));
    p = __builtin_fma(p, z, DM_K(-1.0 / 43.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 41.0));
    p = __builtin_fma(p, z, DM_K(-1.0 / 39.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 37.0));
    p = __builtin_fma(p, z, DM_K(-1.0 / 35.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 33.0));
    p = __builtin_fma(p, z, DM_K(-1.0 / 31.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 29.0));
    p = __builtin_fma(p, z, DM_K(-1.0 / 27.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 25.0));
    p = __builtin_fma(p, z, DM_K(-1.0 / 23.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 21.0));
    p = __builtin_fma(p, z, DM_K(-1.0 / 19.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 17.0));
    p = __builtin_fma(p, z, DM_K(-1.0 / 15.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 13.0));
    p = __builtin_fma(p, z, DM_K(-1.0 / 11.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 9.0));
    p = __builtin_fma(p, z, DM_K(-1.0 / 7.0));
    p = __builtin_fma(p, z, DM_K(1.0 / 5.0));
    p = __builtin_fma(p, z, DM_K(-1.0 / 3.0));
    p = __builtin_fma(p, z, DM_K(1.0));
    return base + u * p;
}

/* f32x4::atan2 stand-in: dm_atan2f(y, x) == y.atan2(x) (src/light.rs:93-94). */
RAYN_HD float dm_atan2f(float yf, float xf) {
    const double PI = 3.14159265358979311600e+00;
    const double PI_2 = 1.57079632679489655800e+00;
    if (dm_isnan(xf) || dm_isnan(yf)) return dm_nanf();
    bool xneg = (dm_f2u(xf) >> 31) != 0;
    bool yneg = (dm_f2u(yf) >> 31) != 0;
    double ax = (double)(xneg ? -xf : xf);
    double ay = (double)(yneg ? -yf : yf);
    double a;
    double inf = (double)dm_inff();
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ax == inf && ay == inf) a = 0.78539816339744827900;
    else if (ay == inf) a = PI_2;
    else if (ax == inf) a = 0.0;
    else if (ay > ax) a = PI_2 - dm_atan_core(ax / ay);
    else a = dm_atan_core(ay / ax);
    if (xneg) a = PI - a;
    float r = (float)a;
    return yneg ? -r : r;
}

/* f32::fract as Rust defines it: x - trunc(x) (src/sampler.rs:63,92; src/filter.rs:231). */
RAYN_HD float dm_fractf(float x) { return x - __builtin_truncf(x); }

#endif /* RAYN_DETMATH_H */
