/* rayn_detmath_fast.h — cheaper evaluation of the pinned elementary functions of rayn_detmath.h WITH THE SAME RESULT BITS.
 *
 * rayn_detmath.h defines what exp / sin_cos / tan / atan2 / powf return: the float nearest to one specific binary64 evaluation
 * (long Horner chains - explicit fused multiply-adds since r2, mul + add in r1 - with truncation error < 1e-15).  The oracle evaluates
 * exactly that.  k_shade_setup spends about a fifth of its time in those functions (binary64 operations cost 1.6x an f32 one on
 * gfx950), so the kernels COULD use this header instead:
 *
 *   1. evaluate a SHORTER polynomial with fused multiply-adds on the same reduced argument (same reduction operations as the
 *      reference evaluation, so the reduced argument is bit-identical): a double d with |d - R| <= EPS * |d|, R = the reference double;
 *   2. rounding-safety test: if (float)(d - EPS*d) == (float)(d + EPS*d), float rounding is monotonic and R lies between the two,
 *      so (float)R is that same float: return it;
 *   3. otherwise (about 3e-5 of the calls), and for every special case (NaN, inf, zero, huge or tiny arguments), evaluate the
 *      reference function itself.
 *
 * STATUS: what the kernels evaluate since r3 (rayn_amd/csrc/device_core.h calls dmf_*; the oracle keeps evaluating rayn_detmath.h, so
 * every GPU parity test also checks this header).  History: with the coefficients as plain literals (r2) the compiler hoisted every
 * coefficient of the fused steps into a vector register pair - k_shade_setup went from 79 to ~125 VGPRs, 228 B of scratch per lane
 * at its 6-waves/SIMD register bound, 1.7x slower; with the coefficients pinned to scalar registers (DM_K in rayn_detmath.h) it needs
 * 79 VGPRs / 16 B of scratch and k_shade_setup is 1.4 % faster than with rayn_detmath.h (153.6 vs 155.7 ms per 1/8 share of config 3).
 * The small gain says the polynomial chains are not where the pinned functions spend their time (range reduction, binary64
 * divisions and conversions are).
 *
 * The result is bit-identical to rayn_detmath.h by construction as long as EPS really bounds |d - R|.  The analytic
 * truncation bounds are stated per function; EPS is at least 3x larger.  tests/test_detmath.py checks on the CPU (this header compiles
 * for the host too) 10^7..10^8 arguments per function for bit equality, measures the largest |d - R| / |d| seen (it must stay below
 * EPS / 3) and the fallback rate; tests/test_gpu_parity.py checks the device build against the oracle through the C-ABI probe.
 */
#ifndef RAYN_DETMATH_FAST_H
#define RAYN_DETMATH_FAST_H

#include "rayn_detmath.h"
#include "rayn_logtab.h"

#if defined(__HIPCC__) && defined(DMF_NOINLINE)
#define RAYN_SLOW static __host__ __device__ __attribute__((noinline))
#elif defined(__HIPCC__)
#define RAYN_SLOW static __host__ __device__ inline
#else
#define RAYN_SLOW static __attribute__((noinline))
#endif

/* out-of-line reference evaluations (the rare path: keeps the fast callers small).  DMF_COUNT_FALLBACKS (host test builds only)
 * counts how often they are taken. */
#ifdef DMF_COUNT_FALLBACKS
static unsigned long long dmf_fallbacks = 0;
#define DMF_NOTE_FALLBACK dmf_fallbacks++
#else
#define DMF_NOTE_FALLBACK (void)0
#endif
RAYN_SLOW float dmf_slow_exp(float x) { DMF_NOTE_FALLBACK; return dm_expf(x); }
RAYN_SLOW float dmf_slow_pow(float x, float y) { DMF_NOTE_FALLBACK; return dm_powf(x, y); }
RAYN_SLOW void dmf_slow_sincos(float x, float* s, float* c) { DMF_NOTE_FALLBACK; dm_sincosf(x, s, c); }
RAYN_SLOW float dmf_slow_tan(float x) { DMF_NOTE_FALLBACK; return dm_tanf(x); }
RAYN_SLOW float dmf_slow_atan2(float y, float x) { DMF_NOTE_FALLBACK; return dm_atan2f(y, x); }
RAYN_SLOW float dmf_slow_log(float x) { DMF_NOTE_FALLBACK; return dm_logf(x); }

/* step 2: d is within eps*|d| of the reference double; true + the float when rounding cannot differ */
/* r6: the test on the BITS of d instead of two more binary64 operations and two conversions.  A binary64 d with a binary32-normal magnitude rounds to
 * binary32 by its 29 low significand bits; the rounding boundaries (midpoints of consecutive floats) are exactly the doubles whose low 29 bits are
 * 0x10000000.  |d - reference| <= eps |d| < eps 2^53 ulp(d), so with MARGIN = the power of two >= eps 2^53 the two round alike whenever the low 29
 * bits of d are at least MARGIN away from 0x10000000 (mod 2^29: a boundary of the neighbouring float is the same pattern).  One conversion and four
 * integer operations; magnitudes outside [2^-126, 2^127) (zero, denormal or overflowing floats, inf, NaN) take the reference path. */
#ifdef DMF_ROUND_SAFE_R5 /* the r3-r5 form, kept for variant builds (tools/variants/README.md): (float)(d - eps d) == (float)(d + eps d) */
RAYN_HD bool dmf_round_safe(double d, double eps, float* out) {
    const double e = d * eps;
    const float lo = (float)(d - e), hi = (float)(d + e);
    *out = lo;
    const float a = lo < 0.0f ? -lo : lo;
    return lo == hi && a > 1.0e-36f && a < 3.0e38f;
}
#else
RAYN_HD bool dmf_round_safe(double d, double eps, float* out) {
    const uint32_t margin = eps <= 1.13e-13 ? 1024u : eps <= 9.09e-13 ? 8192u : eps <= 1.81e-12 ? 16384u : eps <= 3.63e-12 ? 32768u : eps <= 7.27e-12 ? 65536u : 0x08000000u;
    const uint64_t u = dm_d2u(d);
    *out = (float)d;
    const uint32_t t = ((uint32_t)u + margin - 0x10000000u) & 0x1FFFFFFFu;
    const uint32_t ex = (uint32_t)(u >> 52) & 0x7FFu;
    return t >= 2u * margin && ex - 897u < 253u;
}
#endif

/* e^a, a double in (-87, 88): same reduction as dm_exp_core, Taylor through r^10 with fma.
 * |r| <= 0.3466: truncation r^11/11! <= 2.2e-13, relative to e^r >= 0.707: 3.1e-13.  EPS_EXP = 1e-12. */
#define DMF_EPS_EXP 1.0e-12
RAYN_HD double dmf_exp_core(double a) {
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kf = __builtin_floor(a * LOG2E + 0.5);
    double r = (a - kf * LN2_HI) - kf * LN2_LO; /* the reference's operations: identical r */
    double p = 1.0 / 3628800.0;
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
    int k1 = k / 2, k2 = k - k1;
    return (p * dm_pow2i(k1)) * dm_pow2i(k2);
}

RAYN_HD float dmf_expf(float xf) {
    float out;
    if (xf > -87.0f && xf < 88.0f && dmf_round_safe(dmf_exp_core((double)xf), DMF_EPS_EXP, &out)) return out;
    return dmf_slow_exp(xf);
}

/* ln of a positive normal double that came from a float: same reduction as dm_log_core, series through z^8 with fma.
 * z = s^2 <= 0.02944: truncation z^9/19 <= 8.7e-16 relative to the series (>= 1). */
RAYN_HD double dmf_log_core(double x) {
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    uint64_t u = dm_d2u(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = dm_u2d((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.41421356237309514547) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 17.0;
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

/* r6: ln of a positive normal float for dmf_logf (the Mandelbulb distance estimator's logarithm, an extension outside the reference) WITHOUT the binary64
 * division and the 12-term series of dm_log_core: x = 2^e m, interval i = top 7 mantissa bits, r = m inv_i - 1 (one fma, |r| <= 2^-7; intervals >= 53
 * take m / 2 and e + 1 so that x near 1 has e = 0), ln x = e ln 2 + l_i + log1p(r), log1p through r^7 (rayn_logtab.h, tools/gen_log_table.py).
 * Error against the true logarithm: the fma's rounding of r (2^-53 absolute = 2^-45 of a 2^-8 r, but ABSOLUTE 2^-53 in the sum), the table's l_i
 * (2^-54 |l_i|), truncation r^8 / 8 <= 2^-59 |r|, the Horner steps (a few 2^-53 |r|): below 4 x 2^-53 of the result wherever no cancellation occurs - and
 * the only cancellations are factor-2 ones (e ln 2 against l_i of the opposite sign, at most half of it); the two intervals that touch 1 hold inv = 1,
 * l = 0: r = m - 1 is exact and ln x = log1p(r) alone, relative error of the series.  dm_log_core itself is within 1e-15 of the true value.
 * |d - R| <= 3e-15 |d|; EPS_LOG = 1e-13. */
#define DMF_EPS_LOG 1.0e-13
RAYN_HD double dmf_log_tab_core_t(double x, const double* __restrict__ tab /* RAYN_LOGTAB or a copy of it (k_shadow_bulb keeps one in LDS) */) {
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const uint64_t u = dm_d2u(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    const uint32_t i = (uint32_t)(u >> 45) & 127u;
    double m = dm_u2d((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1, 2) */
    if (i >= RAYN_LOGTAB_SPLIT) { m = m * 0.5; e += 1; }
    const double inv = tab[2u * i], l = tab[2u * i + 1u];
    const double r = __builtin_fma(m, inv, -1.0);
    double p = 1.0 / 7.0;
    p = __builtin_fma(p, r, DM_K(-1.0 / 6.0));
    p = __builtin_fma(p, r, DM_K(1.0 / 5.0));
    p = __builtin_fma(p, r, DM_K(-1.0 / 4.0));
    p = __builtin_fma(p, r, DM_K(1.0 / 3.0));
    p = __builtin_fma(p, r, DM_K(-0.5));
    const double lg = __builtin_fma(r * r, p, r); /* log1p(r) */
    const double ef = (double)e;
    return ef * LN2_HI + (ef * LN2_LO + (l + lg));
}
RAYN_HD double dmf_log_tab_core(double x) { return dmf_log_tab_core_t(x, RAYN_LOGTAB); }
RAYN_HD float dmf_logf_t(float xf, const double* __restrict__ tab) {
    float out;
    if (xf > 1.0e-30f && xf < 1.0e30f && dmf_round_safe(dmf_log_tab_core_t((double)xf, tab), DMF_EPS_LOG, &out)) return out;
    return dmf_slow_log(xf);
}
RAYN_HD float dmf_logf(float xf) { return dmf_logf_t(xf, RAYN_LOGTAB); }

/* x^y for finite x > 0, x != 1, finite y != 0 with |y ln x| < 87: the exponent a = y * ln x carries an absolute error
 * <= |a| * 2e-15 <= 1.8e-13 (= relative error of e^a), plus exp's 3.1e-13.  EPS_POW = 2e-12. */
#define DMF_EPS_POW 2.0e-12
RAYN_HD float dmf_powf(float xf, float yf) {
    float out;
    if (xf > 1.0e-30f && xf < 1.0e30f && xf != 1.0f && yf != 0.0f && yf > -1.0e4f && yf < 1.0e4f) {
        const double a = (double)yf * dmf_log_core((double)xf);
        if (a > -87.0 && a < 88.0 && dmf_round_safe(dmf_exp_core(a), DMF_EPS_POW, &out)) return out;
    }
    return dmf_slow_pow(xf, yf);
}

/* sin and cos of a double, |x| < 1e4: same reduction as dm_sincos_core (identical r), sin through r^13, cos through r^12 (fma).
 * |r| <= 0.7854: sin truncation r^15/15! <= 2.1e-14 |r|-relative 2.7e-14; cos truncation r^14/14! <= 3.9e-13, relative to
 * cos r >= 0.707: 5.5e-13.  EPS_SINCOS = 2e-12 (applied to each output relative to itself). */
#define DMF_EPS_SINCOS 2.0e-12
RAYN_HD void dmf_sincos_core(double x, double* sn, double* cs) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;
    const double PIO2_1T = 6.07710050650619224932e-11;
    double kf = __builtin_floor(x * TWO_OVER_PI + 0.5);
    double r = (x - kf * PIO2_1) - kf * PIO2_1T; /* the reference's operations: identical r (a fused form would differ after cancellation) */
    double z = r * r;
    double ps = 1.0 / 6227020800.0;
    ps = __builtin_fma(ps, z, DM_K(-1.0 / 39916800.0));
    ps = __builtin_fma(ps, z, DM_K(1.0 / 362880.0));
    ps = __builtin_fma(ps, z, DM_K(-1.0 / 5040.0));
    ps = __builtin_fma(ps, z, DM_K(1.0 / 120.0));
    ps = __builtin_fma(ps, z, DM_K(-1.0 / 6.0));
    double S = __builtin_fma(r * z, ps, r);
    double pc = 1.0 / 479001600.0;
    pc = __builtin_fma(pc, z, DM_K(-1.0 / 3628800.0));
    pc = __builtin_fma(pc, z, DM_K(1.0 / 40320.0));
    pc = __builtin_fma(pc, z, DM_K(-1.0 / 720.0));
    pc = __builtin_fma(pc, z, DM_K(1.0 / 24.0));
    pc = __builtin_fma(pc, z, DM_K(-0.5));
    double C = __builtin_fma(z, pc, DM_K(1.0));
    long long k = (long long)kf;
    int q = (int)(k & 3);
    if (q == 0) { *sn = S; *cs = C; }
    else if (q == 1) { *sn = C; *cs = -S; }
    else if (q == 2) { *sn = -S; *cs = -C; }
    else { *sn = -C; *cs = S; }
}

RAYN_HD void dmf_sincosf(float xf, float* sn, float* cs) {
    if (xf > -1.0e4f && xf < 1.0e4f) {
        double s, c;
        dmf_sincos_core((double)xf, &s, &c);
        float fs, fc;
        const bool ok_s = dmf_round_safe(s, DMF_EPS_SINCOS, &fs), ok_c = dmf_round_safe(c, DMF_EPS_SINCOS, &fc);
        if (ok_s && ok_c) { *sn = fs; *cs = fc; return; }
    }
    dmf_slow_sincos(xf, sn, cs);
}

/* tan = s / c (binary64 division like the reference): relative error <= the sum of both.  EPS_TAN = 4e-12. */
#define DMF_EPS_TAN 4.0e-12
RAYN_HD float dmf_tanf(float xf) {
    float out;
    if (xf > -1.0e4f && xf < 1.0e4f) {
        double s, c;
        dmf_sincos_core((double)xf, &s, &c);
        if (dmf_round_safe(s / c, DMF_EPS_TAN, &out)) return out;
    }
    return dmf_slow_tan(xf);
}

/* atan of t in [0, 1]: same reduction as dm_atan_core (identical u), series through z^14 with fma.
 * z = u^2 <= 0.1716: truncation z^15/31 <= 1.1e-13 relative to the series (>= 0.94).  The later steps (base + u*p, pi/2 - a,
 * pi - a) add terms of the same sign or subtract from a larger constant: no cancellation.  EPS_ATAN = 1e-12. */
#define DMF_EPS_ATAN 1.0e-12
RAYN_HD double dmf_atan_core(double t) {
    const double PI_4 = 7.85398163397448278999e-01;
    double base = 0.0, u = t;
    if (t > 0.41421356237309503) { u = (t - 1.0) / (t + 1.0); base = PI_4; }
    double z = u * u;
    double p = 1.0 / 29.0;
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

RAYN_HD float dmf_atan2f(float yf, float xf) {
    const double PI = 3.14159265358979311600e+00;
    const double PI_2 = 1.57079632679489655800e+00;
    const float axf = xf < 0.0f ? -xf : xf, ayf = yf < 0.0f ? -yf : yf;
    /* ordinary operands only: both finite, neither tiny nor huge (zeros, infinities, NaN: reference path) */
    if (axf > 1.0e-30f && axf < 1.0e30f && ayf > 1.0e-30f && ayf < 1.0e30f) {
        const bool xneg = xf < 0.0f, yneg = yf < 0.0f;
        const double ax = (double)axf, ay = (double)ayf;
        double a = ay > ax ? PI_2 - dmf_atan_core(ax / ay) : dmf_atan_core(ay / ax);
        if (xneg) a = PI - a;
        float r;
        if (dmf_round_safe(a, DMF_EPS_ATAN, &r)) return yneg ? -r : r;
    }
    return dmf_slow_atan2(yf, xf);
}

#endif /* RAYN_DETMATH_FAST_H */
