/*
 * det_math.h -- TEST INFRASTRUCTURE (oracle).  Not part of the product.
 *
 * Deterministic log / exp / pow used by the oracle's ETS likelihood.  The HIP
 * kernels state the same formulas independently (csrc/det_math.hpp); both sides
 * are compiled with -ffp-contract=off, so every operation below is one
 * correctly-rounded IEEE-754 binary64 operation and the two implementations
 * agree bit for bit.  libm's log()/exp() are NOT used on the parity path because
 * glibc and ROCm's ocml differ in the last ulp, which is enough to flip a
 * Nelder-Mead comparison.
 *
 * The reductions are the classic public-domain fdlibm ones
 * (log: x = 2^k (1+f), s = f/(2+f), minimax R(s^2); exp: x = k ln2 + r,
 * Remez c(r)); accuracy < 1 ulp.  No reference file corresponds to this: the
 * reference delegates to Rust's f64::ln / f64::exp inside the un-vendored
 * anofox-forecast 0.15.3 crate (finding 1).
 */
#ifndef ORACLE_DET_MATH_H
#define ORACLE_DET_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint64_t det_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double det_from_bits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

static inline double det_log(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    uint64_t u = det_bits(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (x != x) return x;                              /* NaN */
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((u << 1) == 0) return -INFINITY;           /* log(+-0) */
        if (hx >> 31) return NAN;                      /* log(x<0) */
        k -= 54;                                       /* subnormal: scale up */
        x *= 18014398509481984.0;                      /* 2^54 */
        u = det_bits(x);
        hx = (uint32_t)(u >> 32);
    } else if (hx >= 0x7ff00000u) {
        return x;                                      /* +inf */
    } else if (hx == 0x3ff00000u && (u << 32) == 0) {
        return 0.0;                                    /* log(1) */
    }
    /* reduce x into [sqrt(2)/2, sqrt(2)) */
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffu);
    x = det_from_bits(u);

    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

/* x * 2^n by exponent arithmetic only (no libm), n in a safe range. */
static inline double det_scalbn(double x, int n)
{
    if (n > 1023) {
        x *= 8.98846567431157953865e307; /* 2^1023 */
        n -= 1023;
        if (n > 1023) {
            x *= 8.98846567431157953865e307;
            n -= 1023;
            if (n > 1023) n = 1023;
        }
    } else if (n < -1022) {
        x *= 2.004168360008972778e-292;  /* 2^-1022 * 2^53 */
        n += 1022 - 53;
        if (n < -1022) {
            x *= 2.004168360008972778e-292;
            n += 1022 - 53;
            if (n < -1022) n = -1022;
        }
    }
    return x * det_from_bits((uint64_t)(0x3ff + n) << 52);
}

static inline double det_exp(double x)
{
    static const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                        invln2 = 1.44269504088896338700e+00,
                        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                        P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                        P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893383973096) return INFINITY;
    if (x < -745.13321910194110842) return 0.0;
    double ax = x < 0 ? -x : x;
    int k = 0;
    double hi, lo;
    if (ax > 0.34657359027997264) {            /* |x| > 0.5 ln2 */
        k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
        hi = x - (double)k * ln2hi;
        lo = (double)k * ln2lo;
        x = hi - lo;
    } else if (ax > 3.725290298461914e-09) {   /* |x| > 2^-28 */
        hi = x;
        lo = 0.0;
    } else {
        return 1.0 + x;
    }
    /* division-free: e^r by its Taylor polynomial of degree 13 in Horner form (|r| <= 0.35, remainder
     * < 5e-18), fused multiply-adds; the fdlibm rational form costs one fp64 division (~20 VALU
     * instructions on gfx950) per call and this function sits inside the per-step b^phi. */
    (void)P1; (void)P2; (void)P3; (void)P4; (void)P5; (void)hi; (void)lo;
    double p = 1.0 / 6227020800.0;
    p = fma(p, x, 1.0 / 479001600.0);
    p = fma(p, x, 1.0 / 39916800.0);
    p = fma(p, x, 1.0 / 3628800.0);
    p = fma(p, x, 1.0 / 362880.0);
    p = fma(p, x, 1.0 / 40320.0);
    p = fma(p, x, 1.0 / 5040.0);
    p = fma(p, x, 1.0 / 720.0);
    p = fma(p, x, 1.0 / 120.0);
    p = fma(p, x, 1.0 / 24.0);
    p = fma(p, x, 1.0 / 6.0);
    p = fma(p, x, 0.5);
    p = fma(p, x, 1.0);
    double y = fma(p, x, 1.0);
    if (k == 0) return y;
    return det_scalbn(y, k);
}

/* pow for a positive base (ETS multiplicative trend, b^phi): exp(y * log(x)). */
static inline double det_pow_pos(double x, double y) { return det_exp(y * det_log(x)); }

/*
 * b^phi of the damped multiplicative trend, evaluated once per observation: x in [2^-1000, 2^1000], 0 < y <= 1 (ets.c
 * rejects the trial point otherwise).  Table driven (tools/gen_pow_tables.py -> pow_tables.inc), because this one function
 * is most of the arithmetic of the five damped multiplicative-trend specs and sits on the critical path of their slowest
 * fits: no division, two short polynomials.
 *   ln x  = e ln2 + LOG_C[j] + log1p(r),  x = 2^e m, m in [1, 2), j = top 7 mantissa bits, r = m INV_C[j] - 1, |r| <= 2^-8,
 *           log1p(r) = r + r^2 (-1/2 + r/3 + r^2 (-1/4 + r/5 + r^2 (-1/6)))          (remainder r^7/7 < 2e-18)
 *   e^t   = 2^(n >> 6) EXP2_T[n & 63] (1 + p(s)),  n = rint(t 64/ln2), s = t - n ln2/64 (two-part), |s| <= ln2/128,
 *           p(s)     = s + s^2 (1/2 + s/6 + s^2 (1/24 + s/120 + s^2/720))             (remainder s^7/5040 < 3e-20)
 * Every operation is an IEEE-754 binary64 +, *, fma or an exact integer step, in a fixed order: csrc/det_math.hpp
 * (dm_pow_step) states the identical sequence and reads the identical tables, so kernel and checker agree bit for bit.
 * Accuracy against the exact power: < 1.5 ulp over the domain (tests/test_host_logic.py).
 */
#include "pow_tables.inc"
static const double det_pow_inv_c[ANOFOX_POW_INV_C_N] = { ANOFOX_POW_INV_C_VALUES };
static const double det_pow_log_c[ANOFOX_POW_LOG_C_N] = { ANOFOX_POW_LOG_C_VALUES };
static const double det_pow_exp2_t[ANOFOX_POW_EXP2_T_N] = { ANOFOX_POW_EXP2_T_VALUES };

static inline double det_pow_step(double x, double y)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double n_per_ln2 = 0x1.71547652b82fep+6;                         /* 64 / ln2 */
    const double L_hi = 0x1.62e42fee00000p-7, L_lo = 0x1.a39ef35793c76p-39;  /* ln2 / 64 = L_hi + L_lo, L_hi has 32 significant bits */
    const uint64_t u = det_bits(x);
    const int e = (int)(u >> 52) - 1023;                                   /* x is a positive normal number */
    const int j = (int)(u >> 45) & 127;
    const double m = det_from_bits((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    const double r = fma(m, det_pow_inv_c[j], -1.0);
    const double r2 = r * r;
    const double pa = fma(r, 1.0 / 3.0, -0.5);
    const double pb = fma(r, 0.2, -0.25);
    const double pl = fma(r2, fma(r2, -1.0 / 6.0, pb), pa);
    const double l1p = fma(r2, pl, r);
    const double de = (double)e;
    const double lg = fma(de, ln2_hi, det_pow_log_c[j]) + fma(de, ln2_lo, l1p);
    const double t = y * lg;
    const double dn = rint(n_per_ln2 * t);              /* round to nearest even; |dn| <= 64000 */
    const double s = fma(-dn, L_lo, fma(-dn, L_hi, t));
    const double s2 = s * s;
    const double qa = fma(s, 1.0 / 6.0, 0.5);
    const double qb = fma(s, 1.0 / 120.0, 1.0 / 24.0);
    const double q = fma(s2, fma(s2, 1.0 / 720.0, qb), qa);
    const double p = fma(s2, q, s);
    const int n = (int)dn;
    const double tv = det_pow_exp2_t[n & 63];
    const double ev = fma(tv, p, tv);
    return ev * det_from_bits((uint64_t)(0x3ff + (n >> 6)) << 52);
}

/*
 * Round 5: b^phi for a growth rate NEAR ONE, which is where growth rates live (|b - 1| <= 1/16 in 99.9 % of the steps of the M5-shape
 * fits, tools/pow_hist).  The binomial series (1 + r)^y = sum_k C(y, k) r^k has coefficients that depend on the exponent only: they
 * are computed ONCE per likelihood pass (det_pow_near1_coef, DET_POW_NEAR1_DEG - 1 multiplications), and the step is a Horner chain
 * of DET_POW_NEAR1_DEG fused multiply-adds -- no table, no range reduction (12 operations against 36 + two table lookups of
 * det_pow_step).  Truncation: |C(y, 12)| 16^-12 < 1e-17 for y in (0, 1]; total error < 1 ulp.  ets.c takes this branch iff
 * |b - 1| <= DET_POW_NEAR1_R (a per-series, per-step decision on the VALUE, so it is the same in the kernels whatever else
 * shares the wave) and det_pow_step otherwise.  csrc/det_math.hpp (dm_pow_near1_coef / dm_pow_near1) states the identical sequence.
 */
#define DET_POW_NEAR1_DEG 11
#define DET_POW_NEAR1_R 0x1p-4
static inline void det_pow_near1_coef(double y, double *c /* [DET_POW_NEAR1_DEG + 1], c[0] unused */)
{
    static const double inv[DET_POW_NEAR1_DEG + 1] = { 0.0, 1.0, 1.0 / 2.0, 1.0 / 3.0, 1.0 / 4.0, 1.0 / 5.0, 1.0 / 6.0, 1.0 / 7.0,
                                                       1.0 / 8.0, 1.0 / 9.0, 1.0 / 10.0, 1.0 / 11.0 };
    c[0] = 1.0;
    c[1] = y;
    for (int k = 2; k <= DET_POW_NEAR1_DEG; k++) c[k] = c[k - 1] * ((y - (double)(k - 1)) * inv[k]);
}
static inline double det_pow_near1(double r, const double *c)
{
    /* Horner's rule.  (Estrin's scheme -- depth 5 instead of 12 -- was built and measured SLOWER on the GPU, tools/ubench/fma_vgpr:
     * 109 against 89 cycles per evaluation on one wave per SIMD: a dependent fp64 FMA completes every ~7.4 cycles, an independent
     * one issues every ~5-6, so the three extra instructions cost more than the shorter chain saves.) */
    double p = c[DET_POW_NEAR1_DEG];
    for (int k = DET_POW_NEAR1_DEG - 1; k >= 1; k--) p = fma(p, r, c[k]);
    return fma(p, r, 1.0);
}

/*
 * Round 5: the ONE reciprocal of a general-class step.  Its value is the correctly rounded 1 / d (this division); what is new is
 * the DOMAIN: a denominator outside [2^-1000, 2^1000] in magnitude (or not a number) makes the trial point inadmissible -- the
 * kernels then compute the quotient with v_rcp_f64 + two Newton steps + one correction, the compiler's own division expansion
 * without its range scaling and special-case fix-up (7 instructions instead of 12), which is the correctly rounded quotient
 * exactly on that domain (csrc/det_math.hpp dm_recip; checked against the hardware division by anofox_hip_selftest).
 */
static inline int det_recip_ok(double d) { double a = fabs(d); return a >= 0x1p-1000 && a <= 0x1p+1000; }
/*
 * A multiplicative-error model divides by the one-step forecast f and accumulates sum log|f|: f must lie in [2^-120, 2^120] in
 * magnitude (1e-36 .. 1e36), else the trial point is inadmissible.  Inside that domain the running product of up to eight factors
 * times a mantissa in [1/2, 1) cannot leave the normal range, so WHERE the product is renormalised (frexp) does not change a bit of
 * (mant, eacc) -- scaling by a power of two is exact and commutes with rounding -- and the kernels may renormalise once every four
 * observations while ets.c below does it at every one.
 */
static inline int det_mulerr_f_ok(double f) { double a = fabs(f); return a >= 0x1p-120 && a <= 0x1p+120; }

#endif /* ORACLE_DET_MATH_H */
