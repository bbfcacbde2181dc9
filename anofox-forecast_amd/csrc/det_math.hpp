// det_math.hpp -- deterministic fp64 log / exp / pow for the gfx950 kernels.
//
// The ETS likelihood needs log(SSE), log|f| and b^phi.  ocml's log/exp and glibc's
// differ in the last ulp, which is enough to flip a Nelder-Mead comparison and send
// the GPU and a CPU checker down different optimiser paths.  These versions use only
// +,-,*,/ (each one IEEE-754 binary64 operation; the build uses -ffp-contract=off),
// so any IEEE machine reproduces them bit for bit.  Reductions are the classic
// fdlibm ones: log via x = 2^k(1+f), s = f/(2+f); exp via x = k ln2 + r.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace anofox {

__device__ __forceinline__ double dm_from_bits(uint64_t u) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ uint64_t dm_bits(double x) { return (uint64_t)__double_as_longlong(x); }

__device__ __forceinline__ double dm_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = dm_bits(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (x != x) return x;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((u << 1) == 0) return -__builtin_huge_val();
        if (hx >> 31) return __builtin_nan("");
        k -= 54;
        x *= 18014398509481984.0;
        u = dm_bits(x);
        hx = (uint32_t)(u >> 32);
    } else if (hx >= 0x7ff00000u) {
        return x;
    } else if (hx == 0x3ff00000u && (u << 32) == 0) {
        return 0.0;
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffull);
    x = dm_from_bits(u);

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

__device__ __forceinline__ double dm_scalbn(double x, int n)
{
    if (n > 1023) {
        x *= 8.98846567431157953865e307;
        n -= 1023;
        if (n > 1023) {
            x *= 8.98846567431157953865e307;
            n -= 1023;
            if (n > 1023) n = 1023;
        }
    } else if (n < -1022) {
        x *= 2.004168360008972778e-292;
        n += 1022 - 53;
        if (n < -1022) {
            x *= 2.004168360008972778e-292;
            n += 1022 - 53;
            if (n < -1022) n = -1022;
        }
    }
    return x * dm_from_bits((uint64_t)(0x3ff + n) << 52);
}

__device__ __forceinline__ double dm_exp(double x)
{
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    if (x != x) return x;
    if (x > 709.782712893383973096) return __builtin_huge_val();
    if (x < -745.13321910194110842) return 0.0;
    double ax = x < 0 ? -x : x;
    int k = 0;
    double hi, lo;
    if (ax > 0.34657359027997264) {
        k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
        hi = x - (double)k * ln2hi;
        lo = (double)k * ln2lo;
        x = hi - lo;
    } else if (ax > 3.725290298461914e-09) {
        hi = x;
        lo = 0.0;
    } else {
        return 1.0 + x;
    }
    // division-free: Taylor polynomial of e^r, degree 13, Horner with fused multiply-adds
    // (|r| <= 0.35: remainder < 5e-18).  Same coefficients and order as the CPU checker.
    (void)hi; (void)lo;
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
    return dm_scalbn(y, k);
}

// x^y for the per-step damped multiplicative growth b^phi: x in [2^-1000, 2^1000], 0 < y <= 1 (the caller rejects the
// trial point otherwise and ignores the value).  This one function is most of the arithmetic of the five damped
// multiplicative-trend specs and sits on the critical path of their slowest fits, so it is table driven: no division,
// two degree-6 polynomials in Estrin form (5 dependent fused multiply-adds each), the tables (tools/gen_pow_tables.py,
// 2.5 KB) in LDS.  oracle/det_math.h (det_pow_step) states the identical sequence of operations on the identical
// tables.   ln x = e ln2 + LOG_C[j] + log1p(m INV_C[j] - 1);   e^t = 2^(n >> 6) EXP2_T[n & 63] (1 + p(t - n ln2/64)).
#include "pow_tables.inc"
static __device__ const double dm_pow_inv_c_src[ANOFOX_POW_INV_C_N] = { ANOFOX_POW_INV_C_VALUES };
static __device__ const double dm_pow_log_c_src[ANOFOX_POW_LOG_C_N] = { ANOFOX_POW_LOG_C_VALUES };
static __device__ const double dm_pow_exp2_t_src[ANOFOX_POW_EXP2_T_N] = { ANOFOX_POW_EXP2_T_VALUES };
// Round 4 experiment, kept as a build switch: the workgroup's copy of the tables can be REPLICATED (-DANOFOX_POW_R128 / _R64: entry j
// stored R times side by side, a lane reads copy (its position in its LDS lane group) mod R -- with 16 / 32 copies no two lanes of a
// ds_read_b128 / ds_read_b64 lane group can meet on a bank, MI355X_MICROARCH.md "LDS").  Measured: NO gain (profiles/r04_README.md:
// 8 / 8 and 2 / 4 copies run the damped multiplicative-trend fit in 137.9 / 137.3 ms against 137.1 ms) -- the two lookups cost their
// LATENCY on the step's critical path (taking them out altogether, with wrong values: 113 ms), not bank-conflict cycles.  Default 1 / 1.
#ifndef ANOFOX_POW_R128
#define ANOFOX_POW_R128 1
#endif
#ifndef ANOFOX_POW_R64
#define ANOFOX_POW_R64 1
#endif
constexpr int DM_POW_R128 = ANOFOX_POW_R128, DM_POW_R64 = ANOFOX_POW_R64;
constexpr int DM_POW_EXP_BASE = 2 * ANOFOX_POW_INV_C_N * DM_POW_R128;
constexpr int DM_POW_TAB_DOUBLES = DM_POW_EXP_BASE + ANOFOX_POW_EXP2_T_N * DM_POW_R64;

// the workgroup's copy: pair (j, copy c) = {INV_C[j], LOG_C[j]} at doubles 2 (j R128 + c) (one 128-bit LDS read), then EXP2_T[n] copy c
// at DM_POW_EXP_BASE + n R64 + c
__device__ __forceinline__ double *dm_pow_tab()
{
    __shared__ __attribute__((aligned(16))) double tab[DM_POW_TAB_DOUBLES];
    return tab;
}

// once per kernel, by ALL threads of the workgroup (any size), before the first dm_pow_step and before any thread leaves
__device__ __forceinline__ void dm_pow_tab_init()
{
    double *tab = dm_pow_tab();
    const int nt = blockDim.x;
    for (int idx = threadIdx.x; idx < ANOFOX_POW_INV_C_N * DM_POW_R128; idx += nt) {
        const int j = idx / DM_POW_R128;
        tab[2 * idx] = dm_pow_inv_c_src[j];
        tab[2 * idx + 1] = dm_pow_log_c_src[j];
    }
    for (int idx = threadIdx.x; idx < ANOFOX_POW_EXP2_T_N * DM_POW_R64; idx += nt) tab[DM_POW_EXP_BASE + idx] = dm_pow_exp2_t_src[idx / DM_POW_R64];
    __syncthreads();
}

// this lane's copies (offsets in doubles): its position inside its ds_read_b128 lane group -- {0-3, 12-15, 20-27} and
// {4-11, 16-19, 28-31} of each 32-lane half -- and inside its ds_read_b64 group (the 32-lane half)
struct DmPowLane { int o128, o64; };
__device__ __forceinline__ DmPowLane dm_pow_lane()
{
    const int l = threadIdx.x & 31;
    const int gp = l < 4 ? l : (l < 12 ? l - 4 : (l < 20 ? l - 8 : (l < 28 ? l - 12 : l - 16)));
    DmPowLane r;
    r.o128 = 2 * (gp % DM_POW_R128);
    r.o64 = DM_POW_EXP_BASE + (l % DM_POW_R64);
    return r;
}

__device__ __forceinline__ double dm_pow_step(double x, double y, const DmPowLane &cp)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double n_per_ln2 = 0x1.71547652b82fep+6;                           // 64 / ln2
    const double L_hi = 0x1.62e42fee00000p-7, L_lo = 0x1.a39ef35793c76p-39;  // ln2 / 64 = L_hi + L_lo, L_hi has 32 significant bits
    const double *tab = dm_pow_tab();
    const uint64_t u = dm_bits(x);
    const uint32_t hx = (uint32_t)(u >> 32);
    const int e = (int)(hx >> 20) - 1023;                                    // x is a positive normal number
    const int j = (int)(hx >> 13) & 127;
    const double m = dm_from_bits((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const d2_t ic = *reinterpret_cast<const d2_t *>(tab + 2 * DM_POW_R128 * j + cp.o128);      // {INV_C[j], LOG_C[j]}, this lane's copy
    const double r = fma(m, ic.x, -1.0);
    const double r2 = r * r;
    const double pa = fma(r, 1.0 / 3.0, -0.5);
    const double pb = fma(r, 0.2, -0.25);
    const double pl = fma(r2, fma(r2, -1.0 / 6.0, pb), pa);
    const double l1p = fma(r2, pl, r);
    const double de = (double)e;
    const double lg = fma(de, ln2_hi, ic.y) + fma(de, ln2_lo, l1p);
    const double t = y * lg;
    const double dn = __builtin_rint(n_per_ln2 * t);    // round to nearest even (v_rndne_f64); |dn| <= 64000
    const double s = fma(-dn, L_lo, fma(-dn, L_hi, t));
    const double s2 = s * s;
    const double qa = fma(s, 1.0 / 6.0, 0.5);
    const double qb = fma(s, 1.0 / 120.0, 1.0 / 24.0);
    const double q = fma(s2, fma(s2, 1.0 / 720.0, qb), qa);
    const double p = fma(s2, q, s);
    const int n = (int)dn;
    const double tv = tab[DM_POW_R64 * (n & 63) + cp.o64];
    const double ev = fma(tv, p, tv);
    // ev 2^(n >> 6): |n >> 6| <= 1000 and ev in [1/2, 2], so the scaling is exact -- v_ldexp_f64, one instruction instead of the three
    // integer operations + multiply that build and apply the power of two (same value: the oracle multiplies by the constructed double)
    return __builtin_ldexp(ev, n >> 6);
}

// Round 5: b^phi for a growth rate near one (|b - 1| <= 1/16: 99.9 % of the steps of the M5-shape fits).  The binomial series
// (1 + r)^y = sum_k C(y, k) r^k: the coefficients depend on the exponent only and are computed once per pass into registers
// (dm_pow_near1_coef), the step is a Horner chain of DM_POW_NEAR1_DEG fused multiply-adds -- no table (no LDS round trip on the
// recursion's critical path), no range reduction: 12 instructions against 36 + two lookups.  oracle/det_math.h (det_pow_near1_coef,
// det_pow_near1) states the identical sequence; the branch is taken on the lane's own VALUE (ets_device.hpp), so a lane's result
// does not depend on what shares its wave.
constexpr int DM_POW_NEAR1_DEG = 11;
constexpr double DM_POW_NEAR1_R = 0x1p-4;
struct DmPowNear1 { double c[DM_POW_NEAR1_DEG + 1]; };
__device__ __forceinline__ void dm_pow_near1_coef(double y, DmPowNear1 &o)
{
    constexpr double inv[DM_POW_NEAR1_DEG + 1] = { 0.0, 1.0, 1.0 / 2.0, 1.0 / 3.0, 1.0 / 4.0, 1.0 / 5.0, 1.0 / 6.0, 1.0 / 7.0,
                                                   1.0 / 8.0, 1.0 / 9.0, 1.0 / 10.0, 1.0 / 11.0 };
    o.c[0] = 1.0;
    o.c[1] = y;
#pragma unroll
    for (int k = 2; k <= DM_POW_NEAR1_DEG; k++) o.c[k] = o.c[k - 1] * ((y - (double)(k - 1)) * inv[k]);
}
__device__ __forceinline__ double dm_pow_near1(double r, const DmPowNear1 &o)
{
    // Horner's rule (the oracle's order).  Estrin's scheme was measured slower (tools/ubench/fma_vgpr: 109 against 89 cycles per
    // evaluation on one wave per SIMD, 156 against 128 on two): a dependent fp64 FMA completes every ~7.4 cycles and an independent one
    // issues every ~5-6, so on this machine a step costs its instruction COUNT, not its dependent depth
    double p = o.c[DM_POW_NEAR1_DEG];
#pragma unroll
    for (int k = DM_POW_NEAR1_DEG - 1; k >= 1; k--) p = fma(p, r, o.c[k]);
    return fma(p, r, 1.0);
}

// Round 5: the one reciprocal of a general-class step, 1 / d for |d| in [2^-1000, 2^1000] (the caller rejects the trial point
// otherwise, as oracle/ets.c does: det_recip_ok).  This is the compiler's own expansion of an fp64 division -- v_rcp_f64, two
// Newton steps, quotient, residual, correction -- without the two v_div_scale, the v_div_fmas select and the v_div_fixup that only
// matter for operands whose reciprocal or residual leaves the normal range: on this domain no operand is scaled (v_div_scale
// returns it unchanged and leaves VCC clear, so v_div_fmas IS the fma below) and v_div_fixup returns its first operand, so the
// result is bit for bit the hardware's correctly rounded quotient, which is the CPU's 1.0 / d.  7 instructions instead of 12.
// (anofox_hip_selftest compares it with the compiled division on 2^28 operands per call.)
__device__ __forceinline__ double dm_recip(double d)
{
    const double x0 = __builtin_amdgcn_rcp(d);
    const double e0 = fma(-d, x0, 1.0);
    const double x1 = fma(x0, e0, x0);
    const double e1 = fma(-d, x1, 1.0);
    const double x2 = fma(x1, e1, x1);          // (the numerator is one: the expansion's n * x2 is x2 itself)
    const double e2 = fma(-d, x2, 1.0);
    return fma(e2, x2, x2);
}
__device__ __forceinline__ bool dm_recip_ok(double d) { const double a = fabs(d); return a >= 0x1p-1000 && a <= 0x1p+1000; }

// general x^y, x > 0 (forecast path: the exponent is a partial geometric sum and may exceed 1)
__device__ __forceinline__ double dm_pow_pos(double x, double y) { return dm_exp(y * dm_log(x)); }

} // namespace anofox
