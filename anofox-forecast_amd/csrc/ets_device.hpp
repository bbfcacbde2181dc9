// ets_device.hpp -- the streamed ETS likelihood pass for gfx950.
//
// Layout: the batch is one time-major fp64 block Y[t * ld + s]; lane <-> series, so the 64
// lanes of a wave read 512 contiguous bytes per time step.  A pass evaluates K = 4 parameter
// candidates per lane against the same y stream (state of all four in VGPRs); the seasonal
// ring is in VGPRs for compile-time periods (loop unrolled by the period so ring indices are
// static) and in LDS for a run-time period.  No MFMA: the recursion is a scalar scan.
//
// Arithmetic contract (mirrors oracle/ets.c, which restates the published innovations
// state-space recursion):
//   additive class (error A, trend N/A/Ad, season N/A) -- error-correction form
//       q = l + phi b ; f = q + s ; e = y - f
//       l' = fma(alpha, e, q) ; b' = fma(alpha beta*, e, phi b) ; s' = fma(gamma*(1-alpha), e, s)
//   other specs -- general form (forecast::etscalc lineage) with beta/alpha = beta*.
//   objective = n log(SSE) [+ 2 sum log|f| for multiplicative error], +inf if inadmissible.
// Built with -ffp-contract=off; fma() only where written.
#pragma once
#include <hip/hip_runtime.h>
#include "det_math.hpp"
#include "nm.hpp"

namespace anofox {

enum { C_NONE = 0, C_ADD = 1, C_MUL = 2 };

constexpr double ETS_TOL = 1.0e-10;
constexpr double ETS_HUGEN = 1.0e10;
constexpr double ETS_LN2 = 0.693147180559945309417232121458;
constexpr double PAR_LO = 1.0e-4, PAR_HI = 0.9999, PHI_LO = 0.8, PHI_HI = 0.98;
constexpr int ETS_MAX_PERIOD = 64;

struct SeriesView {
    const double *y;   // already offset to this lane's series: element t at y[t * ld]
    size_t ld;
    int len;           // this lane's length (0 = no series)
    int wave_len;      // max len over the wave (uniform)
    int wave_min_len;  // min len over the wave's active lanes (uniform)
};

__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { int w = __shfl_xor(v, o); v = w > v ? w : v; }
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { int w = __shfl_xor(v, o); v = w < v ? w : v; }
    return v;
}

template <int ERR, int TREND, bool DAMPED, int SEAS>
struct EtsCfg {
    static constexpr int E = ERR, T = TREND, S = SEAS;
    static constexpr bool D = DAMPED;
    static constexpr int DIM = 1 + (TREND != C_NONE) + (SEAS != C_NONE) + (DAMPED ? 1 : 0);
    static constexpr bool ADDITIVE = (ERR == C_ADD) && (TREND != C_MUL) && (SEAS != C_MUL);
};

struct EtsPar { double alpha, bstar, phi, beta, gamma; };
struct EtsState { double l, b, sse, mant; int eacc; int bad; };

template <class Cfg>
__device__ __forceinline__ void ets_unpack(const double (&x)[Cfg::DIM], EtsPar &p)
{
    int k = 0;
    p.alpha = x[k++];
    p.bstar = 0.0;
    double gstar = 0.0;
    p.phi = 1.0;
    if constexpr (Cfg::T != C_NONE) p.bstar = x[k++];
    if constexpr (Cfg::S != C_NONE) gstar = x[k++];
    if constexpr (Cfg::D) p.phi = x[k++];
    p.beta = p.alpha * p.bstar;
    p.gamma = gstar * (1.0 - p.alpha);
}

// One time step for one candidate.  `s` is the seasonal state of this phase (updated in place).
template <class Cfg>
__device__ __forceinline__ void ets_step(const EtsPar &p, EtsState &st, double y, double &s)
{
    if constexpr (Cfg::ADDITIVE) {
        double phib = 0.0, q = st.l;
        if constexpr (Cfg::T == C_ADD) {
            phib = Cfg::D ? p.phi * st.b : st.b;
            q = st.l + phib;
        }
        double f = q;
        if constexpr (Cfg::S == C_ADD) f = q + s;
        double e = y - f;
        st.sse = fma(e, e, st.sse);
        st.l = fma(p.alpha, e, q);
        if constexpr (Cfg::T == C_ADD) st.b = fma(p.beta, e, phib);
        if constexpr (Cfg::S == C_ADD) s = fma(p.gamma, e, s);
    } else {
        double phib = 0.0, q = st.l;
        if constexpr (Cfg::T == C_ADD) {
            phib = Cfg::D ? p.phi * st.b : st.b;
            q = st.l + phib;
        } else if constexpr (Cfg::T == C_MUL) {
            if (!(st.b > 0.0)) st.bad = 1;
            phib = Cfg::D ? dm_pow_pos(st.b, p.phi) : st.b;
            q = st.l * phib;
        }
        double f = q;
        if constexpr (Cfg::S == C_ADD) f = q + s;
        else if constexpr (Cfg::S == C_MUL) f = q * s;
        // ONE reciprocal per step serves every quotient: 1/f (relative error; 1/s = q/f and 1/q = s/f
        // for a multiplicative season) and 1/l (multiplicative growth) come from 1/(f l), 1/f or 1/l.
        // An fp64 division is ~20 VALU instructions here, so this halves the general recursion.
        constexpr bool need_f = (Cfg::E == C_MUL) || (Cfg::S == C_MUL);
        double rf = 0.0, rl = 0.0;
        if constexpr (need_f && Cfg::T == C_MUL) { double R = 1.0 / (f * st.l); rf = R * st.l; rl = R * f; }
        else if constexpr (need_f) rf = 1.0 / f;
        else if constexpr (Cfg::T == C_MUL) rl = 1.0 / st.l;
        double e = y - f;
        if constexpr (Cfg::E == C_MUL) {
            e = e * rf;
            int ex;
            st.mant = frexp(st.mant * fabs(f), &ex);
            st.eacc += ex;
        }
        st.sse = fma(e, e, st.sse);
        double pp = y;
        if constexpr (Cfg::S == C_ADD) pp = y - s;
        else if constexpr (Cfg::S == C_MUL) pp = y * (rf * q);
        double lnew = fma(p.alpha, pp - q, q);
        if constexpr (Cfg::T == C_ADD) {
            double r = lnew - st.l;
            st.b = fma(p.bstar, r - phib, phib);
        } else if constexpr (Cfg::T == C_MUL) {
            double r = lnew * rl;
            st.b = fma(p.bstar, r - phib, phib);
        }
        if constexpr (Cfg::S == C_ADD) {
            double tt = y - q;
            s = fma(p.gamma, tt - s, s);
        } else if constexpr (Cfg::S == C_MUL) {
            double tt = y * (rf * s);
            s = fma(p.gamma, tt - s, s);
        }
        st.l = lnew;
    }
}

template <class Cfg>
__device__ __forceinline__ double ets_objective_value(const EtsState &st, int n)
{
    if (st.bad || !(fabs(st.sse) <= 1.7976931348623157e308)) return __builtin_huge_val();
    double lik = (double)n * dm_log(st.sse);
    if constexpr (Cfg::E == C_MUL) lik = lik + 2.0 * (dm_log(st.mant) + (double)st.eacc * ETS_LN2);
    if (lik != lik) return __builtin_huge_val();
    if (lik < -1.0e10) lik = -1.0e10;
    return lik;
}

// Initial states of one (series, spec): level, growth and where the seasonal figure lives.
struct EtsInit {
    double l0, b0;
    const double *fig;   // seasonal figure of this lane's series: phase j at fig[j * fig_ld]
    size_t fig_ld;
    int m;               // run-time period (== MS when MS > 0)
};

// Final pass output target (candidate 0 only)
struct EtsFinalOut { double *yhat; int h; double *sse_out; };

// The pass.  MS > 0: compile-time period, ring in VGPRs.  MS == 0: no seasonality.
// MS == -1: run-time period, ring in LDS (`ring`, K * m * 64 doubles).
template <class Cfg, int MS, int K, bool FINAL>
__device__ __forceinline__ void ets_pass(const SeriesView &v, const EtsInit &in,
                                         const double (&cand)[K][Cfg::DIM], double (&fout)[K],
                                         double *ring, const EtsFinalOut *fin)
{
    static_assert((Cfg::S == C_NONE) == (MS == 0), "MS == 0 iff no seasonal component");
    const int lane = threadIdx.x;
    EtsPar par[K];
    EtsState st[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        ets_unpack<Cfg>(cand[k], par[k]);
        st[k].l = in.l0; st[k].b = in.b0; st[k].sse = 0.0; st[k].mant = 1.0; st[k].eacc = 0; st[k].bad = 0;
    }
    const double *yp = v.y;
    const size_t ld = v.ld;

    if constexpr (MS > 0) {
        double s[K][MS];
#pragma unroll
        for (int j = 0; j < MS; j++) {
            double f0 = in.fig[(size_t)j * in.fig_ld];
#pragma unroll
            for (int k = 0; k < K; k++) s[k][j] = f0;
        }
        // Full blocks (every active lane of the wave in range): no per-step predicate.  NB register
        // buffers of one period each keep NB-1 blocks of y loads in flight ahead of the recursion, so a
        // wave that is alone on its SIMD (late rounds) still covers the HBM miss latency.
        constexpr int NB = (K == 1 && Cfg::ADDITIVE) ? 4 : 2;   // division-heavy steps are long enough with one block ahead
        const int nfull = v.wave_min_len / MS;
        double yb[NB][MS];
#pragma unroll
        for (int i = 0; i < NB; i++)
            if (i < nfull) {
#pragma unroll
                for (int j = 0; j < MS; j++) yb[i][j] = yp[(size_t)(i * MS + j) * ld];
            }
        for (int blk = 0; blk < nfull; blk += NB) {
#pragma unroll
            for (int i = 0; i < NB; i++) {
                if (blk + i < nfull) {
#pragma unroll
                    for (int j = 0; j < MS; j++)
#pragma unroll
                        for (int k = 0; k < K; k++) ets_step<Cfg>(par[k], st[k], yb[i][j], s[k][j]);
                    const int nxt = blk + i + NB;
                    if (nxt < nfull) {
#pragma unroll
                        for (int j = 0; j < MS; j++) yb[i][j] = yp[(size_t)(nxt * MS + j) * ld];
                    }
                }
            }
        }
        int t0 = nfull * MS;
        // ragged tail: per-lane predicate
        for (; t0 < v.wave_len; t0 += MS) {
#pragma unroll
            for (int j = 0; j < MS; j++) {
                const int t = t0 + j;
                if (t < v.len) {
                    const double yv = yp[(size_t)t * ld];
#pragma unroll
                    for (int k = 0; k < K; k++) ets_step<Cfg>(par[k], st[k], yv, s[k][j]);
                }
            }
        }
        if constexpr (FINAL) {
            double pp = par[0].phi, phistar = par[0].phi;
            for (int i = 0; i < fin->h; i++) {
                double f;
                if constexpr (Cfg::T == C_NONE) f = st[0].l;
                else if constexpr (Cfg::T == C_ADD) f = st[0].l + phistar * st[0].b;
                else f = (st[0].b > 0.0) ? st[0].l * dm_pow_pos(st[0].b, phistar) : __builtin_nan("");
                const int j = (v.len + i) % MS;
                double sv = s[0][0];
#pragma unroll
                for (int jj = 1; jj < MS; jj++) sv = (j == jj) ? s[0][jj] : sv;
                f = (Cfg::S == C_ADD) ? f + sv : f * sv;
                if (v.len > 0) fin->yhat[i] = f;
                pp = pp * par[0].phi;
                phistar = phistar + pp;
            }
        }
    } else if constexpr (MS == 0) {
        double dummy = 0.0;
        constexpr int U = 8;
        constexpr int NB = (K == 1 && Cfg::ADDITIVE) ? 4 : 2;   // division-heavy steps are long enough with one block ahead
        const int nfull = v.wave_min_len / U;
        double yb[NB][U];
#pragma unroll
        for (int i = 0; i < NB; i++)
            if (i < nfull) {
#pragma unroll
                for (int j = 0; j < U; j++) yb[i][j] = yp[(size_t)(i * U + j) * ld];
            }
        for (int blk = 0; blk < nfull; blk += NB) {
#pragma unroll
            for (int i = 0; i < NB; i++) {
                if (blk + i < nfull) {
#pragma unroll
                    for (int j = 0; j < U; j++)
#pragma unroll
                        for (int k = 0; k < K; k++) ets_step<Cfg>(par[k], st[k], yb[i][j], dummy);
                    const int nxt = blk + i + NB;
                    if (nxt < nfull) {
#pragma unroll
                        for (int j = 0; j < U; j++) yb[i][j] = yp[(size_t)(nxt * U + j) * ld];
                    }
                }
            }
        }
        int t0 = nfull * U;
        for (; t0 < v.wave_len; t0++) {
            if (t0 < v.len) {
                const double yv = yp[(size_t)t0 * ld];
#pragma unroll
                for (int k = 0; k < K; k++) ets_step<Cfg>(par[k], st[k], yv, dummy);
            }
        }
        if constexpr (FINAL) {
            double pp = par[0].phi, phistar = par[0].phi;
            for (int i = 0; i < fin->h; i++) {
                double f;
                if constexpr (Cfg::T == C_NONE) f = st[0].l;
                else if constexpr (Cfg::T == C_ADD) f = st[0].l + phistar * st[0].b;
                else f = (st[0].b > 0.0) ? st[0].l * dm_pow_pos(st[0].b, phistar) : __builtin_nan("");
                if (v.len > 0) fin->yhat[i] = f;
                pp = pp * par[0].phi;
                phistar = phistar + pp;
            }
        }
    } else {
        // run-time period: ring[(k * m + j) * 64 + lane] in LDS
        const int m = in.m;
        for (int j = 0; j < m; j++) {
            double f0 = in.fig[(size_t)j * in.fig_ld];
#pragma unroll
            for (int k = 0; k < K; k++) ring[(k * m + j) * NM_BLOCK + lane] = f0;
        }
        int j = 0;
        for (int t = 0; t < v.wave_len; t++) {
            if (t < v.len) {
                const double yv = yp[(size_t)t * ld];
#pragma unroll
                for (int k = 0; k < K; k++) {
                    double sv = ring[(k * m + j) * NM_BLOCK + lane];
                    ets_step<Cfg>(par[k], st[k], yv, sv);
                    ring[(k * m + j) * NM_BLOCK + lane] = sv;
                }
            }
            j = (j + 1 == m) ? 0 : j + 1;
        }
        if constexpr (FINAL) {
            double pp = par[0].phi, phistar = par[0].phi;
            for (int i = 0; i < fin->h; i++) {
                double f;
                if constexpr (Cfg::T == C_NONE) f = st[0].l;
                else if constexpr (Cfg::T == C_ADD) f = st[0].l + phistar * st[0].b;
                else f = (st[0].b > 0.0) ? st[0].l * dm_pow_pos(st[0].b, phistar) : __builtin_nan("");
                const int jj = (v.len + i) % m;
                double sv = ring[(0 * m + jj) * NM_BLOCK + lane];
                f = (Cfg::S == C_ADD) ? f + sv : f * sv;
                if (v.len > 0) fin->yhat[i] = f;
                pp = pp * par[0].phi;
                phistar = phistar + pp;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; k++) fout[k] = ets_objective_value<Cfg>(st[k], v.len);
    if constexpr (FINAL) { if (fin->sse_out && v.len > 0) *fin->sse_out = st[0].sse; }
}

// Nelder-Mead model adaptor.  CPL = candidates per lane: 4 -> one lane evaluates all four trial points
// of its problem (throughput form, one y load feeds four recursions); 1 -> the four trial points of a
// problem sit in four adjacent lanes (latency form: a quarter of the per-pass latency and VGPRs, four
// times the waves), results exchanged with wave shuffles.  Both give bit-identical objective values.
template <class Cfg, int MS, int CPL>
struct EtsModel {
    static constexpr int DIM = Cfg::DIM;
    SeriesView v;
    EtsInit in;
    double *ring;
    __device__ void bounds(double (&lo)[DIM], double (&hi)[DIM], double (&x0)[DIM]) const
    {
        int k = 0;
        lo[k] = PAR_LO; hi[k] = PAR_HI; x0[k] = 0.3; k++;
        if constexpr (Cfg::T != C_NONE) { lo[k] = PAR_LO; hi[k] = PAR_HI; x0[k] = 0.1; k++; }
        if constexpr (Cfg::S != C_NONE) { lo[k] = PAR_LO; hi[k] = PAR_HI; x0[k] = 0.1; k++; }
        if constexpr (Cfg::D) { lo[k] = PHI_LO; hi[k] = PHI_HI; x0[k] = 0.9; k++; }
    }
    __device__ double eval1(const double (&x)[DIM]) const
    {
        double c1[1][DIM], f1[1];
#pragma unroll
        for (int i = 0; i < DIM; i++) c1[0][i] = x[i];
        ets_pass<Cfg, MS, 1, false>(v, in, c1, f1, ring, nullptr);
        return f1[0];
    }
    __device__ void eval(const double (&cand)[NM_K][DIM], double (&f)[NM_K]) const
    {
        if constexpr (CPL == NM_K) {
            ets_pass<Cfg, MS, NM_K, false>(v, in, cand, f, ring, nullptr);
        } else {
            const int sub = threadIdx.x & 3;
            double mine[1][DIM], f1[1];
#pragma unroll
            for (int i = 0; i < DIM; i++)
                mine[0][i] = sub == 0 ? cand[0][i] : (sub == 1 ? cand[1][i] : (sub == 2 ? cand[2][i] : cand[3][i]));
            ets_pass<Cfg, MS, 1, false>(v, in, mine, f1, ring, nullptr);
            const int base = threadIdx.x & ~3;
#pragma unroll
            for (int k = 0; k < NM_K; k++) f[k] = __shfl(f1[0], base + k);
        }
    }
};

} // namespace anofox
