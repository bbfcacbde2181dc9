// classic_device.hpp -- SES / Holt / additive Holt-Winters / SeasonalES passes (SSE objective).
//
// These are the models the reference reaches through SimpleExponentialSmoothing::{new,auto},
// HoltLinearTrend::auto, HoltWinters::auto(p, Additive) and SeasonalES::{new,optimized}
// (crates/anofox-fcst-core/src/forecast.rs:1102-1144, 1206-1232) and the fallback chain of
// ETS-without-spec / failed AutoETS (forecast.rs:1327-1336, 1631-1640).  Formulations are the
// ones that reproduce the reference KATs (test/sql/ts_model_distinctness.test:116,141):
//   SES            l0 = y0 ; l += alpha (y - l), t >= 1
//   Holt           l0 = y0, b0 = y1 - y0 ; l' = (l+b) + alpha e ; b += beta ((l'-l) - b)
//   Holt-Winters   l0 = mean(season 1), b0 = (mean(season 2) - mean(season 1))/m, s_j = y_j - l0,
//                  t >= m ; l' = q + alpha ((y - s) - q) ; b += beta ((l'-l) - b) ; s += gamma ((y - l') - s)
//   SeasonalES     s_j = y_j ; s_j += alpha (y - s_j), t >= m
// Seasonal rings live in LDS (run-time period).  Same K = 4 candidate sharing as the ETS pass.
#pragma once
#include <utility>
#include <type_traits>
#include "ets_device.hpp"

namespace anofox {

enum ClassicKind { CK_SES = 0, CK_HOLT = 1, CK_HW = 2, CK_SEASONAL_ES = 3 };

template <int KIND> struct ClassicDim { static constexpr int value = KIND == CK_HOLT ? 2 : (KIND == CK_HW ? 3 : 1); };

struct ClassicFinalOut { double *yhat; int h; bool write; };

// The lane's column in blocks of S rows, the next block requested before the current one is consumed (the first version loaded one
// row per step behind a per-lane condition: every step waited for its own HBM round trip, ~1,000 cycles).  Row indices are
// wave-uniform, so the clamp to the wave's last row is scalar arithmetic; a lane past its own length ignores what it read.
template <int S, class Step>
__device__ __forceinline__ void classic_stream(const double *yp, size_t ld, int t_begin, int wave_len, int len, Step step)
{
    double cur[S], nxt[S];
    auto load = [&](double (&dst)[S], int t0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < S; j++) {
            int t = t0 + j;
            t = t < wave_len ? t : wave_len - 1;
            dst[j] = yp[(size_t)t * ld];
        }
    };
    load(cur, t_begin);
    for (int base = t_begin; base < wave_len; base += S) {
        load(nxt, base + S);
#pragma unroll
        for (int j = 0; j < S; j++)
            if (base + j < len) step(cur[j]);
#pragma unroll
        for (int j = 0; j < S; j++) cur[j] = nxt[j];
    }
}
constexpr int CLASSIC_S = 16;

// The same stream with the position inside the block as a compile-time constant (blocks of S = 2 x 7 rows starting at t = 7: the
// phase of a step is its position mod 7, so a weekly seasonal ring can live in registers with static indices).
template <int S, class Step, int... J>
__device__ __forceinline__ void classic_block_steps(const double (&cur)[S], int base, int len, Step &step, std::integer_sequence<int, J...>)
{
    ((base + J < len ? step(std::integral_constant<int, J>{}, cur[J]) : void()), ...);
}
template <int S, class Step>
__device__ __forceinline__ void classic_stream_static(const double *yp, size_t ld, int t_begin, int wave_len, int len, Step step)
{
    double cur[S], nxt[S];
    auto load = [&](double (&dst)[S], int t0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < S; j++) {
            int t = t0 + j;
            t = t < wave_len ? t : wave_len - 1;
            dst[j] = yp[(size_t)t * ld];
        }
    };
    load(cur, t_begin);
    for (int base = t_begin; base < wave_len; base += S) {
        load(nxt, base + S);
        classic_block_steps<S>(cur, base, len, step, std::make_integer_sequence<int, S>{});
#pragma unroll
        for (int j = 0; j < S; j++) cur[j] = nxt[j];
    }
}

template <int KIND, int K, bool FINAL, bool M7 = false>
__device__ __forceinline__ void classic_pass(const SeriesView &v, int m,
                                             const double (&cand)[K][ClassicDim<KIND>::value],
                                             double (&fout)[K], double *ring, const ClassicFinalOut *fin)
{
    const int lane = threadIdx.x;
    const double *yp = v.y;
    const size_t ld = v.ld;
    double sse[K];
#pragma unroll
    for (int k = 0; k < K; k++) sse[k] = 0.0;

    if constexpr (KIND == CK_SES) {
        double l[K];
        const double y0 = yp[0];
#pragma unroll
        for (int k = 0; k < K; k++) l[k] = y0;
        classic_stream<CLASSIC_S>(yp, ld, 1, v.wave_len, v.len, [&](const double yv) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                double e = yv - l[k];
                sse[k] = fma(e, e, sse[k]);
                l[k] = fma(cand[k][0], e, l[k]);
            }
        });
        if constexpr (FINAL) {
            if (fin->write) for (int i = 0; i < fin->h; i++) fin->yhat[i] = l[0];
        }
    } else if constexpr (KIND == CK_HOLT) {
        double l[K], b[K];
        const double y0 = yp[0], y1 = yp[ld];
#pragma unroll
        for (int k = 0; k < K; k++) { l[k] = y0; b[k] = y1 - y0; }
        classic_stream<CLASSIC_S>(yp, ld, 1, v.wave_len, v.len, [&](const double yv) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                double f = l[k] + b[k];
                double e = yv - f;
                sse[k] = fma(e, e, sse[k]);
                double ln = fma(cand[k][0], e, f);
                b[k] = fma(cand[k][1], (ln - l[k]) - b[k], b[k]);
                l[k] = ln;
            }
        });
        if constexpr (FINAL) {
            if (fin->write) for (int i = 1; i <= fin->h; i++) fin->yhat[i - 1] = l[0] + (double)i * b[0];
        }
    } else if constexpr (KIND == CK_HW && M7) {
        // m = 7: the seasonal ring in registers (same operations as the run-time-period branch below)
        double l[K], b[K], sr[K][7];
        double m1 = 0.0, m2 = 0.0;
        for (int i = 0; i < 7; i++) m1 += yp[(size_t)i * ld];
        for (int i = 7; i < 14; i++) m2 += yp[(size_t)i * ld];
        m1 /= 7.0;
        m2 /= 7.0;
#pragma unroll
        for (int k = 0; k < K; k++) { l[k] = m1; b[k] = (m2 - m1) / 7.0; }
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const double s0 = yp[(size_t)i * ld] - m1;
#pragma unroll
            for (int k = 0; k < K; k++) sr[k][i] = s0;
        }
        classic_stream_static<14>(yp, ld, 7, v.wave_len, v.len, [&](auto jc, const double yv) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value % 7;
#pragma unroll
            for (int k = 0; k < K; k++) {
                double s = sr[k][j];
                double q = l[k] + b[k];
                double e = yv - (q + s);
                sse[k] = fma(e, e, sse[k]);
                double ln = fma(cand[k][0], (yv - s) - q, q);
                b[k] = fma(cand[k][1], (ln - l[k]) - b[k], b[k]);
                sr[k][j] = fma(cand[k][2], (yv - ln) - s, s);
                l[k] = ln;
            }
        });
        if constexpr (FINAL) {
            if (fin->write)
                for (int i = 1; i <= fin->h; i++) {
                    const int jj = (v.len + i - 1) % 7;
                    double sv = sr[0][0];
#pragma unroll
                    for (int q = 1; q < 7; q++) sv = (jj == q) ? sr[0][q] : sv;
                    fin->yhat[i - 1] = (l[0] + (double)i * b[0]) + sv;
                }
        }
    } else if constexpr (KIND == CK_SEASONAL_ES && M7) {
        double sr[K][7];
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const double s0 = yp[(size_t)i * ld];
#pragma unroll
            for (int k = 0; k < K; k++) sr[k][i] = s0;
        }
        classic_stream_static<14>(yp, ld, 7, v.wave_len, v.len, [&](auto jc, const double yv) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value % 7;
#pragma unroll
            for (int k = 0; k < K; k++) {
                double s = sr[k][j];
                double e = yv - s;
                sse[k] = fma(e, e, sse[k]);
                sr[k][j] = fma(cand[k][0], e, s);
            }
        });
        if constexpr (FINAL) {
            if (fin->write)
                for (int i = 0; i < fin->h; i++) {
                    const int jj = (v.len + i) % 7;
                    double sv = sr[0][0];
#pragma unroll
                    for (int q = 1; q < 7; q++) sv = (jj == q) ? sr[0][q] : sv;
                    fin->yhat[i] = sv;
                }
        }
    } else if constexpr (KIND == CK_HW) {
        double l[K], b[K];
        double m1 = 0.0, m2 = 0.0;
        for (int i = 0; i < m; i++) m1 += yp[(size_t)i * ld];
        for (int i = m; i < 2 * m; i++) m2 += yp[(size_t)i * ld];
        m1 /= (double)m;
        m2 /= (double)m;
#pragma unroll
        for (int k = 0; k < K; k++) { l[k] = m1; b[k] = (m2 - m1) / (double)m; }
        for (int i = 0; i < m; i++) {
            double s0 = yp[(size_t)i * ld] - m1;
#pragma unroll
            for (int k = 0; k < K; k++) ring[(k * m + i) * NM_BLOCK + lane] = s0;
        }
        int j = 0;              // phase of the step: every step up to the lane's length runs, so it advances with the steps
        classic_stream<CLASSIC_S>(yp, ld, m, v.wave_len, v.len, [&](const double yv) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                double s = ring[(k * m + j) * NM_BLOCK + lane];
                double q = l[k] + b[k];
                double e = yv - (q + s);
                sse[k] = fma(e, e, sse[k]);
                double ln = fma(cand[k][0], (yv - s) - q, q);
                b[k] = fma(cand[k][1], (ln - l[k]) - b[k], b[k]);
                ring[(k * m + j) * NM_BLOCK + lane] = fma(cand[k][2], (yv - ln) - s, s);
                l[k] = ln;
            }
            j = (j + 1 == m) ? 0 : j + 1;
        });
        if constexpr (FINAL) {
            if (fin->write)
                for (int i = 1; i <= fin->h; i++) {
                    int jj = (v.len + i - 1) % m;
                    fin->yhat[i - 1] = (l[0] + (double)i * b[0]) + ring[(0 * m + jj) * NM_BLOCK + lane];
                }
        }
    } else { // CK_SEASONAL_ES
        for (int i = 0; i < m; i++) {
            double s0 = yp[(size_t)i * ld];
#pragma unroll
            for (int k = 0; k < K; k++) ring[(k * m + i) * NM_BLOCK + lane] = s0;
        }
        int j = 0;
        classic_stream<CLASSIC_S>(yp, ld, m, v.wave_len, v.len, [&](const double yv) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                double s = ring[(k * m + j) * NM_BLOCK + lane];
                double e = yv - s;
                sse[k] = fma(e, e, sse[k]);
                ring[(k * m + j) * NM_BLOCK + lane] = fma(cand[k][0], e, s);
            }
            j = (j + 1 == m) ? 0 : j + 1;
        });
        if constexpr (FINAL) {
            if (fin->write)
                for (int i = 0; i < fin->h; i++) fin->yhat[i] = ring[(0 * m + (v.len + i) % m) * NM_BLOCK + lane];
        }
    }
#pragma unroll
    for (int k = 0; k < K; k++) fout[k] = sse[k];
}

// The family on the ETS round kernels (sub-cases of the same machinery with their own start values and the
// SSE as objective): ClassicCfg<KIND> stands where an EtsCfg stands, ClassicRoundModel where EtsModel<Cfg, MS, 1> stands.
template <int KIND_>
struct ClassicCfg {
    static constexpr int KIND = KIND_;
    static constexpr int DIM = ClassicDim<KIND_>::value;
    static constexpr int E = C_ADD;
    static constexpr int T = (KIND_ == CK_HOLT || KIND_ == CK_HW) ? C_ADD : C_NONE;
    static constexpr int S = (KIND_ == CK_HW || KIND_ == CK_SEASONAL_ES) ? C_ADD : C_NONE;
    static constexpr bool D = false;
    static constexpr bool ADDITIVE = true;
    static constexpr bool CLASSIC = true;
};

template <int KIND, int MS = -1>
struct ClassicRoundModel {
    static constexpr int DIM = ClassicDim<KIND>::value;
    SeriesView v;
    EtsInit in;             // only in.m is used (the start values come from the series itself)
    double *ring;
    __device__ void bounds(double (&lo)[DIM], double (&hi)[DIM], double (&x0)[DIM]) const
    {
#pragma unroll
        for (int i = 0; i < DIM; i++) { lo[i] = PAR_LO; hi[i] = PAR_HI; }
        if constexpr (KIND == CK_SES || KIND == CK_SEASONAL_ES) x0[0] = 0.5;
        else if constexpr (KIND == CK_HOLT) { x0[0] = 0.3; x0[1] = 0.1; }
        else { x0[0] = 0.3; x0[1] = 0.1; x0[2] = 0.1; }
    }
    __device__ double eval1(const double (&x)[DIM]) const
    {
        double c1[1][DIM], f1[1];
#pragma unroll
        for (int i = 0; i < DIM; i++) c1[0][i] = x[i];
        classic_pass<KIND, 1, false, MS == 7>(v, in.m, c1, f1, ring, nullptr);
        return f1[0];
    }
    // the four trial points of an iteration in four adjacent lanes (same exchange as EtsModel::eval)
    __device__ void eval(const double (&cand)[NM_K][DIM], double (&f)[NM_K]) const
    {
        const int sub = threadIdx.x & 3;
        double mine[DIM];
#pragma unroll
        for (int i = 0; i < DIM; i++) mine[i] = sub == 0 ? cand[0][i] : (sub == 1 ? cand[1][i] : (sub == 2 ? cand[2][i] : cand[3][i]));
        const double f1 = eval1(mine);
        const int base = threadIdx.x & ~3;
#pragma unroll
        for (int k = 0; k < NM_K; k++) f[k] = __shfl(f1, base + k);
    }
};

template <int KIND>
struct ClassicModel {
    static constexpr int DIM = ClassicDim<KIND>::value;
    SeriesView v;
    int m;
    double *ring;
    __device__ void bounds(double (&lo)[DIM], double (&hi)[DIM], double (&x0)[DIM]) const
    {
#pragma unroll
        for (int i = 0; i < DIM; i++) { lo[i] = PAR_LO; hi[i] = PAR_HI; }
        if constexpr (KIND == CK_SES || KIND == CK_SEASONAL_ES) x0[0] = 0.5;
        else if constexpr (KIND == CK_HOLT) { x0[0] = 0.3; x0[1] = 0.1; }
        else { x0[0] = 0.3; x0[1] = 0.1; x0[2] = 0.1; }
    }
    __device__ void eval(const double (&cand)[NM_K][DIM], double (&f)[NM_K]) const
    {
        classic_pass<KIND, NM_K, false>(v, m, cand, f, ring, nullptr);
    }
};

} // namespace anofox
