// prep.hip -- per-series statistics and ETS initial states in ONE streamed sweep over the time-major block (+ one more for the
// intervals' sd wherever no final pass can carry it).
//
//   the sweep: sum (-> mean), positivity / constancy flags; the classical decomposition -- a sliding window of the last
//           L = 2*(m/2)+1 observations gives the centred moving average of every interior point, the detrended value (y - trend,
//           y / trend) is added to its phase's accumulator; the sums of the full-sample least-squares line of the plain series and,
//           per phase, of the seasonally adjusted one (the figure is applied once per phase when the sweep has ended); the level-only
//           start from the first max(10, 2 m) values.  Period 7: everything in registers (prep_kernel<7>).  Any other period: the
//           figures and the per-phase sums come from season_figures_kernel (a workgroup per series, the series in LDS), the sweep
//           (prep_kernel<0>) reads them.
//   sd:     sum of squared deviations from the mean (forecast.rs:2558-2591: a two-pass sum by definition) -- a second sweep here, or
//           none when the batch's one final pass carries it (PrepArgs::skip_sd).
// Same operations in the same order as oracle/ets.c (ets_init_states) and oracle/forecast.c (intervals), so the results are
// bit-identical; only the number of times y is streamed changed (1-2 instead of ~22).
#include <type_traits>
#include "ets_device.hpp"
#include "kernels.hpp"

namespace anofox {

// Rows of the time-major block, S at a time, for a wave whose lanes own adjacent columns: range-checked buffer loads
// (descriptor per block from scalars, row stride in the scalar offset, the lane's column in the vector offset; rows past
// `rows` read as zeros) -- the loader of ets_pass (ets_device.hpp).  The S rows of the NEXT block are requested before the
// current block is consumed, so a wave keeps S loads in flight instead of waiting for every row.
template <int S>
struct RowLoader {
    const char *base; unsigned col_bytes; size_t row_bytes, total_bytes;
    __device__ __forceinline__ void load(double (&buf)[S], const int row0) const
    {
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
        const size_t off = (size_t)row0 * row_bytes;
        const size_t rem = off < total_bytes ? total_bytes - off : 0;
        const unsigned nrec = rem > 0xffffffffull ? 0xffffffffu : (unsigned)rem;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(base + (rem ? off : 0)), 0, nrec, 0x00020000);
#pragma unroll
        for (int j = 0; j < S; j++) {
            const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, col_bytes, (unsigned)(j * row_bytes), 0);
            buf[j] = __builtin_bit_cast(double, w);
        }
    }
};
// block length of the streamed rows: 16, or three revolutions of a compile-time period MR (ring slots and phases of a block are
// compile-time constants then).  One wave per 64 series is all the parallelism this kernel has (477 waves for the M5 block: every
// sum is sequential in time, and the mean / sd are the reference's own in-tree arithmetic, forecast.rs:2558-2591, so the time axis
// is not split).  Two buffers alternate without a copy.  What bounds the m = 7 sweep is the wave's own instruction stream and its
// registers (26 instructions a step, 256 VGPRs + AGPR traffic for the unrolled block's accumulators), not bytes in flight -- measured
// on the M5 block (one sweep of 487 MB, fixed-parameter ETS(A,A,A) step in brackets): blocks of 2 / 3 / 4 / 6 revolutions 0.295 / 0.288 /
// 0.310 / 0.318 ms per step; THREE buffers of 28 rows (two blocks always in flight) 238 us for the sweep against 138, FOUR of 14 rows 212 us.
#ifndef ANOFOX_PREP_REVS
#define ANOFOX_PREP_REVS 3
#endif
template <int MR> constexpr int prep_block() { return MR > 0 ? ANOFOX_PREP_REVS * MR : 16; }

// Every period but 7 (round 5; periods above the LDS ring limit only, until then): the classical decomposition costs T * (m + 1)
// multiply-adds per series (every centred moving average is its own sequential sum, as the oracle writes it) -- 1.4 M for T = 1,913,
// m = 755 -- and with one wave per 64 series, window ring in an HBM scratch, prep_kernel spent 206 ms on the 36,000 columns of a merged
// batch before its first fit could start.  season_figures_kernel gives a series a WORKGROUP instead: the series in LDS, one centre per
// thread (its window summed in the oracle's order: acc = acc + w_k y_k, k ascending), then one phase per thread (the detrended values of
// a phase added in time order), then the normalisation and the per-phase sums of the trend start -- every sum keeps the order of
// oracle/ets.c ets_init_states, so the results are the same bits.  prep_kernel then runs with pre_fig = 1: one sweep without a window.
// The period is read PER SERIES (m_col), the block-of-64 grouping of a merged batch is not needed here.
constexpr int SEASON_THREADS = 256;
constexpr int SEASON_LDS_DOUBLES = 12288;        // 96 KB of the CU's 160
constexpr int SEASON_LONG_GRID = 1024;
template <bool USE_LDS>
__global__ __launch_bounds__(SEASON_THREADS) void season_figures_kernel(const PrepArgs a, size_t scratch_stride)
{
    extern __shared__ double season_lds[];
    __shared__ double sh_fmean[2];
    const int tid = threadIdx.x;
    double *x = USE_LDS ? season_lds : a.scratch + (size_t)blockIdx.x * scratch_stride;
    double *tr = x + a.t_rows;
    double *fa = tr + a.t_rows;
    double *fm = fa + a.m;
    for (int s = blockIdx.x; s < a.n_series; s += gridDim.x) {
        const int n = a.len[s];
        const int m = a.m_col ? a.m_col[s] : a.m;
        if (n <= 0 || m < 2 || m > ETS_MAX_PERIOD || n < 2 * m) continue;          // (block-uniform) not seasonal: nobody reads its figures
        const int half = m / 2, L = 2 * half + 1;
        const double w = 1.0 / (double)m;
        const double wend = (m % 2 == 0) ? 0.5 / (double)m : w;
        int pos = 1;
        for (int i = tid; i < n; i += SEASON_THREADS) {
            const double v = a.y[(size_t)i * a.ld + s];
            x[i] = v;
            if (!(v > 0.0)) pos = 0;
        }
        const bool positive = __syncthreads_and(pos) != 0 && !(a.skip_types & 2);     // (multiplicative figures: strictly positive series, and somebody to use them)
        const int c_last = n - 1 - half;                                         // centres half .. c_last
        for (int c = half + tid; c <= c_last; c += SEASON_THREADS) {
            const double *q = x + (c - half);
            double acc = 0.0;
            acc = acc + wend * q[0];
#pragma unroll 8
            for (int k = 1; k < L - 1; k++) acc = acc + w * q[k];
            acc = acc + wend * q[L - 1];
            tr[c] = acc;
        }
        __syncthreads();
        for (int p = tid; p < m; p += SEASON_THREADS) {
            int c = p >= half ? p : p + ((half - p + m - 1) / m) * m;            // first centre of phase p (centre c has phase c mod m)
            double sa = 0.0, sm = 0.0, cn = 0.0;
            for (; c <= c_last; c += m) {
                const double yc = x[c], t = tr[c];
                sa = sa + (yc - t);
                if (positive) sm = sm + (yc / t);
                cn += 1.0;
            }
            fa[p] = sa / cn;
            if (positive) fm[p] = sm / cn;
        }
        __syncthreads();
        if (tid < 2 && (tid == 0 || positive)) {
            const double *f = tid == 0 ? fa : fm;
            double tot = 0.0;
            for (int j = 0; j < m; j++) tot = tot + f[j];
            sh_fmean[tid] = tot / (double)m;
        }
        __syncthreads();
        const double mean_a = sh_fmean[0], mean_m = positive ? sh_fmean[1] : 1.0;
        // normalised figures -> HBM (and kept in fa / fm), and the per-phase sums of the least-squares start (oracle/ets.c ets_init_states):
        // one phase per thread, the additions of a phase in time order; `tr` is free by now
        double *psy = tr, *psxy = tr + m;
        for (int p = tid; p < m; p += SEASON_THREADS) {
            const double fja = fa[p] - mean_a;
            fa[p] = fja;
            a.fig_add[(size_t)p * a.ld + s] = fja;
            if (positive) {
                double fj = fm[p] / mean_m;
                if (!(fj >= 1.0e-2)) fj = 1.0e-2;
                fm[p] = fj;
                a.fig_mul[(size_t)p * a.ld + s] = fj;
            }
            double sy = 0.0, sxy = 0.0;
            for (int t = p; t < n; t += m) {
                const double v = x[t];
                sy = sy + v;
                sxy = sxy + (double)(t + 1) * v;
            }
            psy[p] = sy; psxy[p] = sxy;
        }
        __syncthreads();
        // sy = sum_p (Sy_p - n_p fig_p), sxy = sum_p (Sxy_p - Sx_p fig_p) (additive), sum_p Sy_p / fig_p (multiplicative), p ascending: left
        // in the (season type, additive trend) slots of l0 / b0, which prep_kernel reads before it writes the states there
        if (a.l0 && tid < 2 && (tid == 0 || positive)) {
            double sy = 0.0, sxy = 0.0;
            for (int j = 0; j < m; j++) {
                if (tid == 0) {
                    const long long cnt = ((long long)n - j + m - 1) / m;
                    const long long sx = cnt * (long long)(j + 1) + (long long)m * (cnt * (cnt - 1) / 2);
                    sy = sy + (psy[j] - (double)cnt * fa[j]);
                    sxy = sxy + (psxy[j] - (double)sx * fa[j]);
                } else {
                    sy = sy + psy[j] / fm[j];
                    sxy = sxy + psxy[j] / fm[j];
                }
            }
            const size_t slot = (size_t)((tid == 0 ? 1 : 2) * 3 + 1) * a.ld + s;
            a.l0[slot] = sy;
            a.b0[slot] = sxy;
        }
        __syncthreads();                    // the next series overwrites x
    }
}

size_t season_scratch_doubles(int n_series, int t_rows, int m_max)
{
    const size_t per = 2 * (size_t)t_rows + 2 * (size_t)m_max;
    if (per <= (size_t)SEASON_LDS_DOUBLES) return 0;
    return (size_t)std::min(n_series, SEASON_LONG_GRID) * per;
}

void launch_season_figures(const PrepArgs &a, hipStream_t stream)
{
    if (a.n_series <= 0) return;
    const size_t per = 2 * (size_t)a.t_rows + 2 * (size_t)a.m;
    if (per <= (size_t)SEASON_LDS_DOUBLES) {
        const size_t bytes = per * sizeof(double);
        if (bytes > 48 * 1024)
            anofox_check_attr(hipFuncSetAttribute((const void *)season_figures_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL(season_figures_kernel<true>, dim3((unsigned)std::min(a.n_series, 65536)), dim3(SEASON_THREADS), bytes, stream, a, (size_t)0);
    } else {
        if (!a.scratch) throw std::runtime_error("season figures: long series need the scratch area");
        hipLaunchKernelGGL(season_figures_kernel<false>, dim3((unsigned)std::min(a.n_series, SEASON_LONG_GRID)), dim3(SEASON_THREADS), 0, stream, a, per);
    }
}

// One sweep (round 5; two until then).  What the second sweep used to hold:
//   * the least-squares start of the seasonally adjusted series needs the NORMALISED figures, known only at the end of a sweep -- it is
//     taken over per-phase sums now (oracle/ets.c ets_init_states: Sy_p = sum of y_t, Sxy_p = sum of (t + 1) y_t over the t of phase p,
//     in time order; sy = sum_p (Sy_p - n_p fig_p), sxy = sum_p (Sxy_p - Sx_p fig_p); a multiplicative figure divides the phase sums),
//     which one sweep accumulates beside the decomposition;
//   * the level-only start is the mean of the first max(10, 2 m) adjusted values: m = 7 reads those rows again (a few KB per wave), the others know their figures in passing;
//   * the intervals' population sd is a two-pass sum by definition (forecast.rs:2558-2591): a batch with one candidate spec lets its
//     final pass carry it (skip_sd; ets_final_kernel), any other batch keeps a second sweep here that does nothing else.
// MR > 0: compile-time odd period (7): the window ring and every per-phase accumulator are VGPR arrays with compile-time indices (the
// block is three revolutions long) -- no LDS traffic at all.  MR == 0: any other period, the figures (and the per-phase sums of the
// start) come from season_figures_kernel (pre_fig), the period is read per lane.
template <int MR = 0>
__global__ __launch_bounds__(NM_BLOCK) void prep_kernel(const PrepArgs a)
{
    constexpr int PREP_S = prep_block<MR>();
    static_assert(MR == 0 || (MR % 2 == 1 && MR >= 3), "register variant: odd periods");
    const int lane = threadIdx.x;
    const int s = blockIdx.x * NM_BLOCK + lane;
    const bool valid = s < a.n_series;
    const int n = valid ? a.len[s] : 0;
    const int wave_n = wave_max_i32(n);
    if (wave_n == 0) return;
    const size_t ld = a.ld;
    const int wave_rows = __builtin_amdgcn_readfirstlane(wave_n);
    RowLoader<PREP_S> rows;
    rows.base = (const char *)a.y;
    rows.col_bytes = (unsigned)(valid ? s : 0) * 8u;
    rows.row_bytes = ld * 8;
    rows.total_bytes = (size_t)wave_rows * rows.row_bytes;
    double buf0[PREP_S], buf1[PREP_S];
    // fn(block, first row) over the series' rows on alternating buffers: the rows of the block after next are requested before a block is consumed
    auto stream_rows = [&](auto &&fn) __attribute__((always_inline)) {
        for (int base = 0; base < wave_rows; base += 2 * PREP_S) {
            rows.load(buf1, base + PREP_S);
            fn(buf0, base);
            rows.load(buf0, base + 2 * PREP_S);
            if (base + PREP_S < wave_rows) fn(buf1, base + PREP_S);
        }
    };
    const bool states = a.l0 != nullptr;
    const int m = MR > 0 ? MR : (a.m_col ? (valid ? a.m_col[s] : 1) : a.m);       // (merged batch: the series' own period; a.m bounds the sizes)
    const bool want_season = states && m >= 2 && m <= ETS_MAX_PERIOD;
    constexpr int MRA = MR > 0 ? MR : 1;
    // window ring: the raw values (the centre is read from it) and their products with the weight 1 / m -- the window mean adds the SAME
    // seven rounded products in the same order whether they are formed when the value arrives or when the window is summed (six
    // multiplications a step fewer); per-phase accumulators of the figures (the counts are closed forms) and of the start's sums
    double rR[MRA], rW[MRA], sAR[MRA], sMR[MRA], pSy[MRA], pSxy[MRA];
#pragma unroll
    for (int j = 0; j < MRA; j++) { rR[j] = 0.0; rW[j] = 0.0; sAR[j] = 0.0; sMR[j] = 0.0; pSy[j] = 0.0; pSxy[j] = 0.0; }
    const double w = want_season ? 1.0 / (double)m : 0.0;
    const bool seasonal = want_season && n >= 2 * m && (MR > 0 || a.pre_fig != 0);
    const bool want_mul = !(a.skip_types & 2);          // some candidate spec has a multiplicative season
    const int K0 = 10 > n ? n : 10;
    int Km = 2 * m > 10 ? 2 * m : 10;
    if (Km > n) Km = n;
    const double *figA = (MR == 0 && seasonal) ? a.fig_add + s : nullptr;
    const double *figM = (MR == 0 && seasonal) ? a.fig_mul + s : nullptr;

    // ---- the sweep ----------------------------------------------------------------------------------------
    double sum = 0.0;
    bool positive = true, constant = true, has_nan = false;
    double sy[3] = {0, 0, 0}, sxy[3] = {0, 0, 0}, sk[3] = {0, 0, 0}, ysa0[3] = {0, 0, 0}, ysa1[3] = {0, 0, 0};
    rows.load(buf0, 0);
    const double y0 = n > 0 ? buf0[0] : 0.0;
    int jph = 0;                        // t % m (generic variant: per lane)
    auto sweep_block = [&](const double (&buf)[PREP_S], const int base) __attribute__((always_inline)) {
      // the multiplicative figure (one IEEE division per step) only matters for strictly positive series: once every series of
      // the wave has shown a value <= 0 (intermittent demand: within the first blocks) the wave stops computing it
      const bool wave_pos = want_mul && __any(positive && base < n);
#pragma unroll
      for (int jj = 0; jj < PREP_S; jj++) {
        const int t = base + jj;
        if (t < n) {
            const double v = buf[jj];
            sum += v;
            if (!(v > 0.0)) positive = false;
            if (v != y0) constant = false;
            if (v != v) has_nan = true;
            if (states) {
                const double tv = (double)(t + 1) * v;
                sy[0] = sy[0] + v;
                sxy[0] = sxy[0] + tv;
                if (t < K0) sk[0] = sk[0] + v;
                if (t == 0) ysa0[0] = v;
                if (t == 1) ysa1[0] = v;
                if constexpr (MR > 0) {
                    if (seasonal) {
                        constexpr int HALF = MR / 2;
                        const int sl = jj % MR;                                  // compile-time after unrolling: base is a multiple of MR
                        const double wv = w * v;
#pragma unroll
                        for (int q = 0; q < MR; q++) {
                            rR[q] = (q == sl) ? v : rR[q];
                            rW[q] = (q == sl) ? wv : rW[q];
                            pSy[q] = (q == sl) ? pSy[q] + v : pSy[q];
                            pSxy[q] = (q == sl) ? pSxy[q] + tv : pSxy[q];
                        }
                        if (t >= MR - 1) {
                            double acc = 0.0;
#pragma unroll
                            for (int k = 0; k < MR; k++) acc = acc + rW[(sl + 1 + k) % MR];         // odd period: every weight is 1 / m
                            const double yc = rR[(sl + 1 + HALF) % MR];
                            const int phc = ((jj - HALF) % MR + MR) % MR;
#pragma unroll
                            for (int q = 0; q < MR; q++) sAR[q] = (q == phc) ? sAR[q] + (yc - acc) : sAR[q];
                            if (wave_pos) {
                                const double ratio = yc / acc;
#pragma unroll
                                for (int q = 0; q < MR; q++) sMR[q] = (q == phc) ? sMR[q] + ratio : sMR[q];
                            }
                        }
                    }
                } else {
                    // figures known (season_figures_kernel): the level-only start and the first two adjusted values in passing
                    if (seasonal && t < Km) {
                        const double va = v - figA[(size_t)jph * ld];
                        const double vm = v / figM[(size_t)jph * ld];       // (used for strictly positive series only: their figures exist)
                        sk[1] = sk[1] + va;
                        sk[2] = sk[2] + vm;
                        if (t == 0) { ysa0[1] = va; ysa0[2] = vm; }
                        if (t == 1) { ysa1[1] = va; ysa1[2] = vm; }
                        jph = jph + 1 == m ? 0 : jph + 1;
                    }
                }
            }
        }
      }
    };
    // Full blocks of the register variant run as STRAIGHT-LINE code: once every series of the wave is inside its sample for the whole
    // block (and past the first one, so the window is full) no step needs a per-lane branch -- flags become mask operations, the
    // per-phase updates have compile-time targets -- and the 21 steps of a block form ONE basic block.  The per-lane `if (t < n)` of
    // the gated form makes every step its own basic block, so the seven dependent additions of a step's window mean (same order as
    // the oracle: the sum is sequential by definition) cannot overlap with the neighbouring steps' chains: a wave alone on its
    // SIMD then waits out every fp64 latency -- 283 cycles per step measured.  Same operations on the same values: a lane without a
    // series (padding) computes into registers nobody reads.
    const int wave_min_n = __builtin_amdgcn_readfirstlane(wave_min_i32(n > 0 ? n : 0x7fffffff));
    auto sweep_fast = [&](const double (&buf)[PREP_S], const int base, auto pos_tag) __attribute__((always_inline)) {
        constexpr bool POS = decltype(pos_tag)::value;
        if constexpr (MR > 0) {
            constexpr int HALF = MR / 2;
#pragma unroll
            for (int jj = 0; jj < PREP_S; jj++) {
                const double v = buf[jj];
                sum += v;
                positive = positive & (v > 0.0);
                constant = constant & (v == y0);
                has_nan = has_nan | (v != v);
                const double tv = (double)(base + jj + 1) * v;
                sy[0] = sy[0] + v;
                sxy[0] = sxy[0] + tv;
                const int sl = jj % MR;
                rR[sl] = v;
                rW[sl] = w * v;
                pSy[sl] = pSy[sl] + v;
                pSxy[sl] = pSxy[sl] + tv;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < MR; k++) acc = acc + rW[(sl + 1 + k) % MR];
                const double yc = rR[(sl + 1 + HALF) % MR];
                const int phc = ((jj - HALF) % MR + MR) % MR;
                sAR[phc] = sAR[phc] + (yc - acc);
                if constexpr (POS) sMR[phc] = sMR[phc] + yc / acc;
            }
        }
    };
    auto sweep_any = [&](const double (&buf)[PREP_S], const int base) __attribute__((always_inline)) {
        // (every series of such a block is seasonal: n >= base + PREP_S >= 6 m; past the first block no step takes part in the level-only
        //  start or is t = 0, 1)
        if (MR > 0 && states && base >= PREP_S && base + PREP_S <= wave_min_n) {
            if (want_mul && __any(positive && base < n)) sweep_fast(buf, base, std::true_type{});
            else sweep_fast(buf, base, std::false_type{});
        } else sweep_block(buf, base);
    };
    stream_rows(sweep_any);
    if (n <= 0) return;
    const double mean = sum / (double)n;
    a.mean[s] = mean;
    a.flags[s] = (positive ? SF_POSITIVE : 0u) | (constant ? SF_CONSTANT : 0u) | (has_nan ? SF_HAS_NAN : 0u);
    const bool useA = seasonal, useM = seasonal && positive && want_mul;

    if constexpr (MR > 0) {
        if (seasonal) {
            // seasonal figures (normalised), written to HBM for the fit kernels
#pragma unroll
            for (int type = 1; type <= 2; type++) {
                if (type == 2 && !useM) break;
                double tot = 0.0;
#pragma unroll
                for (int j = 0; j < MR; j++) {
                    // centres c = HALF .. n - 1 - HALF of phase j (c mod m == j): first = j (j >= HALF) or j + m
                    const int c_first = j >= MR / 2 ? j : j + MR, c_last = n - 1 - MR / 2;
                    const double cn = (double)(c_last >= c_first ? (c_last - c_first) / MR + 1 : 0);
                    const double fj = (type == 1 ? sAR[j] : sMR[j]) / cn;
                    if (type == 1) sAR[j] = fj; else sMR[j] = fj;
                    tot = tot + fj;
                }
                const double fmean = tot / (double)m;
                double *fig = (type == 1 ? a.fig_add : a.fig_mul) + s;
#pragma unroll
                for (int j = 0; j < MR; j++) {
                    double fj = type == 1 ? sAR[j] : sMR[j];
                    if (type == 1) fj = fj - fmean;
                    else {
                        fj = fj / fmean;
                        if (!(fj >= 1.0e-2)) fj = 1.0e-2;
                    }
                    if (type == 1) sAR[j] = fj; else sMR[j] = fj;
                    fig[(size_t)j * ld] = fj;
                }
            }
            // the least-squares sums of the adjusted series from the per-phase sums (oracle/ets.c ets_init_states, same order) ...
#pragma unroll
            for (int j = 0; j < MR; j++) {
                const long long cnt = ((long long)n - j + MR - 1) / MR;                       // t = j, j + m, ... < n
                const long long sx = cnt * (long long)(j + 1) + (long long)MR * (cnt * (cnt - 1) / 2);
                sy[1] = sy[1] + (pSy[j] - (double)cnt * sAR[j]);
                sxy[1] = sxy[1] + (pSxy[j] - (double)sx * sAR[j]);
                if (useM) {
                    sy[2] = sy[2] + pSy[j] / sMR[j];
                    sxy[2] = sxy[2] + pSxy[j] / sMR[j];
                }
            }
            // ... and the level-only start: the first rows again (Km <= 2 m = 14)
            for (int t = 0; t < Km; t++) {
                const double v = a.y[(size_t)t * ld + s];
                const int p = t % MR;
                double fa = sAR[0], fm = sMR[0];
#pragma unroll
                for (int q = 1; q < MR; q++) { fa = (q == p) ? sAR[q] : fa; fm = (q == p) ? sMR[q] : fm; }
                const double va = v - fa;
                sk[1] = sk[1] + va;
                if (t == 0) ysa0[1] = va;
                if (t == 1) ysa1[1] = va;
                if (useM) {
                    const double vm = v / fm;
                    sk[2] = sk[2] + vm;
                    if (t == 0) ysa0[2] = vm;
                    if (t == 1) ysa1[2] = vm;
                }
            }
        }
    } else if (seasonal) {
        // season_figures_kernel left the sums of the adjusted series in the slots this kernel is about to fill
        sy[1] = a.l0[(size_t)(1 * 3 + 1) * ld + s];
        sxy[1] = a.b0[(size_t)(1 * 3 + 1) * ld + s];
        if (useM) {
            sy[2] = a.l0[(size_t)(2 * 3 + 1) * ld + s];
            sxy[2] = a.b0[(size_t)(2 * 3 + 1) * ld + s];
        }
    }

    // ---- the intervals' sd: a second sweep unless the batch's final pass carries it -------------------------------
    if (!a.skip_sd) {
        double var = 0.0;
        rows.load(buf0, 0);
        auto var_block = [&](const double (&buf)[PREP_S], const int base) __attribute__((always_inline)) {
#pragma unroll
            for (int jj = 0; jj < PREP_S; jj++) {
                if (base + jj < n) {
                    const double dv = buf[jj] - mean;
                    var += dv * dv;
                }
            }
        };
        stream_rows(var_block);
        a.sd[s] = sqrt(var / (double)n);
    }
    if (!states) return;
    const double dn = (double)n;
    const double sx = dn * (dn + 1.0) / 2.0;
    const double sxx = dn * (dn + 1.0) * (2.0 * dn + 1.0) / 6.0;
    for (int st = 0; st <= 2; st++) {
        if (st == 1 && !useA) break;
        if (st == 2 && !useM) break;
        const double slope = (dn * sxy[st] - sx * sy[st]) / (dn * sxx - sx * sx);
        const double icpt = (sy[st] - slope * sx) / dn;
        const int K = st == 0 ? K0 : Km;
        a.l0[(size_t)(st * 3 + 0) * ld + s] = sk[st] / (double)K;
        a.b0[(size_t)(st * 3 + 0) * ld + s] = 0.0;
        {
            double l0 = icpt, b0 = slope;
            if (fabs(l0 + b0) < 1.0e-8) { l0 = l0 * (1.0 + 1.0e-3); b0 = b0 * (1.0 - 1.0e-3); }
            a.l0[(size_t)(st * 3 + 1) * ld + s] = l0;
            a.b0[(size_t)(st * 3 + 1) * ld + s] = b0;
        }
        {
            double l0 = icpt + slope;
            if (fabs(l0) < 1.0e-8) l0 = 1.0e-7;
            double b0 = (icpt + 2.0 * slope) / l0;
            l0 = l0 / b0;
            if (fabs(b0) > 1.0e10) b0 = (b0 < 0.0 ? -1.0e10 : 1.0e10);
            if (l0 < 1.0e-8 || b0 < 1.0e-8) {
                l0 = ysa0[st] > 1.0e-3 ? ysa0[st] : 1.0e-3;
                double r = ysa1[st] / ysa0[st];
                b0 = r > 1.0e-3 ? r : 1.0e-3;
            }
            a.l0[(size_t)(st * 3 + 2) * ld + s] = l0;
            a.b0[(size_t)(st * 3 + 2) * ld + s] = b0;
        }
    }
}

void launch_prep(const PrepArgs &a, hipStream_t stream)
{
    const int grid = (a.n_series + NM_BLOCK - 1) / NM_BLOCK;
    const bool states = a.l0 != nullptr;
    if (states && a.m == 7 && !a.m_col) {                 // the M5 / weekly period: ring and accumulators in registers
        hipLaunchKernelGGL((prep_kernel<7>), dim3(grid), dim3(NM_BLOCK), 0, stream, a);
        return;
    }
    PrepArgs b = a;
    if (states && a.m >= 2 && a.m <= ETS_MAX_PERIOD) {
        // any other period: figures and per-phase sums from one workgroup per series, then the sweep
        launch_season_figures(b, stream);
        b.pre_fig = 1;
    }
    hipLaunchKernelGGL((prep_kernel<0>), dim3(grid), dim3(NM_BLOCK), 0, stream, b);
}

} // namespace anofox
