// prep.hip -- per-series statistics and ETS initial states in TWO streamed passes over the time-major block.
//
//   pass A: sum (-> mean), positivity / constancy flags, and the classical decomposition: a sliding window of
//           the last L = 2*(m/2)+1 observations (LDS ring) gives the centred moving average of every interior
//           point; the detrended value (y - trend, y / trend) is added to its phase's accumulator (LDS).
//   pass B: sum of squared deviations (-> sd, forecast.rs:2558-2591) and, for the three season types at once,
//           the sums of the full-sample least-squares line of the seasonally adjusted series and of its first
//           K values (level-only start).
// Same operations in the same order as oracle/ets.c (ets_init_states) and oracle/forecast.c (intervals), so the
// results are bit-identical; only the number of times y is streamed changed (2 instead of ~22).
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
// block length of the streamed rows: 16, or four revolutions of a compile-time period MR (ring slots and phases of a block are
// compile-time constants then).  One wave per 64 series is all the parallelism this kernel has (477 waves for the M5 block: every
// sum is sequential in time, and the mean / sd are the reference's own in-tree arithmetic, forecast.rs:2558-2591, so the time axis
// is not split), which makes it a question of bytes in flight: with 14 rows (7 KB) per wave in flight the two passes ran at
// 2.0-2.7 TB/s (249 + 182 us per 487 MB, rocprofv3), whatever the arithmetic per step -- dealing the accumulators of a step to
// three waves changed nothing (431 us against 454: three waves read every row three times).  28 rows per block, two blocks
// alternating without a copy, is what the one-pass ets_final_kernel streams the same block with at 5.5 TB/s.
template <int MR> constexpr int prep_block() { return MR > 0 ? 4 * MR : 16; }

// Long periods (above ETS_LDS_PERIOD): the classical decomposition costs T * (m + 1) multiply-adds per series (every centred
// moving average is its own sequential sum, as the oracle writes it) -- 1.4 M for T = 1,913, m = 755 -- and with one wave per 64
// series, window ring in an HBM scratch, prep_kernel spent 206 ms on the 36,000 columns of a merged batch before its first fit
// could start.  season_figures_kernel gives a series a WORKGROUP instead: the series in LDS, one centre per thread (its window
// summed in the oracle's order: acc = acc + w_k y_k, k ascending), then one phase per thread (the detrended values of a phase added
// in time order), then the normalisation -- every sum keeps the order of oracle/ets.c ets_init_states, so the figures are the same
// bits.  prep_kernel then runs with pre_fig = 1: pass A without the window, pass B reading the figures from fig_add / fig_mul.
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
        const bool positive = __syncthreads_and(pos) != 0;
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
        for (int p = tid; p < m; p += SEASON_THREADS) {
            a.fig_add[(size_t)p * a.ld + s] = fa[p] - mean_a;
            if (positive) {
                double fj = fm[p] / mean_m;
                if (!(fj >= 1.0e-2)) fj = 1.0e-2;
                a.fig_mul[(size_t)p * a.ld + s] = fj;
            }
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

// MR > 0: compile-time odd period (7): the window ring and the per-phase accumulators are VGPR arrays with compile-time
// indices (the block is two revolutions long) -- no LDS traffic at all; same operations in the same order as the generic path.
template <int MR = 0>
__global__ __launch_bounds__(NM_BLOCK) void prep_kernel(const PrepArgs a)
{
    constexpr int PREP_S = prep_block<MR>();
    static_assert(MR == 0 || (MR % 2 == 1 && MR >= 3), "register variant: odd periods");
    extern __shared__ double lds_dyn[];
    double *lds = lds_dyn;
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
    double cur[PREP_S], nxt[PREP_S];
    const bool states = a.l0 != nullptr;
    const int m = a.m_col ? a.m_col[(size_t)blockIdx.x * NM_BLOCK] : a.m;      // merged batch: this workgroup's own period (a.m bounds the sizes)
    const bool want_season = states && m >= 2 && m <= ETS_MAX_PERIOD;
    const int half = m / 2, L = 2 * half + 1;
    // LDS (lane-minor): window ring [L], sumA [m], sumM [m], cnt [m]
    double *ring = lds;
    double *sumA = lds + (size_t)(want_season ? L : 0) * NM_BLOCK;
    double *sumM = sumA + (size_t)(want_season ? m : 0) * NM_BLOCK;
    double *cnt = sumM + (size_t)(want_season ? m : 0) * NM_BLOCK;
    constexpr int MRA = MR > 0 ? MR : 1;
    double rR[MRA], sAR[MRA], sMR[MRA], cNR[MRA];
#pragma unroll
    for (int j = 0; j < MRA; j++) { rR[j] = 0.0; sAR[j] = 0.0; sMR[j] = 0.0; cNR[j] = 0.0; }
    if (MR == 0 && want_season && !a.pre_fig)
        for (int j = 0; j < m; j++) { sumA[j * NM_BLOCK + lane] = 0.0; sumM[j * NM_BLOCK + lane] = 0.0; cnt[j * NM_BLOCK + lane] = 0.0; }
    const double w = want_season ? 1.0 / (double)m : 0.0;
    const double wend = (want_season && m % 2 == 0) ? 0.5 / (double)m : w;
    const bool seasonal = want_season && n >= 2 * m;
    const bool pre_fig = a.pre_fig != 0;                        // figures already in fig_add / fig_mul (season_figures_kernel)
    const bool decompose = seasonal && !pre_fig;

    // ---- pass A ----------------------------------------------------------------------------------------
    double sum = 0.0;
    bool positive = true, constant = true, has_nan = false;
    rows.load(cur, 0);
    const double y0 = n > 0 ? cur[0] : 0.0;
    int slot = 0;                       // t % L
    int ph = 0;                         // (t - half) % m, phase of the window centre
    if (want_season) ph = ((-half) % m + m) % m;
    auto pass_a_block = [&](const double (&buf)[PREP_S], const int base) __attribute__((always_inline)) {
      // the multiplicative figure (one IEEE division per step) only matters for strictly positive series: once every series of
      // the wave has shown a value <= 0 (intermittent demand: within the first blocks) the wave stops computing it
      const bool wave_pos = __any(positive && base < n);
#pragma unroll
      for (int jj = 0; jj < PREP_S; jj++) {
        const int t = base + jj;
        if (t < n) {
            const double v = buf[jj];
            sum += v;
            if (!(v > 0.0)) positive = false;
            if (v != y0) constant = false;
            if (v != v) has_nan = true;
            if constexpr (MR > 0) {
                if (seasonal) {
                    constexpr int HALF = MR / 2;
                    const int sl = jj % MR;                                  // compile-time after unrolling: base is a multiple of MR
#pragma unroll
                    for (int q = 0; q < MR; q++) rR[q] = (q == sl) ? v : rR[q];
                    if (t >= MR - 1) {
                        double acc = 0.0;
#pragma unroll
                        for (int k = 0; k < MR; k++) acc = acc + w * rR[(sl + 1 + k) % MR];     // odd period: every weight is 1 / m
                        const double yc = rR[(sl + 1 + HALF) % MR];
                        const int phc = ((jj - HALF) % MR + MR) % MR;
#pragma unroll
                        for (int q = 0; q < MR; q++) {
                            sAR[q] = (q == phc) ? sAR[q] + (yc - acc) : sAR[q];
                            cNR[q] = (q == phc) ? cNR[q] + 1.0 : cNR[q];
                        }
                        if (wave_pos) {
                            const double ratio = yc / acc;
#pragma unroll
                            for (int q = 0; q < MR; q++) sMR[q] = (q == phc) ? sMR[q] + ratio : sMR[q];
                        }
                    }
                }
            } else
            if (decompose) {
                ring[slot * NM_BLOCK + lane] = v;
                if (t >= L - 1) {
                    double acc = 0.0;
                    int k0 = slot + 1 == L ? 0 : slot + 1;           // slot of t - L + 1
                    int kc = 0;
                    double yc = 0.0;
                    for (int k = 0; k < L; k++) {
                        const double wk = (k == 0 || k == L - 1) ? wend : w;
                        const double yv = ring[k0 * NM_BLOCK + lane];
                        acc = acc + wk * yv;
                        if (kc == half) yc = yv;
                        kc++;
                        k0 = k0 + 1 == L ? 0 : k0 + 1;
                    }
                    sumA[ph * NM_BLOCK + lane] = sumA[ph * NM_BLOCK + lane] + (yc - acc);
                    if (wave_pos) sumM[ph * NM_BLOCK + lane] = sumM[ph * NM_BLOCK + lane] + (yc / acc);
                    cnt[ph * NM_BLOCK + lane] += 1.0;
                }
            }
        }
        slot = slot + 1 == L ? 0 : slot + 1;
        if (want_season) ph = ph + 1 == m ? 0 : ph + 1;
      }
    };
    // Full blocks of the register variant run as STRAIGHT-LINE code: once every series of the wave is inside its sample for the whole
    // block (and past the first one, so the window is full) no step needs a per-lane branch -- flags become mask operations, the
    // per-phase updates have compile-time targets -- and the 28 steps of a block form ONE basic block.  The per-lane `if (t < n)` of
    // the gated form made every step its own basic block, so the seven dependent additions of a step's window mean (same order as
    // the oracle: the sum is sequential by definition) could not overlap with the neighbouring steps' chains: a wave alone on its
    // SIMD then waits out every fp64 latency -- 283 cycles per step measured (453 us for the two passes of the M5 block, whatever
    // the number of rows in flight).  Same operations on the same values: a lane without a series (padding) computes into
    // registers nobody reads.
    const int wave_min_n = __builtin_amdgcn_readfirstlane(wave_min_i32(n > 0 ? n : 0x7fffffff));
    auto pass_a_fast = [&](const double (&buf)[PREP_S], auto pos_tag) __attribute__((always_inline)) {
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
                const int sl = jj % MR;
                rR[sl] = v;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < MR; k++) acc = acc + w * rR[(sl + 1 + k) % MR];
                const double yc = rR[(sl + 1 + HALF) % MR];
                const int phc = ((jj - HALF) % MR + MR) % MR;
                sAR[phc] = sAR[phc] + (yc - acc);
                cNR[phc] = cNR[phc] + 1.0;
                if constexpr (POS) sMR[phc] = sMR[phc] + yc / acc;
            }
        }
    };
    auto pass_a_any = [&](const double (&buf)[PREP_S], const int base) __attribute__((always_inline)) {
        if (MR > 0 && base >= PREP_S && base + PREP_S <= wave_min_n) {
            if (__any(positive && base < n)) pass_a_fast(buf, std::true_type{});
            else pass_a_fast(buf, std::false_type{});
        } else pass_a_block(buf, base);
    };
    // two blocks per iteration on alternating buffers: the rows of the block after next are requested before a block is consumed
    for (int base = 0; base < wave_rows; base += 2 * PREP_S) {
        rows.load(nxt, base + PREP_S);
        pass_a_any(cur, base);
        rows.load(cur, base + 2 * PREP_S);
        if (base + PREP_S < wave_rows) pass_a_any(nxt, base + PREP_S);
    }
    if (n <= 0) return;
    const double mean = sum / (double)n;
    a.mean[s] = mean;
    a.flags[s] = (positive ? SF_POSITIVE : 0u) | (constant ? SF_CONSTANT : 0u) | (has_nan ? SF_HAS_NAN : 0u);

    // seasonal figures (normalised), kept in LDS for pass B and written to HBM for the fit kernels
    if constexpr (MR > 0) {
        if (seasonal) {
#pragma unroll
            for (int type = 1; type <= 2; type++) {
                if (type == 2 && !positive) break;
                double tot = 0.0;
#pragma unroll
                for (int j = 0; j < MR; j++) {
                    const double fj = (type == 1 ? sAR[j] : sMR[j]) / cNR[j];
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
        }
    } else
    if (decompose) {
        for (int type = 1; type <= 2; type++) {
            if (type == 2 && !positive) break;
            double *acc = type == 1 ? sumA : sumM;
            double tot = 0.0;
            for (int j = 0; j < m; j++) {
                const double fj = acc[j * NM_BLOCK + lane] / cnt[j * NM_BLOCK + lane];
                acc[j * NM_BLOCK + lane] = fj;
                tot = tot + fj;
            }
            const double fmean = tot / (double)m;
            double *fig = (type == 1 ? a.fig_add : a.fig_mul) + s;
            for (int j = 0; j < m; j++) {
                double fj = acc[j * NM_BLOCK + lane];
                if (type == 1) fj = fj - fmean;
                else {
                    fj = fj / fmean;
                    if (!(fj >= 1.0e-2)) fj = 1.0e-2;
                }
                acc[j * NM_BLOCK + lane] = fj;
                fig[(size_t)j * ld] = fj;
            }
        }
    }

    // ---- pass B ----------------------------------------------------------------------------------------
    double var = 0.0;
    double sy[3] = {0, 0, 0}, sxy[3] = {0, 0, 0}, sk[3] = {0, 0, 0}, ysa0[3] = {0, 0, 0}, ysa1[3] = {0, 0, 0};
    const bool useA = seasonal, useM = seasonal && positive;
    const bool wave_useM = __any(useM);            // no strictly positive series in the wave: no multiplicative states, no division per step
    int K0 = 10 > n ? n : 10;
    int Km = 2 * m > 10 ? 2 * m : 10;
    if (Km > n) Km = n;
    int j = 0;
    const double *figA = pre_fig ? a.fig_add + (valid ? s : 0) : sumA + lane;
    const double *figM = pre_fig ? a.fig_mul + (valid ? s : 0) : sumM + lane;
    const size_t fig_stride = pre_fig ? ld : (size_t)NM_BLOCK;
    rows.load(cur, 0);
    auto pass_b_block = [&](const double (&buf)[PREP_S], const int base) __attribute__((always_inline)) {
#pragma unroll
      for (int jj = 0; jj < PREP_S; jj++) {
        const int t = base + jj;
        if (t >= n) continue;
        const double v = buf[jj];
        const double dv = v - mean;
        var += dv * dv;
        if (states) {
            double vs[3];
            vs[0] = v;
            if constexpr (MR > 0) {
                vs[1] = useA ? v - sAR[jj % MR] : 0.0;
                vs[2] = 0.0;
                if (wave_useM) vs[2] = useM ? v / sMR[jj % MR] : 0.0;
            } else {
                vs[1] = useA ? v - figA[(size_t)j * fig_stride] : 0.0;
                vs[2] = 0.0;
                if (wave_useM) vs[2] = useM ? v / figM[(size_t)j * fig_stride] : 0.0;
            }
#pragma unroll
            for (int st = 0; st < 3; st++) {
                sy[st] = sy[st] + vs[st];
                sxy[st] = sxy[st] + (double)(t + 1) * vs[st];
                if (t < (st == 0 ? K0 : Km)) sk[st] = sk[st] + vs[st];
                if (t == 0) ysa0[st] = vs[st];
                if (t == 1) ysa1[st] = vs[st];
            }
            if (want_season) j = j + 1 == m ? 0 : j + 1;
        }
      }
    };
    // ... and the same for pass B: past the first block no step takes part in the level-only start (t < K <= 2 m) or is t = 0, 1
    auto pass_b_fast = [&](const double (&buf)[PREP_S], const int base, auto m_tag) __attribute__((always_inline)) {
        constexpr bool USEM = decltype(m_tag)::value;
        if constexpr (MR > 0) {
#pragma unroll
            for (int jj = 0; jj < PREP_S; jj++) {
                const double v = buf[jj];
                const double dv = v - mean;
                var += dv * dv;
                const double tt = (double)(base + jj + 1);
                double vs[3];
                vs[0] = v;
                vs[1] = v - sAR[jj % MR];
                vs[2] = 0.0;
                if constexpr (USEM) vs[2] = useM ? v / sMR[jj % MR] : 0.0;
#pragma unroll
                for (int st = 0; st < (USEM ? 3 : 2); st++) {
                    sy[st] = sy[st] + vs[st];
                    sxy[st] = sxy[st] + tt * vs[st];
                }
            }
        }
    };
    auto pass_b_any = [&](const double (&buf)[PREP_S], const int base) __attribute__((always_inline)) {
        // (every series of such a block is seasonal: n >= base + PREP_S >= 8 m)
        if (MR > 0 && states && base >= PREP_S && base + PREP_S <= wave_min_n && __all(n <= 0 || useA)) {
            if (wave_useM) pass_b_fast(buf, base, std::true_type{});
            else pass_b_fast(buf, base, std::false_type{});
        } else pass_b_block(buf, base);
    };
    for (int base = 0; base < wave_rows; base += 2 * PREP_S) {
        rows.load(nxt, base + PREP_S);
        pass_b_any(cur, base);
        rows.load(cur, base + 2 * PREP_S);
        if (base + PREP_S < wave_rows) pass_b_any(nxt, base + PREP_S);
    }
    a.sd[s] = sqrt(var / (double)n);
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
    size_t lds_bytes = 0;
    if (a.l0 != nullptr && a.m > ETS_LDS_PERIOD && a.m <= ETS_MAX_PERIOD) {
        // long periods: the figures from one workgroup per series, then the two streamed passes without the decomposition
        PrepArgs b = a;
        launch_season_figures(b, stream);
        b.pre_fig = 1;
        hipLaunchKernelGGL((prep_kernel<0>), dim3(grid), dim3(NM_BLOCK), 0, stream, b);
        return;
    }
    if (a.l0 != nullptr && a.m == 7 && !a.m_col) {        // the M5 / weekly period: ring and accumulators in registers
        hipLaunchKernelGGL((prep_kernel<7>), dim3(grid), dim3(NM_BLOCK), 0, stream, a);
        return;
    }
    if (a.l0 != nullptr && a.m >= 2 && a.m <= ETS_LDS_PERIOD) lds_bytes = sizeof(double) * (size_t)((2 * (a.m / 2) + 1) + 3 * a.m) * NM_BLOCK;
    if (lds_bytes > 48 * 1024) anofox_check_attr(hipFuncSetAttribute((const void *)prep_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL((prep_kernel<0>), dim3(grid), dim3(NM_BLOCK), lds_bytes, stream, a);
}

} // namespace anofox
