// kernels.hip -- the non-templated kernels of the batch path: series statistics + ETS initial
// states (prep), AICc selection, the classic SES/Holt/Holt-Winters/SeasonalES fits, the closed-form
// baselines and the prediction intervals.  All one-wave workgroups, lane <-> series, coalesced
// 512-byte rows of the time-major block per time step.
#include "classic_device.hpp"
#include "fit_units.hpp"
#include "kernels.hpp"

namespace anofox {

// ------------------------------------------------------------------------------------------
// AICc selection over the fitted spec slots (first minimum wins, oracle_auto_ets_search).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NM_BLOCK) void select_kernel(const SelectArgs a)
{
    const int s = blockIdx.x * NM_BLOCK + threadIdx.x;
    if (s >= a.n_series) return;
    if (a.len[s] <= 0) { a.fallback_mask[s] = 0u; return; }
    int best = -1;
    double best_aicc = __builtin_huge_val();
    int passes = 0, evals = 0;
    for (int k = 0; k < a.n_slots; k++) {
        passes += a.passes_slots[(size_t)k * a.ld + s];
        evals += a.evals_slots[(size_t)k * a.ld + s];
        if (a.status_slots[(size_t)k * a.ld + s] != FIT_OK) continue;
        double v = a.aicc[(size_t)k * a.ld + s];
        if (v < best_aicc) { best_aicc = v; best = k; }
    }
    a.passes_total[s] = passes;
    a.evals_total[s] = evals;
    if (best < 0) {
        a.status[s] = -1;
        a.model_code[s] = 0;
        a.fallback_mask[s] = 1u;
        return;
    }
    a.fallback_mask[s] = 0u;
    a.status[s] = 0;
    a.model_code[s] = 100 + a.slot_spec[best];
    const double *src = a.yhat_slots + ((size_t)best * a.n_series + s) * a.h;
    double *dst = a.yhat + (size_t)s * a.h;
    for (int i = 0; i < a.h; i++) dst[i] = src[i];
}

void launch_select(const SelectArgs &a, hipStream_t stream)
{
    const int grid = (a.n_series + NM_BLOCK - 1) / NM_BLOCK;
    hipLaunchKernelGGL(select_kernel, dim3(grid), dim3(NM_BLOCK), 0, stream, a);
}

// ------------------------------------------------------------------------------------------
// classic fits
// ------------------------------------------------------------------------------------------
// GRING: the seasonal ring (NM_K candidates x m phases x 64 lanes) lives in an HBM scratch area of the workgroup (long periods)
template <int KIND, bool GRING = false>
__global__ __launch_bounds__(NM_BLOCK) void classic_kernel(const ClassicArgs a)
{
    extern __shared__ double lds[];
    constexpr int DIM = ClassicDim<KIND>::value;
    const int s = blockIdx.x * NM_BLOCK + threadIdx.x;
    const int len = (s < a.n_series) ? a.len[s] : 0;
    bool selected = (s < a.n_series) && len > 0 && (a.mask == nullptr || a.mask[s] == a.want);
    const int m = a.m_col ? a.m_col[(size_t)blockIdx.x * NM_BLOCK] : a.m;      // merged batch: the workgroup's own period
    const int min_len = !a.m_col ? a.min_len : (KIND == CK_HW ? 2 * m : (KIND == CK_SEASONAL_ES ? m : a.min_len));
    int st = FIT_OK;
    if (selected && len < min_len) st = FIT_SHORT;
    const bool active = selected && st == FIT_OK;

    SeriesView v;
    v.y = a.y + (s < a.n_series ? s : 0);
    v.ld = a.ld;
    v.len = active ? len : 0;
    v.wave_len = wave_max_i32(v.len);
    v.wave_min_len = 0;
    if (v.wave_len == 0) {
        if (selected) a.status[s] = st;
        return;
    }
    ClassicModel<KIND> mdl;
    mdl.v = v;
    mdl.m = m;
    mdl.ring = GRING ? a.ring_scratch + (size_t)blockIdx.x * (size_t)NM_K * (size_t)(a.m > 0 ? a.m : 1) * NM_BLOCK : lds + nm_lds_doubles<DIM>();

    double xbest[DIM], fbest;
    NmStats ns = {0, 0, 0};
    const bool optimise = (KIND == CK_HOLT || KIND == CK_HW) ? true : (a.optimized != 0);
    if (optimise) nm_minimize(mdl, active, lds, xbest, fbest, ns);
    else xbest[0] = a.fixed_alpha;

    double cand[NM_K][DIM], f[NM_K];
#pragma unroll
    for (int k = 0; k < NM_K; k++)
#pragma unroll
        for (int i = 0; i < DIM; i++) cand[k][i] = xbest[i];
    ClassicFinalOut fin;
    fin.h = a.h;
    fin.write = active;
    fin.yhat = a.yhat + (size_t)(s < a.n_series ? s : 0) * a.h;
    classic_pass<KIND, NM_K, true>(v, m, cand, f, mdl.ring, &fin);
    if (selected) {
        a.status[s] = st;
        if (active) {
            a.passes[s] += ns.passes + 1;
            if (a.model_code_out) a.model_code_out[s] = a.model_code;
        }
    }
}

// the classic kernels hold NM_K = 4 candidate rings per lane: 2 KB of LDS per phase and wave
constexpr int CLASSIC_LDS_PERIOD = 48;

void launch_classic(int kind, const ClassicArgs &a, hipStream_t stream)
{
    const int grid = (a.n_series + NM_BLOCK - 1) / NM_BLOCK;
    const int mm = a.m > 0 ? a.m : 1;
    size_t ring = sizeof(double) * (size_t)NM_K * mm * NM_BLOCK;
    switch (kind) {
    case CK_SES:
        hipLaunchKernelGGL(classic_kernel<CK_SES>, dim3(grid), dim3(NM_BLOCK), sizeof(double) * nm_lds_doubles<1>(), stream, a);
        break;
    case CK_HOLT:
        hipLaunchKernelGGL(classic_kernel<CK_HOLT>, dim3(grid), dim3(NM_BLOCK), sizeof(double) * nm_lds_doubles<2>(), stream, a);
        break;
    case CK_HW: {
        if (mm > CLASSIC_LDS_PERIOD) {
            if (!a.ring_scratch) throw std::runtime_error("Holt-Winters: a period above the LDS limit needs the ring scratch area");
            hipLaunchKernelGGL((classic_kernel<CK_HW, true>), dim3(grid), dim3(NM_BLOCK), sizeof(double) * nm_lds_doubles<3>(), stream, a);
            break;
        }
        size_t bytes = sizeof(double) * nm_lds_doubles<3>() + ring;
        if (bytes > 48 * 1024)
            anofox_check_attr(hipFuncSetAttribute((const void *)classic_kernel<CK_HW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL(classic_kernel<CK_HW>, dim3(grid), dim3(NM_BLOCK), bytes, stream, a);
        break;
    }
    default: {
        if (mm > CLASSIC_LDS_PERIOD) {
            if (!a.ring_scratch) throw std::runtime_error("SeasonalES: a period above the LDS limit needs the ring scratch area");
            hipLaunchKernelGGL((classic_kernel<CK_SEASONAL_ES, true>), dim3(grid), dim3(NM_BLOCK), sizeof(double) * nm_lds_doubles<1>(), stream, a);
            break;
        }
        size_t bytes = sizeof(double) * nm_lds_doubles<1>() + ring;
        if (bytes > 48 * 1024)
            anofox_check_attr(hipFuncSetAttribute((const void *)classic_kernel<CK_SEASONAL_ES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL(classic_kernel<CK_SEASONAL_ES>, dim3(grid), dim3(NM_BLOCK), bytes, stream, a);
        break;
    }
    }
}

// ------------------------------------------------------------------------------------------
// closed-form baselines (forecast.rs:1026-1100, toy ARIMA :1391-1431)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NM_BLOCK) void simple_kernel(const SimpleArgs a)
{
    const int s = blockIdx.x * NM_BLOCK + threadIdx.x;
    if (s >= a.n_series) return;
    const int n = a.len[s];
    if (n <= 0) return;
    const double *y = a.y + s;
    const size_t ld = a.ld;
    double *out = a.yhat + (size_t)s * a.h;
    const double last = y[(size_t)(n - 1) * ld];
    switch (a.kind) {
    case SK_NAIVE:
        for (int i = 0; i < a.h; i++) out[i] = last;
        break;
    case SK_SEASONAL_NAIVE: {
        int p = a.period < 1 ? 1 : a.period;
        if (p > n) p = n;
        for (int i = 0; i < a.h; i++) out[i] = y[(size_t)(n - p + (i % p)) * ld];
        break;
    }
    case SK_SMA: {
        int w = a.window < n ? a.window : n;
        double sum = 0.0;
        for (int k = 0; k < w; k++) sum += y[(size_t)(n - 1 - k) * ld];
        double v = sum / (double)w;
        for (int i = 0; i < a.h; i++) out[i] = v;
        break;
    }
    case SK_DRIFT: {
        double drift = (last - y[0]) / (double)(n - 1);
        for (int i = 1; i <= a.h; i++) out[i - 1] = last + drift * (double)i;
        break;
    }
    default: { // SK_TOY_ARIMA
        if (n < 5) { for (int i = 0; i < a.h; i++) out[i] = last; break; }
        double sd = 0.0;
        for (int i = 1; i < n; i++) sd += y[(size_t)i * ld] - y[(size_t)(i - 1) * ld];
        double mean_diff = sd / (double)(n - 1);
        double prev = last - y[(size_t)(n - 2) * ld], cum = last;
        for (int i = 0; i < a.h; i++) {
            double nd = mean_diff + 0.5 * (prev - mean_diff);
            cum += nd;
            out[i] = cum;
            prev = nd;
        }
        break;
    }
    }
    a.status[s] = 0;
}

void launch_simple(const SimpleArgs &a, hipStream_t stream)
{
    const int grid = (a.n_series + NM_BLOCK - 1) / NM_BLOCK;
    hipLaunchKernelGGL(simple_kernel, dim3(grid), dim3(NM_BLOCK), 0, stream, a);
}

// ------------------------------------------------------------------------------------------
// intervals: yhat -/+ z sd sqrt(step)   (forecast.rs:2558-2591)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interval_kernel(const IntervalArgs a)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)a.n_series * a.h;
    if (idx >= total) return;
    const int s = (int)(idx / a.h), i = (int)(idx % a.h);
    if (a.status[s] != 0) { a.lower[idx] = __builtin_nan(""); a.upper[idx] = __builtin_nan(""); return; }
    const double wd = a.z * a.sd[s] * sqrt((double)(i + 1));
    const double f = a.yhat[idx];
    a.lower[idx] = f - wd;
    a.upper[idx] = f + wd;
}

void launch_intervals(const IntervalArgs &a, hipStream_t stream)
{
    const size_t total = (size_t)a.n_series * a.h;
    if (total == 0) return;
    hipLaunchKernelGGL(interval_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
}

// ------------------------------------------------------------------------------------------
FitLaunchers ets_fit_launcher(int spec_id, int m, int yt)
{
    FitLaunchers f = fit_unit_nonseasonal(spec_id, m, yt);
    if (!f.final) f = fit_unit_seasonal_add(spec_id, m, yt);
    if (!f.final) f = fit_unit_seasonal_gen_a(spec_id, m, yt);
    if (!f.final) f = fit_unit_seasonal_gen_m(spec_id, m, yt);
    return f;
}

// ------------------------------------------------------------------------------------------
// compaction of the running problems (ballot per wave + one atomic; order of survivors not preserved)
// ------------------------------------------------------------------------------------------
// One wave per 64 candidates: ballot, one atomic per wave to reserve a slot range, ordered within the
// wave.  The order of the waves' ranges is not deterministic -- it only decides which column a problem
// occupies in the next round, never a result (every problem is independent of its position).
__global__ __launch_bounds__(NM_BLOCK) void compact_kernel(const int32_t *series_prev, const int32_t *n_prev_ptr, int n_series,
                                                           const int32_t *done, int32_t *series_next, int32_t *n_next_ptr, int32_t *n_clear)
{
    // the counter of the round after next is cleared here (three counters rotate), so no memset sits between the rounds
    if (n_clear && blockIdx.x == 0 && threadIdx.x == 0) *n_clear = 0;
    const int n_prev = n_prev_ptr ? *n_prev_ptr : n_series;
    if ((int)blockIdx.x * NM_BLOCK >= n_prev) return;
    const int lane = threadIdx.x;
    const int idx = blockIdx.x * NM_BLOCK + lane;
    int s = 0;
    bool keep = false;
    if (idx < n_prev) {
        s = series_prev ? series_prev[idx] : idx;
        keep = done[s] == 0;
    }
    const unsigned long long bal = __ballot(keep);
    const int cnt = __popcll(bal);
    if (cnt == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(n_next_ptr, cnt);
    base = __shfl(base, 0);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (keep) series_next[base + before] = s;
}

void launch_compact(const int32_t *series_prev, const int32_t *n_prev, int n_series, const int32_t *done,
                    int32_t *series_next, int32_t *n_next, hipStream_t stream, int32_t *n_clear)
{
    if (!n_clear) anofox_check_attr(hipMemsetAsync(n_next, 0, sizeof(int32_t), stream));   // stand-alone use: clear the output counter here
    const int grid = (n_series + NM_BLOCK - 1) / NM_BLOCK;
    hipLaunchKernelGGL(compact_kernel, dim3(grid), dim3(NM_BLOCK), 0, stream, series_prev, n_prev, n_series, done, series_next, n_next, n_clear);
}

// ------------------------------------------------------------------------------------------
// column gather: rebuild a dense time-major block of the running problems so that the next round
// still reads 512 contiguous bytes per wave and time step (reads here are the only uncoalesced ones)
// ------------------------------------------------------------------------------------------
constexpr int GATHER_TB = 32;
template <class E>
__global__ __launch_bounds__(NM_BLOCK) void gather_columns_kernel(const E *y, size_t ld, const int32_t *series_of,
                                                                  const int32_t *n_active, int t_max, E *out, size_t ld_out, int cap)
{
    const int n_act = *n_active;
    if ((int)blockIdx.x * NM_BLOCK >= n_act || n_act > cap) return;       // (more running problems than `out` holds: the round kernel indexes y by series)
    const int p = blockIdx.x * NM_BLOCK + threadIdx.x;
    const int s = series_of[p < n_act ? p : n_act - 1];
    const int t0 = blockIdx.y * GATHER_TB;
    const int t1 = t0 + GATHER_TB < t_max ? t0 + GATHER_TB : t_max;
    for (int t = t0; t < t1; t++) out[(size_t)t * ld_out + p] = y[(size_t)t * ld + s];
}

void launch_gather_columns(const void *y, size_t ld, const int32_t *series_of, const int32_t *n_active, int n_series,
                           int t_max, void *out, size_t ld_out, hipStream_t stream, int cap, int elem_bytes)
{
    if (n_series <= 0 || t_max <= 0) return;          // a batch of empty series: nothing to copy (and no zero-sized grid)
    const int n_cols = std::min(n_series, cap);
    if (n_cols <= 0) return;
    dim3 grid((n_cols + NM_BLOCK - 1) / NM_BLOCK, (t_max + GATHER_TB - 1) / GATHER_TB);
    if (elem_bytes == 8)
        hipLaunchKernelGGL(gather_columns_kernel<double>, grid, dim3(NM_BLOCK), 0, stream, (const double *)y, ld, series_of, n_active, t_max, (double *)out, ld_out, cap);
    else if (elem_bytes == 4)
        hipLaunchKernelGGL(gather_columns_kernel<float>, grid, dim3(NM_BLOCK), 0, stream, (const float *)y, ld, series_of, n_active, t_max, (float *)out, ld_out, cap);
    else
        hipLaunchKernelGGL(gather_columns_kernel<unsigned short>, grid, dim3(NM_BLOCK), 0, stream, (const unsigned short *)y, ld, series_of, n_active, t_max, (unsigned short *)out, ld_out, cap);
}

// ------------------------------------------------------------------------------------------
// compact copies of the block (kernels.hpp launch_compact_block): one streamed sweep, a thread per column and 32 rows; the cells
// beyond a series' length are copied too (never read as observations) but do not count as misfits
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NM_BLOCK) void compact_block_kernel(const double *y, size_t ld, const int32_t *len, int n_series, int t_rows, float *out32,
                                                                 unsigned short *out16, unsigned int *misfit)
{
    const int s = blockIdx.x * NM_BLOCK + threadIdx.x;
    const int t0 = blockIdx.y * GATHER_TB;
    const int t1 = t0 + GATHER_TB < t_rows ? t0 + GATHER_TB : t_rows;
    const int L = s < n_series ? len[s] : 0;
    unsigned bad32 = 0, bad16 = 0;
    if ((size_t)s < ld) {
        for (int t = t0; t < t1; t++) {
            const double v = y[(size_t)t * ld + s];
            const float f = (float)v;
            // (a value outside 0 .. 65,535, or not a number, converts to something that does not compare equal: counted below)
            const unsigned short u = (v >= 0.0 && v <= 65535.0) ? (unsigned short)v : (unsigned short)0;
            out32[(size_t)t * ld + s] = f;
            if (out16) out16[(size_t)t * ld + s] = u;
            if (t < L) {
                // bit patterns, not values: -0.0 widens back from float as -0.0 but from an integer as +0.0
                bad32 += (__double_as_longlong((double)f) == __double_as_longlong(v)) ? 0u : 1u;
                bad16 += (__double_as_longlong((double)u) == __double_as_longlong(v)) ? 0u : 1u;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { bad32 += __shfl_xor(bad32, o); bad16 += __shfl_xor(bad16, o); }
    if (threadIdx.x == 0) {
        if (bad32) atomicAdd(misfit, bad32);
        if (bad16) atomicAdd(misfit + 1, bad16);
    }
}

void launch_compact_block(const double *y, size_t ld, const int32_t *len, int n_series, int t_rows, float *out32, unsigned short *out16,
                          unsigned int *misfit, hipStream_t stream)
{
    if (ld == 0 || t_rows <= 0) return;
    dim3 grid((unsigned)((ld + NM_BLOCK - 1) / NM_BLOCK), (unsigned)((t_rows + GATHER_TB - 1) / GATHER_TB));
    hipLaunchKernelGGL(compact_block_kernel, grid, dim3(NM_BLOCK), 0, stream, y, ld, len, n_series, t_rows, out32, out16, misfit);
}

// ------------------------------------------------------------------------------------------
// seasonal period detection (seasonality.rs:323-377 detect_seasonality_first; oracle/forecast.c oracle_detect_seasonality_first):
// the lag of the strongest local maximum above 0.1 of the autocorrelation over lags 1 .. n / 2, or 0.  O(n^2 / 2) multiply-adds
// per series -- 1.8 M for an M5 series, 0.6 s of 32 host threads for the 30,490 of them -- so it runs here: one workgroup per
// series, the centred series in LDS (a scratch block in HBM above DETECT_LDS_ROWS observations), one lag per lane (the lanes of a
// wave read consecutive LDS words, the other factor is a broadcast), every sum in the order the scalar loop takes (separate
// multiply and add, one accumulator), so the chosen lag is the oracle's bit for bit.  The mean, the variance and the final scan
// are sequential sums / a sequential scan on one lane (2 n + n / 2 dependent steps against n^2 / 512 per lane for the lags).
// ------------------------------------------------------------------------------------------
constexpr int DETECT_THREADS = 256;
template <bool USE_LDS>
__global__ __launch_bounds__(DETECT_THREADS) void detect_period_kernel(const double *y, size_t ld, const int32_t *len, int n_series, int t_rows,
                                                                       double *scratch, int32_t *period, double *best_acf_out)
{
    extern __shared__ double detect_lds[];
    __shared__ double sh_mean, sh_var;
    const int tid = threadIdx.x;
    double *c = USE_LDS ? detect_lds : scratch + (size_t)blockIdx.x * ((size_t)t_rows + (size_t)t_rows / 2 + 2);
    double *acf = c + t_rows;
    for (int s = blockIdx.x; s < n_series; s += gridDim.x) {
        const int n = len[s];
        int best = 0;
        double best_acf = 0.0;
        const int max_lag = n / 2;
        if (n >= 4 && max_lag >= 2) {
            for (int i = tid; i < n; i += DETECT_THREADS) c[i] = y[(size_t)i * ld + s];
            __syncthreads();
            if (tid == 0) {
                double mean = 0.0;
                for (int i = 0; i < n; i++) mean += c[i];
                mean /= (double)n;
                double var = 0.0;
                for (int i = 0; i < n; i++) { const double d = c[i] - mean; var += d * d; }
                sh_mean = mean; sh_var = var;
            }
            __syncthreads();
            const double mean = sh_mean, var = sh_var;
            if (!(fabs(var) < 2.220446049250313e-16)) {
                for (int i = tid; i < n; i += DETECT_THREADS) c[i] = c[i] - mean;
                __syncthreads();
                for (int lag = 1 + tid; lag <= max_lag; lag += DETECT_THREADS) {
                    const double *q = c + lag;
                    const int cnt = n - lag;
                    double sum = 0.0;
#pragma unroll 8
                    for (int i = 0; i < cnt; i++) sum += c[i] * q[i];
                    acf[lag - 1] = sum / var;
                }
                __syncthreads();
                if (tid == 0) {
                    for (int i = 1; i + 1 < max_lag; i++) {
                        const double v = acf[i];
                        if (v > acf[i - 1] && v > acf[i + 1] && v > 0.1)
                            if (best == 0 || v > best_acf) { best = i + 1; best_acf = v; }
                    }
                }
            }
            __syncthreads();            // the next series overwrites c
        }
        if (tid == 0) {
            period[s] = best;
            if (best_acf_out) best_acf_out[s] = best_acf;
        }
    }
}

size_t detect_scratch_doubles(int n_series, int t_rows)
{
    if (t_rows <= DETECT_LDS_ROWS) return 0;
    return (size_t)std::min(n_series, DETECT_LONG_GRID) * ((size_t)t_rows + (size_t)t_rows / 2 + 2);
}

void launch_detect_periods(const double *y, size_t ld, const int32_t *len, int n_series, int t_rows, double *scratch, int32_t *period,
                           double *best_acf, hipStream_t stream)
{
    if (n_series <= 0) return;
    if (t_rows <= DETECT_LDS_ROWS) {
        const size_t bytes = sizeof(double) * ((size_t)t_rows + (size_t)t_rows / 2 + 2);
        if (bytes > 48 * 1024)
            anofox_check_attr(hipFuncSetAttribute((const void *)detect_period_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL(detect_period_kernel<true>, dim3((unsigned)std::min(n_series, 65536)), dim3(DETECT_THREADS), bytes, stream, y, ld, len, n_series,
                           t_rows, (double *)nullptr, period, best_acf);
    } else {
        hipLaunchKernelGGL(detect_period_kernel<false>, dim3((unsigned)std::min(n_series, DETECT_LONG_GRID)), dim3(DETECT_THREADS), 0, stream, y, ld, len,
                           n_series, t_rows, scratch, period, best_acf);
    }
}

// ---- self test of dm_recip (det_math.hpp): the short division sequence against the compiled IEEE division -------------------------
// operand i of `n`: a 64-bit mix of (seed, i) as the significand and sign, the exponent swept over the whole admissible domain
// [2^-1000, 2^1000]; counts the operands where the two quotients differ in any bit
__global__ __launch_bounds__(256) void recip_selftest_kernel(unsigned long long n, unsigned long long seed, unsigned long long *mismatches, double *first_bad)
{
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long z = seed + 0x9e3779b97f4a7c15ull * (i + 1);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z ^= z >> 31;
        const long long e = (long long)((z >> 52) % 2001ull) - 1000;            // exponent in [-1000, 1000]
        unsigned long long bits = (z & 0x800fffffffffffffull) | ((unsigned long long)(e + 1023) << 52);
        if (e == 1000) bits &= 0xfff0000000000000ull;                           // |d| <= 2^1000 exactly
        if ((i & 1023) == 0) bits |= 0x000fffffffffffffull * ((i >> 10) & 1);   // all-ones / all-zeros significands among them
        if (e == 1000) bits &= 0xfff0000000000000ull;
        const double d = dm_from_bits(bits);
        if (!dm_recip_ok(d)) continue;
        const double want = 1.0 / d, got = dm_recip(d);
        if (dm_bits(want) != dm_bits(got)) { if (bad == 0 && atomicAdd(mismatches + 1, 1ull) == 0) *first_bad = d; bad++; }
    }
    if (bad) atomicAdd(mismatches, bad);
}

unsigned long long recip_selftest(unsigned long long n, unsigned long long seed, double *first_bad_out, hipStream_t stream)
{
    unsigned long long *d_cnt = nullptr; double *d_bad = nullptr;
    if (hipMalloc(&d_cnt, 2 * sizeof(unsigned long long)) != hipSuccess || hipMalloc(&d_bad, sizeof(double)) != hipSuccess) return ~0ull;
    (void)hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), stream);
    (void)hipMemsetAsync(d_bad, 0, sizeof(double), stream);
    hipLaunchKernelGGL(recip_selftest_kernel, dim3(4096), dim3(256), 0, stream, n, seed, d_cnt, d_bad);
    unsigned long long h[2] = {~0ull, 0};
    double hb = 0.0;
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(h, d_cnt, sizeof h, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&hb, d_bad, sizeof hb, hipMemcpyDeviceToHost) != hipSuccess) h[0] = ~0ull;
    (void)hipFree(d_cnt); (void)hipFree(d_bad);
    if (first_bad_out) *first_bad_out = hb;
    return h[0];
}

} // namespace anofox
