// kernels.hpp -- kernel argument blocks and launch entry points shared by the .hip units
// and the host layer.  Everything device-side is one-wave (64-thread) workgroups: lane <->
// series, so a workgroup covers 64 consecutive columns of the time-major block.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <atomic>

#include <stdexcept>
#include <string>

namespace anofox {

// results of the runtime calls made while preparing a launch (raising a kernel's LDS limit, clearing a counter) are
// never discarded: a refused request would otherwise surface as a silent launch failure
inline void anofox_check_attr(hipError_t e)
{
    if (e != hipSuccess) throw std::runtime_error(std::string("launch preparation failed: ") + hipGetErrorString(e));
}

// per-series flag bits produced by the prep kernel
enum : uint32_t { SF_POSITIVE = 1u, SF_CONSTANT = 2u, SF_HAS_NAN = 4u };

// per-(series, spec) fit status
enum : int32_t { FIT_OK = 0, FIT_SHORT = 1, FIT_NONPOSITIVE = 2, FIT_NONFINITE = 3, FIT_PERIOD = 4, FIT_SKIPPED = 5 };

struct PrepArgs {
    const double *y; size_t ld; const int32_t *len; int n_series;
    int m;                 // seasonal period to prepare figures for (<= 1: none); with m_col: the LARGEST period of the block (sizes)
    const int32_t *m_col;  // [ld] period of every column, constant within each group of 64 columns (a merged batch of several periods); NULL = m
    double *mean, *sd;     // population mean / sd of the series (forecast.rs:2558-2591)
    uint32_t *flags;
    double *fig_add, *fig_mul;   // [m x ld] seasonal figures (additive / multiplicative)
    double *l0, *b0;       // [9 x ld]: index (season_type * 3 + trend_type)
    double *scratch;       // season_figures scratch for series too long for its LDS (season_scratch_doubles), else unused
    int pre_fig;           // 1: fig_add / fig_mul already hold the figures (launch_season_figures): no decomposition in the sweep
    int t_rows;            // rows of the block (the figures kernel sizes its LDS by it)
    int skip_sd;           // 1: the batch's ONE final pass computes sd (FitArgs::sd_out): prep_kernel is a single sweep then
    int skip_types;        // season types NO candidate spec of the batch has (bit 0 additive, bit 1 multiplicative): their figures and start
                           // states are not computed (the multiplicative figure is an IEEE division per time step)
};

// seasonal figures of every period but 7, one workgroup per series (prep.hip season_figures_kernel); `scratch`: season_scratch_doubles
// doubles when the series and its trend do not fit LDS (else unused)
size_t season_scratch_doubles(int n_series, int t_rows, int m_max);
void launch_season_figures(const PrepArgs &a, hipStream_t stream);

// Nelder-Mead state parked in HBM between rounds, indexed by series (stride ld)
struct NmStateBuf {
    double *sim;                 // [(D+1)*D x ld], vertex k coordinate i at row k*D+i
    double *fs;                  // [(D+1) x ld]
    int32_t *phase, *evals, *iters, *passes, *done;   // [ld]
};

struct FitArgs {
    const double *y; size_t ld; const int32_t *len; int n_series;
    int t_rows;                  // rows held by the blocks y and y_round (streaming loads clamp to the last one)
    // round-specific view: the block this round streams (original or gathered columns), the column ->
    // series map (NULL = identity) and the device-side count of running problems (NULL = n_series)
    const double *y_round; size_t ld_round;
    const int32_t *series_of;
    const int32_t *n_active;
    int budget, first_round;
    const int32_t *m_col;        // [ld] per-column seasonal period (constant within 64 columns) of a merged batch of several periods; NULL = m (then m is the size bound)
    int spec_below;              // device-side driver choice: the speculative kernel runs iff n_active <= spec_below,
                                 // the sequential one iff n_active > spec_below (both are enqueued; -1 = unconditional)
    const uint32_t *mask;        // SES / Holt / Holt-Winters / SeasonalES on the round kernels: run only where mask[s] == want (NULL = all series)
    uint32_t want;
    int min_len;                 // ... and fail shorter series (Holt-Winters and SeasonalES derive theirs from the period)
    int budget_seq;              // passes per round when the device-side choice (round_auto) lands on the sequential driver
    int spec2_below;             // ... and the one-problem-per-wave driver (two iterations per pass) iff n_active <= spec2_below
    int gathered;                // y_round holds the running problems' columns densely (column p), else index by series
    int gather_cap;              // columns y_round has room for: the gather (and this flag) only apply while n_active <= gather_cap
    NmStateBuf st;
    double *ring_scratch;        // periods above ETS_LDS_PERIOD: m * 64 doubles per workgroup of the launch (seasonal ring in HBM)
    size_t ring_scratch_doubles; //   doubles it holds: every launch checks that its workgroups fit (ets_round_launch; round 6 -- a scratch sized for
                                 //   fewer workgroups than a launch has is a stray device write, found as an intermittent memory fault)
    double *nm_scratch;          // PARK kernels (ets_fit_kernel.hpp RoundTraits): nm_lds_doubles<DIM>() doubles per workgroup of the launch,
    size_t nm_scratch_doubles;   //   where the lanes' simplices rest between passes (else they rest in LDS); doubles it holds
    int m, h;
    const double *l0, *b0;       // [ld] for this spec's (season, trend) class
    const double *fig; size_t fig_ld;
    const uint32_t *flags;
    int need_positive;           // spec has a multiplicative component
    int skip_constant;           // AutoETS: constant series go to the fallback chain
    int n_param;                 // k of the information criteria
    double *aicc;                // [ld]
    double *yhat;                // [n_series x h] for this spec
    int32_t *status;             // [ld]
    int32_t *evals, *iters, *passes; // [ld]
    // inspection pass of the final kernel (all NULL in a normal run): only series whose selected model is `insp_code` take part
    const int32_t *insp_sel;     // [n_series] selected model code
    int32_t insp_code;
    double *insp_fitted;         // [t_rows x ld] one-step fitted values (time-major)
    double *insp_states;         // [(2 + m) x ld] final level, growth, seasonal states by phase
    double *insp_info;           // [8 x ld] alpha, beta, gamma, phi, aic, aicc, bic, sse
    // lane-level efficiency of the round kernels (round 5; NULL = not counted): [0] += passes the wave streamed, [1] += lane-passes
    // that evaluated a trial point of a running problem -- live_lane_passes / (64 wave_passes) is the share of the issued lanes that
    // did work (a converged or parked lane idles until its wave leaves; a wave of the one-wave-per-problem driver counts 64 live lanes)
    unsigned long long *lane_stats;
    // a batch with ONE candidate spec (explicit ETS, fitted or with given parameters): its final pass is the second sweep of the intervals'
    // population sd (forecast.rs:2558-2591: sum of (y - mean)^2 in time order), so prep_kernel needs none (PrepArgs::skip_sd).  NULL otherwise.
    const double *mean;          // [ld] series means (prep_kernel)
    double *sd_out;              // [ld]
    // developer instrument (round 6; NULL = off: ANOFOX_HIP_TUNE wave_trace=<file>): every wave of a round kernel that streams at least one
    // pass appends one record {tag, start, end (s_memrealtime ticks, 100 MHz), HW_ID | XCC_ID << 32} -- who is resident when, and how long a
    // launch's waves wait for a SIMD with room.  wave_trace[0] = records used (atomic), [1] = capacity, records from [4] on.
    int wave_prio;               // issue priority of the round kernel's waves (s_setprio 0..3): the chains that end the step run ahead of their SIMD's other wave
    unsigned long long *wave_trace;
    unsigned long long wave_trace_tag;   // spec order index << 32 | round << 16 | workgroups of the launch are not needed: blockIdx goes in
};

struct SelectArgs {
    int n_series, h, n_slots;
    size_t ld;
    const int32_t *len;          // series with len <= 0 are not part of this group
    const double *aicc;          // [n_slots x ld]
    const double *yhat_slots;    // [n_slots x n_series x h]
    const int32_t *slot_spec;    // [n_slots] spec id
    const int32_t *status_slots; // [n_slots x ld]
    const int32_t *passes_slots; // [n_slots x ld]
    const int32_t *evals_slots;
    double *yhat;                // [n_series x h]
    int32_t *model_code;         // [n_series]
    int32_t *status;             // [n_series]  0 ok, -1 needs fallback
    uint32_t *fallback_mask;     // [n_series] 1 where the fallback chain must run
    int32_t *passes_total;       // [n_series] streamed passes summed over specs (+1 final each)
    int32_t *evals_total;
};

struct ClassicArgs {
    const double *y; size_t ld; const int32_t *len; int n_series;
    int m, h;
    int optimized;               // SES / SeasonalES: optimise alpha (else fixed_alpha)
    double fixed_alpha;
    const uint32_t *mask;        // NULL = all series; else run only where mask[s] == want
    uint32_t want;
    int min_len;                 // series shorter than this fail (status = FIT_SHORT)
    double *yhat;                // [n_series x h]
    int32_t *status;             // [n_series]
    int32_t *passes;             // [n_series] (accumulated)
    int32_t model_code;          // written to model_code[s] when not NULL
    int32_t *model_code_out;
    double *ring_scratch;        // periods whose K * m * 64 ring does not fit LDS: that many doubles per workgroup in HBM
    const int32_t *m_col;        // [ld] per-column period (constant within 64 columns) of a merged batch; NULL = m.  min_len is then 2 m (Holt-Winters) of the column's own period
};

enum SimpleKind { SK_NAIVE = 0, SK_SEASONAL_NAIVE = 1, SK_SMA = 2, SK_DRIFT = 3, SK_TOY_ARIMA = 4 };
struct SimpleArgs {
    const double *y; size_t ld; const int32_t *len; int n_series;
    int kind, h, period, window;
    double *yhat; int32_t *status;
};

struct IntervalArgs {
    int n_series, h;
    const double *yhat, *sd;
    const int32_t *status;
    double z;
    double *lower, *upper;
};

// ETS fit kernels, one per (spec id, ring variant).  ring: 0 = none / VGPR ring for the
// compile-time period given, -1 = LDS ring.  Returns NULL when not instantiated.
typedef void (*FitLaunchFn)(const FitArgs &, hipStream_t);
struct FitLaunchers { FitLaunchFn round_seq, round_spec, round_spec2, round_auto, final; size_t nm_scratch_per_wg; FitLaunchFn round_k4, round_auto_k4; };   // sequential / speculative / two-level speculative rounds, all three behind a device-side choice, final pass; doubles of global simplex scratch per workgroup (0: the simplex rests in LDS); the K4 forms of round_seq / round_auto (additive class, NULL otherwise: ets_fit_kernel.hpp)
// `m`: the period (7 and 12 have compile-time variants), or ETS_PERLANE_LDS / ETS_PERLANE_HBM for the round kernels of a merged
// batch of several periods (per-lane period, ring in LDS / in HBM scratch sized by the batch's largest period)
constexpr int ETS_PERLANE_LDS = -3, ETS_PERLANE_HBM = -4;
// `yt`: storage type of the block the kernels stream (ets_device.hpp YT_F64 / YT_F32 / YT_U16); every entry is NULL when the
// combination is not instantiated (compact types x per-lane period variants)
FitLaunchers ets_fit_launcher(int spec_id, int m, int yt = 0);
FitLaunchers classic_fit_launcher(int kind, int m);          // fit_classic.hip: the SES / Holt / Holt-Winters / SeasonalES family on the round kernels (final = NULL)
struct ClassicArgs;
void launch_classic_final(int kind, const FitArgs &a, const ClassicArgs &c, hipStream_t stream);

// compaction of the unfinished problems: series_next[0..n_next) = the series of the previous map whose done flag
// is 0 (one ballot + one atomic per wave; the order of the survivors is not preserved, results do not depend on it)
void launch_compact(const int32_t *series_prev, const int32_t *n_prev, int n_series, const int32_t *done,
                    int32_t *series_next, int32_t *n_next, hipStream_t, int32_t *n_clear = nullptr);
// out[t * ld_out + p] = y[t * ld + series_of[p]] for p < *n_active, t < t_max
// seasonal period detection (first / strongest autocorrelation peak, 0 = none) of every column of a time-major block; series of up
// to DETECT_LDS_ROWS observations are held in LDS, longer ones in `scratch` (detect_scratch_doubles doubles, else unused)
constexpr int DETECT_LDS_ROWS = 8192;        // 1.5 x 8 B x 8,192 = 96 KB of the CU's 160 KB
constexpr int DETECT_LONG_GRID = 1024;       // workgroups of the scratch variant (each owns 1.5 t_rows doubles of scratch)
size_t detect_scratch_doubles(int n_series, int t_rows);
void launch_detect_periods(const double *y, size_t ld, const int32_t *len, int n_series, int t_rows, double *scratch, int32_t *period,
                           double *best_acf, hipStream_t stream);
// `cap`: columns `out` has room for -- the copy is skipped (the round kernel then indexes `y` by series) while more problems run
// `elem_bytes`: 8 (fp64 block), 4 or 2 (compact copy: ld / ld_out still count columns)
void launch_gather_columns(const void *y, size_t ld, const int32_t *series_of, const int32_t *n_active, int n_series,
                           int t_max, void *out, size_t ld_out, hipStream_t stream, int cap, int elem_bytes = 8);
// Compact copies of the time-major block for the round kernels (round 6): out32[t * ld + s] = (float)y, out16 = (uint16_t)y for every
// cell of the block, and misfit[0] / misfit[1] += the OBSERVATIONS (t < len[s]) that do not survive the round trip through float /
// uint16_t exactly.  A batch streams a compact copy only when its counter is zero (host_api.hip): bit-identical by construction.
void launch_compact_block(const double *y, size_t ld, const int32_t *len, int n_series, int t_rows, float *out32, unsigned short *out16,
                          unsigned int *misfit, hipStream_t stream);

// AutoARIMA (arima.hip): prep (D, d, differenced block), stepwise CSS search (advance / fit sweeps), forecast + integration
struct ArimaArgs {
    const double *y; size_t ld; const int32_t *len; int n_series;
    int m, h;
    int t_max;                          // rows of y (longest series of the batch)
    void *ws; size_t ws_bytes;          // search workspace, arima_workspace_bytes(n_series, t_max) bytes
    int32_t *wlen, *d, *D;              // [ld]
    double *wmean, *wsd, *last_d0, *last_d1;
    int32_t *order;                     // [5 x ld] p, q, P, Q, constant
    double *xbest;                      // [6 x ld] optimiser coordinates of the selected model
    double *aicc;
    int32_t *status, *evals, *passes, *models;
    double *yhat;                       // [n_series x h]
    int32_t *model_code;                // 1000000 + p*1e5 + d*1e4 + q*1e3 + P*100 + D*10 + Q
    int trace;                          // debugging: the refit kernel prints per-wave timings (ANOFOX_HIP_TUNE arima_trace=1|2)
    double lookahead, spec_factor;      // schedule knobs of the search (host_api.hip Tunables: lookahead once the queue fits the resident lanes
    int lookahead_depth;                //   this many times over; four lanes per problem below spec_factor x the resident groups)
    const std::atomic<int> *concurrent; // host only: AutoARIMA runs in flight in this process (the parts of a call with detected periods run side by side)
    double shared_chunk_rounds;         // ... with more than one, a fit launch takes at most this many rounds of the resident lanes (tune
                                        // arima_shared_chunk_rounds; 0 = whole queue): see launch_arima
    int prep_lanes;                     // series per wave of arima_prep_kernel (tune arima_prep_lanes; default 64 = one full wave per 64 series)
    int queue_sort;                     // order of the fit queue (tune arima_queue_sort; arima.hip ar_bucket): 0 as emitted, 1..3 by series within a bucket
    int refit_budget;                   // exact-likelihood refit: evaluations per series in the sequential launch before the speculative one takes over (0: one launch)
    double *long_scratch;               // seasonal period above 24: HBM scratch of arima_long_scratch_doubles() doubles (rings, polynomials), else NULL
    int ml_refit;                       // exact-likelihood (Kalman / Chandrasekhar) refit of the selected models; 0 keeps the CSS estimates
    const int32_t *m_col;               // [ld] merged batch of several LONG periods (all above 24): the period of every series; `m` is then the
                                        // largest (scratch sizes); NULL = one period `m` for the whole batch
};
size_t arima_workspace_bytes(int n_series, int t_max);
size_t arima_long_scratch_doubles(int n_series, int m, int max_fit_waves);   // 0 for periods whose rings live in LDS (m <= 24)
int arima_max_fit_waves();                                                  // resident waves of the fit kernels (what the scratch is sized for)
int launch_arima(const ArimaArgs &, hipStream_t);   // returns the number of kernel launches; synchronises the stream between sweeps

// dm_recip (det_math.hpp) against the compiled division on `n` generated operands of the admissible domain: the number of operands
// whose quotients differ (0 is the claim), ~0 when the device could not run it; *first_bad: one of them
unsigned long long recip_selftest(unsigned long long n, unsigned long long seed, double *first_bad, hipStream_t stream);

void launch_prep(const PrepArgs &, hipStream_t);
void launch_select(const SelectArgs &, hipStream_t);
void launch_classic(int kind, const ClassicArgs &, hipStream_t);
void launch_simple(const SimpleArgs &, hipStream_t);
void launch_intervals(const IntervalArgs &, hipStream_t);

} // namespace anofox
