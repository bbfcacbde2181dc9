// arima.hip -- AutoARIMA on gfx950: differencing tests, stepwise (p,q,P,Q,constant) search with a
// conditional-sum-of-squares fit per candidate, forecast + integration.
//
// Reference call site: crates/anofox-fcst-core/src/forecast.rs:1435-1521 (AutoARIMAConfig::default()
// [.with_seasonal_period(m)]); the arithmetic is in the un-vendored anofox-forecast 0.15.3 crate, so this
// follows the published Hyndman-Khandakar procedure exactly as restated by the CPU checker (oracle/arima.c):
// D by seasonal strength > 0.64, d by KPSS (lag trunc(3 sqrt(n)/13), 0.463), CSS over the coefficients themselves, boxed to
// [-0.99, 0.99] by clipping, minimised by Nelder-Mead (absolute initial steps), stepwise neighbourhood search on AICc with the
// lineage's root check (a candidate with an AR or MA root inside radius 1.001 is inadmissible).  Round 4: box + root check replace
// the tanh-PACF transform -- this is the procedure that reproduces the reference's known answer (oracle/arima.h).
//
// Organisation (problem-parallel): the fit of one candidate order depends on nothing but the series (fixed start,
// fixed steps), so the stepwise search is split into
//   advance  one lane per series: replays the sequential search against a per-series cache of candidate results; when
//            it meets a candidate that has not been fitted yet it queues that one and every other unfitted candidate
//            of the same sweep (speculation: the sequential search would stop the sweep at its first improvement);
//   fit      persistent lanes pull (series, order) problems from the queue, longest dimension first, and run
//            Nelder-Mead to completion -- every objective evaluation is one streamed pass of the CSS recursion over the
//            lane's row of the series-major differenced block W[s * tw + t]; results go to the cache;
// repeated until no series queues anything.  Replay reproduces the sequential search exactly (same candidates tried,
// same order, same model count), so the selected model and its coefficients are those of the checker.  No MFMA
// (scalar recursions); the pass is VALU-issue bound.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "det_math.hpp"
#include "kernels.hpp"
#include "nm.hpp"

namespace anofox {

constexpr int AR_MAXP = 5, AR_MAXSP = 2, AR_MAXORDER = 5, AR_MAXDIM = 6, AR_MAXMODELS = 94;
// Seasonal periods up to AR_LDS_PERIOD keep the ring of seasonal lags in LDS (2 m + 4 slots of {e, v} per lane: 52 KB per wave at
// m = 24); longer ones -- weekly data with a yearly period of 52, hourly 168, daily 365: the reference takes any period,
// forecast.rs:1447-1451 -- keep it in an HBM scratch area of the wave (same lane-minor layout: a slot is one coalesced 1 KB access),
// as the ETS kernels do above ETS_LDS_PERIOD.  Above AR_MAX_PERIOD the host fails the series loudly (oracle: ARIMA_MAX_PERIOD).
constexpr int AR_LDS_PERIOD = 24, AR_MAX_PERIOD = 2048;
// Nelder-Mead budgets (evaluations / iterations), oracle/arima.h: every candidate of the stepwise search gets a bounded run
// (ARIMA_SEARCH_EVALS: an approximate criterion, which keeps the whole search inside the reference's measured cost), the
// selected model's CSS estimates then run to convergence (ARIMA_POLISH_NM_CAP x dim)
__host__ __device__ inline int ar_search_cap(int dim) { return 30 + 15 * dim; }
constexpr double AR_COEF_BOX = 0.99;                  // oracle/arima.h ARIMA_COEF_BOX
constexpr double AR_ROOT_MIN = 1.001;                 // oracle/arima.h ARIMA_ROOT_MIN
__host__ __device__ inline int ar_polish_cap(int dim) { return 100 * dim; }
constexpr int AR_KEYS = 6 * 6 * 3 * 3 * 2;            // order keys (p, q, P, Q, constant)
constexpr int AR_KEYWORDS = (AR_KEYS + 31) / 32;      // bitmap words
constexpr int AR_SWEEP = 17;                          // candidates of one sweep (8 seasonal, 8 non-seasonal, constant)
constexpr int AR_S = 32;                              // steps per streamed block of the CSS pass (run-time period)
constexpr int AR_SPARE = 96;                          // spare elements per row of W (two blocks of the longest block length)

__device__ __forceinline__ int ar_wave_max(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { int w = __shfl_xor(v, o); v = w > v ? w : v; }
    return v;
}

struct ArOrd { int p, q, P, Q, c; };
__device__ __forceinline__ int ar_dim(const ArOrd &o) { return o.p + o.q + o.P + o.Q + o.c; }
__device__ __forceinline__ int ar_key(const ArOrd &o) { return (((o.p * 6 + o.q) * 3 + o.P) * 3 + o.Q) * 2 + o.c; }
// Shape classes of the CSS pass (ar_css_pass_impl NP, NQ, NSP, NSQ): a wave runs the cheapest variant that covers its live lanes
// (ar_css_pass_shaped).  Chosen on the candidates the stepwise search of M5-like series fits (oracle built with -DARIMA_TRACE, 300
// series, 498k evaluations): fused multiply-adds per step 15 -> 6.7 on average with the four seasonal rows, 4.05 if every order had its
// own.  The two rows WITHOUT seasonal factors also drop the ring of seasonal lags -- in the HBM-ring variants of the long periods that is
// 48 of the 56 bytes a step moves.
constexpr int AR_NCLS = 6;
__host__ __device__ inline int ar_shape_class(int p, int q, int P, int Q)
{
    if (P == 0 && Q == 0) {
        if (p <= 1 && q <= 1) return 0;                         // pass <1, 1, 0, 0>:  3 per step, no ring
        if (p <= 2 && q <= 3) return 1;                         // pass <2, 3, 0, 0>:  6, no ring
        return 5;
    }
    if (p <= 1 && q <= 1 && P <= 1 && Q <= 1) return 2;         // pass <1, 1, 1, 1>:  5
    if (p <= 1 && q <= 2 && P <= 1 && Q <= 2) return 3;         // pass <1, 2, 1, 2>:  7
    if (p <= 2 && q <= 3) return 4;                             // pass <2, 3, 2, 2>: 10
    return 5;                                                   // pass <5, 5, 2, 2>: 15
}
// problem queues: one per (dimension, shape class), so that the 64 (or 16) problems a wave starts with share a pass variant; fetched
// longest first (dimension, then class, descending: the tail of a launch is made of the cheap problems)
constexpr int AR_NBUCKETS = (AR_MAXDIM + 1) * AR_NCLS;
constexpr int AR_QC = 32;                                       // ws.counts[AR_QC + bucket]: queue lengths
// Queue order (ArimaArgs::queue_sort).  The candidates of one series read the same row of W, and the big sweeps are bound by exactly
// that traffic (round 4, once the pass stopped paying for absent terms: 6 TB/s of algorithmic row bytes on the two largest launches), so
// the queue of a sweep is sorted by series within each bucket: neighbouring lanes then hold candidates of the same series, their row
// loads carry the same addresses and the memory pipeline fetches the row once.
//   0  as emitted (atomic order), buckets (dimension, shape class)
//   1  buckets (dimension, shape class), by series within a bucket
//   2  buckets by dimension only (shape classes mix in a wave: a wider pass variant, more lanes per row)
//   3  one bucket: by series only
__device__ __forceinline__ int ar_bucket(const ArOrd &o, int sort_mode)
{
    if (sort_mode == 3) return 0;
    if (sort_mode == 2) return ar_dim(o) * AR_NCLS;
    return ar_dim(o) * AR_NCLS + ar_shape_class(o.p, o.q, o.P, o.Q);
}

// the optimiser's coordinates are the coefficients, read through the box (oracle/arima.c box_coef), k <= 5
__device__ __forceinline__ void ar_pacf(const double *u, int k, double *phi)
{
    for (int j = 0; j < k; j++) {
        double a = u[j];
        if (a < -AR_COEF_BOX) a = -AR_COEF_BOX;
        if (a > AR_COEF_BOX) a = AR_COEF_BOX;
        phi[j] = a;
    }
}

// All roots of 1 - sum_i c_i z^i outside the circle of radius r1 (oracle/arima.c roots_outside): coefficients scaled by r1^i, then
// the step-down recursion -- every reflection coefficient inside (-1, 1).  Compile-time indices only (the arrays stay in registers);
// entries beyond k are never read by an active step, so the active arithmetic is the oracle's operation for operation.
template <int N>
__device__ __forceinline__ bool ar_roots_outside(const double (&c)[N], int k, double r1)
{
    double al[N], tmp[N];
    double sc = 1.0;
#pragma unroll
    for (int i = 0; i < N; i++) { sc = sc * r1; al[i] = i < k ? c[i] * sc : 0.0; tmp[i] = 0.0; }
    bool ok = true;
#pragma unroll
    for (int j = N; j >= 1; j--) {
        if (j <= k) {
            const double kj = al[j - 1];
            if (!(fabs(kj) < 1.0)) ok = false;
            const double den = 1.0 - kj * kj;
#pragma unroll
            for (int i = 1; i <= j - 1; i++) tmp[i - 1] = fma(kj, al[j - i - 1], al[i - 1]) / den;
#pragma unroll
            for (int i = 1; i <= j - 1; i++) al[i - 1] = tmp[i - 1];
        }
    }
    return ok;
}

// LDS layout per wave (lane-minor): [ring: R slots x 64 lanes x {e, v}][simplex 42 + values 7][acoef L1][bcoef L1][tried 11 words]
// The ring holds e_t and v_t of step t side by side in slot t & (R - 1), R = the power of two >= 2 m + 6, so that one
// 128-bit LDS access moves both and the slot index is a wave-uniform mask (no wrap arithmetic per lane).

// LDS per wave (lane-minor): [ring: R slots x 64 lanes x {e, v}][simplex vertices 42]
// The ring holds e_t and v_t of step t side by side in slot t mod R, R = 2 m + 6 (the two seasonal lags of a 4-step
// sub-block plus the q + m Q residuals the forecast needs), so that one 128-bit LDS access moves both; slot indices are
// wave-uniform and kept in scalar registers.  (Forecast kernel: R = 2 m + 6; fit kernels: ar_fit_ring_slots.)
typedef double ar_ev_t __attribute__((ext_vector_type(2)));
__host__ __device__ inline int ar_ring_slots(int m) { return 2 * m + 6; }
struct ArLds {
    double *base; int R;
    int col;              // lane column holding this lane's simplex (its own, or its group leader's in the speculative fit)
    double *gring;        // the ring of R slots in HBM scratch (long periods), else NULL: the ring is the first 2 R x 64 doubles of `base`
    double *tile;         // fit kernels with the cooperative row loader: LDS tile behind the simplex (ArCoop), else NULL
    bool lane_m = false;  // merged batch of several long periods: R and the period are PER-LANE quantities (pass variant 6: the ring is
                          // still slot k of lane l at gring[k * 64 + l], sized by the batch's largest period; only the slot arithmetic is per lane)
    // (round 4, measured: the simplex in an HBM scratch area instead -- it is only touched between passes -- costs 25 % per pass on a
    //  lone wave, 258 -> 340 ms on the M5 batch: the few dozen dependent accesses of a Nelder-Mead update are L2 round trips then)
    __device__ double *smp() const { return gring ? base : base + (size_t)2 * R * NM_BLOCK; }
    __device__ double &sim(int k, int i) const { return smp()[(k * AR_MAXDIM + i) * NM_BLOCK + col]; }
    __device__ double e_at(int t) const { return (gring ? gring : base)[((size_t)(t % R) * NM_BLOCK + threadIdx.x) * 2]; }
};
// the fit kernels need the two seasonal lags of a 4-step sub-block only: 2 m + 4 slots (39 KB per wave at m = 7 with the
// 42 simplex coordinates: four waves per CU, one per SIMD); the function values of the simplex stay in registers
__host__ __device__ inline int ar_fit_ring_slots(int m) { return 2 * m + 4 < 8 ? 8 : 2 * m + 4; }
static size_t ar_lds_bytes(int m)
{
    const bool ring_in_lds = !(m == 7 || m > AR_LDS_PERIOD);
    const bool coop = m == 7 || m <= 1 || m > AR_LDS_PERIOD;          // ar_fit_coop of the period's pass variant
    // the tile of the longest block any cooperative variant uses (S = 32: 64 x 17 units of 16 bytes)
    return sizeof(double) * ((size_t)((ring_in_lds ? 2 * ar_fit_ring_slots(m) : 0) + (AR_MAXDIM + 1) * AR_MAXDIM) * NM_BLOCK + (coop ? (size_t)2 * NM_BLOCK * 17 : 0));
}
// HBM scratch of a long period (doubles): the fit kernels' rings (one per resident wave), the forecast kernel's ring + expanded
// polynomials and the prep kernel's seasonal figure (one per 64 series) -- the stages run one after the other and share it
__host__ __device__ inline size_t ar_fc_scratch_doubles(int m) { return (size_t)(2 * ar_ring_slots(m) + 2 * (AR_MAXP + AR_MAXSP * m + 1)) * NM_BLOCK; }
size_t arima_long_scratch_doubles(int n_series, int m, int max_fit_waves)
{
    if (m <= AR_LDS_PERIOD) return 0;
    const size_t grid = (size_t)((n_series + NM_BLOCK - 1) / NM_BLOCK);
    const size_t fit = (size_t)max_fit_waves * 2 * ar_fit_ring_slots(m) * NM_BLOCK;
    const size_t fc = grid * ar_fc_scratch_doubles(m);
    return fit > fc ? fit : fc;                  // (the prep kernel's m x 64 per workgroup is below the forecast kernel's need)
}

// simplex function values in registers; run-time index by select chains (a handful of v_cndmask per access, against
// tens of thousands of instructions per pass)
struct ArFs {
    double v[AR_MAXDIM + 1];
    __device__ __forceinline__ double get(int k) const
    {
        double r = v[0];
#pragma unroll
        for (int j = 1; j <= AR_MAXDIM; j++) r = (k == j) ? v[j] : r;
        return r;
    }
    __device__ __forceinline__ void set(int k, double x)
    {
#pragma unroll
        for (int j = 0; j <= AR_MAXDIM; j++) v[j] = (k == j) ? x : v[j];
    }
};

// the four factor polynomials of a trial point, zero padded (registers)
struct ArFac { double phi[AR_MAXP], th[AR_MAXP], Phi[AR_MAXSP], Th[AR_MAXSP]; double mu; int nc; };
__device__ __forceinline__ void ar_factors(const ArOrd &o, int m, const double *x, ArFac &f)
{
#pragma unroll
    for (int i = 0; i < AR_MAXP; i++) { f.phi[i] = 0.0; f.th[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < AR_MAXSP; i++) { f.Phi[i] = 0.0; f.Th[i] = 0.0; }
    int k = 0;
    ar_pacf(x + k, o.p, f.phi); k += o.p;
    ar_pacf(x + k, o.q, f.th); k += o.q;
    ar_pacf(x + k, o.P, f.Phi); k += o.P;
    ar_pacf(x + k, o.Q, f.Th); k += o.Q;
    f.mu = o.c ? x[k] : 0.0;
    f.nc = o.p + m * o.P;
}

// the lineage's admissibility rule on the factors of a point (oracle/arima.c model_roots_ok): the seasonal factors are polynomials
// in z^m, so their radius is AR_ROOT_MIN^m (built by m multiplications, like the oracle)
__device__ __forceinline__ bool ar_model_roots_ok(const ArOrd &o, int m, const ArFac &f)
{
    double rm = 1.0;
    for (int i = 0; i < m; i++) rm = rm * AR_ROOT_MIN;
    const bool a1 = ar_roots_outside<AR_MAXP>(f.phi, o.p, AR_ROOT_MIN), a2 = ar_roots_outside<AR_MAXP>(f.th, o.q, AR_ROOT_MIN);
    const bool a3 = ar_roots_outside<AR_MAXSP>(f.Phi, o.P, rm), a4 = ar_roots_outside<AR_MAXSP>(f.Th, o.Q, rm);
    return a1 && a2 && a3 && a4;
}

// the refit's estimates replace the CSS ones only where they are admissible (oracle refit_ml: fb <= f0 && model_roots_ok)
template <class LDS>
__device__ __forceinline__ bool ar_simplex_best_roots_ok(const ArOrd &o, int m, int D, const LDS &L)
{
    double xb[AR_MAXDIM] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < D; i++) xb[i] = L.sim(0, i);
    ArFac fb;
    ar_factors(o, m, xb, fb);
    return ar_model_roots_ok(o, m, fb);
}

// expanded lag polynomials of the fitted model (forecast only), in LDS after the ring
struct ArPolyLds {
    double *base; int L1;
    __device__ double &a(int k) const { return base[k * NM_BLOCK + threadIdx.x]; }
    __device__ double &b(int k) const { return base[(L1 + k) * NM_BLOCK + threadIdx.x]; }
};
__device__ __forceinline__ void ar_build_poly(const ArOrd &o, int m, const double *x, const ArPolyLds &L, int &La, int &Lb, double &mu)
{
    double phi[AR_MAXP], th[AR_MAXP], Phi[AR_MAXSP], Th[AR_MAXSP];
    int k = 0;
    ar_pacf(x + k, o.p, phi); k += o.p;
    ar_pacf(x + k, o.q, th); k += o.q;
    ar_pacf(x + k, o.P, Phi); k += o.P;
    ar_pacf(x + k, o.Q, Th); k += o.Q;
    mu = o.c ? x[k] : 0.0;
    La = o.p + m * o.P;
    Lb = o.q + m * o.Q;
    for (int i = 0; i <= La; i++) L.a(i) = 0.0;
    for (int i = 1; i <= o.p; i++) L.a(i) = phi[i - 1];
    for (int I = 1; I <= o.P; I++) {
        L.a(m * I) = L.a(m * I) + Phi[I - 1];
        for (int i = 1; i <= o.p; i++) L.a(m * I + i) = L.a(m * I + i) - phi[i - 1] * Phi[I - 1];
    }
    for (int i = 0; i <= Lb; i++) L.b(i) = 0.0;
    for (int i = 1; i <= o.q; i++) L.b(i) = th[i - 1];
    for (int I = 1; I <= o.Q; I++) {
        L.b(m * I) = L.b(m * I) + Th[I - 1];
        for (int i = 1; i <= o.q; i++) L.b(m * I + i) = L.b(m * I + i) - th[i - 1] * Th[I - 1];
    }
    for (int i = 0; i <= Lb; i++) L.b(i) = -L.b(i);
}

// One CSS pass for the whole wave, CASCADED form (oracle/arima.c css_eval): fixed 5 + 2 + 5 + 2 fused
// multiply-adds per step whatever the lane's orders (absent coefficients are exact zeros), the short lags in
// VGPR shift registers, the seasonal lags in the LDS ring, w streamed from the lane's own row.
//     v_t = w'_t - sum phi_i w'_{t-i} ; z_t = v_t - sum Phi_I v_{t-mI} ; u_t = z_t + sum theta_j u_{t-j} ;
//     e_t = u_t + sum Theta_J e_{t-mJ}   (z, u, e from t >= nc = p + m P on)
// MODE 0: seasonal lags read from the ring step by step (m = 2, 3: a lag can fall inside a sub-block);
// MODE 1: m >= 4, every seasonal lag of a 4-step sub-block was produced before it, so its 8 ring slots are fetched up
//         front (independent LDS reads) and the four steps run in registers; MODE 2: m = 1, no seasonal factors at all.
// The step is branch-free.  A block of S steps runs ungated when every live lane is inside its sample for the whole block
// and past its warm-up (base >= max nc, base + S <= min len over the live lanes; idle lanes compute into their own
// slots and are ignored); otherwise the gated variant predicates the ring stores on t < len and selects exact zeros for
// the warm-up and past-the-end steps (selects, not arithmetic on possibly non-finite padding).
// Lanes of a wave work on different series (and different lengths): each streams its own row.  The row is double-buffered
// in registers: the S values of the NEXT block are requested (eight 128-bit loads... per lane) before the S steps of the
// current one run and are only touched afterwards, so the HBM/L2 latency sits behind S steps of recursion.  Rows carry
// 2 S spare elements, so the loads are unconditional.
// M > 0: compile-time period with the fit kernels' ring of R = 2 M + 4 slots and a block length S that is a multiple of R,
// so every ring slot of a block is a compile-time constant (LDS accesses with immediate offsets, no index arithmetic).
template <int M>
struct ArBlockLen { static constexpr int R = 2 * M + 4; static constexpr int value = (R >= 24 ? R : 2 * R); };
template <>
struct ArBlockLen<0> { static constexpr int R = 0; static constexpr int value = AR_S; };

// MODE 3 block length in revolutions of the 2 M ring.  Two (28 steps at m = 7: 112 VGPRs of row buffers) with one wave per SIMD is
// the measured best: ONE revolution brings the weekly fit kernels down to 256 registers and two waves per SIMD (371 registers
// before; 17 doubles of spill), and the M5 batch then takes 381 ms against 358 ms -- the pass issues fp64 operations 68 % of the time
// already (tools/pmc_arima.sh), a second wave only adds its spill traffic.  ar_fit_waves / AR_MODE3_REVS = 2 / 1 rebuilds that variant.
// Round 4, after the pass variants: tried again, with the simplex in an HBM scratch so that LDS would allow the second wave -- 258 -> 339 ms
// (the Nelder-Mead bookkeeping between passes turns into L2 round trips, and the 14-step block adds loop overhead); reverted.
constexpr int AR_MODE3_REVS = 2;
// COOP (fit kernels, where LDS has room): the rows of a block are fetched COOPERATIVELY.  The lanes of a fit wave hold unrelated
// series, so "every lane streams its own row" makes each 128-bit load instruction touch 64 different cache lines: rocprofv3 showed the
// sequential fit kernel waiting on memory 55 % of its time at 35 % VALU issue (profiles/r03_pmc_traffic_arima.json), the four-lanes-
// per-problem kernel (16 distinct rows per wave) at 37 % / 55 % -- the address pipeline of the CU, not HBM (traffic is 1.1x the
// algorithmic bytes), is what the scattered form saturates.  Cooperatively, one load instruction covers the block of 64 / (S / 2)
// consecutive rows' S-step segments (for S = 28: four rows x 224 contiguous bytes = 18 lines instead of 64), the segments pass through
// an LDS tile of 64 x (S / 2 + 1) 16-byte units (odd row stride: conflict-free 128-bit reads) and every lane reads its own row's
// segment back into registers.  Same values, same arithmetic.
template <int S_> struct ArCoop {
    static constexpr int U = S_ / 2;                       // 16-byte units per row segment
    static constexpr int RPI = NM_BLOCK / U;               // rows per load instruction
    static constexpr int NI = (NM_BLOCK + RPI - 1) / RPI;  // load instructions per block
    static constexpr int TS = (U % 2) ? U : U + 1;         // tile row stride in units
    static constexpr int TILE_DOUBLES = 2 * NM_BLOCK * TS;
};
// NP, NQ, NSP, NSQ (round 4): the SHAPE of the pass -- upper bounds on the orders of every live lane of the wave.  The fixed 5 + 2 + 5 + 2
// form pays 15 fused multiply-adds per step for candidates that have 4 on average (M5-like batch: (0,1)(1,1), (0,1)(0,1), (1,0)(1,0) ...
// carry most of the evaluations); the terms beyond a lane's orders multiply exact zeros, the oracle does not compute them at all, so a pass
// that leaves out the terms NO lane of the wave has is the same arithmetic.  The fit kernels queue their problems by shape class
// (ar_shape_class) so that the lanes of a wave agree, and pick the variant per pass from the wave's live lanes (ar_css_pass_shaped).
template <int MODE, int M, bool COOP = false, int NP = AR_MAXP, int NQ = AR_MAXP, int NSP = AR_MAXSP, int NSQ = AR_MAXSP, bool KEEP_E = true>
__device__ __noinline__ double ar_css_pass_impl(const double *wrow, int len, int wave_len_v, bool live, const ArFac &fin, int m_v,
                                                const ArLds &L)
{
    constexpr bool SEAS = MODE != 2 && (NSP > 0 || NSQ > 0);      // some lane may have a seasonal factor: the ring of lags is kept
    constexpr bool LAG2 = MODE != 2 && (NSP > 1 || NSQ > 1);      // ... and its second lag is read
    constexpr bool RING = SEAS || (KEEP_E && MODE != 3);           // KEEP_E: the caller reads the last residuals from the ring (forecast)
    // out of line on purpose: the pass gets its own register allocation (the fit kernel around it is a large state
    // machine), and the coefficients are copied out of the caller's (scratch-resident) block once per pass
    double phi[AR_MAXP], th[AR_MAXP], Phi[AR_MAXSP], Th[AR_MAXSP];
#pragma unroll
    for (int q = 0; q < AR_MAXP; q++) { phi[q] = -fin.phi[q]; th[q] = fin.th[q]; }
#pragma unroll
    for (int q = 0; q < AR_MAXSP; q++) { Phi[q] = -fin.Phi[q]; Th[q] = fin.Th[q]; }
    const double mu = fin.mu;
    const int nc = fin.nc;
    // wave-uniform quantities in scalar registers
    const int wave_len = __builtin_amdgcn_readfirstlane(wave_len_v);
    constexpr bool CT = M > 0;
    // MODE 6 (round 4): MODE 5 with the period -- hence the ring length and every slot index -- a PER-LANE quantity: the series of all
    // detected long periods of a call run as ONE batch (each used to be a latency-bound batch of its own: ~400 of them on the M5 shape)
    constexpr bool LANE_M = MODE == 6, HB = MODE == 5 || MODE == 6;
    const int m = CT ? M : (LANE_M ? m_v : __builtin_amdgcn_readfirstlane(m_v));
    const int R = CT ? ArBlockLen<M>::R : (LANE_M ? L.R : __builtin_amdgcn_readfirstlane(L.R));
    const int lim = live ? len : 0;
    const int nc_max = __builtin_amdgcn_readfirstlane(ar_wave_max(live ? nc : 0));
    const int len_min = -__builtin_amdgcn_readfirstlane(ar_wave_max(live ? -len : -0x3fffffff));
    // explicit address spaces: through the call boundary the pointers are generic, and generic (flat) loads would make
    // every LDS wait also wait for the row prefetch in flight
    typedef const __attribute__((address_space(1))) ar_ev_t *gptr_t;
    typedef typename std::conditional<HB, __attribute__((address_space(1))) ar_ev_t *, __attribute__((address_space(3))) ar_ev_t *>::type lptr_t;
    gptr_t wp_next = (gptr_t)wrow;
    const lptr_t ring = (lptr_t)(HB ? L.gring : L.base) + threadIdx.x;        // slot k of this lane: ring[k * NM_BLOCK]
    double css = 0.0;
    double wl[AR_MAXP] = {0, 0, 0, 0, 0}, ul[AR_MAXP] = {0, 0, 0, 0, 0};
    if (MODE != 3 && RING)
        for (int k = 0; k < R; k++) ring[k * NM_BLOCK] = ar_ev_t{0.0, 0.0};
    // MODE 3: compile-time period with BOTH seasonal lags in registers (shift rings of 2 M values of e and of v, block =
    // two revolutions so every ring index is a constant): no LDS traffic in the pass at all
    constexpr int S = (MODE == 3) ? AR_MODE3_REVS * 2 * M : ArBlockLen<M>::value;
    constexpr int RR = (MODE == 3) ? 2 * M : 1;
    static_assert(S % 2 == 0 && 3 * S <= AR_SPARE, "block length (the row prefetch runs two blocks ahead)");
    double er[RR], vr[RR];
#pragma unroll
    for (int k = 0; k < RR; k++) { er[k] = 0.0; vr[k] = 0.0; }
    typedef ArCoop<S> CO;
    // two blocks in flight (round 4): with one block the loads of block k + 1 had the ~28 steps of block k to arrive -- a few hundred
    // cycles once the pass carries 5-10 multiply-adds per step instead of 15, against a memory round trip of a microsecond and more;
    // a launch of few problems (the late sweeps, the polish) ran at the pace of that round trip
    ar_ev_t cur[S / 2], nxa[COOP ? CO::NI : S / 2], nxb[COOP ? CO::NI : S / 2];
    // COOP: this lane loads unit `cu` of the segments of rows i * RPI + cr (i = 0 .. NI - 1); lanes past RPI * U and rows past 63
    // repeat an in-range address and drop the value
    typedef __attribute__((address_space(3))) ar_ev_t *tile_t;
    const tile_t tile = (tile_t)(L.tile);
    const int cr = (int)threadIdx.x / CO::U, cu = (int)threadIdx.x % CO::U;
    gptr_t rowp[COOP ? CO::NI : 1];
    if (COOP) {
        // the row pointers of the rows this lane fetches from (wave shuffles of the 64-bit pointers, once per pass)
        const unsigned long long mine = (unsigned long long)wrow;
#pragma unroll
        for (int i = 0; i < CO::NI; i++) {
            int row = i * CO::RPI + cr;
            row = row < NM_BLOCK ? row : NM_BLOCK - 1;
            const unsigned lo = (unsigned)__shfl((int)(unsigned)(mine & 0xffffffffull), row);
            const unsigned hi = (unsigned)__shfl((int)(unsigned)(mine >> 32), row);
            rowp[i] = (gptr_t)(((unsigned long long)hi << 32) | lo) + cu;
        }
    }
    auto coop_issue = [&](ar_ev_t (&nxt)[COOP ? CO::NI : S / 2], const int unit0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CO::NI; i++) nxt[i] = rowp[i][unit0];
    };
    auto coop_commit = [&](const ar_ev_t (&nxt)[COOP ? CO::NI : S / 2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CO::NI; i++)
            if (cr < CO::RPI && i * CO::RPI + cr < NM_BLOCK) tile[(i * CO::RPI + cr) * CO::TS + cu] = nxt[i];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < S / 2; j++) cur[j] = tile[(int)threadIdx.x * CO::TS + j];
        __builtin_amdgcn_wave_barrier();
    };
    // issue the loads of the block that starts at 16-byte unit `unit0` of the rows into one of the two buffers / make a buffer current
    auto issue = [&](ar_ev_t (&nxt)[COOP ? CO::NI : S / 2], const int unit0) __attribute__((always_inline)) {
        if (COOP) coop_issue(nxt, unit0);
        else {
#pragma unroll
            for (int j = 0; j < S / 2; j++) nxt[j] = wp_next[unit0 + j];
        }
    };
    auto commit = [&](const ar_ev_t (&nxt)[COOP ? CO::NI : S / 2]) __attribute__((always_inline)) {
        if (COOP) coop_commit(nxt);
        else {
#pragma unroll
            for (int j = 0; j < S / 2; j++) cur[j] = nxt[j];
        }
    };
    // (not where the registers are spoken for: the HBM-ring variants already hold two sub-blocks of lags in flight and would spill
    //  400 bytes more; the four-lane kernel without a period keeps its two waves per SIMD)
    constexpr bool DEEP = !HB && !(MODE == 2 && !COOP);      // (the four-lane kernel without it: +2 % on the M5 batch, same-box A/B)
    issue(nxa, 0);
    commit(nxa);
    if (DEEP) issue(nxa, S / 2);
    // ring slots of t0, t0 - m, t0 - 2m (scalar, advanced by 4 per sub-block)
    int s0 = 0, s1 = (R - m % R) % R, s2 = (R - (2 * m) % R) % R;
    auto wrap = [&](int x) __attribute__((always_inline)) { return x >= R ? x - R : x; };
    // slot of step (sb + j) of a block and of its two seasonal lags: constants when the period is a template parameter
    auto slot0 = [&](int sbj, int j) __attribute__((always_inline)) { return CT ? sbj % (CT ? ArBlockLen<M>::R : 1) : wrap(s0 + j); };
    auto slot1 = [&](int sbj, int j) __attribute__((always_inline)) { return CT ? (sbj + 4 * (CT ? ArBlockLen<M>::R : 1) - M) % (CT ? ArBlockLen<M>::R : 1) : wrap(s1 + j); };
    auto slot2 = [&](int sbj, int j) __attribute__((always_inline)) { return CT ? (sbj + 4 * (CT ? ArBlockLen<M>::R : 1) - 2 * M) % (CT ? ArBlockLen<M>::R : 1) : wrap(s2 + j); };

    // Ring in HBM (MODE 5 / 6): the two seasonal lags of a sub-block are requested TWO sub-blocks ahead.  Loaded where they are used,
    // every group of four steps waited out a memory round trip (the long-period fits ran at a third of the weekly kernels' rate);
    // the values exist long before -- the period is above 24, so the lags of steps t0 + 8 .. t0 + 11 were stored at least three
    // sub-blocks ago, and their slots are not written again before step t0 + 12 (R = 2 m + 4 slots) -- so the loads of sub-block
    // k + 2 are issued at the start of sub-block k and arrive behind eight steps of arithmetic.  Same values, same arithmetic.
    static_assert(!HB || S % 4 == 0, "the lag prefetch walks whole sub-blocks");
    ar_ev_t l1p[HB ? 4 : 1], l2p[HB ? 4 : 1], l1q[HB ? 4 : 1], l2q[HB ? 4 : 1];
    if (HB) {
#pragma unroll
        for (int j = 0; j < 4; j++) {       // the lags of the first two sub-blocks lie before the series: the ring was cleared above
            l1p[j] = ar_ev_t{0.0, 0.0}; l2p[j] = ar_ev_t{0.0, 0.0};
            l1q[j] = ar_ev_t{0.0, 0.0}; l2q[j] = ar_ev_t{0.0, 0.0};
            if (SEAS) l1q[j] = ring[wrap(wrap(s1 + 4) + j) * NM_BLOCK];
            if (LAG2) l2q[j] = ring[wrap(wrap(s2 + 4) + j) * NM_BLOCK];
        }
    }
    auto block = [&](const int base, auto gated_tag) __attribute__((always_inline)) {
        constexpr bool GATED = decltype(gated_tag)::value;
#pragma unroll
        for (int sb = 0; sb < S; sb += 4) {
            const int t0 = base + sb;
            ar_ev_t l1[4], l2[4];                 // slots t - m and t - 2m: {e, v}
            double vnew[4], enew[4];
            if (HB) {
                const int n1 = wrap(wrap(s1 + 4) + 4), n2 = wrap(wrap(s2 + 4) + 4);      // slots of sub-block k + 2
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    l1[j] = l1p[j]; l2[j] = l2p[j];
                    l1p[j] = l1q[j]; l2p[j] = l2q[j];
                    if (SEAS) l1q[j] = ring[wrap(n1 + j) * NM_BLOCK];
                    if (LAG2) l2q[j] = ring[wrap(n2 + j) * NM_BLOCK];
                }
            }
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (sb + j >= S) continue;
                    if (SEAS) l1[j] = ring[slot1(sb + j, j) * NM_BLOCK];
                    if (LAG2) l2[j] = ring[slot2(sb + j, j) * NM_BLOCK];
                }
            }
            // STAGED (round 4): the four steps of a sub-block stage by stage instead of step by step.  A wave issues in order, and a step is
            // a chain -- w' -> v -> z -> u -> e -> e^2: written step by step every instruction waited for the one before it (~50 cycles per
            // step for 6 instructions on a wave alone on its SIMD).  Only u (one multiply-add on u_{t-1}) and the sum of squares are
            // recursions; v, z and e are feed-forward once the lags of the sub-block are in hand (they are: m >= 4), so the four v's, the four
            // z's, the four e's are independent instructions next to each other.  Same operations on the same operands.
            constexpr bool STAGED = MODE != 0;
            if constexpr (STAGED) {
                double vtj[4], uj[4];
                bool onj[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (sb + j >= S) continue;
                    const double wv = ((sb + j) & 1) ? cur[(sb + j) / 2].y : cur[(sb + j) / 2].x;
                    const double wp = wv - mu;
                    double vt = wp;
#pragma unroll
                    for (int q = 0; q < NP; q++) vt = fma(phi[q], wl[q], vt);
#pragma unroll
                    for (int q = NP - 1; q > 0; q--) wl[q] = wl[q - 1];
                    if (NP > 0) wl[0] = wp;
                    vtj[j] = vt;
                    if (MODE == 3) {
                        l2[j] = ar_ev_t{er[(sb + j) % RR], vr[(sb + j) % RR]};
                        l1[j] = ar_ev_t{er[(sb + j + M) % RR], vr[(sb + j + M) % RR]};
                    }
                }
                double zj[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (sb + j >= S) continue;
                    double z = vtj[j];
                    if (MODE != 2) {
                        if (NSP > 0) z = fma(Phi[0], l1[j].y, z);
                        if (NSP > 1) z = fma(Phi[1], l2[j].y, z);
                    }
                    zj[j] = z;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (sb + j >= S) continue;
                    double u = zj[j];
#pragma unroll
                    for (int q = NQ - 1; q >= 0; q--) u = fma(th[q], ul[q], u);
                    onj[j] = t0 + j >= nc;
                    if (GATED) u = onj[j] ? u : 0.0;
#pragma unroll
                    for (int q = NQ - 1; q > 0; q--) ul[q] = ul[q - 1];
                    if (NQ > 0) ul[0] = u;
                    uj[j] = u;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (sb + j >= S) continue;
                    const int t = t0 + j;
                    double et = uj[j];
                    if (MODE != 2) {
                        if (NSQ > 0) et = fma(Th[0], l1[j].x, et);
                        if (NSQ > 1) et = fma(Th[1], l2[j].x, et);
                        if (GATED) et = onj[j] ? et : 0.0;
                    }
                    if (MODE == 3 && SEAS) {
                        const bool keep = GATED && !(t < lim);
                        er[(sb + j) % RR] = keep ? er[(sb + j) % RR] : et;
                        vr[(sb + j) % RR] = keep ? vr[(sb + j) % RR] : vtj[j];
                    }
                    vnew[j] = vtj[j]; enew[j] = et;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (sb + j >= S) continue;
                    const double ec = (!GATED || t0 + j < lim) ? enew[j] : 0.0;
                    css = fma(ec, ec, css);
                }
            } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (sb + j >= S) continue;              // (a block length that is not a multiple of four: the last sub-block is short)
                const int t = t0 + j;
                const double wv = ((sb + j) & 1) ? cur[(sb + j) / 2].y : cur[(sb + j) / 2].x;
                const double wp = wv - mu;
                double vt = wp;
#pragma unroll
                for (int q = 0; q < NP; q++) vt = fma(phi[q], wl[q], vt);
#pragma unroll
                for (int q = NP - 1; q > 0; q--) wl[q] = wl[q - 1];
                if (NP > 0) wl[0] = wp;
                double z = vt;
                if (MODE == 0) {
                    if (SEAS) l1[j] = ring[slot1(sb + j, j) * NM_BLOCK];
                    if (LAG2) l2[j] = ring[slot2(sb + j, j) * NM_BLOCK];
                }
                if (MODE == 3) {
                    l2[j] = ar_ev_t{er[(sb + j) % RR], vr[(sb + j) % RR]};
                    l1[j] = ar_ev_t{er[(sb + j + M) % RR], vr[(sb + j + M) % RR]};
                }
                if (MODE != 2) {
                    if (NSP > 0) z = fma(Phi[0], l1[j].y, z);
                    if (NSP > 1) z = fma(Phi[1], l2[j].y, z);
                }
                // the newest lag enters last, so consecutive steps are one fused multiply-add apart
                double u = z;
#pragma unroll
                for (int q = NQ - 1; q >= 0; q--) u = fma(th[q], ul[q], u);
                const bool on = t >= nc;
                if (GATED) u = on ? u : 0.0;
#pragma unroll
                for (int q = NQ - 1; q > 0; q--) ul[q] = ul[q - 1];
                if (NQ > 0) ul[0] = u;
                double et = u;
                if (MODE != 2) {
                    if (NSQ > 0) et = fma(Th[0], l1[j].x, et);
                    if (NSQ > 1) et = fma(Th[1], l2[j].x, et);
                    if (GATED) et = on ? et : 0.0;
                }
                if (MODE == 0) {
                    if (RING && (!GATED || t < lim)) ring[slot0(sb + j, j) * NM_BLOCK] = ar_ev_t{et, vt};
                }
                if (MODE == 3 && SEAS) {
                    const bool keep = GATED && !(t < lim);
                    er[(sb + j) % RR] = keep ? er[(sb + j) % RR] : et;
                    vr[(sb + j) % RR] = keep ? vr[(sb + j) % RR] : vt;
                }
                vnew[j] = vt; enew[j] = et;
                const double ec = (!GATED || t < lim) ? et : 0.0;
                css = fma(ec, ec, css);
            }
            }
            if ((MODE == 1 || MODE == 2 || HB) && RING) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (sb + j < S && (!GATED || t0 + j < lim)) ring[slot0(sb + j, j) * NM_BLOCK] = ar_ev_t{enew[j], vnew[j]};
            }
            if (!CT) { s0 = wrap(s0 + 4); s1 = wrap(s1 + 4); s2 = wrap(s2 + 4); }
        }
    };

    auto run_block = [&](const int base) __attribute__((always_inline)) {
        if (base >= nc_max && base + S <= len_min) block(base, std::false_type{});
        else block(base, std::true_type{});
    };
    if (DEEP) {
        for (int base = 0; base < wave_len; base += 2 * S) {
            issue(nxb, (base + 2 * S) / 2);
            run_block(base);
            commit(nxa);
            if (base + S >= wave_len) break;
            issue(nxa, (base + 3 * S) / 2);
            run_block(base + S);
            commit(nxb);
        }
    } else {
        for (int base = 0; base < wave_len; base += S) {
            issue(nxa, (base + S) / 2);
            run_block(base);
            commit(nxa);
        }
    }
    return css;
}

__device__ __forceinline__ double ar_css_pass(const double *wrow, int len, int wave_len, bool live, const ArFac &f, int m, const ArLds &L)
{
    if (L.gring) return L.lane_m ? ar_css_pass_impl<6, 0>(wrow, len, wave_len, live, f, m, L)       // long periods, one per lane
                                 : ar_css_pass_impl<5, 0>(wrow, len, wave_len, live, f, m, L);      // long period: ring in HBM scratch
    if (L.R == 2 * m + 4) {      // fit kernels: common periods compiled in
        if (m == 7) return ar_css_pass_impl<3, 7>(wrow, len, wave_len, live, f, m, L);
        if (m == 12) return ar_css_pass_impl<1, 12>(wrow, len, wave_len, live, f, m, L);
        if (m == 4) return ar_css_pass_impl<1, 4>(wrow, len, wave_len, live, f, m, L);
    }
    if (m >= 4) return ar_css_pass_impl<1, 0>(wrow, len, wave_len, live, f, m, L);
    if (m <= 1) return ar_css_pass_impl<2, 0>(wrow, len, wave_len, live, f, m, L);
    return ar_css_pass_impl<0, 0>(wrow, len, wave_len, live, f, m, L);
}

// ------------------------------------------------------------------------------------------------
// workspace (one allocation per batch, carved by ArWs::carve; see arima_workspace_bytes)
// ------------------------------------------------------------------------------------------------
// waves the refit kernel is launched with at most (its simplex scratch is sized for them)
__host__ __device__ inline int ar_ml_sim_waves(int n) { const int g = (2 * n + NM_BLOCK - 1) / NM_BLOCK + 8; return g < 2048 ? g : 2048; }
struct ArWs {
    double *W;            // [n x tw] differenced series, series-major, 2 S spare elements per row
    double *cache_aicc;   // [n x AR_KEYS] AICc of every fitted candidate (+inf: fit failed)
    double *cache_x;      // [n x AR_KEYS x 6] optimiser coordinates of every fitted candidate
    int32_t *cache_evals; // [n x AR_KEYS]
    uint32_t *computed;   // [n x AR_KEYWORDS] candidate has been fitted or queued
    uint32_t *tried;      // [n x AR_KEYWORDS] candidate has been tried by the (replayed) sequential search
    double *best_aicc;    // [n]
    int32_t *state;       // [n x 8] stage, idx, base key, best key, have, improved, n_models, fin
    int32_t *q_series, *q_key;   // [AR_NBUCKETS x cap] problem queues by (dimension, shape class)
    int32_t *counts;      // [128] AR_QC + bucket: queue lengths (42), 8 fetch cursor, 96.. trace counters, 9..13 refit class cursors, 16..20 refit class sizes,
                          //      21..25 class cursors of the refit's second (speculative) launch, 26 series parked for it
    int32_t *q_rank;      // [AR_NBUCKETS x cap] rank of a queued problem among its series' problems of the same bucket (emission order)
    int32_t *q2_series, *q2_key;   // the queues sorted by series within a bucket (arima_queue_scan / _scatter_kernel)
    int32_t *hist;        // [AR_NBUCKETS x n] problems per (bucket, series) of the sweep, then their exclusive prefix over the series
    int nser;
    int32_t *qp_series, *qp_key;   // [AR_NBUCKETS x n] polish queues: the selected model of every series whose search has ended, by (dimension, shape class)
    int32_t *counts_p;    // [128] their `counts` (same layout: AR_QC + bucket = queue lengths, 8 = fetch cursor of the polish launch)
    double *ml_sim;       // [waves x 42 x 64] simplex scratch of the exact-likelihood refit
    double *ml_park;      // [n x 64] Nelder-Mead state of the series the first refit launch parks for the second
    int32_t *ml_park_list;   // [n] those series
    size_t tw, cap;
    static size_t align(size_t x) { return (x + 255) & ~(size_t)255; }
    size_t carve(char *p, int n, int t_max)
    {
        tw = (size_t)((t_max + AR_S - 1) / AR_S) * AR_S + AR_SPARE;
        // per (dimension, shape class) queue: six problems per series.  A sweep emits up to 17 per series (more with lookahead) over
        // 42 queues; a queue that is full drops the candidate (its `computed` bit is cleared) and the next sweep queues it again --
        // same results, one more sweep -- which the single-bucket queue order (tune arima_queue_sort = 3) does run into
        // A floor of 512 (round 5): a call with one or a few series runs its sweeps with lookahead 1 or 2 -- up to
        // 18 or 18 x 18 candidates per series, of which more than six can land in one bucket; with cap = 6 n they were dropped and
        // re-queued, and every extra sweep is a host round trip of the latency-bound path the lookahead exists to shorten
        cap = std::max<size_t>((size_t)n * 6, 512);
        size_t off = 0;
        auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += align(bytes); return r; };
        W = (double *)take(sizeof(double) * (size_t)n * tw);
        cache_aicc = (double *)take(sizeof(double) * (size_t)n * AR_KEYS);
        cache_x = (double *)take(sizeof(double) * (size_t)n * AR_KEYS * AR_MAXDIM);
        cache_evals = (int32_t *)take(sizeof(int32_t) * (size_t)n * AR_KEYS);
        computed = (uint32_t *)take(sizeof(uint32_t) * (size_t)n * AR_KEYWORDS);
        tried = (uint32_t *)take(sizeof(uint32_t) * (size_t)n * AR_KEYWORDS);
        best_aicc = (double *)take(sizeof(double) * (size_t)n);
        state = (int32_t *)take(sizeof(int32_t) * (size_t)n * 8);
        q_series = (int32_t *)take(sizeof(int32_t) * AR_NBUCKETS * cap);
        q_key = (int32_t *)take(sizeof(int32_t) * AR_NBUCKETS * cap);
        counts = (int32_t *)take(sizeof(int32_t) * 128);
        q_rank = (int32_t *)take(sizeof(int32_t) * AR_NBUCKETS * cap);
        q2_series = (int32_t *)take(sizeof(int32_t) * AR_NBUCKETS * cap);
        q2_key = (int32_t *)take(sizeof(int32_t) * AR_NBUCKETS * cap);
        hist = (int32_t *)take(sizeof(int32_t) * (size_t)AR_NBUCKETS * n);
        qp_series = (int32_t *)take(sizeof(int32_t) * (size_t)AR_NBUCKETS * n);
        qp_key = (int32_t *)take(sizeof(int32_t) * (size_t)AR_NBUCKETS * n);
        counts_p = (int32_t *)take(sizeof(int32_t) * 128);
        nser = n;
        ml_sim = (double *)take(sizeof(double) * (size_t)ar_ml_sim_waves(n) * 64 * NM_BLOCK);   // AR_ML_CTX <= 64 slots per lane
        ml_park = (double *)take(sizeof(double) * (size_t)n * 64);
        ml_park_list = (int32_t *)take(sizeof(int32_t) * (size_t)n);
        return off;
    }
};
size_t arima_workspace_bytes(int n_series, int t_max) { ArWs w; return w.carve(nullptr, n_series > 0 ? n_series : 1, t_max > 0 ? t_max : 1); }

enum { AS_STAGE = 0, AS_IDX, AS_BASE, AS_BEST, AS_HAVE, AS_IMPROVED, AS_NMODELS, AS_FIN };

__device__ __forceinline__ ArOrd ar_unkey(int key)
{
    ArOrd o;
    o.c = key & 1; key >>= 1;
    o.Q = key % 3; key /= 3;
    o.P = key % 3; key /= 3;
    o.q = key % 6; key /= 6;
    o.p = key;
    return o;
}

// ------------------------------------------------------------------------------------------------
// prep: D (seasonal strength), d (KPSS), differenced block W, moments, integration constants
// ------------------------------------------------------------------------------------------------
__device__ bool ar_kpss_reject(const double *x, size_t ld, int n)
{
    if (n < 4) return false;
    // (unrolled sweeps: the loads of eight steps are in flight together; the sums stay sequential)
    double s = 0.0;
#pragma unroll 8
    for (int i = 0; i < n; i++) s = s + x[(size_t)i * ld];
    const double mean = s / (double)n;
    double cum = 0.0, eta = 0.0, s2 = 0.0;
#pragma unroll 8
    for (int i = 0; i < n; i++) {
        double e = x[(size_t)i * ld] - mean;
        cum = cum + e;
        eta = fma(cum, cum, eta);
        s2 = fma(e, e, s2);
    }
    const double dn = (double)n;
    eta = eta / (dn * dn);
    s2 = s2 / dn;
    const int lag = (int)(3.0 * sqrt(dn) / 13.0);
    // autocovariances of lags 1 .. lag, eight lags per sweep of the series (round 4; it was one sweep per lag: two loads and one
    // dependent multiply-add per term, 2 x lag x n loads -- most of the prep kernel's time).  Every lag keeps its own accumulator and
    // receives its terms in the order t = k .. n - 1, so each sum is the oracle's sum; the window holds the centred values the eight
    // lags of the sweep pair with the current one.
    constexpr int KW = 8;
    for (int k0 = 0; k0 < lag; k0 += KW) {
        double acc[KW], win[KW];                       // win[j] = x[t - (k0 + 1 + j)] - mean
#pragma unroll
        for (int j = 0; j < KW; j++) { acc[j] = 0.0; win[j] = 0.0; }
#pragma unroll 4
        for (int t = k0 + 1; t < n; t++) {
#pragma unroll
            for (int j = KW - 1; j > 0; j--) win[j] = win[j - 1];
            win[0] = x[(size_t)(t - k0 - 1) * ld] - mean;
            const double ct = x[(size_t)t * ld] - mean;
#pragma unroll
            for (int j = 0; j < KW; j++)
                if (k0 + 1 + j <= lag && t >= k0 + 1 + j) acc[j] = fma(ct, win[j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < KW; j++) {
            const int k = k0 + 1 + j;
            if (k <= lag) {
                double wgt = 1.0 - (double)k / ((double)lag + 1.0);
                s2 = s2 + 2.0 * wgt * (acc[j] / dn);
            }
        }
    }
    if (!(s2 > 0.0)) return false;
    return (eta / s2) > 0.463;
}

// Seasonal strength of the classical decomposition (oracle/arima.c oracle_arima_seasonal_strength).  Round 4: the detrended value
// d_i = y_i - (centred moving average at i) is computed ONCE per i, in one sweep with the window of the last L values in a lane-private
// ring (fig + m entries: LDS, or the HBM scratch of a long period), and parked in the series' row of W (free until the copy below); the
// per-phase sums of the first stage are accumulated in that same sweep (each phase still receives its terms in ascending i), the second
// and third stage read the parked d_i.  It used to be three sweeps, each re-reading L values of y per i through a loop of dependent
// loads: 9.5 of the prep kernel's 14 ms on the M5 batch.  Same operations on the same operands, in the same order per accumulator.
__device__ double ar_seasonal_strength(const double *y, size_t ld, int n, int m, double *fig /* [m + L] lane-private scratch, stride NM_BLOCK */,
                                       double *dpark /* [n] lane-private: the series' row of W */)
{
    if (m < 2 || n < 3 * m) return 0.0;
    const int half = m / 2;
    const int L = (m % 2 == 0) ? m + 1 : m;
    const double w = 1.0 / (double)m;
    const double wend = (m % 2 == 0) ? 0.5 / (double)m : w;
    double *const ring = fig + (size_t)m * NM_BLOCK;           // slot t mod L holds y_t for the L values around the current centre
    for (int j = 0; j < m; j++) fig[j * NM_BLOCK] = 0.0;
    for (int t = 0; t < L - 1; t++) ring[t * NM_BLOCK] = y[(size_t)t * ld];
    {
        int start = 0, fresh = L - 1, centre = half, ph = half % m;
        double ynext = y[(size_t)(2 * half) * ld];              // the value that enters the window at the next centre, requested a step ahead
        for (int i = half; i < n - half; i++) {
            ring[fresh * NM_BLOCK] = ynext;
            if (i + 1 < n - half) ynext = y[(size_t)(i + 1 + half) * ld];
            double acc = 0.0;
            int sl = start;
#pragma unroll 4
            for (int k = 0; k < L; k++) {
                acc = acc + ((k == 0 || k == L - 1) ? wend : w) * ring[sl * NM_BLOCK];
                sl = (sl + 1 == L) ? 0 : sl + 1;
            }
            const double d = ring[centre * NM_BLOCK] - acc;
            dpark[i] = d;
            fig[ph * NM_BLOCK] = fig[ph * NM_BLOCK] + d;
            start = (start + 1 == L) ? 0 : start + 1;
            fresh = (fresh + 1 == L) ? 0 : fresh + 1;
            centre = (centre + 1 == L) ? 0 : centre + 1;
            ph = (ph + 1 == m) ? 0 : ph + 1;
        }
    }
    double tot = 0.0;
    for (int j = 0; j < m; j++) {
        const int first = j >= half ? j : j + m;
        const int cnt = (n - half - 1 - first) / m + 1;        // centres i = first, first + m, ... below n - half (n >= 3 m: at least one)
        fig[j * NM_BLOCK] = fig[j * NM_BLOCK] / (double)cnt;
        tot = tot + fig[j * NM_BLOCK];
    }
    const double fmean = tot / (double)m;
    for (int j = 0; j < m; j++) fig[j * NM_BLOCK] = fig[j * NM_BLOCK] - fmean;
    const int nv = n - 2 * half;
    double sd = 0.0, sr = 0.0;
    int ph = half % m;
#pragma unroll 8
    for (int i = half; i < n - half; i++) {
        const double d = dpark[i];
        sd = sd + d;
        sr = sr + (d - fig[ph * NM_BLOCK]);
        ph = (ph + 1 == m) ? 0 : ph + 1;
    }
    const double md = sd / (double)nv, mr = sr / (double)nv;
    double vd = 0.0, vr = 0.0;
    ph = half % m;
#pragma unroll 8
    for (int i = half; i < n - half; i++) {
        const double d = dpark[i];
        const double r = d - fig[ph * NM_BLOCK];
        vd = fma(d - md, d - md, vd);
        vr = fma(r - mr, r - mr, vr);
        ph = (ph + 1 == m) ? 0 : ph + 1;
    }
    if (!(vd > 0.0)) return 0.0;
    double f = 1.0 - vr / vd;
    if (f < 0.0) f = 0.0;
    if (f > 1.0) f = 1.0;
    return f;
}

__global__ __launch_bounds__(NM_BLOCK) void arima_prep_kernel(const ArimaArgs a, const ArWs ws)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int s = blockIdx.x * blockDim.x + threadIdx.x;      // (blockDim.x: 64, or fewer series per wave where that fills the chip -- launch_arima)
    if (s >= a.n_series) return;
    // search state of this series (a series outside this period group keeps wlen = 0 and is skipped everywhere)
    for (int i = 0; i < 8; i++) ws.state[(size_t)s * 8 + i] = 0;
    for (int i = 0; i < AR_KEYWORDS; i++) { ws.computed[(size_t)s * AR_KEYWORDS + i] = 0u; ws.tried[(size_t)s * AR_KEYWORDS + i] = 0u; }
    ws.best_aicc[s] = __builtin_huge_val();
    const int n = a.len[s];
    if (n <= 0) { a.wlen[s] = 0; ws.state[(size_t)s * 8 + AS_FIN] = 1; return; }
    const double *y = a.y + s;
    double *w = ws.W + (size_t)s * ws.tw;
    const size_t ld = a.ld;
    const int m = a.m_col ? a.m_col[s] : a.m;          // (merged batch of long periods: the series' own; a.m is the largest and sizes the scratch)
    int len = n, D = 0, d = 0;
    long long tk[6] = {0, 0, 0, 0, 0, 0};
    tk[0] = wall_clock64();
    double *const fig = a.long_scratch ? a.long_scratch + (size_t)blockIdx.x * ar_fc_scratch_doubles(a.m) + threadIdx.x : lds + threadIdx.x;
    const double strength = m > 1 ? ar_seasonal_strength(y, ld, n, m, fig, w) : 0.0;
    tk[1] = wall_clock64();
    if (m > 1 && strength > 0.64 && n > m + 2) {
        D = 1;
#pragma unroll 8
        for (int t = m; t < n; t++) w[t - m] = y[(size_t)t * ld] - y[(size_t)(t - m) * ld];
        len = n - m;
    } else {
#pragma unroll 8
        for (int t = 0; t < n; t++) w[t] = y[(size_t)t * ld];
    }
    // integration constants of the seasonally differenced series (before the ordinary differences)
    a.last_d0[s] = w[len - 1];
    a.last_d1[s] = len >= 2 ? w[len - 1] - w[len - 2] : 0.0;
    tk[2] = wall_clock64();
    while (d < 2 && len > 3 && ar_kpss_reject(w, 1, len)) {
        double prev = w[0];
#pragma unroll 8
        for (int t = 1; t < len; t++) {
            double cur = w[t];
            w[t - 1] = cur - prev;
            prev = cur;
        }
        len -= 1;
        d++;
    }
    tk[3] = wall_clock64();
    for (int t = len; t < (int)ws.tw; t++) w[t] = 0.0;       // padding read by the streamed blocks (never used)
    double sum = 0.0;
#pragma unroll 8
    for (int i = 0; i < len; i++) sum = sum + w[i];
    const double wmean = sum / (double)len;
    double v = 0.0;
#pragma unroll 8
    for (int i = 0; i < len; i++) { double dd = w[i] - wmean; v = fma(dd, dd, v); }
    a.wlen[s] = len;
    a.d[s] = d;
    a.D[s] = D;
    a.wmean[s] = wmean;
    a.wsd[s] = sqrt(v / (double)len);
    if (len < 3) ws.state[(size_t)s * 8 + AS_FIN] = 1;
    tk[4] = wall_clock64();
    if (a.trace >= 3 && threadIdx.x == 0 && (blockIdx.x % 97) == 0)      // 100 MHz wall clock: 100 ticks per microsecond
        printf("prep wave %d: seasonal strength %lld us, copy %lld us, KPSS + differences %lld us (d = %d), moments + padding %lld us\n", (int)blockIdx.x,
               (tk[1] - tk[0]) / 100, (tk[2] - tk[1]) / 100, (tk[3] - tk[2]) / 100, d, (tk[4] - tk[3]) / 100);
}

// ------------------------------------------------------------------------------------------------
// advance: replay of the sequential stepwise search against the cache; queues what is missing
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ar_candidate(int stage, int idx, const ArOrd &base, int allow_c, int maxP, ArOrd &o)
{
    if (stage == 0) {
        const int sP = maxP ? 1 : 0;
        if (idx == 0) o = ArOrd{2, 2, sP, sP, allow_c};
        else if (idx == 1) o = ArOrd{0, 0, 0, 0, allow_c};
        else if (idx == 2) o = ArOrd{1, 0, sP, 0, allow_c};
        else if (idx == 3) o = ArOrd{0, 1, 0, sP, allow_c};
        else o = ArOrd{0, 0, 0, 0, allow_c ? 0 : -1};   // only when a constant is allowed
    } else {
        const int dPv[8] = {-1, 0, 1, 0, -1, -1, 1, 1}, dQv[8] = {0, -1, 0, 1, -1, 1, -1, 1};
        if (idx < 8) o = ArOrd{base.p, base.q, base.P + dPv[idx], base.Q + dQv[idx], base.c};
        else if (idx < 16) o = ArOrd{base.p + dPv[idx - 8], base.q + dQv[idx - 8], base.P, base.Q, base.c};
        else o = ArOrd{base.p, base.q, base.P, base.Q, 1 - base.c};
    }
    return !(o.c < 0 || o.p < 0 || o.q < 0 || o.P < 0 || o.Q < 0 || o.p > AR_MAXP || o.q > AR_MAXP || o.P > maxP || o.Q > maxP ||
             o.p + o.q + o.P + o.Q > AR_MAXORDER || (o.c && !allow_c));
}

__global__ __launch_bounds__(256) void arima_advance_kernel(const ArimaArgs a, const ArWs ws, const int lookahead)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_series) return;
    int32_t *st = ws.state + (size_t)s * 8;
    if (st[AS_FIN]) return;
    const int len = a.wlen[s];
    const int m = a.m_col ? a.m_col[s] : a.m;
    const int allow_c = (a.d[s] + a.D[s] <= 1) ? 1 : 0;
    const int maxP = m > 1 ? AR_MAXSP : 0;
    uint32_t *tried = ws.tried + (size_t)s * AR_KEYWORDS;
    uint32_t *computed = ws.computed + (size_t)s * AR_KEYWORDS;
    int stage = st[AS_STAGE], idx = st[AS_IDX], n_models = st[AS_NMODELS];
    bool have = st[AS_HAVE] != 0, improved = st[AS_IMPROVED] != 0, fin = false;
    int best_key = st[AS_BEST];
    ArOrd base = ar_unkey(st[AS_BASE]);
    double best_aicc = ws.best_aicc[s];
    int evals = a.evals[s];
    auto impossible = [&](const ArOrd &o) { return len - (o.p + m * o.P) <= 0 || len - (ar_dim(o) + 1) - 1 <= 0; };

    for (;;) {
        if (stage == 0) {
            if (idx >= 5) {
                if (!have) { fin = true; break; }
                stage = 1; idx = 0; base = ar_unkey(best_key); improved = false;
                continue;
            }
        } else {
            if (improved) { base = ar_unkey(best_key); idx = 0; improved = false; }
            if (idx >= AR_SWEEP) { fin = true; break; }
        }
        ArOrd o;
        if (!ar_candidate(stage, idx, base, allow_c, maxP, o) || n_models >= AR_MAXMODELS) { idx++; continue; }
        const int key = ar_key(o);
        if (tried[key >> 5] & (1u << (key & 31))) { idx++; continue; }
        const bool imp = impossible(o);
        if (!imp && !(computed[key >> 5] & (1u << (key & 31)))) {
            // not fitted yet: queue it together with every other unfitted candidate of this sweep, then wait for the fit
            const int last = stage == 0 ? 5 : AR_SWEEP;
            auto emit = [&](const ArOrd &c) {
                const int ck = ar_key(c);
                if ((tried[ck >> 5] | computed[ck >> 5]) & (1u << (ck & 31))) return;
                if (impossible(c)) return;
                computed[ck >> 5] |= (1u << (ck & 31));
                const int bk = ar_bucket(c, a.queue_sort);
                const int pos = atomicAdd(&ws.counts[AR_QC + bk], 1);
                if ((size_t)pos >= ws.cap) { atomicSub(&ws.counts[AR_QC + bk], 1); computed[ck >> 5] &= ~(1u << (ck & 31)); return; }
                ws.q_series[(size_t)bk * ws.cap + pos] = s;
                ws.q_key[(size_t)bk * ws.cap + pos] = ck;
                if (a.queue_sort) {             // (this lane is the only one that touches its series' counters)
                    int32_t *hc = ws.hist + (size_t)bk * ws.nser + s;
                    ws.q_rank[(size_t)bk * ws.cap + pos] = *hc;
                    *hc = *hc + 1;
                }
            };
            for (int j = idx; j < last; j++) {
                ArOrd c;
                if (ar_candidate(stage, j, base, allow_c, maxP, c)) emit(c);
            }
            if (lookahead) {
                // idle lanes ahead (short queue): also fit the sweep that would follow whichever candidate is accepted,
                // so that the next replay usually runs two sweeps deep before it has to wait again
                for (int j = idx - 1; j < last; j++) {
                    ArOrd nb;
                    if (j < idx) { if (!(stage == 0 && have)) continue; nb = ar_unkey(best_key); }   // initial stage: the best so far may stay
                    else {
                        if (!ar_candidate(stage, j, base, allow_c, maxP, nb)) continue;
                        const int nk = ar_key(nb);
                        if (tried[nk >> 5] & (1u << (nk & 31))) continue;
                    }
                    for (int jj = 0; jj < AR_SWEEP; jj++) {
                        ArOrd c;
                        if (!ar_candidate(1, jj, nb, allow_c, maxP, c)) continue;
                        emit(c);
                        if (lookahead >= 2) {
                            // a handful of series left: one sweep deeper still (the neighbours of that neighbour)
                            const int ck = ar_key(c);
                            if (tried[ck >> 5] & (1u << (ck & 31))) continue;
                            for (int kk = 0; kk < AR_SWEEP; kk++) {
                                ArOrd c2;
                                if (ar_candidate(1, kk, c, allow_c, maxP, c2)) emit(c2);
                            }
                        }
                    }
                }
            }
            break;
        }
        // the sequential search tries this candidate now
        tried[key >> 5] |= (1u << (key & 31));
        n_models++;
        idx++;
        if (!imp) {
            evals += ws.cache_evals[(size_t)s * AR_KEYS + key];
            const double aicc = ws.cache_aicc[(size_t)s * AR_KEYS + key];
            if (fabs(aicc) <= 1.7976931348623157e308 && aicc < best_aicc) { best_aicc = aicc; best_key = key; have = true; improved = true; }
        }
    }
    st[AS_STAGE] = stage; st[AS_IDX] = idx; st[AS_BASE] = ar_key(base); st[AS_BEST] = best_key;
    st[AS_HAVE] = have ? 1 : 0; st[AS_IMPROVED] = improved ? 1 : 0; st[AS_NMODELS] = n_models; st[AS_FIN] = fin ? 1 : 0;
    ws.best_aicc[s] = best_aicc;
    a.evals[s] = evals;
    if (fin) {
        const size_t ld = a.ld;
        const ArOrd best = ar_unkey(best_key);
        a.status[s] = have ? FIT_OK : FIT_SHORT;
        a.aicc[s] = best_aicc;
        a.order[(size_t)0 * ld + s] = best.p; a.order[(size_t)1 * ld + s] = best.q; a.order[(size_t)2 * ld + s] = best.P;
        a.order[(size_t)3 * ld + s] = best.Q; a.order[(size_t)4 * ld + s] = best.c;
        for (int i = 0; i < AR_MAXDIM; i++)
            a.xbest[(size_t)i * ld + s] = have ? ws.cache_x[((size_t)s * AR_KEYS + best_key) * AR_MAXDIM + i] : 0.0;
        a.models[s] = n_models;
        // the selected model goes to the polish queue the moment the search ends (round 4: most series are done sweeps before the last
        // one; their polish runs beside the late, latency-bound sweeps -- launch_arima)
        if (have && ar_dim(best) > 0) {
            const int bk = ar_bucket(best, 1);
            const int pos = atomicAdd(&ws.counts_p[AR_QC + bk], 1);
            ws.qp_series[(size_t)bk * ws.nser + pos] = s;
            ws.qp_key[(size_t)bk * ws.nser + pos] = best_key;
        }
    }
}

// series that never enter the search (too short / not in this group)
__global__ __launch_bounds__(256) void arima_skip_kernel(const ArimaArgs a)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_series || a.len[s] <= 0) return;
    if (a.wlen[s] < 3) { a.status[s] = FIT_SHORT; a.models[s] = 0; }
}

// the queue of a sweep, sorted by series within each bucket: exclusive prefix of the per-(bucket, series) counts over the series (one
// workgroup per bucket), then every problem moves to prefix[its series] + its rank
__global__ __launch_bounds__(1024) void arima_queue_scan_kernel(const ArWs ws)
{
    __shared__ int part[1024];
    const int b = blockIdx.x, n = ws.nser;
    if (ws.counts[AR_QC + b] == 0) return;
    int32_t *h = ws.hist + (size_t)b * n;
    const int per = (n + 1023) / 1024;
    const int i0 = (int)threadIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    int sum = 0;
    for (int i = i0; i < i1; i++) sum += h[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int i = i0; i < i1; i++) { const int c = h[i]; h[i] = run; run += c; }
}
__global__ __launch_bounds__(256) void arima_queue_scatter_kernel(const ArWs ws)
{
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ws.counts[AR_QC + b]) return;
    const size_t src = (size_t)b * ws.cap + i;
    const int s = ws.q_series[src];
    const size_t dst = (size_t)b * ws.cap + (size_t)(ws.hist[(size_t)b * ws.nser + s] + ws.q_rank[src]);
    ws.q2_series[dst] = s;
    ws.q2_key[dst] = ws.q_key[src];
}

enum { PH_NEXT = 0, PH_INIT, PH_ITER, PH_E, PH_OC, PH_IC, PH_SHRINK, PH_FINAL };

// item `item` of the launch in fetch order: the queues one after the other, dimension 6 first (longest fits first, so the tail of the
// launch is made of short ones), within a dimension the widest shape class first
__device__ __forceinline__ size_t ar_queue_slot(const ArWs &ws, int item)
{
    int c[AR_NBUCKETS];
#pragma unroll
    for (int b = 0; b < AR_NBUCKETS; b++) c[b] = ws.counts[AR_QC + b];       // (independent loads, then a scan in registers)
    int rem = item, bk = 0;
    bool found = false;
#pragma unroll
    for (int b = AR_NBUCKETS - 1; b >= 0; b--) {
        const bool here = !found && rem < c[b];
        bk = here ? b : bk;
        found = found || here;
        rem = found ? rem : rem - c[b];
    }
    return (size_t)bk * ws.cap + (size_t)rem;
}

// the pass variant of a wave: the cheapest one that covers the orders of its live lanes (wave-uniform branches)
template <int MODE, int M, bool COOP>
__device__ __forceinline__ double ar_css_pass_shaped(const double *wrow, int len, int wave_len, bool live, const ArFac &f, int m, const ArLds &L, const ArOrd &o,
                                                     int32_t *trace_counts)
{
    constexpr int S1 = MODE == 2 ? 0 : 1, S2 = MODE == 2 ? 0 : 2;           // (no seasonal factors at all without a period)
    const int p = live ? o.p : 0, q = live ? o.q : 0, P = live ? o.P : 0, Q = live ? o.Q : 0;
    const bool seas = MODE != 2 && __any(P + Q > 0);
    const bool pq11 = !__any(p > 1 || q > 1), pq12 = !__any(p > 1 || q > 2), pq23 = !__any(p > 2 || q > 3);
    const bool s11 = !__any(P > 1 || Q > 1), s12 = !__any(P > 1);
    int v;                                                                   // 0..5: the rows of ar_shape_class
    if (!seas) v = pq11 ? 0 : (pq23 ? 1 : 5);
    else v = (pq11 && s11) ? 2 : ((pq12 && s12) ? 3 : (pq23 ? 4 : 5));
    if (trace_counts && threadIdx.x == 0) atomicAdd(&trace_counts[v], 1);   // (tune arima_trace = 2) wave-passes per variant
    if (trace_counts && live) atomicAdd(&trace_counts[8 + ar_shape_class(o.p, o.q, o.P, o.Q)], 1);      // ... and live lane-passes per class
    if (v == 0) return ar_css_pass_impl<MODE, M, COOP, 1, 1, 0, 0, false>(wrow, len, wave_len, live, f, m, L);
    if (v == 1) return ar_css_pass_impl<MODE, M, COOP, 2, 3, 0, 0, false>(wrow, len, wave_len, live, f, m, L);
    if (v == 2) return ar_css_pass_impl<MODE, M, COOP, 1, 1, S1, S1, false>(wrow, len, wave_len, live, f, m, L);
    if (v == 3) return ar_css_pass_impl<MODE, M, COOP, 1, 2, S1, S2, false>(wrow, len, wave_len, live, f, m, L);
    if (v == 4) return ar_css_pass_impl<MODE, M, COOP, 2, 3, S2, S2, false>(wrow, len, wave_len, live, f, m, L);
    return ar_css_pass_impl<MODE, M, COOP, AR_MAXP, AR_MAXP, S2, S2, false>(wrow, len, wave_len, live, f, m, L);
}

template <class LT>
__device__ __forceinline__ double ar_trial(const LT &L, int D, int which, int i)
{
    double s = L.sim(0, i);
    for (int k = 1; k < D; k++) s = s + L.sim(k, i);
    const double xb = s / (double)D;
    const double xw = L.sim(D, i);
    const double a = which == 0 ? 2.0 : (which == 1 ? 3.0 : (which == 2 ? 1.5 : 0.5));
    const double b = which == 0 ? 1.0 : (which == 1 ? 2.0 : 0.5);
    return which == 3 ? a * xb + b * xw : a * xb - b * xw;
}

template <class LT, class FT>
__device__ __forceinline__ void ar_accept(const LT &L, FT &F, int D, int which, double fnew)
{
    double xn[AR_MAXDIM];
    for (int i = 0; i < D; i++) xn[i] = ar_trial(L, D, which, i);
    int j = D;
    while (j > 0 && fnew < F.get(j - 1)) {
        F.set(j, F.get(j - 1));
        for (int i = 0; i < D; i++) L.sim(j, i) = L.sim(j - 1, i);
        j--;
    }
    F.set(j, fnew);
    for (int i = 0; i < D; i++) L.sim(j, i) = xn[i];
}

template <class LT, class FT>
__device__ __forceinline__ void ar_sort(const LT &L, FT &F, int D)
{
    for (int k = 1; k <= D; k++) {
        const double fk = F.get(k);
        double tmp[AR_MAXDIM];
        for (int i = 0; i < D; i++) tmp[i] = L.sim(k, i);
        int j = k;
        while (j > 0 && fk < F.get(j - 1)) {
            F.set(j, F.get(j - 1));
            for (int i = 0; i < D; i++) L.sim(j, i) = L.sim(j - 1, i);
            j--;
        }
        F.set(j, fk);
        for (int i = 0; i < D; i++) L.sim(j, i) = tmp[i];
    }
}

// `polish`: the problems are the SELECTED models (one per series, queued by arima_polish_queue_kernel): start at the search's
// estimates (a.xbest) with steps of 0.1, run to convergence, write the estimates and their criterion back (oracle polish_css)
// One instantiation per pass variant (MODE, M as in ar_css_pass_impl: the launch picks it from the period), so that a kernel carries
// the registers of ITS pass only (see AR_MODE3_REVS for the two-waves-per-SIMD experiment).
// the cooperative row loader needs an LDS tile (15-17 KB per wave): on where the ring is not in LDS -- the weekly period (registers),
// no period (8 slots) and the long periods (HBM scratch); the LDS-ring periods keep per-lane row loads (the tile would halve their waves per CU)
constexpr bool ar_fit_coop(int mode) { return mode == 3 || mode == 2 || mode == 5 || mode == 6; }
constexpr int ar_fit_waves(int mode) { return (mode == 3 && AR_MODE3_REVS == 1) ? 2 : 1; }
template <int MODE, int M>
__global__ __launch_bounds__(NM_BLOCK, ar_fit_waves(MODE)) void arima_fit_kernel(const ArimaArgs a, const ArWs ws, const int first, const int total, const int polish)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int m = a.m;
    // MODE 3: both seasonal lags live in registers, MODE 5: in the wave's HBM scratch ring -- LDS holds the simplex only
    ArLds L{lds, MODE == 3 ? 0 : ar_fit_ring_slots(m), (int)threadIdx.x,
            (MODE == 5 || MODE == 6) ? a.long_scratch + (size_t)blockIdx.x * 2 * ar_fit_ring_slots(m) * NM_BLOCK : nullptr, nullptr};
    L.lane_m = MODE == 6;
    int pm = m;                 // the period of this lane's problem (MODE 6: its series' own; a.m is the batch's largest, the ring's size)
    if (ar_fit_coop(MODE)) L.tile = L.smp() + (size_t)(AR_MAXDIM + 1) * AR_MAXDIM * NM_BLOCK;
    ArFs F;
    for (int k = 0; k <= AR_MAXDIM; k++) F.v[k] = 0.0;

    bool fin = false;
    int s = 0, key = 0, len = 0, D = 0;
    ArOrd cur{0, 0, 0, 0, 0};
    const double *wrow = ws.W;
    int ph = PH_NEXT, vi = 0, nm_evals = 0, nm_iters = 1, passes = 0;
    double fxr = 0.0;

    for (;;) {
        // ---- 1. next problem of this lane ------------------------------------------------------------
        if (!fin && ph == PH_NEXT) {
            const int item = first + atomicAdd(&ws.counts[8], 1);           // (items first .. total - 1 of the queue are this launch's)
            if (item >= total) fin = true;
            else {
                const size_t qi = ar_queue_slot(ws, item);
                s = ws.q_series[qi]; key = ws.q_key[qi];
                cur = ar_unkey(key);
                D = ar_dim(cur);
                len = a.wlen[s];
                if (MODE == 6) { pm = a.m_col[s]; L.R = ar_fit_ring_slots(pm); }
                wrow = ws.W + (size_t)s * ws.tw;
                const double wmean = a.wmean[s], wsd = a.wsd[s];
                for (int i = 0; i < D; i++) L.sim(0, i) = polish ? a.xbest[(size_t)i * a.ld + s] : 0.0;
                if (cur.c && !polish) L.sim(0, D - 1) = wmean;
                for (int k = 0; k < D; k++) {
                    for (int i = 0; i < D; i++) L.sim(k + 1, i) = L.sim(0, i);
                    const double step = (cur.c && k == D - 1) ? (wsd > 0.0 ? 0.1 * wsd : 1.0e-4) : (polish ? 0.1 : 0.25);
                    L.sim(k + 1, k) = L.sim(0, k) + step;
                }
                nm_evals = 0; nm_iters = 1; vi = 0; passes = 0;
                if (D == 0) { ph = PH_FINAL; nm_evals = 1; }
                else ph = PH_INIT;
            }
        }
        // ---- 2. trial point of the running fit ------------------------------------------------------
        double x[AR_MAXDIM] = {0, 0, 0, 0, 0, 0};
        if (!fin) {
            if (ph == PH_ITER) {
                const int cap = polish ? ar_polish_cap(D) : ar_search_cap(D);
                bool stop = !(nm_evals < cap && nm_iters < cap);
                if (!stop) {
                    bool small = true;
                    for (int k = 1; k <= D; k++) {
                        for (int i = 0; i < D; i++)
                            if (!(fabs(L.sim(k, i) - L.sim(0, i)) <= 1.0e-4)) small = false;
                        if (!(fabs(F.get(0) - F.get(k)) <= 1.0e-8)) small = false;
                    }
                    stop = small;
                }
                if (stop) ph = PH_FINAL;
            }
            if (ph == PH_INIT) { for (int i = 0; i < D; i++) x[i] = L.sim(vi, i); }
            else if (ph == PH_ITER) { for (int i = 0; i < D; i++) x[i] = ar_trial(L, D, 0, i); }
            else if (ph == PH_E || ph == PH_OC || ph == PH_IC) {
                const int which = ph == PH_E ? 1 : (ph == PH_OC ? 2 : 3);
                for (int i = 0; i < D; i++) x[i] = ar_trial(L, D, which, i);
            } else if (ph == PH_SHRINK) { for (int i = 0; i < D; i++) x[i] = L.sim(1 + vi, i); }
            else { for (int i = 0; i < D; i++) x[i] = L.sim(0, i); }      // PH_FINAL
        }
        ArFac fac;
        ar_factors(cur, pm, x, fac);
        if (__all(fin)) break;

        // ---- 3. one streamed pass -------------------------------------------------------------------
        const int wave_len = ar_wave_max(fin ? 0 : len);
        const double css = ar_css_pass_shaped<MODE, M, ar_fit_coop(MODE)>(wrow, len, wave_len, !fin, fac, pm, L, cur, a.trace >= 2 ? ws.counts + 96 : nullptr);
        if (fin) continue;
        passes++;
        const int nu = len - fac.nc;
        double v = css / (double)nu;
        double f = __builtin_huge_val();
        if (fabs(css) <= 1.7976931348623157e308) {
            if (v < 1.0e-300) v = 1.0e-300;
            f = 0.5 * dm_log(v);
        }

        // ---- 4. consume -----------------------------------------------------------------------------
        if (ph == PH_INIT) {
            F.set(vi, f); vi++; nm_evals++;
            if (vi == D + 1) { ar_sort(L, F, D); ph = PH_ITER; }
        } else if (ph == PH_ITER) {
            fxr = f; nm_evals++;
            if (fxr < F.get(0)) ph = PH_E;
            else if (fxr < F.get(D - 1)) { ar_accept(L, F, D, 0, fxr); nm_iters++; }
            else if (fxr < F.get(D)) ph = PH_OC;
            else ph = PH_IC;
        } else if (ph == PH_E) {
            nm_evals++;
            if (f < fxr) ar_accept(L, F, D, 1, f); else ar_accept(L, F, D, 0, fxr);
            nm_iters++; ph = PH_ITER;
        } else if (ph == PH_OC || ph == PH_IC) {
            nm_evals++;
            const bool ok = (ph == PH_OC) ? (f <= fxr) : (f < F.get(D));
            if (ok) { ar_accept(L, F, D, ph == PH_OC ? 2 : 3, f); nm_iters++; ph = PH_ITER; }
            else {
                for (int k = 1; k <= D; k++)
                    for (int i = 0; i < D; i++) L.sim(k, i) = L.sim(0, i) + 0.5 * (L.sim(k, i) - L.sim(0, i));
                vi = 0; ph = PH_SHRINK;
            }
        } else if (ph == PH_SHRINK) {
            F.set(1 + vi, f); vi++; nm_evals++;
            if (vi == D) { nm_iters++; ar_sort(L, F, D); ph = PH_ITER; }
        } else { // PH_FINAL: information criterion of the fitted candidate -> cache
            double aicc = __builtin_huge_val();
            if (fabs(css) <= 1.7976931348623157e308) {
                const double dn = (double)len, dk = (double)(D + 1);
                aicc = dn * dm_log(v) + 2.0 * dk + 2.0 * dk * (dk + 1.0) / (dn - dk - 1.0);
                if (!(fabs(aicc) <= 1.7976931348623157e308)) aicc = __builtin_huge_val();
            }
            // admissibility (root check): an inadmissible candidate has an infinite criterion; an inadmissible POLISHED point leaves
            // the search's own estimates (and their criterion) in place
            const bool roots_ok = ar_model_roots_ok(cur, pm, fac);
            if (!roots_ok) aicc = __builtin_huge_val();
            if (polish) {
                a.evals[s] += nm_evals;
                if (roots_ok) {
                    a.aicc[s] = aicc;
                    for (int i = 0; i < D; i++) a.xbest[(size_t)i * a.ld + s] = L.sim(0, i);
                }
            } else {
                const size_t ci = (size_t)s * AR_KEYS + key;
                ws.cache_aicc[ci] = aicc;
                ws.cache_evals[ci] = nm_evals;
                for (int i = 0; i < AR_MAXDIM; i++) ws.cache_x[ci * AR_MAXDIM + i] = i < D ? L.sim(0, i) : 0.0;
            }
            atomicAdd(&a.passes[s], passes);
            ph = PH_NEXT;
        }
    }
}

// Speculative variant for short queues (fewer problems than a quarter of the resident lanes): four adjacent lanes share
// one problem and evaluate the four possible trial points of a Nelder-Mead iteration (reflection, expansion, outside and
// inside contraction) in ONE pass; the decision then replays the sequential rules, counting only the evaluations the
// sequential method would have made, so iterates, evaluation counts and the stopping point are those of the sequential
// fit.  An iteration costs one pass instead of ~1.7: the critical path of a sweep's slowest fit shortens accordingly.
// The simplex lives in the group leader's LDS column; all four lanes run the same bookkeeping on it (identical values).
template <int MODE, int M>
__global__ __launch_bounds__(NM_BLOCK, ar_fit_waves(MODE)) void arima_fit_spec_kernel(const ArimaArgs a, const ArWs ws, const int first, const int total, const int polish)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int m = a.m;
    const int lane = threadIdx.x, g = lane & 3, leader = lane & ~3;
    ArLds L{lds, MODE == 3 ? 0 : ar_fit_ring_slots(m), leader,
            (MODE == 5 || MODE == 6) ? a.long_scratch + (size_t)blockIdx.x * 2 * ar_fit_ring_slots(m) * NM_BLOCK : nullptr, nullptr};
    L.lane_m = MODE == 6;
    int pm = m;
    // (per-lane row loads here: the four lanes of a problem read the same row, 16 distinct rows per wave -- measured 173 ms with them
    //  against 183 ms with the cooperative loader over the nine four-lane launches of the M5 batch; the sequential kernel 163 -> 132 ms)
    ArFs F;
    for (int k = 0; k <= AR_MAXDIM; k++) F.v[k] = 0.0;
    bool fin = false;
    int s = 0, key = 0, len = 0, D = 0;
    ArOrd cur{0, 0, 0, 0, 0};
    const double *wrow = ws.W;
    int ph = PH_NEXT, vi = 0, nm_evals = 0, nm_iters = 1, passes = 0;

    for (;;) {
        if (!fin && ph == PH_NEXT) {
            int item = 0;
            if (g == 0) item = first + atomicAdd(&ws.counts[8], 1);
            item = __shfl(item, leader);
            if (item >= total) fin = true;
            else {
                const size_t qi = ar_queue_slot(ws, item);
                s = ws.q_series[qi]; key = ws.q_key[qi];
                cur = ar_unkey(key);
                D = ar_dim(cur);
                len = a.wlen[s];
                if (MODE == 6) { pm = a.m_col[s]; L.R = ar_fit_ring_slots(pm); }
                wrow = ws.W + (size_t)s * ws.tw;
                const double wmean = a.wmean[s], wsd = a.wsd[s];
                for (int i = 0; i < D; i++) L.sim(0, i) = polish ? a.xbest[(size_t)i * a.ld + s] : 0.0;
                if (cur.c && !polish) L.sim(0, D - 1) = wmean;
                for (int k = 0; k < D; k++) {
                    for (int i = 0; i < D; i++) L.sim(k + 1, i) = L.sim(0, i);
                    const double step = (cur.c && k == D - 1) ? (wsd > 0.0 ? 0.1 * wsd : 1.0e-4) : (polish ? 0.1 : 0.25);
                    L.sim(k + 1, k) = L.sim(0, k) + step;
                }
                nm_evals = 0; nm_iters = 1; vi = 0; passes = 0;
                if (D == 0) { ph = PH_FINAL; nm_evals = 1; }
                else ph = PH_INIT;
            }
        }
        // ---- trial point of this lane ---------------------------------------------------------------
        double x[AR_MAXDIM] = {0, 0, 0, 0, 0, 0};
        bool mine = false;                      // this lane's evaluation is a real one
        if (!fin) {
            if (ph == PH_ITER) {
                const int cap = polish ? ar_polish_cap(D) : ar_search_cap(D);
                bool stop = !(nm_evals < cap && nm_iters < cap);
                if (!stop) {
                    bool small = true;
                    for (int k = 1; k <= D; k++) {
                        for (int i = 0; i < D; i++)
                            if (!(fabs(L.sim(k, i) - L.sim(0, i)) <= 1.0e-4)) small = false;
                        if (!(fabs(F.get(0) - F.get(k)) <= 1.0e-8)) small = false;
                    }
                    stop = small;
                }
                if (stop) ph = PH_FINAL;
            }
            if (ph == PH_INIT) { mine = vi + g <= D; if (mine) for (int i = 0; i < D; i++) x[i] = L.sim(vi + g, i); }
            else if (ph == PH_ITER) { mine = true; for (int i = 0; i < D; i++) x[i] = ar_trial(L, D, g, i); }
            else if (ph == PH_SHRINK) { mine = 1 + vi + g <= D; if (mine) for (int i = 0; i < D; i++) x[i] = L.sim(1 + vi + g, i); }
            else { mine = true; for (int i = 0; i < D; i++) x[i] = L.sim(0, i); }      // PH_FINAL
        }
        ArFac fac;
        ar_factors(cur, pm, x, fac);
        if (__all(fin)) break;

        const int wave_len = ar_wave_max(fin ? 0 : len);
        const double css = ar_css_pass_shaped<MODE, M, false>(wrow, len, wave_len, !fin, fac, pm, L, cur, a.trace >= 2 ? ws.counts + 112 : nullptr);
        if (fin) continue;
        passes++;
        const int nu = len - fac.nc;
        double v = css / (double)nu;
        double f = __builtin_huge_val();
        if (fabs(css) <= 1.7976931348623157e308) {
            if (v < 1.0e-300) v = 1.0e-300;
            f = 0.5 * dm_log(v);
        }

        if (ph == PH_INIT) {
            for (int j = 0; j < 4; j++) { const double fj = __shfl(f, leader + j); if (vi + j <= D) F.set(vi + j, fj); }
            const int cnt = (D + 1 - vi) < 4 ? (D + 1 - vi) : 4;
            vi += cnt; nm_evals += cnt;
            if (vi == D + 1) { ar_sort(L, F, D); ph = PH_ITER; }
        } else if (ph == PH_ITER) {
            const double fr = __shfl(f, leader), fe = __shfl(f, leader + 1), foc = __shfl(f, leader + 2), fic = __shfl(f, leader + 3);
            nm_evals++;
            bool shrink = false;
            if (fr < F.get(0)) { nm_evals++; if (fe < fr) ar_accept(L, F, D, 1, fe); else ar_accept(L, F, D, 0, fr); }
            else if (fr < F.get(D - 1)) ar_accept(L, F, D, 0, fr);
            else if (fr < F.get(D)) { nm_evals++; if (foc <= fr) ar_accept(L, F, D, 2, foc); else shrink = true; }
            else { nm_evals++; if (fic < F.get(D)) ar_accept(L, F, D, 3, fic); else shrink = true; }
            if (!shrink) nm_iters++;
            else {
                for (int k = 1; k <= D; k++)
                    for (int i = 0; i < D; i++) L.sim(k, i) = L.sim(0, i) + 0.5 * (L.sim(k, i) - L.sim(0, i));
                vi = 0; ph = PH_SHRINK;
            }
        } else if (ph == PH_SHRINK) {
            for (int j = 0; j < 4; j++) { const double fj = __shfl(f, leader + j); if (1 + vi + j <= D) F.set(1 + vi + j, fj); }
            const int cnt = (D - vi) < 4 ? (D - vi) : 4;
            vi += cnt; nm_evals += cnt;
            if (vi == D) { nm_iters++; ar_sort(L, F, D); ph = PH_ITER; }
        } else { // PH_FINAL
            double aicc = __builtin_huge_val();
            if (fabs(css) <= 1.7976931348623157e308) {
                const double dn = (double)len, dk = (double)(D + 1);
                aicc = dn * dm_log(v) + 2.0 * dk + 2.0 * dk * (dk + 1.0) / (dn - dk - 1.0);
                if (!(fabs(aicc) <= 1.7976931348623157e308)) aicc = __builtin_huge_val();
            }
            const bool roots_ok = ar_model_roots_ok(cur, pm, fac);
            if (!roots_ok) aicc = __builtin_huge_val();
            if (g == 0) {
                if (polish) {
                    a.evals[s] += nm_evals;
                    if (roots_ok) {
                        a.aicc[s] = aicc;
                        for (int i = 0; i < D; i++) a.xbest[(size_t)i * a.ld + s] = L.sim(0, i);
                    }
                } else {
                    const size_t ci = (size_t)s * AR_KEYS + key;
                    ws.cache_aicc[ci] = aicc;
                    ws.cache_evals[ci] = nm_evals;
                    for (int i = 0; i < AR_MAXDIM; i++) ws.cache_x[ci * AR_MAXDIM + i] = i < D ? L.sim(0, i) : 0.0;
                }
                atomicAdd(&a.passes[s], passes);
            }
            ph = PH_NEXT;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forecast of the differenced series with the selected model, then integration
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NM_BLOCK) void arima_forecast_kernel(const ArimaArgs a, const ArWs ws)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x;
    const int s = blockIdx.x * NM_BLOCK + lane;
    const bool valid = s < a.n_series;
    const int len = valid ? a.wlen[s] : 0;
    const bool live = valid && len >= 3 && a.status[s] == FIT_OK;
    // (merged batch of long periods: the lane's own period for every index, the batch's largest, a.m, for every region size)
    const int m = (a.m_col && valid) ? a.m_col[s] : a.m;
    // long period: ring and polynomials in the workgroup's HBM scratch
    double *const wg_scratch = a.long_scratch ? a.long_scratch + (size_t)blockIdx.x * ar_fc_scratch_doubles(a.m) : nullptr;
    ArLds L{lds, ar_ring_slots(m), (int)threadIdx.x, wg_scratch, nullptr};
    L.lane_m = a.m_col != nullptr;
    ArPolyLds PL{(wg_scratch ? wg_scratch : lds) + (size_t)2 * ar_ring_slots(a.m) * NM_BLOCK, AR_MAXP + AR_MAXSP * a.m + 1};
    const double *w = ws.W + (size_t)(valid ? s : 0) * ws.tw;
    const size_t ld = a.ld;
    const int wave_len = ar_wave_max(live ? len : 0);
    if (wave_len == 0) return;
    ArOrd o{0, 0, 0, 0, 0};
    double x[AR_MAXDIM] = {0, 0, 0, 0, 0, 0};
    if (live) {
        o.p = a.order[(size_t)0 * ld + s]; o.q = a.order[(size_t)1 * ld + s]; o.P = a.order[(size_t)2 * ld + s];
        o.Q = a.order[(size_t)3 * ld + s]; o.c = a.order[(size_t)4 * ld + s];
        for (int i = 0; i < AR_MAXDIM; i++) x[i] = a.xbest[(size_t)i * ld + s];
    }
    int La = 0, Lb = 0;
    double mu = 0.0;
    ArFac fac;
    ar_factors(o, m, x, fac);
    if (live) ar_build_poly(o, m, x, PL, La, Lb, mu);
    (void)ar_css_pass(w, len, wave_len, live, fac, m, L);
    if (!live) return;
    const int n = a.len[s], h = a.h;
    const int d = a.d[s], Dd = a.D[s];
    double *out = a.yhat + (size_t)s * h;
    for (int j = 0; j < h; j++) {
        const int t = len + j;
        double acc = mu;
        for (int k = 1; k <= La; k++)
            if (t - k >= 0) {
                const double wv = (t - k < len) ? w[t - k] : out[t - k - len];
                acc = fma(PL.a(k), wv - mu, acc);
            }
        for (int k = 1; k <= Lb; k++)
            if (t - k >= 0 && t - k < len) acc = fma(PL.b(k), L.e_at(t - k), acc);
        out[j] = acc;
    }
    double last_d0 = a.last_d0[s], last_d1 = a.last_d1[s];
    const double *y = a.y + s;
    for (int j = 0; j < h; j++) {
        double val = out[j];
        if (d == 2) { last_d1 = last_d1 + val; val = last_d1; }
        if (d >= 1) { last_d0 = last_d0 + val; val = last_d0; }
        if (Dd) val = val + ((j - m < 0) ? y[(size_t)(n + j - m) * ld] : out[j - m]);
        out[j] = val;
    }
    a.model_code[s] = 1000000 + o.p * 100000 + d * 10000 + o.q * 1000 + o.P * 100 + Dd * 10 + o.Q;
}


// ------------------------------------------------------------------------------------------------
// exact Gaussian likelihood refit of the selected model (oracle/arima.c ml_eval / refit_ml)
// ------------------------------------------------------------------------------------------------
// "CSS for the search, exact likelihood for the final estimates."  One lane per series; every objective evaluation is the
// Kalman filter of the Harvey state space (state dimension r = max(p + m P, q + m Q + 1) <= AR_ML_MAX_R) run through the
// Chandrasekhar recursions, so the state covariance is never formed: O(r) work per step, and the four r-vectors (expanded AR
// coefficients, gain K, increment L, state) stay in VGPRs -- the loops over the state are unrolled to the compile-time
// bound RM of the launch's period (coefficients past a lane's own r are exact zeros), no LDS or HBM traffic in the pass
// besides the lane's row of W.  The stationary start (F_1 and K_1 from the ARMA autocovariances: inverse Levinson recursion on
// the AR polynomial, then the MA filter) is O(r^2) per evaluation in a small LDS scratch.  Same sequence of IEEE operations
// as the oracle, so the estimates are bit-identical.
constexpr int AR_ML_MAX_R = 32;
constexpr int AR_ML_NM_CAP = 50;        // the refit starts at the CSS optimum: 50 x dim evaluations / iterations at most (oracle: ARIMA_ML_NM_CAP)
// LDS of the refit kernel per wave (lane-minor), in doubles per lane: rm + (l1 - 1) + (2 rm - 1), rm = the state-dimension class
// of the launch's period, l1 = AR_MAXP + AR_MAXSP m + 1 -- 78 at m = 7 (39 KB per wave: FOUR waves per CU, one per SIMD; the
// first version kept the simplex, both polynomials with their unused slot 0 and five scratch vectors there: 187, ONE wave
// per CU).  The Nelder-Mead simplex lives in a global scratch (a few dozen L2 hits per evaluation against a 1,900-step pass).
__host__ __device__ inline int ar_ml_l1(int m) { return AR_MAXP + AR_MAXSP * m + 1; }
__host__ __device__ inline int ar_ml_rm(int m) { const int l1 = ar_ml_l1(m); return l1 <= 8 ? 8 : (l1 <= 12 ? 12 : (l1 <= 16 ? 16 : (l1 <= 20 ? 20 : AR_ML_MAX_R))); }
__host__ __device__ inline size_t ar_ml_lds_doubles(int m) { return (size_t)(3 * ar_ml_rm(m) + ar_ml_l1(m) - 2); }

// s1 [rm]: the AR coefficients a_1.. (slot i = a_{i+1}), then the working polynomial of the Levinson recursions in place (its
//          slot j - 1 keeps the reflection coefficient of order j), then gx;  cb [l1 - 1]: the MA coefficients b_1..;
// gu [2 rm - 1]: autocovariances of the AR part, then psi.
struct ArMlLds {
    double *base; int rm, l1;
    __device__ double &s1(int i) const { return base[(size_t)i * NM_BLOCK + threadIdx.x]; }
    __device__ double &cb(int i) const { return base[(size_t)(rm + i - 1) * NM_BLOCK + threadIdx.x]; }      // i >= 1
    __device__ double &gu(int i) const { return base[(size_t)(rm + l1 - 1 + i) * NM_BLOCK + threadIdx.x]; }
    __device__ double &al(int i) const { return s1(i); }
    __device__ double &gx(int i) const { return s1(i); }
    __device__ double &psi(int i) const { return gu(i); }
};

// the expanded lag polynomials of ar_build_poly (same operations), without the unused slots 0
__device__ __forceinline__ void ar_build_poly_ml(const ArOrd &o, int m, const double *x, const ArMlLds &Q, int &La, int &Lb, double &mu)
{
    double phi[AR_MAXP], th[AR_MAXP], Phi[AR_MAXSP], Th[AR_MAXSP];
    int k = 0;
    ar_pacf(x + k, o.p, phi); k += o.p;
    ar_pacf(x + k, o.q, th); k += o.q;
    ar_pacf(x + k, o.P, Phi); k += o.P;
    ar_pacf(x + k, o.Q, Th); k += o.Q;
    mu = o.c ? x[k] : 0.0;
    La = o.p + m * o.P;
    Lb = o.q + m * o.Q;
    for (int i = 1; i <= La; i++) Q.s1(i - 1) = 0.0;
    for (int i = 1; i <= o.p; i++) Q.s1(i - 1) = phi[i - 1];
    for (int I = 1; I <= o.P; I++) {
        Q.s1(m * I - 1) = Q.s1(m * I - 1) + Phi[I - 1];
        for (int i = 1; i <= o.p; i++) Q.s1(m * I + i - 1) = Q.s1(m * I + i - 1) - phi[i - 1] * Phi[I - 1];
    }
    for (int i = 1; i <= Lb; i++) Q.cb(i) = 0.0;
    for (int i = 1; i <= o.q; i++) Q.cb(i) = th[i - 1];
    for (int I = 1; I <= o.Q; I++) {
        Q.cb(m * I) = Q.cb(m * I) + Th[I - 1];
        for (int i = 1; i <= o.q; i++) Q.cb(m * I + i) = Q.cb(m * I + i) - th[i - 1] * Th[I - 1];
    }
    for (int i = 1; i <= Lb; i++) Q.cb(i) = -Q.cb(i);
}

// Two adjacent lanes (2k, 2k + 1) work on ONE series in the refit: the even lane holds entries [0, RM/2) of the filter's
// r-vectors, the odd lane entries [RM/2, RM).  Values cross the pair through DPP quad permutes (full-rate VALU, no LDS).
__device__ __forceinline__ int ar_pair_lo(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xA0, 0xF, 0xF, false); }   // quad_perm [0,0,2,2]
__device__ __forceinline__ int ar_pair_hi(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xF5, 0xF, 0xF, false); }   // quad_perm [1,1,3,3]
__device__ __forceinline__ double ar_pair_lo(double v) { return __hiloint2double(ar_pair_lo(__double2hiint(v)), ar_pair_lo(__double2loint(v))); }
__device__ __forceinline__ double ar_pair_hi(double v) { return __hiloint2double(ar_pair_hi(__double2hiint(v)), ar_pair_hi(__double2loint(v))); }

// 0.5 (log(ssq / n) + sumlog / n) of the model whose expanded polynomials ar_build_poly_ml left in Q; +inf when not computable.
// Called by the WHOLE wave (lanes without a problem pass act = false): the filter loop runs to the longest row of the wave
// with wave-uniform control flow only -- a lane that has reached its steady state, the end of its row or a non-computable
// point keeps stepping with its updates switched off by selects (exact no-ops), so a step is straight-line code without
// per-lane branches (the first version branched per lane and step: 390 instructions per step, 88 of them register copies
// at the joins, 2,100 cycles; this one ~170).
template <int RM>
__device__ double ar_ml_eval(bool act, int La, int Lb, double mu, const double *w, int n, const ArMlLds &Q)
{
    int r = La > Lb + 1 ? La : Lb + 1;
    bool bad = false;
    if (!act || r > AR_ML_MAX_R || r > RM || n < 1) { bad = true; La = 0; Lb = 0; r = 1; n = 0; mu = 0.0; }
    auto Cc = [&](int i) { return (i == 0) ? 1.0 : ((i <= Lb) ? Q.cb(i) : 0.0); };
    // the AR coefficients go to VGPRs first (entries at and above the lane's own La are exact zeros): their LDS slots become
    // the working polynomial
    double A[RM];
#pragma unroll
    for (int i = 0; i < RM; i++) A[i] = (i < La) ? Q.s1(i) : 0.0;
    auto Ad = [&](int i) {                                   // run-time index: select chain
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < RM; q++) v = (q == i) ? A[q] : v;
        return v;
    };
    // ---- stationary start (oracle/arima.c ml_eval): autocovariances by the inverse Levinson recursion, O(r^2) ----
    // step-down, in place: new al(p) and al(j - 2 - p) need each other's old values only
    double E0 = 1.0;
    for (int j = La; j >= 1; j--) {
        const double kj = Q.al(j - 1);                       // stays in slot j - 1: the reflection coefficient of order j
        const double den = 1.0 - kj * kj;
        if (!(den > 0.0)) { bad = true; break; }             // not stationary: the point is rejected (the rest runs on garbage)
        E0 = E0 / den;
        for (int p = 0; 2 * p <= j - 2; p++) {
            const int q = j - 2 - p;
            const double op = Q.al(p), oq = Q.al(q);
            Q.al(p) = fma(kj, oq, op) / den;
            if (q != p) Q.al(q) = fma(kj, op, oq) / den;
        }
    }
    Q.gu(0) = E0;
    {
        double Ej = E0;
        for (int j = 1; j <= La; j++) {
            const double kj = Q.al(j - 1);
            double acc = kj * Ej;
            for (int i = 1; i <= j - 1; i++) acc = fma(Q.al(i - 1), Q.gu(j - i), acc);
            Q.gu(j) = acc;
            for (int p = 0; 2 * p <= j - 2; p++) {
                const int q = j - 2 - p;
                const double op = Q.al(p), oq = Q.al(q);
                Q.al(p) = fma(-kj, oq, op);
                if (q != p) Q.al(q) = fma(-kj, op, oq);
            }
            Ej = Ej * (1.0 - kj * kj);
        }
    }
    const int G = r - 1 + Lb;                                // the largest lag gx needs
    for (int k = La + 1; k <= G; k++) {
        double acc = 0.0;
#pragma unroll
        for (int l = 1; l <= RM; l++)
            if (l <= La) acc = fma(A[l - 1], Q.gu(k - l), acc);
        Q.gu(k) = acc;
    }
    // MA autocovariances in VGPRs (compile-time lag, run-time sum), then gx over the working polynomial's slots
    double BB[RM];
#pragma unroll
    for (int k = 0; k < RM; k++) {
        double acc = 0.0;
        for (int i = 0; i + k <= Lb; i++) acc = fma(Cc(i), Cc(i + k), acc);
        BB[k] = acc;
    }
    for (int k = 0; k < r; k++) {
        double acc = BB[0] * Q.gu(k);
#pragma unroll
        for (int d = 1; d < RM; d++)
            if (d <= Lb) {
                const int km = k - d < 0 ? d - k : k - d;
                acc = fma(BB[d], Q.gu(k + d) + Q.gu(km), acc);
            }
        Q.gx(k) = acc;
    }
    double F = Q.gx(0);
    for (int k = 0; k < r; k++) {                            // gu is free now: psi takes its slots
        double acc = Cc(k);
#pragma unroll
        for (int l = 1; l < RM; l++)
            if (l <= k) acc = fma(A[l - 1], Q.psi(k - l), acc);
        Q.psi(k) = acc;
    }
    if (!(F > 0.0) || !(F <= 1.7976931348623157e308)) { bad = true; F = 1.0; }
    // K_1 of the whole state, then this lane's half of the four r-vectors (the pair's other lane runs the same code on the
    // same values up to here; entries at and above the lane's own r are exact zeros)
    constexpr int H = RM / 2;
    const bool hi = (threadIdx.x & 1) != 0;
    double Ah[H], Kh[H], Lh[H + 1], sth[H + 1];
    {
        double K[RM];
#pragma unroll
        for (int i = 0; i < RM; i++) K[i] = 0.0;
        double gnext = 0.0;
        for (int i = r - 1; i >= 0; i--) {
            const double kv = fma(Ad(i), F, gnext);
#pragma unroll
            for (int q = 0; q < RM; q++) K[q] = (q == i) ? kv : K[q];
            if (i == 0) break;                               // nothing follows row 0
            double acc = 0.0;
#pragma unroll
            for (int l = 1; l <= RM; l++)
                if (l >= i + 1 && l <= r) acc = fma(A[l - 1], Q.gx(l - i), acc);
            for (int l = i; l <= r - 1; l++) acc = fma(Cc(l), Q.psi(l - i), acc);
            gnext = acc;
        }
#pragma unroll
        for (int i = 0; i < H; i++) {
            Ah[i] = hi ? A[H + i] : A[i];
            Kh[i] = hi ? K[H + i] : K[i];
            Lh[i] = Kh[i];
            sth[i] = 0.0;
        }
        Lh[H] = 0.0; sth[H] = 0.0;
    }
    if (bad) n = 0;
    // the filter: one division per step while F still moves anywhere in the wave, none once every lane is in its steady state.
    // The pair's row of W comes in blocks of 4 values (128-bit loads, the next block requested before the current one is
    // consumed; a row has tw >= t_max + spare elements, so reading on to the longest row of the wave stays inside the lane's own).
    double rF = 1.0 / F;
    double M = -rF;
    double ssq = 0.0, mant = 1.0;
    int eacc = 0, t = 0;
    bool steady = false;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    constexpr int WB = 4;
    double wc[WB], wn[WB];
    auto load_w = [&](double (&buf)[WB], int t0) __attribute__((always_inline)) {
        const d2_t *p = reinterpret_cast<const d2_t *>(w + t0);
#pragma unroll
        for (int j = 0; j < WB / 2; j++) { const d2_t v2 = p[j]; buf[2 * j] = v2.x; buf[2 * j + 1] = v2.y; }
    };
    int n_max = n;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl_xor(n_max, o); n_max = other > n_max ? other : n_max; }
    load_w(wc, 0);
    for (int base = 0; base < n_max; base += WB) {
        load_w(wn, base + WB);
        if (__any(!steady && base < n)) {
#pragma unroll
            for (int j = 0; j < WB; j++) {
                const bool inr = base + j < n;
                const bool live = inr && !steady;            // this series' F, K, L still move
                // entry 0 of the state and of L (even lane) for both; entry H (the odd lane's first) closes the even lane's shift
                const double s0 = sth[0], l0 = Lh[0];
                const double a0 = ar_pair_lo(s0), c = ar_pair_lo(l0);
                const double sH = ar_pair_hi(s0), lH = ar_pair_hi(l0);
                sth[H] = hi ? 0.0 : sH;
                Lh[H] = hi ? 0.0 : lH;
                const double v = inr ? (wc[j] - mu) - a0 : 0.0;
                const double vf = v * rF;
                ssq = fma(v, vf, ssq);
                int ex;
                const double mm = frexp(mant * F, &ex);
                mant = live ? mm : mant;
                eacc += live ? ex : 0;
#pragma unroll
                for (int i = 0; i < H; i++) sth[i] = fma(Kh[i], vf, fma(Ah[i], a0, sth[i + 1]));
                const double cm = live ? c * M : 0.0;        // 0: K and F stay as they are, bit for bit
                const double dF = c * cm;
                const double Fn = live ? F + dF : F;
                bad = bad | (live & !(Fn > 0.0));
                const double cf = c * rF;
#pragma unroll
                for (int i = 0; i < H; i++) {
                    const double tl = fma(Ah[i], c, Lh[i + 1]);
                    const double kold = Kh[i];
                    Kh[i] = fma(tl, cm, kold);                // cm = 0 off the transient: K stays (tl is finite there: L decays)
                    Lh[i] = fma(-kold, cf, tl);
                }
                const double rFn = 1.0 / Fn;
                const double Mn = (M * F) * rFn;
                M = live ? Mn : M;
                rF = live ? rFn : rF;
                F = Fn;
                t = live ? base + j + 1 : t;                 // steps filtered in the transient so far
                if (j == WB - 1) {                           // every fourth step: can any entry of L still move F?
                    // sum of squares in the oracle's order: entries [0, H) in the even lane, the odd lane carries on from there
                    double lsq = 0.0;
#pragma unroll
                    for (int i = 0; i < H; i++) lsq = fma(Lh[i], Lh[i], lsq);
                    double lsq2 = ar_pair_lo(lsq);
#pragma unroll
                    for (int i = 0; i < H; i++) lsq2 = fma(Lh[i], Lh[i], lsq2);
                    lsq = ar_pair_hi(lsq2);
                    steady = steady | (live & !(lsq * fabs(M) > 1.0e-12 * F));
                }
                __builtin_amdgcn_sched_barrier(0);           // one step at a time: interleaving steps only adds live registers
            }
        } else {
#pragma unroll
            for (int j = 0; j < WB; j++) {
                const double s0 = sth[0];
                const double a0 = ar_pair_lo(s0), sH = ar_pair_hi(s0);
                sth[H] = hi ? 0.0 : sH;
                const double v = (base + j < n) ? (wc[j] - mu) - a0 : 0.0;
                const double vf = v * rF;
                ssq = fma(v, vf, ssq);
#pragma unroll
                for (int i = 0; i < H; i++) sth[i] = fma(Kh[i], vf, fma(Ah[i], a0, sth[i + 1]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int j = 0; j < WB; j++) wc[j] = wn[j];
    }
    if (bad || !(fabs(ssq) <= 1.7976931348623157e308) || !(ssq >= 0.0)) return __builtin_huge_val();
    double s2 = ssq / (double)n;
    if (s2 < 1.0e-300) s2 = 1.0e-300;
    double sumlog = dm_log(mant) + (double)eacc * 0.693147180559945309417232121458;
    const int n_steady = steady ? n - t : 0;
    if (n_steady > 0) sumlog = fma((double)n_steady, dm_log(F), sumlog);
    const double f = 0.5 * (dm_log(s2) + sumlog / (double)n);
    return (f == f) ? f : __builtin_huge_val();
}

enum { RP_F0 = 0, RP_INIT, RP_ITER, RP_E, RP_OC, RP_IC, RP_SHRINK, RP_DONE };

// the refit's Nelder-Mead state per lane in a global scratch (lane-minor per wave, like the LDS layout of the fit kernels):
// 42 simplex coordinates, 7 function values, the reflection value, the value at the start, the step of the constant --
// everything the filter loop does not need stays out of its register budget (the r-vectors alone take 164 VGPRs at r = 20)
constexpr int AR_ML_CTX = (AR_MAXDIM + 1) * AR_MAXDIM + (AR_MAXDIM + 1) + 3;
struct ArSimG {
    double *base; int col;
    __device__ double &sim(int k, int i) const { return base[(size_t)(k * AR_MAXDIM + i) * NM_BLOCK + col]; }
    __device__ double &fv(int k) const { return base[(size_t)((AR_MAXDIM + 1) * AR_MAXDIM + k) * NM_BLOCK + col]; }
    __device__ double &fxr() const { return fv(AR_MAXDIM + 1); }
    __device__ double &f0() const { return fv(AR_MAXDIM + 2); }
    __device__ double &wsd() const { return fv(AR_MAXDIM + 3); }
    __device__ double get(int k) const { return fv(k); }
    __device__ void set(int k, double x) const { fv(k) = x; }
};

// Series whose state dimension r lies in (r_lo, RM] are refitted by the RM instantiation (the unrolled loops cost RM whatever
// the lane's own r, so the common small models -- r = 9 for (0,1,1)(1,0,1)[7] -- run in a narrower kernel); `cursor` is the
// queue cursor of this launch in ws.counts.
// `budget` > 0: a series that has used this many evaluations when an iteration starts is PARKED -- its simplex, function values and
// counters go to ws.ml_park, its index to ws.ml_park_list -- and the second launch (ar_refit_body_spec) finishes it with four
// trial points per pass.
template <int RM>
__device__ __noinline__ int ar_refit_body(const ArimaArgs &a, const ArWs &ws, const int r_lo, const int cursor, double *lds, const int budget)
{
    const int lane = threadIdx.x;
    const bool even = (lane & 1) == 0;                     // the pair's lanes run the same Nelder-Mead on the same values; the even one talks to memory
    const int m = a.m;
    const size_t ld = a.ld;
    static_assert(AR_ML_CTX <= 64, "refit context slots");
    const ArSimG L{ws.ml_sim + (size_t)blockIdx.x * (size_t)64 * NM_BLOCK, lane};
    const ArSimG &F = L;                                   // the function values live next to the simplex
    ArMlLds Q{lds, RM, ar_ml_l1(m > AR_LDS_PERIOD ? 1 : m)};        // (long period: only models without seasonal terms get here)

    // persistent lane pairs: a pair that has finished its series takes the next one from the cursor (evaluation counts differ
    // several-fold between series, so a wave tied to 32 fixed series would idle most of its lanes)
    int s = 0, len = 0, D = 0;
    ArOrd o{0, 0, 0, 0, 0};
    const double *w = ws.W;
    bool fin = false;
    int ph = RP_DONE, vi = 0, nm_evals = 0, nm_iters = 1, evals = 0;

    int loops = 0;
    for (;; loops++) {
        // ---- next series of this lane ----
        for (int attempt = 0; attempt < 4 && !fin && ph == RP_DONE; attempt++) {
            if (evals > 0) { if (even) { a.passes[s] += evals; a.evals[s] += evals; } evals = 0; }
            int item = 0;
            if (even) item = atomicAdd(&ws.counts[cursor], 1);
            item = ar_pair_lo(item);
            if (item >= a.n_series) { fin = true; break; }
            s = item;
            len = a.wlen[s];
            if (!(len >= 3 && a.status[s] == FIT_OK)) continue;
            o.p = a.order[(size_t)0 * ld + s]; o.q = a.order[(size_t)1 * ld + s]; o.P = a.order[(size_t)2 * ld + s];
            o.Q = a.order[(size_t)3 * ld + s]; o.c = a.order[(size_t)4 * ld + s];
            D = o.p + o.q + o.P + o.Q + o.c;
            if (D == 0) continue;
            {
                const int la = o.p + m * o.P, lb1 = o.q + m * o.Q + 1;
                const int rr = la > lb1 ? la : lb1;
                if (rr <= r_lo || rr > RM) continue;              // another instantiation's series (or beyond AR_ML_MAX_R: CSS estimates stay)
                if (m > AR_LDS_PERIOD && (o.P || o.Q)) continue;
            }
            w = ws.W + (size_t)s * ws.tw;
            L.wsd() = a.wsd[s];
            ph = RP_F0; vi = 0; nm_evals = 0; nm_iters = 1;
        }
        if (__all(fin && ph == RP_DONE)) break;
        double x[AR_MAXDIM] = {0, 0, 0, 0, 0, 0};
        if (ph != RP_DONE) {
            if (ph == RP_ITER) {
                bool stop = !(nm_evals < AR_ML_NM_CAP * D && nm_iters < AR_ML_NM_CAP * D);
                if (!stop) {
                    bool small = true;
                    for (int k = 1; k <= D; k++) {
                        for (int i = 0; i < D; i++)
                            if (!(fabs(L.sim(k, i) - L.sim(0, i)) <= 1.0e-4)) small = false;
                        if (!(fabs(F.get(0) - F.get(k)) <= 1.0e-8)) small = false;
                    }
                    stop = small;
                }
                if (stop) {
                    // the exact-likelihood estimates replace the CSS ones when they are at least as good as the start
                    const double fb = F.get(0);
                    if (even && fabs(fb) <= 1.7976931348623157e308 && fb <= L.f0() && ar_simplex_best_roots_ok(o, m, D, L))
                        for (int i = 0; i < D; i++) a.xbest[(size_t)i * ld + s] = L.sim(0, i);
                    ph = RP_DONE;
                } else if (budget > 0 && nm_evals >= budget) {
                    if (even) {
                        double *pk = ws.ml_park + (size_t)s * 64;
                        for (int c = 0; c < AR_ML_CTX; c++) pk[c] = L.base[(size_t)c * NM_BLOCK + L.col];
                        pk[60] = (double)nm_evals; pk[61] = (double)nm_iters;
                        ws.ml_park_list[atomicAdd(&ws.counts[26], 1)] = s;
                    }
                    ph = RP_DONE;
                }
            }
            if (ph == RP_F0) { for (int i = 0; i < D; i++) x[i] = a.xbest[(size_t)i * ld + s]; }      // the CSS optimum
            else if (ph == RP_INIT) { for (int i = 0; i < D; i++) x[i] = L.sim(vi, i); }
            else if (ph == RP_ITER) { for (int i = 0; i < D; i++) x[i] = ar_trial(L, D, 0, i); }
            else if (ph == RP_E || ph == RP_OC || ph == RP_IC) {
                const int which = ph == RP_E ? 1 : (ph == RP_OC ? 2 : 3);
                for (int i = 0; i < D; i++) x[i] = ar_trial(L, D, which, i);
            } else if (ph == RP_SHRINK) { for (int i = 0; i < D; i++) x[i] = L.sim(1 + vi, i); }
        }
        int La = 0, Lb = 0;
        double mu = 0.0;
        if (ph != RP_DONE) { ar_build_poly_ml(o, m, x, Q, La, Lb, mu); evals++; }
        const double f = ar_ml_eval<RM>(ph != RP_DONE, La, Lb, mu, w, len, Q);     // the whole wave: wave-uniform filter loop
        if (ph == RP_F0) {
            L.f0() = f;
            if (!(fabs(f) <= 1.7976931348623157e308)) ph = RP_DONE;       // not computable at the start: the CSS estimates stay
            else {
                const double wsd = L.wsd();
                for (int i = 0; i < D; i++) L.sim(0, i) = a.xbest[(size_t)i * ld + s];
                for (int k = 0; k < D; k++) {
                    for (int i = 0; i < D; i++) L.sim(k + 1, i) = L.sim(0, i);
                    const double step = (o.c && k == D - 1) ? (wsd > 0.0 ? 0.1 * wsd : 1.0e-4) : 0.1;
                    L.sim(k + 1, k) = L.sim(0, k) + step;
                }
                nm_evals = 0; nm_iters = 1; vi = 0;
                ph = RP_INIT;
            }
        } else if (ph == RP_INIT) {
            F.set(vi, f); vi++; nm_evals++;
            if (vi == D + 1) { ar_sort(L, F, D); ph = RP_ITER; }
        } else if (ph == RP_ITER) {
            const double fxr = f;
            L.fxr() = fxr; nm_evals++;
            if (fxr < F.get(0)) ph = RP_E;
            else if (fxr < F.get(D - 1)) { ar_accept(L, F, D, 0, fxr); nm_iters++; }
            else if (fxr < F.get(D)) ph = RP_OC;
            else ph = RP_IC;
        } else if (ph == RP_E) {
            nm_evals++;
            const double fxr = L.fxr();
            if (f < fxr) ar_accept(L, F, D, 1, f); else ar_accept(L, F, D, 0, fxr);
            nm_iters++; ph = RP_ITER;
        } else if (ph == RP_OC || ph == RP_IC) {
            nm_evals++;
            const bool ok = (ph == RP_OC) ? (f <= L.fxr()) : (f < F.get(D));
            if (ok) { ar_accept(L, F, D, ph == RP_OC ? 2 : 3, f); nm_iters++; ph = RP_ITER; }
            else {
                for (int k = 1; k <= D; k++)
                    for (int i = 0; i < D; i++) L.sim(k, i) = L.sim(0, i) + 0.5 * (L.sim(k, i) - L.sim(0, i));
                vi = 0; ph = RP_SHRINK;
            }
        } else if (ph == RP_SHRINK) {
            F.set(1 + vi, f); vi++; nm_evals++;
            if (vi == D) { nm_iters++; ar_sort(L, F, D); ph = RP_ITER; }
        }
    }
    if (evals > 0 && even) { a.passes[s] += evals; a.evals[s] += evals; }
    return loops;
}

// Second launch of the refit: the series the first one parked (those still running after its evaluation budget -- the few that
// decide how long the refit takes) continue with FOUR trial points per filter pass: reflection, expansion, outside and inside
// contraction by four lane pairs (eight lanes per series) instead of the sequential driver's ~1.7 passes per iteration, the four
// vertices of a shrink likewise.  Every pair keeps its own copy of the simplex and applies the same decisions to it (the function
// values cross the octet through wave shuffles) and the evaluation counts advance as in the sequential driver: same estimates, bit for bit.
template <int RM>
__device__ __noinline__ int ar_refit_body_spec(const ArimaArgs &a, const ArWs &ws, const int r_lo, const int cursor, double *lds)
{
    const int lane = threadIdx.x;
    const int g = (lane >> 1) & 3, lead = lane & ~7;       // pair g of the octet evaluates trial point g
    const int m = a.m;
    const size_t ld = a.ld;
    const ArSimG L{ws.ml_sim + (size_t)blockIdx.x * (size_t)64 * NM_BLOCK, lane};
    const ArSimG &F = L;
    ArMlLds Q{lds, RM, ar_ml_l1(m > AR_LDS_PERIOD ? 1 : m)};
    const int n_parked = ws.counts[26];
    int s = 0, len = 0, D = 0;
    ArOrd o{0, 0, 0, 0, 0};
    const double *w = ws.W;
    bool fin = false;
    int ph = RP_DONE, vi = 0, nm_evals = 0, nm_iters = 1, evals = 0, passes = 0;

    int loops = 0;
    for (;; loops++) {
        for (int attempt = 0; attempt < 4 && !fin && ph == RP_DONE; attempt++) {
            if (passes > 0) { if (lane == lead) { a.passes[s] += passes; a.evals[s] += evals; } evals = 0; passes = 0; }
            int item = 0;
            if (lane == lead) item = atomicAdd(&ws.counts[cursor], 1);
            item = __shfl(item, lead);
            if (item >= n_parked) { fin = true; break; }
            s = ws.ml_park_list[item];
            len = a.wlen[s];
            o.p = a.order[(size_t)0 * ld + s]; o.q = a.order[(size_t)1 * ld + s]; o.P = a.order[(size_t)2 * ld + s];
            o.Q = a.order[(size_t)3 * ld + s]; o.c = a.order[(size_t)4 * ld + s];
            D = o.p + o.q + o.P + o.Q + o.c;
            {
                const int la = o.p + m * o.P, lb1 = o.q + m * o.Q + 1;
                const int rr = la > lb1 ? la : lb1;
                if (rr <= r_lo || rr > RM) continue;              // another instantiation's series
            }
            w = ws.W + (size_t)s * ws.tw;
            const double *pk = ws.ml_park + (size_t)s * 64;
            for (int c = 0; c < AR_ML_CTX; c++) L.base[(size_t)c * NM_BLOCK + L.col] = pk[c];
            nm_evals = (int)pk[60]; nm_iters = (int)pk[61];
            vi = 0;
            ph = RP_ITER;
        }
        if (__all(fin && ph == RP_DONE)) break;
        double x[AR_MAXDIM] = {0, 0, 0, 0, 0, 0};
        bool mine = false;                                   // this pair's evaluation is a real one
        if (ph != RP_DONE) {
            if (ph == RP_ITER) {
                bool stop = !(nm_evals < AR_ML_NM_CAP * D && nm_iters < AR_ML_NM_CAP * D);
                if (!stop) {
                    bool small = true;
                    for (int k = 1; k <= D; k++) {
                        for (int i = 0; i < D; i++)
                            if (!(fabs(L.sim(k, i) - L.sim(0, i)) <= 1.0e-4)) small = false;
                        if (!(fabs(F.get(0) - F.get(k)) <= 1.0e-8)) small = false;
                    }
                    stop = small;
                }
                if (stop) {
                    const double fb = F.get(0);
                    if (lane == lead && fabs(fb) <= 1.7976931348623157e308 && fb <= L.f0() && ar_simplex_best_roots_ok(o, m, D, L))
                        for (int i = 0; i < D; i++) a.xbest[(size_t)i * ld + s] = L.sim(0, i);
                    ph = RP_DONE;
                }
            }
            if (ph == RP_ITER) { mine = true; for (int i = 0; i < D; i++) x[i] = ar_trial(L, D, g, i); }
            else if (ph == RP_SHRINK) { mine = 1 + vi + g <= D; if (mine) for (int i = 0; i < D; i++) x[i] = L.sim(1 + vi + g, i); }
        }
        int La = 0, Lb = 0;
        double mu = 0.0;
        if (ph != RP_DONE) { ar_build_poly_ml(o, m, x, Q, La, Lb, mu); passes++; }
        const double f = ar_ml_eval<RM>(ph != RP_DONE && mine, La, Lb, mu, w, len, Q);     // the whole wave: wave-uniform filter loop
        double fj[4];
        for (int j = 0; j < 4; j++) fj[j] = __shfl(f, lead + 2 * j);
        if (ph == RP_ITER) {
            const double fr = fj[0], fe = fj[1], foc = fj[2], fic = fj[3];
            nm_evals++; evals++;
            bool shrink = false;
            if (fr < F.get(0)) { nm_evals++; evals++; if (fe < fr) ar_accept(L, F, D, 1, fe); else ar_accept(L, F, D, 0, fr); }
            else if (fr < F.get(D - 1)) ar_accept(L, F, D, 0, fr);
            else if (fr < F.get(D)) { nm_evals++; evals++; if (foc <= fr) ar_accept(L, F, D, 2, foc); else shrink = true; }
            else { nm_evals++; evals++; if (fic < F.get(D)) ar_accept(L, F, D, 3, fic); else shrink = true; }
            if (!shrink) nm_iters++;
            else {
                for (int k = 1; k <= D; k++)
                    for (int i = 0; i < D; i++) L.sim(k, i) = L.sim(0, i) + 0.5 * (L.sim(k, i) - L.sim(0, i));
                vi = 0; ph = RP_SHRINK;
            }
        } else if (ph == RP_SHRINK) {
            for (int j = 0; j < 4; j++) if (1 + vi + j <= D) F.set(1 + vi + j, fj[j]);
            const int cnt = (D - vi) < 4 ? (D - vi) : 4;
            vi += cnt; nm_evals += cnt; evals += cnt;
            if (vi == D) { nm_iters++; ar_sort(L, F, D); ph = RP_ITER; }
        }
    }
    if (passes > 0 && lane == lead) { a.passes[s] += passes; a.evals[s] += evals; }
    return loops;
}

// class of a series' state dimension: 0..4 = r <= 8 / 12 / 16 / 20 / 32, -1 = nothing to refit
__device__ __forceinline__ int ar_ml_class(const ArimaArgs &a, int s)
{
    if (!(a.wlen[s] >= 3 && a.status[s] == FIT_OK)) return -1;
    const size_t ld = a.ld;
    const int p = a.order[(size_t)0 * ld + s], q = a.order[(size_t)1 * ld + s], P = a.order[(size_t)2 * ld + s], Qs = a.order[(size_t)3 * ld + s];
    if (p + q + P + Qs + a.order[(size_t)4 * ld + s] == 0) return -1;
    if (a.m > AR_LDS_PERIOD && (P || Qs)) return -1;          // seasonal terms of a long period: CSS estimates stay (oracle refit_ml)
    const int la = p + a.m * P, lb1 = q + a.m * Qs + 1;
    const int rr = la > lb1 ? la : lb1;
    return rr <= 8 ? 0 : (rr <= 12 ? 1 : (rr <= 16 ? 2 : (rr <= 20 ? 3 : (rr <= AR_ML_MAX_R ? 4 : -1))));
}
// series per class -> ws.counts[16..20] (the refit deals its waves to the classes in that proportion)
__global__ void arima_refit_count_kernel(const ArimaArgs a, const ArWs ws)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_series) return;
    const int c = ar_ml_class(a, s);
    if (c >= 0) atomicAdd(&ws.counts[16 + c], 1);
}

// One persistent launch for all state-dimension classes (cursors ws.counts[9..13]).  A wave starts on the class its block
// index falls into when the grid is dealt to the classes by their series counts, then walks round the other classes (whose
// cursors it mostly finds exhausted): the classes run side by side and the kernel ends with the slowest class, not with the
// sum of their critical paths.  256 registers (two waves per SIMD): with 512 allowed the scheduler spread the r-vectors over
// VGPRs, AGPRs and scratch -- 15 scratch reloads per filter step at r = 20, 2,500 cycles per step, measured.
template <bool SPEC>
__global__ __launch_bounds__(NM_BLOCK, 2) void arima_refit_kernel(const ArimaArgs a, const ArWs ws, const int budget)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (SPEC && ws.counts[26] == 0) return;                  // nothing was parked
    int first = 0;
    {
        long tot = 0;
        for (int c = 0; c < 5; c++) tot += ws.counts[16 + c];
        if (tot == 0) return;
        const long pos = ((long)blockIdx.x * tot) / (long)gridDim.x;
        long cum = 0;
        for (int c = 0; c < 5; c++) { cum += ws.counts[16 + c]; if (pos < cum) { first = c; break; } }
    }
    long long tc[6];
    int it[5] = {0, 0, 0, 0, 0};
    tc[0] = wall_clock64();
    for (int k = 0; k < 5; k++) {
        const int c = (first + k) % 5;
        if (ws.counts[16 + c] == 0) { tc[k + 1] = tc[k]; continue; }
        if (c == 0) it[0] += SPEC ? ar_refit_body_spec<8>(a, ws, 0, 21, lds) : ar_refit_body<8>(a, ws, 0, 9, lds, budget);
        else if (c == 1) it[1] += SPEC ? ar_refit_body_spec<12>(a, ws, 8, 22, lds) : ar_refit_body<12>(a, ws, 8, 10, lds, budget);
        else if (c == 2) it[2] += SPEC ? ar_refit_body_spec<16>(a, ws, 12, 23, lds) : ar_refit_body<16>(a, ws, 12, 11, lds, budget);
        else if (c == 3) it[3] += SPEC ? ar_refit_body_spec<20>(a, ws, 16, 24, lds) : ar_refit_body<20>(a, ws, 16, 12, lds, budget);
        else it[4] += SPEC ? ar_refit_body_spec<AR_ML_MAX_R>(a, ws, 20, 25, lds) : ar_refit_body<AR_ML_MAX_R>(a, ws, 20, 13, lds, budget);
        tc[k + 1] = wall_clock64();
    }
    if (a.trace && threadIdx.x == 0 && (blockIdx.x % 32) == 0)      // 100 MHz wall clock: 1e5 ticks per ms
        printf("refit wave %d: loops class8 %d class12 %d class16 %d class20 %d class32 %d, ticks of its 1st..5th class %lld %lld %lld %lld %lld\n", (int)blockIdx.x,
               it[0], it[1], it[2], it[3], it[4], tc[1] - tc[0], tc[2] - tc[1], tc[3] - tc[2], tc[4] - tc[3], tc[5] - tc[4]);
}

#define AR_HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP: ") + hipGetErrorString(e_) + " at " #x); } while (0)

int arima_max_fit_waves()
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus * 8;                   // no variant of the fit kernels runs more than two waves per SIMD
}

int launch_arima(const ArimaArgs &a, hipStream_t stream)
{
    ArWs ws;
    const size_t need = ws.carve((char *)a.ws, a.n_series, a.t_max);
    if (!a.ws || need > a.ws_bytes) throw std::runtime_error("AutoARIMA workspace too small");
    const int grid = (a.n_series + NM_BLOCK - 1) / NM_BLOCK;
    const int grid256 = (a.n_series + 255) / 256;
    const size_t fit_lds = ar_lds_bytes(a.m);
    const bool long_m = a.m > AR_LDS_PERIOD;
    if (a.m > AR_MAX_PERIOD) throw std::runtime_error("AutoARIMA: seasonal period above the cap");
    if (a.m_col && !long_m) throw std::runtime_error("AutoARIMA: a batch of several periods needs all of them above the LDS limit");
    if (long_m && !a.long_scratch) throw std::runtime_error("AutoARIMA: a period above the LDS limit needs the scratch area");
    const size_t fc_lds = long_m ? 0 : sizeof(double) * ar_fc_scratch_doubles(a.m);
    // the pass variant of the period (ar_css_pass: the fit kernels' ring has 2 m + 4 slots)
    typedef void (*fit_fn_t)(const ArimaArgs, const ArWs, const int, const int, const int);
    fit_fn_t fit_seq, fit_spec;
    int fit_waves = 1;
    if (a.m == 7) { fit_seq = arima_fit_kernel<3, 7>; fit_spec = arima_fit_spec_kernel<3, 7>; fit_waves = ar_fit_waves(3); }
    else if (a.m == 12) { fit_seq = arima_fit_kernel<1, 12>; fit_spec = arima_fit_spec_kernel<1, 12>; }
    else if (a.m == 4) { fit_seq = arima_fit_kernel<1, 4>; fit_spec = arima_fit_spec_kernel<1, 4>; }
    else if (a.m > AR_LDS_PERIOD && a.m_col) { fit_seq = arima_fit_kernel<6, 0>; fit_spec = arima_fit_spec_kernel<6, 0>; }      // long periods, one per series
    else if (a.m > AR_LDS_PERIOD) { fit_seq = arima_fit_kernel<5, 0>; fit_spec = arima_fit_spec_kernel<5, 0>; }
    else if (a.m >= 4) { fit_seq = arima_fit_kernel<1, 0>; fit_spec = arima_fit_spec_kernel<1, 0>; }
    else if (a.m <= 1) { fit_seq = arima_fit_kernel<2, 0>; fit_spec = arima_fit_spec_kernel<2, 0>; }
    else { fit_seq = arima_fit_kernel<0, 0>; fit_spec = arima_fit_spec_kernel<0, 0>; }
    if (fit_lds > 48 * 1024) {
        AR_HIPCHECK(hipFuncSetAttribute((const void *)fit_seq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fit_lds));
        AR_HIPCHECK(hipFuncSetAttribute((const void *)fit_spec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fit_lds));
    }
    if (fc_lds > 48 * 1024) AR_HIPCHECK(hipFuncSetAttribute((const void *)arima_forecast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fc_lds));
    int dev = 0, cus = 256;
    AR_HIPCHECK(hipGetDevice(&dev));
    AR_HIPCHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int per_cu = (int)((160 * 1024) / fit_lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 * fit_waves ? 4 * fit_waves : per_cu);         // waves per SIMD the kernel's registers allow
    const int max_waves = cus * per_cu;
    // the schedule thresholds (four lanes per problem, lookahead) were measured in units of one wave per SIMD: they stay in those units
    // when a variant fits two waves per SIMD (with twice the resident waves the same thresholds sent nine of eleven sweeps to the
    // four-lane kernel, 2.35x the lane-passes: 358 -> 410 ms on the M5 batch)
    const int sched_waves = cus * (per_cu > 4 ? 4 : per_cu);

    const size_t prep_lds = long_m ? 0 : sizeof(double) * (size_t)(2 * (a.m > 1 ? a.m : 1) + 1) * NM_BLOCK;      // figure (m) + window ring (m or m + 1)
    if (a.trace >= 2) AR_HIPCHECK(hipMemsetAsync(ws.counts + 96, 0, 32 * sizeof(int32_t), stream));
    // one lane per series, every loop a chain of dependent loads and adds: the kernel is as fast as the number of waves that hide each
    // other's latency.  64 series per wave are 477 waves on the M5 batch -- half the SIMDs idle, the others with one wave; with 16 series
    // per wave it would be 1,906 waves -- measured in round 4 and NOT faster once the ring window and the eight-lag KPSS sweep had cut the
    // kernel from 20.5 to 5.3 ms (profiles/r04_arima_experiments.txt), so Tunables::arima_prep_lanes defaults to 64; the knob stays for
    // the schedule-variant test.  (Long periods keep 64: their figure scratch is laid out per 64 series.)
    const int prep_block = long_m ? NM_BLOCK : a.prep_lanes;
    AR_HIPCHECK(hipMemsetAsync(ws.counts_p, 0, 128 * sizeof(int32_t), stream));
    hipLaunchKernelGGL(arima_prep_kernel, dim3((a.n_series + prep_block - 1) / prep_block), dim3(prep_block), prep_lds, stream, a, ws);
    hipLaunchKernelGGL(arima_skip_kernel, dim3(grid256), dim3(256), 0, stream, a);
    int launches = 2;
    long prev_total = -1;
    ArWs ws_sorted = ws;                 // the fit kernels of a sweep read the sorted copy of the queues
    ws_sorted.q_series = ws.q2_series; ws_sorted.q_key = ws.q2_key;
    // A launch runs persistent lanes over the whole queue -- and holds every SIMD's registers while it does.  Alone on the device that is
    // what we want; next to other AutoARIMA runs (the parts of a call with detected periods, one per period, each a chain of ~40 short
    // launches with host round trips between them) it is a convoy: the other chains' advance / sort / forecast kernels -- a few waves,
    // microseconds of work -- waited 28 ms on average (163 ms at most) for a long launch of somebody else to END (kernel trace of the
    // default call shape, round 4).  With company, a launch therefore takes `shared_chunk_rounds` rounds of the resident lanes at most
    // and the queue goes out in several launches: waves retire every few milliseconds and the other streams get in.
    auto launch_fit = [&](long from, long total, int polish, const ArWs &ws, hipStream_t stream) {      // items from .. total - 1 of the queue
        const double spec_factor = a.spec_factor;   // (tune arima_spec_factor, default 4)
        const bool spec = (double)(total - from) <= spec_factor * (double)sched_waves * (NM_BLOCK / 4);
        // short queue (up to a few problems per resident 4-lane group): four lanes per problem, one pass per Nelder-Mead
        // iteration -- such a launch is bound by its slowest fit, not by throughput
        const long per_wave = spec ? NM_BLOCK / 4 : NM_BLOCK;
        // (a negative value: always, |value| rounds per launch -- the parity test of the chunked form)
        const double rounds = a.shared_chunk_rounds < 0.0 ? -a.shared_chunk_rounds : a.shared_chunk_rounds;
        const bool shared = a.shared_chunk_rounds < 0.0 || (rounds > 0.0 && a.concurrent && a.concurrent->load() > 1);
        long chunk = shared ? (long)(rounds * (double)max_waves * (double)per_wave) : total - from;
        if (chunk < per_wave) chunk = per_wave;
        int n = 0;
        for (long first = from; first < total; first += chunk) {
            const long end = first + chunk < total ? first + chunk : total;
            if (first > 0) AR_HIPCHECK(hipMemsetAsync(ws.counts + 8, 0, sizeof(int32_t), stream));
            long waves = (end - first + per_wave - 1) / per_wave;
            if (waves > max_waves) waves = max_waves;
            hipLaunchKernelGGL(spec ? fit_spec : fit_seq, dim3((int)waves), dim3(NM_BLOCK), fit_lds, stream, a, ws, (int)first, (int)end, polish);
            n++;
        }
        return n;
    };
    // The selected models' CSS estimates, to convergence (one problem per series: the polish): advance queues a series' polish when its
    // search ends, so the queue is complete when the last sweep is -- no queue kernel, no extra host round trip.
    // (Round 4, measured: polishing the early finishers on a second stream BESIDE the remaining sweeps -- 20,623 of the 30,490 series have
    //  ended their search by the fourth sweep of the M5 batch -- loses: 230 -> 238-255 ms for every share of the chip and every trigger
    //  tried.  The persistent polish waves hold their SIMDs for the whole launch and the sweeps, which are the critical path, queue
    //  behind them; the polish's own critical path -- its slowest fits, 600 iterations -- does not get shorter by starting earlier.)
    ArWs ws_pol = ws;
    ws_pol.q_series = ws.qp_series; ws_pol.q_key = ws.qp_key; ws_pol.counts = ws.counts_p; ws_pol.cap = (size_t)a.n_series;
    long np_now = 0;
    for (int sweep = 0; sweep < 4 * AR_MAXMODELS; sweep++) {
        AR_HIPCHECK(hipMemsetAsync(ws.counts, 0, 96 * sizeof(int32_t), stream));
        if (a.queue_sort) AR_HIPCHECK(hipMemsetAsync(ws.hist, 0, sizeof(int32_t) * (size_t)AR_NBUCKETS * a.n_series, stream));
        // look one sweep ahead once the previous sweep's queue times the fan-out (~18 candidates per series) fits the
        // resident lanes `la_factor` times over: the extra fits cost idle lanes, the saved sweeps cost ~0.1-0.2 s each
        const double la_factor = a.lookahead;   // (tune arima_lookahead, default 6: profiles/r04_arima_experiments.txt; earlier rounds, default 12:) measured 0 / 0.25 / 1 / 4 / 16: 2.16 / 2.03 / 1.93 / 1.92 / 1.90 s on the M5 batch (round 1); with the refit 4 / 8 / 12 / 16 / 24 / 32 / 64: 2.23 / 2.23 / 2.14 / 2.14 / 2.17 / 2.17 / 2.92 s
        int lookahead = (prev_total >= 0 && (double)prev_total * (AR_SWEEP + 1) <= la_factor * (double)sched_waves * NM_BLOCK) ? 1 : 0;
        // ... and two sweeps ahead once even that fan-out squared fits the resident lanes (the late sweeps of a few hundred series are
        // each bound by their slowest fit, ~0.1 s: 5 of them on the M5 batch)
        const int la_depth = a.lookahead_depth;   // (tune arima_lookahead_depth, default 2)
        if (lookahead && la_depth >= 2 && (double)prev_total * (AR_SWEEP + 1) * (AR_SWEEP + 1) <= la_factor * (double)sched_waves * NM_BLOCK) lookahead = 2;
        hipLaunchKernelGGL(arima_advance_kernel, dim3(grid256), dim3(256), 0, stream, a, ws, lookahead);
        int32_t counts[AR_NBUCKETS], queued_p[AR_NBUCKETS];
        AR_HIPCHECK(hipMemcpyAsync(counts, ws.counts + AR_QC, sizeof counts, hipMemcpyDeviceToHost, stream));
        AR_HIPCHECK(hipMemcpyAsync(queued_p, ws.counts_p + AR_QC, sizeof queued_p, hipMemcpyDeviceToHost, stream));
        AR_HIPCHECK(hipStreamSynchronize(stream));
        launches++;
        np_now = 0;
        for (int i = 0; i < AR_NBUCKETS; i++) np_now += queued_p[i];
        long total = 0;
        for (int i = 0; i < AR_NBUCKETS; i++) total += counts[i];
        if (total == 0) break;
        if (a.trace) std::fprintf(stderr, "[anofox-hip] AutoARIMA sweep %d: %ld problems queued (lookahead %d)\n", sweep, total, lookahead);
        prev_total = total;
        if (a.queue_sort) {
            int longest = 0;
            for (int i = 0; i < AR_NBUCKETS; i++) longest = counts[i] > longest ? counts[i] : longest;
            hipLaunchKernelGGL(arima_queue_scan_kernel, dim3(AR_NBUCKETS), dim3(1024), 0, stream, ws);
            hipLaunchKernelGGL(arima_queue_scatter_kernel, dim3((longest + 255) / 256, AR_NBUCKETS), dim3(256), 0, stream, ws);
            launches += 2;
        }
        launches += launch_fit(0, total, 0, a.queue_sort ? ws_sorted : ws, stream);
    }
    if (np_now > 0) {
        if (a.trace) std::fprintf(stderr, "[anofox-hip] AutoARIMA polish: %ld selected models\n", np_now);
        AR_HIPCHECK(hipMemsetAsync(ws.counts_p + 8, 0, sizeof(int32_t), stream));
        launches += launch_fit(0, np_now, 1, ws_pol, stream);
    }
    // on request (ANOFOX_ARIMA_CSS_ML): final estimates of the selected models on the exact Gaussian likelihood
    if (a.ml_refit) {
        const size_t lds_b = sizeof(double) * ar_ml_lds_doubles(long_m ? 1 : a.m) * NM_BLOCK;
        if (lds_b > 48 * 1024) {
            AR_HIPCHECK(hipFuncSetAttribute((const void *)arima_refit_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
            AR_HIPCHECK(hipFuncSetAttribute((const void *)arima_refit_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
        }
        int per = (int)((160 * 1024) / lds_b);
        per = per < 1 ? 1 : (per > 8 ? 8 : per);                            // 256 registers: two waves per SIMD at most
        // persistent lane pairs, no more waves than fit the chip; two lanes per series and some slack, so that (nearly) every
        // series finds a pair among the waves that start with its class
        const int want = (2 * a.n_series + NM_BLOCK - 1) / NM_BLOCK + 8;
        const int g = std::min(std::min(cus * per, ar_ml_sim_waves(a.n_series)), want);
        AR_HIPCHECK(hipMemsetAsync(ws.counts + 9, 0, 18 * sizeof(int32_t), stream));
        hipLaunchKernelGGL(arima_refit_count_kernel, dim3(grid256), dim3(256), 0, stream, a, ws);
        // two launches: the sequential driver up to `budget` evaluations per series (most series are done by then), then the
        // speculative one -- four trial points per pass, eight lanes per series -- for the ones that are not: the refit lasts as
        // long as its slowest series, and those need one pass per iteration instead of ~1.7 once the chip has room for them
        const int budget = a.refit_budget;
        hipLaunchKernelGGL(arima_refit_kernel<false>, dim3(g), dim3(NM_BLOCK), lds_b, stream, a, ws, budget);
        if (budget > 0) hipLaunchKernelGGL(arima_refit_kernel<true>, dim3(g), dim3(NM_BLOCK), lds_b, stream, a, ws, 0);
        launches += 3;
    }
    hipLaunchKernelGGL(arima_forecast_kernel, dim3(grid), dim3(NM_BLOCK), fc_lds, stream, a, ws);
    if (a.trace >= 2) {
        int32_t tc[32];
        AR_HIPCHECK(hipMemcpyAsync(tc, ws.counts + 96, sizeof tc, hipMemcpyDeviceToHost, stream));
        AR_HIPCHECK(hipStreamSynchronize(stream));
        for (int k = 0; k < 2; k++) {
            const int32_t *c = tc + 16 * k;
            std::fprintf(stderr, "[anofox-hip] AutoARIMA %s kernel: wave-passes by variant <1,1,0,0> %d <2,3,0,0> %d <1,1,1,1> %d <1,2,1,2> %d <2,3,2,2> %d full %d; live lane-passes by class %d %d %d %d %d %d\n",
                         k ? "four-lane" : "sequential", c[0], c[1], c[2], c[3], c[4], c[5], c[8], c[9], c[10], c[11], c[12], c[13]);
        }
    }
    return launches + 1;
}

} // namespace anofox
