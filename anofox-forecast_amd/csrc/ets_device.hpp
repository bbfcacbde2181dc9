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
#include <type_traits>
#include "det_math.hpp"
#include "nm.hpp"

namespace anofox {

enum { C_NONE = 0, C_ADD = 1, C_MUL = 2 };

constexpr double ETS_TOL = 1.0e-10;
constexpr double ETS_HUGEN = 1.0e10;
constexpr double ETS_LN2 = 0.693147180559945309417232121458;
constexpr double PAR_LO = 1.0e-4, PAR_HI = 0.9999, PHI_LO = 0.8, PHI_HI = 0.98;
constexpr int ETS_MERGED_LDS_PERIOD = 16;   // ... of a merged batch of several periods: its LDS ring is sized by the largest period for EVERY wave
constexpr int ETS_LDS_PERIOD = 64;      // run-time periods up to this keep the seasonal ring in LDS (MS == -1); longer ones
                                        // (weekly 52 fits, hourly 168, yearly-on-daily 365 do not) keep it in an HBM scratch
                                        // area of the wave (MS == -2: same code, the ring pointer is a global one)
constexpr int ETS_MAX_PERIOD = 2048;    // beyond this the series reports COMPUTATION_ERROR naming the cap (the reference takes any period)

struct SeriesView {
    const double *y;   // already offset to this lane's series: element t at y[t * ld]
    size_t ld;
    int len;           // this lane's length (0 = no series)
    int wave_len;      // max len over the wave (uniform)
    int wave_min_len;  // min len over the wave's active lanes (uniform)
    int rows;          // rows the block holds (row indices are clamped to rows - 1 by the streaming loads)
    const double *yb;  // the block itself (wave-uniform; a compact block's elements are YT, the pass addresses it in bytes) and this lane's column: y == yb + col.  The streaming loads
    int col;           // address rows from the scalar base so the per-load address work stays on the scalar unit
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

// Storage type of the streamed block (round 6).  The arithmetic is fp64 whatever the block holds; a batch whose every observation
// survives the round trip through a narrower type exactly -- counts: the M5 shape -- is streamed from a COMPACT copy of the block
// (kernels.hip compact_block_kernel makes it and counts the observations that do not survive; one that does not keeps the batch on
// the fp64 block).  float: every value with <= 24 significant bits; uint16_t: integers 0 .. 65,535.  Half / a quarter of the bytes per
// pass and of the registers the staged rows take; the step widens the value (one conversion) and runs the same operations on the
// same numbers: bit-identical (tests/test_gpu_parity.py test_compact_storage_is_bit_identical).
enum { YT_F64 = 0, YT_F32 = 1, YT_U16 = 2 };
template <class YT> struct YtCode;
template <> struct YtCode<double> { static constexpr int value = YT_F64; };
template <> struct YtCode<float> { static constexpr int value = YT_F32; };
template <> struct YtCode<unsigned short> { static constexpr int value = YT_U16; };
// ANOFOX_DM_RETRY (round 6): the damped multiplicative-trend step WITHOUT its per-step branch.  A block of S steps first runs with
// the near-one binomial b^phi for every lane (no branch: the steps of the unrolled block schedule across each other) while every
// lane tracks the largest |b - 1| it met; one wave-uniform test after the block -- did ANY lane leave |b - 1| <= 1/16, or end with
// a growth rate that is not a number -- and only then the block is run AGAIN from the state it started with, by the step with the
// branch (the table-driven power and its domain checks for the far lanes): 3-5 % of the blocks of the M5-shape fits
// (tools/far_blocks.py).  Same operations on the same numbers for every lane either way: the first run's results are used only
// when no lane needed the other formula.  0 (SHIPPED): the per-step branch (rounds 4-5).
// MEASURED, round 6 (profiles/r06_ab_retry_prio.txt, same box, alternating): bit-identical (84 parity tests) and 17 % SLOWER on the
// 25-spec batch (448-459 against 382-389 ms per step); no spills (183-248 registers), so it is not register pressure -- the block
// is simply issued twice as code (two unrolled copies per loop body) and the first copy's results stay live across the test.
// Together with round 5's wave-uniform ballot branch this closes the "step without its branch" line of attack on exact arithmetic.
#ifndef ANOFOX_DM_RETRY
#define ANOFOX_DM_RETRY 0
#endif
#ifndef ANOFOX_TWO_BLOCK
#define ANOFOX_TWO_BLOCK 1
#endif
#ifndef ANOFOX_S_DM
#define ANOFOX_S_DM 8           // block length (rows in flight) of the damped multiplicative-trend pass
#endif
#ifndef ANOFOX_S_GEN
#define ANOFOX_S_GEN 16         // ... of the other general-class passes
#endif
#ifndef ANOFOX_S_HBM_RING_COMPACT
#define ANOFOX_S_HBM_RING_COMPACT 16    // block length of the HBM-ring pass (one long period) over a compact block
#endif
#ifndef ANOFOX_S_COMPACT_MUL
#define ANOFOX_S_COMPACT_MUL 2      // block length factor of the general-class and damped-M passes over a compact (float / uint16) block
#endif
#ifndef ANOFOX_S_COMPACT_MUL_ADD
#define ANOFOX_S_COMPACT_MUL_ADD 1  // ... of the additive-class passes
#endif
#ifndef ANOFOX_K4_S_SEAS
#define ANOFOX_K4_S_SEAS 8      // block length of the four-candidates-per-lane pass with a seasonal ring in registers (4 x m more doubles of state)
#endif
template <int ERR, int TREND, bool DAMPED, int SEAS>
struct EtsCfg {
    static constexpr int E = ERR, T = TREND, S = SEAS;
    static constexpr bool D = DAMPED;
    static constexpr int DIM = 1 + (TREND != C_NONE) + (SEAS != C_NONE) + (DAMPED ? 1 : 0);
    static constexpr bool ADDITIVE = (ERR == C_ADD) && (TREND != C_MUL) && (SEAS != C_MUL);
    static constexpr bool CLASSIC = false;      // (classic_device.hpp: the SES / Holt / Holt-Winters / SeasonalES family on the same round kernels)
};

struct EtsPar { double alpha, bstar, phi, beta, gamma; DmPowLane pl; DmPowNear1 pc; };      // pl: this lane's copies of the b^phi tables, pc: the binomial coefficients of phi (damped multiplicative trend only)
// f: last one-step forecast (inspection only).  dlo / dhi: smallest / largest magnitude the step's reciprocal has been asked for, bmin: the
// smallest undamped multiplicative growth rate seen -- the pass checks them ONCE at its end (ets_objective_value) instead of comparing in
// every step: a comparison feeds a scalar mask that feeds a select, and with one or two waves per SIMD that round trip between the vector
// and the scalar unit stalls the recursion (~30 cycles a step, profiles/r05_step_anatomy.txt); min / max stay on the vector unit, off the chain
struct EtsState { double l, b, sse, mant; int eacc; int bad; double f; double dlo, dhi, bmin; double rfar; };      // rfar: largest |b - 1| of the running block (ANOFOX_DM_RETRY)

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
// NEAR: the damped multiplicative-trend step with the near-one power for every lane and no branch (the caller checks st.rfar after
// the block and re-runs it with NEAR = false when a lane was far: ANOFOX_DM_RETRY)
template <class Cfg, bool NEAR = false>
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
        st.f = f;
        if constexpr (Cfg::T == C_ADD) st.b = fma(p.beta, e, phib);
        if constexpr (Cfg::S == C_ADD) s = fma(p.gamma, e, s);
    } else {
        // every other spec, in error-correction form (oracle/ets.c ets_lik states the same operations in the same order)
        double phib = 0.0, q = st.l;
        if constexpr (Cfg::T == C_ADD) {
            phib = Cfg::D ? p.phi * st.b : st.b;
            q = st.l + phib;
        } else if constexpr (Cfg::T == C_MUL) {
            if constexpr (Cfg::D) {
                // growth rates live next to one: the binomial series in r = b - 1 (round 5, det_math.hpp), for every lane; the rare
                // lane outside |r| <= 1/16 takes the table-driven power (and the checks that only matter there: growth rates
                // outside [2^-1000, 2^1000] or not positive reject the trial point, as oracle/ets.c does)
                // (a per-lane branch: the block is skipped when no lane of the wave is far.  Measured alternatives, profiles/r05_step_anatomy.txt:
                //  ANY branch in the step costs ~96 cycles of a one-wave-per-SIMD step, taken or not; a wave-uniform ballot branch with
                //  the cold block out of line is 25-30 % slower over a whole fit -- the far case then pays two far jumps)
                const double r = st.b - 1.0;
                phib = dm_pow_near1(r, p.pc);
                if constexpr (NEAR) st.rfar = __builtin_fmax(st.rfar, fabs(r));
                else if (!(fabs(r) <= DM_POW_NEAR1_R)) {
                    if (!(st.b >= 0x1p-1000 && st.b <= 0x1p+1000)) st.bad = 1;
                    phib = dm_pow_step(st.b, p.phi, p.pl);
                }
            } else {
                st.bmin = __builtin_fmin(st.bmin, st.b);          // (checked at the end of the pass: a growth rate <= 0 rejects the trial point)
                phib = st.b;
            }
            q = st.l * phib;
        }
        double f = q;
        if constexpr (Cfg::S == C_ADD) f = q + s;
        else if constexpr (Cfg::S == C_MUL) f = q * s;
        st.f = f;
        // ONE reciprocal per step (dm_recip: the division's own instruction sequence without range scaling and fix-up); a denominator
        // outside [2^-1000, 2^1000] rejects the trial point here and in oracle/ets.c -- tracked as a running min / max of its magnitude
        auto recip = [&](const double d) __attribute__((always_inline)) {
            const double ad = fabs(d);
            st.dlo = __builtin_fmin(st.dlo, ad);
            st.dhi = __builtin_fmax(st.dhi, ad);
            return dm_recip(d);
        };
        if constexpr (Cfg::E == C_MUL) {
            static_assert(Cfg::S != C_ADD, "multiplicative error with an additive season is not a valid spec");
            const double rf = recip(f);
            const double eps = (y - f) * rf;
            st.mant = st.mant * fabs(f);                            // renormalised every fourth observation by the pass (ets_renorm)
            st.sse = fma(eps, eps, st.sse);
            const double qe = q * eps;
            st.l = fma(p.alpha, qe, q);
            if constexpr (Cfg::T == C_ADD) st.b = fma(p.beta, qe, phib);
            else if constexpr (Cfg::T == C_MUL) st.b = fma(p.beta, phib * eps, phib);
            if constexpr (Cfg::S == C_MUL) s = fma(p.gamma, s * eps, s);
        } else {
            const double e = y - f;
            st.sse = fma(e, e, st.sse);
            if constexpr (Cfg::S == C_MUL && Cfg::T == C_MUL) {
                const double R = recip(f * st.l);                   // = 1 / (q s l): 1 / (s l) = R q, 1 / s = R q l, 1 / q = R (s l)
                const double esl = e * (R * q);
                const double es = esl * st.l;
                const double sl = s * st.l;
                st.l = fma(p.alpha, es, q);
                st.b = fma(p.beta, esl, phib);
                s = fma(p.gamma, e * (R * sl), s);
            } else if constexpr (Cfg::S == C_MUL) {
                const double R = recip(f);                          // 1 / s = R q, 1 / q = R s
                const double es = e * (R * q);
                st.l = fma(p.alpha, es, q);
                if constexpr (Cfg::T == C_ADD) st.b = fma(p.beta, es, phib);
                s = fma(p.gamma, e * (R * s), s);
            } else {
                static_assert(Cfg::T == C_MUL, "general class: a multiplicative component");
                const double rl = recip(st.l);
                st.l = fma(p.alpha, e, q);
                st.b = fma(p.beta, e * rl, phib);
                if constexpr (Cfg::S == C_ADD) s = fma(p.gamma, e, s);
            }
        }
    }
}

// sum log|f| of a multiplicative-error pass is carried as mant * 2^eacc.  oracle/ets.c renormalises the product (frexp) at every
// observation; here it happens after every fourth step of a block and at the block's end -- the same (mant, eacc) bit for bit, because
// a multiplicative-error trial point is only admissible while every |f| lies in [2^-120, 2^120] (checked at the end of the pass through
// dlo / dhi): up to four such factors times a mantissa in [1/2, 1) stay normal numbers, and scaling by a power of two is exact
template <class Cfg>
__device__ __forceinline__ void ets_renorm(EtsState &st)
{
    if constexpr (Cfg::E == C_MUL) {
        int ex;
        st.mant = frexp(st.mant, &ex);
        st.eacc += ex;
    }
}

template <class Cfg>
__device__ __forceinline__ double ets_objective_value(const EtsState &st, int n)
{
    if (st.bad || !(fabs(st.sse) <= 1.7976931348623157e308)) return __builtin_huge_val();
    if constexpr (!Cfg::ADDITIVE) {
        // the domain of the step's reciprocal and of an undamped multiplicative growth rate, checked once (EtsState).  n == 0: no step ran
        // (multiplicative error: the denominator is the one-step forecast, whose domain is the narrower one of the log-likelihood product)
        constexpr double D_LO = Cfg::E == C_MUL ? 0x1p-120 : 0x1p-1000, D_HI = Cfg::E == C_MUL ? 0x1p+120 : 0x1p+1000;
        if (n > 0 && !(st.dlo >= D_LO && st.dhi <= D_HI)) return __builtin_huge_val();
        if constexpr (Cfg::T == C_MUL && !Cfg::D) { if (n > 0 && !(st.bmin > 0.0)) return __builtin_huge_val(); }
    }
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
struct EtsFinalOut {
    double *yhat; int h; double *sse_out;
    // inspection (NULL otherwise): one-step fitted values of this lane's series at fitted[t * fitted_ld], final states at
    // states[r * states_ld] with r = 0 level, 1 growth, 2 + j seasonal state of phase j
    double *fitted; size_t fitted_ld; double *states; size_t states_ld;
    // the sweep also carries the second pass of the series' population variance (NULL: no): *var_out = sum over t < len of (y_t - mean)^2,
    // the additions in time order as calculate_confidence_intervals has them (forecast.rs:2558-2591)
    double mean; double *var_out;
};

// The pass.  MS > 0: compile-time period, ring in VGPRs.  MS == 0: no seasonality.
// MS == -1: run-time period, ring in LDS (`ring`, K * m * 64 doubles).  MS == -2: run-time period, `ring` points to HBM scratch.
// MS == -3 / -4 (round kernels of a merged batch of several periods, K == 1): the same two, with the period a PER-LANE quantity --
// the lanes of a wave may then hold series of different periods, which is what lets a merged batch use the ordinary round schedule
// (compaction + dense re-gather pack survivors of different periods into one wave).  Costs a vector phase counter and its wrap
// per step; the ring of a lane is still `ring[j * 64 + lane]`, sized by the largest period of the batch.
template <class Cfg, int MS, int K, bool FINAL, class YT = double>
__device__ __forceinline__ void ets_pass(const SeriesView &v, const EtsInit &in,
                                         const double (&cand)[K][Cfg::DIM], double (&fout)[K],
                                         double *ring, const EtsFinalOut *fin)
{
    static_assert((Cfg::S == C_NONE) == (MS == 0), "MS == 0 iff no seasonal component");
    const int lane = threadIdx.x & (NM_BLOCK - 1);       // (round kernels may run several waves per workgroup)
    EtsPar par[K];
    EtsState st[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        ets_unpack<Cfg>(cand[k], par[k]);
        if constexpr (Cfg::T == C_MUL && Cfg::D) { par[k].pl = dm_pow_lane(); dm_pow_near1_coef(par[k].phi, par[k].pc); }
        st[k].l = in.l0; st[k].b = in.b0; st[k].sse = 0.0; st[k].mant = 1.0; st[k].eacc = 0; st[k].bad = 0;
        st[k].f = 0.0; st[k].dlo = __builtin_huge_val(); st[k].dhi = 0.0; st[k].bmin = __builtin_huge_val(); st[k].rfar = 0.0;

    }
    const double *yp = v.yb;            // wave-uniform base; the lane's column is added as a 32-bit offset
    const unsigned col_bytes = (unsigned)v.col * (unsigned)sizeof(YT);     // ld < 2^29 columns: the byte offset of a column fits 32 bits
    const size_t ld = v.ld;

    // y is streamed through two register buffers of S steps: the S rows of the NEXT block are requested before the S
    // steps of the current one run and are only touched (copied into cur) after them, so every wave keeps S loads in
    // flight behind S steps of recursion and the loop has one, already satisfied, wait per block.  The loads are
    // unconditional (row index clamped to the block's last row, wave-uniform scalar arithmetic); a block that reaches past
    // the shortest active series of the wave runs with a per-step predicate.
    const int wave_len = __builtin_amdgcn_readfirstlane(v.wave_len);
    const int wave_min_len = __builtin_amdgcn_readfirstlane(v.wave_min_len);
    const int row_max = __builtin_amdgcn_readfirstlane(v.rows) - 1;
    // block length by the weight of a step: ~28-32 steps for the additive class (5-10 operations a step), half that when
    // a step holds a reciprocal, a quarter when it holds a pow (damped multiplicative trend) -- those are long enough to
    // cover the latency with fewer rows in flight, and their own register needs leave less room for the buffers
    // (ring in HBM: the ring values stream through two more buffers of S -- half the block length, or the four buffers spill)
    //  (four candidates per lane, K = 4: half the block, the four recursions' states take the registers)
    //  (a compact block: a staged row takes ONE register, so twice the rows are in flight at the register cost of the fp64 block's
    //   length -- measured, round 6, 25-spec batch on the uint16 copy: damped-M 8 -> 16 rows 388-392 -> 377-382 ms, general class
    //   16 -> 32 rows 368-374 ms: profiles/r06_ab_block_len.txt)
    constexpr int S_WIDE = std::is_same_v<YT, double> ? 1 : (Cfg::ADDITIVE ? ANOFOX_S_COMPACT_MUL_ADD : ANOFOX_S_COMPACT_MUL);
    constexpr int S_FULL = (Cfg::ADDITIVE ? (K == 4 ? (MS > 0 ? ANOFOX_K4_S_SEAS : 16) : 32) : ((Cfg::T == C_MUL && Cfg::D) ? ANOFOX_S_DM : ANOFOX_S_GEN)) * S_WIDE;
    //  and at most 8 -- the ring prefetch needs periods of two blocks, a merged batch keeps periods from 17 up in HBM)
    // (a uniform batch of ONE long period, MS == -2, has m > ETS_LDS_PERIOD = 64 >= 2 S for blocks up to 32 steps; over a compact block the
    //  two y buffers leave the registers for two ring buffers of ANOFOX_S_HBM_RING_COMPACT steps)
    //  (the ROUND kernels only: the final pass of a merged batch of several periods also runs MS == -2, block by block, with periods from 17 up)
    constexpr int S_HBM_CAP = (MS == -2 && !FINAL && !std::is_same_v<YT, double>) ? ANOFOX_S_HBM_RING_COMPACT : 8;
    constexpr int S_TARGET = ((MS == -2 || MS == -4) && K == 1 && S_FULL > S_HBM_CAP) ? S_HBM_CAP : S_FULL;
    constexpr int S = (MS > 0) ? ((S_TARGET / MS > 0 ? S_TARGET / MS : 1) * MS) : S_TARGET;
    // (a staged row rests in the storage type: a float takes one register, a double two; 16-bit values are zero-extended by the load)
    typedef std::conditional_t<std::is_same_v<YT, unsigned short>, unsigned, YT> ybuf_t;
    auto widen = [](const ybuf_t x) __attribute__((always_inline)) { return (double)x; };
    ybuf_t cur[S], nxt[S];
    // One loader, no branch (a conditional load in the loop makes the compiler wait for every outstanding load at once)
    // and no per-load address arithmetic on the vector unit: the rows of a block are fetched with buffer loads whose
    // descriptor is rebuilt once per block from scalars (base = first row of the block, range = what is left of the
    // allocation), the row stride goes in the scalar offset operand and the lane's column in the 32-bit vector offset.
    // The hardware range check returns zeros for rows past the allocation, so the prefetch of the block after the last
    // one needs no clamp (those values are never used).
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    // The range arithmetic of the additive class runs in 32-bit ROW counts: gfx950 has no scalar 64-bit ordered compare, and byte
    // offsets compared as 64-bit values go through the vector unit and back -- three v_cmp -> s_cselect round trips per block, 22
    // instructions instead of 7; a block of the additive class is 5-10 instructions a step, so that is 7 % of a fitted ETS(A,A,A)
    // (17.7-17.8 -> 16.6 ms, profiles/r05_step_anatomy.txt section 6).  The general class keeps the byte form: its steps are 3-6 times
    // longer, the saving is under 1 %, and with the row form in every class the 25-spec batch measured 1.5-2.7 % SLOWER (443 / 442 ->
    // 456 / 449 ms, same box, alternating runs; not explained) where the additive-only form is neutral (455 / 459 -> 451 / 461 ms).
    const size_t row_bytes = ld * sizeof(YT);                      // (ld < 2^29 columns: fits 32 bits)
    const size_t total_bytes = (size_t)(row_max + 1) * row_bytes;
    const int rows_total = row_max + 1;
    auto load_block = [&](ybuf_t (&buf)[S], const int row0) __attribute__((always_inline)) {
        unsigned nrec;
        size_t off;
        if constexpr (Cfg::ADDITIVE) {
            const unsigned rows_left = row0 < rows_total ? (unsigned)(rows_total - row0) : 0u;
            const unsigned long long rem = (unsigned long long)rows_left * (unsigned long long)(unsigned)row_bytes;
            nrec = (unsigned)(rem >> 32) != 0u ? 0xffffffffu : (unsigned)rem;
            off = rows_left ? (size_t)((unsigned long long)(unsigned)row0 * (unsigned long long)(unsigned)row_bytes) : (size_t)0;
        } else {
            const size_t o = (size_t)row0 * row_bytes;
            const size_t rem = o < total_bytes ? total_bytes - o : 0;
            nrec = rem > 0xffffffffull ? 0xffffffffu : (unsigned)rem;
            off = rem ? o : 0;
        }
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)yp + off), 0, nrec, 0x00020000);
#pragma unroll
        for (int j = 0; j < S; j++) {
            if constexpr (std::is_same_v<YT, double>) {
                const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, col_bytes, (unsigned)(j * row_bytes), 0);
                buf[j] = __builtin_bit_cast(double, w);
            } else if constexpr (std::is_same_v<YT, float>) {
                const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rsrc, col_bytes, (unsigned)(j * row_bytes), 0);
                buf[j] = __builtin_bit_cast(float, w);
            } else {
                buf[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsrc, col_bytes, (unsigned)(j * row_bytes), 0);
            }
        }
    };
    load_block(cur, 0);
    double var_acc = 0.0;
    auto keep_fit = [&](const int t, const double yv) __attribute__((always_inline)) {
        if constexpr (FINAL) {
            if (fin->fitted && t < v.len) fin->fitted[(size_t)t * fin->fitted_ld] = st[0].f;
            // (unconditional: two operations of a step; a test here would make every step of the unrolled block its own basic block.  Every
            //  caller runs this under `t < len` of the lanes it writes for; with var_out == NULL the sum is never stored)
            const double dv = yv - fin->mean;
            var_acc += dv * dv;
        }
    };

    if constexpr (MS >= 0) {
        constexpr int MR = MS > 0 ? MS : 1;
        double s[K][MR];
        if constexpr (MS > 0) {
#pragma unroll
            for (int j = 0; j < MS; j++) {
                double f0 = in.fig[(size_t)j * in.fig_ld];
#pragma unroll
                for (int k = 0; k < K; k++) s[k][j] = f0;
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) s[k][0] = 0.0;
        }
        // Two blocks per iteration on alternating buffers -- the copy between the buffers, one of a step's instructions, disappears: SHIPPED
        // for the additive class only (ANOFOX_TWO_BLOCK = 1; for every class it measured slower in the 25-spec batch, round 5).  The main
        // loop runs the blocks that every active lane of the wave covers without a predicate; the tail runs them predicated.
        auto run_block = [&](const ybuf_t (&buf)[S], const int base, auto pred_tag) __attribute__((always_inline)) {
            constexpr bool PRED = decltype(pred_tag)::value;
            if constexpr (ANOFOX_DM_RETRY && Cfg::T == C_MUL && Cfg::D && K == 1 && !PRED && !FINAL) {
                // (unpredicated blocks of the round kernels: the optimiser's passes; tail blocks and the final pass keep the per-step branch)
                const EtsState st0 = st[0];
                double s0[MR];
#pragma unroll
                for (int jj = 0; jj < MR; jj++) s0[jj] = s[0][jj];
                st[0].rfar = 0.0;
#pragma unroll
                for (int j = 0; j < S; j++) {
                    ets_step<Cfg, true>(par[0], st[0], widen(buf[j]), s[0][MS > 0 ? j % MR : 0]);
                    if ((j & 3) == 3 || j == S - 1) ets_renorm<Cfg>(st[0]);
                }
                // any lane far (or with a growth rate that is not a number: the comparison in the other step sends it to the far side)?
                // (a lane without a series -- the last wave of a launch -- runs on zeros: it has no say)
                const bool again = v.len > 0 && (!(st[0].rfar <= DM_POW_NEAR1_R) || st[0].b != st[0].b);
                if (__builtin_amdgcn_ballot_w64(again) == 0ull) return;
                st[0] = st0;
#pragma unroll
                for (int jj = 0; jj < MR; jj++) s[0][jj] = s0[jj];
            }
#pragma unroll
            for (int j = 0; j < S; j++) {
                if (!PRED || base + j < v.len) {
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        ets_step<Cfg>(par[k], st[k], widen(buf[j]), s[k][MS > 0 ? j % MR : 0]);
                        if ((j & 3) == 3 || j == S - 1) ets_renorm<Cfg>(st[k]);     // (a constant once the loop is unrolled)
                    }
                    keep_fit(base + j, widen(buf[j]));
                }
            }
        };
        int base = 0;
        // ANOFOX_TWO_BLOCK: 0 one block per iteration and a copy between the buffers (half the code: the unrolled blocks of the ~25
        // round kernels that share a CU's instruction cache are what it holds), 1 two blocks for the additive class only (round 4),
        // 2 for every class
        constexpr bool TWO = ANOFOX_TWO_BLOCK == 2 || (ANOFOX_TWO_BLOCK == 1 && Cfg::ADDITIVE);
        if constexpr (TWO) {
            for (; base + 2 * S <= wave_min_len; base += 2 * S) {
                load_block(nxt, base + S);
                run_block(cur, base, std::false_type{});
                load_block(cur, base + 2 * S);
                run_block(nxt, base + S, std::false_type{});
            }
        }
        for (; base < wave_len; base += S) {
            load_block(nxt, base + S);
            if (base + S <= wave_min_len) run_block(cur, base, std::false_type{});
            else run_block(cur, base, std::true_type{});
#pragma unroll
            for (int j = 0; j < S; j++) cur[j] = nxt[j];
        }
        if constexpr (FINAL) {
            if constexpr (MS > 0) {
                if (fin->states && v.len > 0) {
#pragma unroll
                    for (int jj = 0; jj < MS; jj++) fin->states[(size_t)(2 + jj) * fin->states_ld] = s[0][jj];
                }
            }
            double pp = par[0].phi, phistar = par[0].phi;
            for (int i = 0; i < fin->h; i++) {
                double f;
                if constexpr (Cfg::T == C_NONE) f = st[0].l;
                else if constexpr (Cfg::T == C_ADD) f = st[0].l + phistar * st[0].b;
                else f = (st[0].b > 0.0) ? st[0].l * dm_pow_pos(st[0].b, phistar) : __builtin_nan("");
                if constexpr (MS > 0) {
                    const int j = (v.len + i) % MS;
                    double sv = s[0][0];
#pragma unroll
                    for (int jj = 1; jj < MS; jj++) sv = (j == jj) ? s[0][jj] : sv;
                    f = (Cfg::S == C_ADD) ? f + sv : f * sv;
                }
                if (v.len > 0) fin->yhat[i] = f;
                pp = pp * par[0].phi;
                phistar = phistar + pp;
            }
        }
    } else {
        // run-time period: ring[(k * m + j) * 64 + lane] in LDS, phase j advanced as a wave-uniform counter (per lane: MS <= -3)
        constexpr bool LANE_M = MS <= -3;
        static_assert(!LANE_M || K == 1, "per-lane periods: one candidate per lane");
        const int m = LANE_M ? in.m : __builtin_amdgcn_readfirstlane(in.m);
        const int m_max = LANE_M ? __builtin_amdgcn_readfirstlane(wave_max_i32(in.m)) : m;
        for (int j = 0; j < m_max; j++) {
            if (LANE_M && j >= m) continue;
            double f0 = in.fig[(size_t)j * in.fig_ld];
#pragma unroll
            for (int k = 0; k < K; k++) ring[(k * m + j) * NM_BLOCK + lane] = f0;
        }
        int j = 0;
        constexpr bool HBM_RING = (MS == -2 || MS == -4);
        if constexpr (HBM_RING && K == 1) {
            // Ring in HBM: a load of the ring inside the step would put a memory round trip on the recursion's critical path at
            // every step (the compiler cannot move it above the previous step's store to the same array).  The ring values of a block
            // are S DISTINCT phases and those of the next block S others (m > 16 >= 2 S), so they stream exactly like y: the next
            // block's S values are requested before the current block's steps run, the updated values go out as stores.
            static_assert(2 * S <= ((MS == -2 && !FINAL) ? ETS_LDS_PERIOD : ETS_MERGED_LDS_PERIOD), "the ring prefetch needs periods of at least two blocks");
            double rc[S], rn[S];
            auto ring_load = [&](double (&buf)[S], int j0) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < S; i++) {
                    buf[i] = ring[(size_t)j0 * NM_BLOCK + lane];
                    j0 = (j0 + 1 == m) ? 0 : j0 + 1;
                }
                return j0;
            };
            int jn = ring_load(rc, 0);
            int base = 0;
            for (; base < wave_len; base += S) {
                load_block(nxt, base + S);
                const int jnn = ring_load(rn, jn);
                const bool full = base + S <= wave_min_len;
#pragma unroll
                for (int i = 0; i < S; i++) {
                    if (full || base + i < v.len) {
                        double sv = rc[i];
                        ets_step<Cfg>(par[0], st[0], widen(cur[i]), sv);
                        if ((i & 3) == 3 || i == S - 1) ets_renorm<Cfg>(st[0]);
                        ring[(size_t)j * NM_BLOCK + lane] = sv;
                        keep_fit(base + i, widen(cur[i]));
                    }
                    j = (j + 1 == m) ? 0 : j + 1;
                }
#pragma unroll
                for (int i = 0; i < S; i++) { cur[i] = nxt[i]; rc[i] = rn[i]; }
                jn = jnn;
            }
        } else {
        auto ring_block = [&](const int base, const bool full) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < S; i++) {
                if (full || base + i < v.len) {
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        double sv = ring[(k * m + j) * NM_BLOCK + lane];
                        ets_step<Cfg>(par[k], st[k], widen(cur[i]), sv);
                        if ((i & 3) == 3 || i == S - 1) ets_renorm<Cfg>(st[k]);
                        ring[(k * m + j) * NM_BLOCK + lane] = sv;
                    }
                    keep_fit(base + i, widen(cur[i]));
                }
                j = (j + 1 == m) ? 0 : j + 1;
            }
        };
        int base = 0;
        for (; base < wave_len; base += S) {
            load_block(nxt, base + S);
            ring_block(base, base + S <= wave_min_len);
#pragma unroll
            for (int i = 0; i < S; i++) cur[i] = nxt[i];
        }
        }
        if constexpr (FINAL) {
            if (fin->states && v.len > 0)
                for (int jj = 0; jj < m; jj++) fin->states[(size_t)(2 + jj) * fin->states_ld] = ring[(0 * m + jj) * NM_BLOCK + lane];
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
    for (int k = 0; k < K; k++) {
        ets_renorm<Cfg>(st[k]);        // (a lane whose series ended inside a block: its last steps were not followed by a renormalisation)
        fout[k] = ets_objective_value<Cfg>(st[k], v.len);
    }
    if constexpr (FINAL) {
        if (fin->sse_out && v.len > 0) *fin->sse_out = st[0].sse;
        if (fin->var_out) *fin->var_out = var_acc;
        if (fin->states && v.len > 0) { fin->states[0] = st[0].l; fin->states[fin->states_ld] = st[0].b; }
    }
}

// Nelder-Mead model adaptor.  CPL = candidates per lane: 4 -> one lane evaluates all four trial points
// of its problem (throughput form, one y load feeds four recursions); 1 -> the four trial points of a
// problem sit in four adjacent lanes (latency form: a quarter of the per-pass latency and VGPRs, four
// times the waves), results exchanged with wave shuffles.  Both give bit-identical objective values.
template <class Cfg, int MS, int CPL, class YT = double>
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
        ets_pass<Cfg, MS, 1, false, YT>(v, in, c1, f1, ring, nullptr);
        return f1[0];
    }
    __device__ void eval(const double (&cand)[NM_K][DIM], double (&f)[NM_K]) const
    {
        if constexpr (CPL == NM_K) {
            ets_pass<Cfg, MS, NM_K, false, YT>(v, in, cand, f, ring, nullptr);
        } else {
            const int sub = threadIdx.x & 3;                 // (lane & 3: a wave is 64 consecutive threads)
            double mine[1][DIM], f1[1];
#pragma unroll
            for (int i = 0; i < DIM; i++)
                mine[0][i] = sub == 0 ? cand[0][i] : (sub == 1 ? cand[1][i] : (sub == 2 ? cand[2][i] : cand[3][i]));
            ets_pass<Cfg, MS, 1, false, YT>(v, in, mine, f1, ring, nullptr);
            const int base = (threadIdx.x & (NM_BLOCK - 1)) & ~3;
#pragma unroll
            for (int k = 0; k < NM_K; k++) f[k] = __shfl(f1[0], base + k);
        }
    }
};

} // namespace anofox
