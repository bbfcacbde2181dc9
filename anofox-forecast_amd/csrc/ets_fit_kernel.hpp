// ets_fit_kernel.hpp -- fit one ETS spec for every series of the batch, in resumable rounds.
//
// round kernel : grid = ceil(n_active / 64) one-wave workgroups; lane <-> one still-running
//                (series, spec) problem.  Each lane advances its own Nelder-Mead (nm.hpp) by at most
//                `budget` streamed passes over its column of the time-major block (ets_device.hpp),
//                then parks the simplex in HBM.  Between rounds the host enqueues a compaction
//                of the unfinished problems and a column gather (kernels.hip), so every wave of the next
//                round is full again and still reads 512 contiguous bytes per time step.
// final kernel : one K = 1 pass per series with the optimum -> final states, h forecasts, AICc.
//
// Dominant cost: passes x 8 T bytes per problem -> HBM/L2 stream + fp64 VALU recursion; no MFMA
// (a scan, not a contraction).
#pragma once
#include "ets_device.hpp"
#include "classic_device.hpp"
#include "kernels.hpp"

#ifndef ANOFOX_ROUND_WAVES
#define ANOFOX_ROUND_WAVES 2   // min waves per SIMD of the round kernels (caps VGPRs at 256); measured best of 1/2/4
#endif
namespace anofox {

// Per-class residency policy of the round kernels (round 4).  PARK: the Nelder-Mead simplex of every lane rests in a global scratch
// of the workgroup between passes instead of LDS (nm.hpp), so LDS holds only the b^phi tables and the kernel may be resident four
// times per SIMD; WAVES: the residency the register allocation is asked to allow (caps VGPRs at 512 / WAVES).
// -DANOFOX_PARK_CLASS: 0 none (the DEFAULT: every PARK variant measured slower, profiles/r04_ab_experiments.txt), 1 damped multiplicative
// trend, 2 + every general-class spec, 3 all ETS specs.
#ifndef ANOFOX_PARK_CLASS
#define ANOFOX_PARK_CLASS 0
#endif
#ifndef ANOFOX_PARK_WAVES
#define ANOFOX_PARK_WAVES 4
#endif
// WPB: waves per workgroup of the damped multiplicative-trend kernels (-DANOFOX_DM_WPB; default 1).  Several independent waves may
// share one copy of the b^phi tables: nothing else is shared -- a wave owns its 64 lanes' problems, its slice of the dynamic LDS
// and its slices of the scratch areas (indexed by its VIRTUAL block number blockIdx.x * WPB + wave); the only barrier is the one
// behind the table fill.  Measured (profiles/r04_README.md): four waves per workgroup run a lone damped-M fit 6.5 % faster
// (128.2 against 137.1 ms) and the 25-spec batch 12 % SLOWER (614-626 against 548-551 ms: a 71 KB workgroup waits for a CU with
// that much LDS free while the other specs' 15 KB workgroups keep taking it).
#ifndef ANOFOX_DM_WPB
#define ANOFOX_DM_WPB 1
#endif
// ANOFOX_SETPRIO (experiment switch, round 6): wave issue priority (s_setprio) of the round kernels by spec class -- 0 none (shipped),
// 1 damped multiplicative trend 3, 2 damped-M 3 / other general class 1, 3 damped-M 3 / general 2 / additive-class K4 kernels stay 0
#ifndef ANOFOX_SETPRIO
#define ANOFOX_SETPRIO 0
#endif
// ANOFOX_ROUND_WAVES_COMPACT: the same for the kernels that stream a compact copy of the block (a staged row takes one register instead of two)
#ifndef ANOFOX_ROUND_WAVES_COMPACT
#define ANOFOX_ROUND_WAVES_COMPACT ANOFOX_ROUND_WAVES
#endif
// ANOFOX_SEQ_WAVES_COMPACT (experiment, 0 = off): residency asked of the one-lane-per-problem-only kernels (the first round) over a
// compact block.  Measured with 3 (profiles/r06_ab_three_waves_first_round.txt): 395 against 377 ms -- the kernels that fit 168
// registers are no faster with a third wave per SIMD (9-10 ms per wave either way), the four that do not fit spill (a first round of
// ETS(A,Md,M) 92 ms per wave): with two waves per SIMD the vector unit is already busy, residency is not the lever on this chip
#ifndef ANOFOX_SEQ_WAVES_COMPACT
#define ANOFOX_SEQ_WAVES_COMPACT 0
#endif
template <class Cfg, class YT = double> struct RoundTraits {
    static constexpr bool DAMPED_MUL = !Cfg::CLASSIC && Cfg::T == C_MUL && Cfg::D;
    static constexpr bool PARK = !Cfg::CLASSIC && (ANOFOX_PARK_CLASS >= 3 || (ANOFOX_PARK_CLASS == 2 && !Cfg::ADDITIVE) || (ANOFOX_PARK_CLASS == 1 && DAMPED_MUL));
    static constexpr int WAVES = PARK ? ANOFOX_PARK_WAVES : (std::is_same_v<YT, double> ? ANOFOX_ROUND_WAVES : ANOFOX_ROUND_WAVES_COMPACT);
    static constexpr int WPB = DAMPED_MUL ? ANOFOX_DM_WPB : 1;
};

// the model behind a Cfg: an ETS spec, or one of the SES / Holt / Holt-Winters / SeasonalES family (SSE objective, own start values)
template <class Cfg, int MS, class YT = double, bool CLASSIC = Cfg::CLASSIC> struct RoundModelOf { using type = EtsModel<Cfg, MS, 1, YT>; };
template <class Cfg, int MS, class YT> struct RoundModelOf<Cfg, MS, YT, true> { using type = ClassicRoundModel<Cfg::KIND, MS>; };      // (the classic family streams the fp64 block)

// SPEC = 0: sequential Nelder-Mead, one lane per problem (64 problems per wave)
// SPEC = 1: speculative, the four trial points of a problem in four adjacent lanes (16 problems per wave)
// SPEC = 2: two-level speculative, two iterations per pass, one problem per wave (the last problems of a spec)
// SPEC = 3: all three in one kernel, picked from the device-side count of running problems -- ONE launch per round and spec:
//           a launch whose workgroups only find out that another driver owns the round still has to be dispatched, and on a
//           saturated chip that stalls the spec's chain for milliseconds (6 ms measured for 1,024 empty workgroups)
// K4 (round 4; SPEC 0 / 3, additive class, no run-time ring): the one-lane-per-problem driver is the SPECULATIVE one with all four
//           trial points of an iteration evaluated by the SAME lane in one pass (ets_pass<.., K = 4>: one y load feeds four
//           recursions) -- one pass per iteration instead of ~1.7.  The additive-class pass is 6-10 instructions per 8-byte load:
//           memory bound, so on a batch whose live specs are all additive (intermittent counts: the real M5 shape) the passes
//           ARE the cost and four times the arithmetic is free; in a mix with the general-class specs (VALU bound) it is not,
//           and the sequential driver stays.  Same iterates, same evaluation counts (the speculative driver's bookkeeping).
// YT: storage type of the block the round streams (ets_device.hpp: double, or float / uint16_t for a compact copy of a batch of counts)
template <class Cfg, int MS, int SPEC, bool K4 = false, class YT = double>
__global__ __launch_bounds__(NM_BLOCK * RoundTraits<Cfg>::WPB, ((ANOFOX_SEQ_WAVES_COMPACT > 0 && SPEC == 0 && !K4 && !std::is_same_v<YT, double> && !Cfg::CLASSIC ? ANOFOX_SEQ_WAVES_COMPACT : RoundTraits<Cfg, YT>::WAVES) + RoundTraits<Cfg>::WPB - 1) / RoundTraits<Cfg>::WPB) void ets_round_kernel(const FitArgs a)
{
    extern __shared__ double lds_all[];
    constexpr int D = Cfg::DIM;
    constexpr bool PARK = RoundTraits<Cfg>::PARK;
    constexpr int WPB = RoundTraits<Cfg>::WPB;
    if constexpr (ANOFOX_SETPRIO >= 1 && RoundTraits<Cfg>::DAMPED_MUL) __builtin_amdgcn_s_setprio(3);
    else if constexpr (ANOFOX_SETPRIO >= 2 && !Cfg::CLASSIC && !Cfg::ADDITIVE) __builtin_amdgcn_s_setprio(ANOFOX_SETPRIO == 2 ? 1 : 2);
    // per chain (host_api.hip launch_fit_slots, tune prio_top): the instruction takes an immediate
    if (a.wave_prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (a.wave_prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.wave_prio == 1) __builtin_amdgcn_s_setprio(1);
    if constexpr (Cfg::T == C_MUL && Cfg::D) dm_pow_tab_init();     // b^phi tables -> LDS, by every thread of the workgroup, before any wave leaves
    const int lane = threadIdx.x & (NM_BLOCK - 1);
    const int wave = WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int vblock = (int)blockIdx.x * WPB + wave, vgrid = (int)gridDim.x * WPB;      // the wave's virtual one-wave workgroup
    // this wave's slice of the dynamic LDS: [simplex store unless it rests in global scratch][seasonal ring of a run-time period]
    const int lds_per_wave = (PARK ? 0 : nm_lds_doubles<D>()) + ((MS == -1 || MS == -3) ? a.m * NM_BLOCK : 0);
    double *const lds = lds_all + (size_t)wave * (size_t)lds_per_wave;
    double *const nmst = PARK ? a.nm_scratch + (size_t)vblock * (size_t)nm_lds_doubles<D>() : lds;
    double *const lds_ring = PARK ? lds : lds + nm_lds_doubles<D>();
    const unsigned long long trace_t0 = a.wave_trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int n_act = a.n_active ? *a.n_active : a.n_series;
    int mode = SPEC;
    if constexpr (SPEC == 3) mode = n_act > a.spec_below ? 0 : ((a.spec2_below > 0 && n_act <= a.spec2_below && n_act <= vgrid) ? 2 : 1);
    else if (a.spec_below >= 0) {                                           // another driver may own this round
        const int owner = n_act > a.spec_below ? 0 : (n_act > a.spec2_below ? 1 : 2);
        if (owner != SPEC) return;
    }
    const int LPP = mode == 0 ? 1 : (mode == 1 ? NM_K : NM_BLOCK);         // lanes per problem (a constant unless SPEC == 3)
    const int PPB = NM_BLOCK / LPP;                                         // problems per workgroup
    if (vblock * PPB >= n_act) return;
    const int p = vblock * PPB + lane / LPP;
    const bool valid = p < n_act;
    const int s = valid ? (a.series_of ? a.series_of[p] : p) : 0;
    const int len = valid ? a.len[s] : 0;

    // merged batch of several periods: the workgroup's problems share one (lane 0 always holds a problem)
    // (per-lane period variants: the lane's own column's period; else the workgroup's, from lane 0)
    const int m = a.m_col ? (MS <= -3 ? a.m_col[s] : __builtin_amdgcn_readfirstlane(a.m_col[s])) : a.m;
    const int n_param = a.n_param - ((Cfg::S != C_NONE && a.m_col) ? a.m - m : 0);      // a.n_param counts the m - 1 seasonal states of a.m
    bool active = valid;
    // ... and it runs without compaction: every round sweeps all columns and skips the finished ones
    if (!a.first_round && a.series_of == nullptr && valid && a.st.done[s]) active = false;
    if (a.first_round) {
        // admissibility of this (series, spec) -- mirrors the preconditions of oracle ets_fit
        int st = FIT_OK;
        if constexpr (Cfg::CLASSIC) {
            // the family's own preconditions (oracle/forecast.c m_holt_winters / m_seasonal_es): a series the caller's mask does
            // not select is left alone, Holt-Winters needs two seasons, SeasonalES one
            const int min_len = Cfg::KIND == CK_HW ? 2 * m : (Cfg::KIND == CK_SEASONAL_ES ? m : a.min_len);
            if (len <= 0 || (a.mask != nullptr && valid && a.mask[s] != a.want)) st = FIT_SKIPPED;
            else if (len < min_len) st = FIT_SHORT;
        } else {
            const uint32_t fl = valid ? a.flags[s] : 0u;
            if (len <= 0) st = FIT_SKIPPED;
            else if (Cfg::S != C_NONE && len < 2 * m) st = FIT_SHORT;
            else if (len < n_param + 2) st = FIT_SHORT;
            else if (a.need_positive && !(fl & SF_POSITIVE)) st = FIT_NONPOSITIVE;
            else if (a.skip_constant && (fl & SF_CONSTANT)) st = FIT_SKIPPED;
        }
        active = valid && st == FIT_OK;
        if (valid) {
            a.status[s] = st;
            if (!active) { a.st.done[s] = 1; a.st.passes[s] = 0; a.st.evals[s] = 0; a.st.iters[s] = 0; }
        }
    }

    SeriesView v;
    const bool gathered = a.gathered != 0 && n_act <= a.gather_cap;     // (else the gather kernel left y_round alone: no room)
    v.col = valid ? (gathered ? p : s) : 0;
    v.yb = gathered ? a.y_round : a.y;
    v.y = v.yb + v.col;
    v.ld = gathered ? a.ld_round : a.ld;
    v.len = active ? len : 0;
    v.wave_len = wave_max_i32(v.len);
    v.wave_min_len = wave_min_i32(active ? len : 0x7fffffff);
    v.rows = a.t_rows;
    if (v.wave_len == 0) return;

    typename RoundModelOf<Cfg, MS, YT>::type mdl;
    mdl.v = v;
    mdl.in.l0 = (active && !Cfg::CLASSIC) ? a.l0[s] : 0.0;
    mdl.in.b0 = (active && !Cfg::CLASSIC && Cfg::T != C_NONE) ? a.b0[s] : 0.0;
    mdl.in.fig = a.fig ? a.fig + s : nullptr;
    mdl.in.fig_ld = a.fig_ld;
    mdl.in.m = m;
    mdl.ring = (MS == -2 || MS == -4) ? a.ring_scratch + (size_t)vblock * (size_t)a.m * NM_BLOCK : lds_ring;

    NmRun r;
    if (a.first_round) nm_init_simplex(mdl, nmst, r, active);
    else {
        // resume: simplex and counters were parked in HBM, indexed by series
#pragma unroll
        for (int k = 0; k <= D; k++) {
#pragma unroll
            for (int i = 0; i < D; i++) nmst[(k * D + i) * NM_BLOCK + lane] = active ? a.st.sim[(size_t)(k * D + i) * a.ld + s] : 0.0;
            nmst[((D + 1) * D + k) * NM_BLOCK + lane] = active ? a.st.fs[(size_t)k * a.ld + s] : 0.0;
        }
        r.phase = active ? a.st.phase[s] : NM_ITER;
        r.evals = active ? a.st.evals[s] : 0;
        r.iters = active ? a.st.iters[s] : 0;
        r.passes = active ? a.st.passes[s] : 0;
        r.done = !active;
    }
    const int passes_in = r.passes;

    // one problem per wave (the last problems of a spec) runs to completion
    const int budget = mode == 2 ? (1 << 30) : ((SPEC == 3 && mode == 0 && !K4) ? a.budget_seq : a.budget);
    // mode 0, one lane per problem: the sequential driver, or (K4) the speculative one on a model that evaluates four points per lane
    // (written out per branch: a lambda capturing the model by reference kept it, and the run state, in scratch memory)
#define ANOFOX_ADVANCE_LANE()                                                                          \
    do {                                                                                               \
        if constexpr (K4) {                                                                            \
            EtsModel<Cfg, MS, NM_K, YT> mdl4;                                                          \
            mdl4.v = mdl.v; mdl4.in = mdl.in; mdl4.ring = mdl.ring;                                    \
            nm_advance_spec(mdl4, nmst, r, budget);                                                    \
        } else nm_advance_seq(mdl, nmst, r, budget);                                                   \
    } while (0)
    if constexpr (SPEC == 3) {
        if (mode == 2) nm_advance_spec2(mdl, nmst, r, budget);
        else if (mode == 1) nm_advance_spec(mdl, nmst, r, budget);
        else ANOFOX_ADVANCE_LANE();
    } else if constexpr (SPEC == 2) nm_advance_spec2(mdl, nmst, r, budget);
    else if constexpr (SPEC == 1) nm_advance_spec(mdl, nmst, r, budget);
    else ANOFOX_ADVANCE_LANE();
#undef ANOFOX_ADVANCE_LANE
    nm_fence();

    if (a.wave_trace && lane == 0) {
        const unsigned long long slot = atomicAdd(a.wave_trace, 1ull);
        if (slot < a.wave_trace[1]) {
            unsigned long long *rec = a.wave_trace + 4 + 4 * slot;
            rec[0] = a.wave_trace_tag | ((unsigned long long)mode << 8) | (unsigned long long)(K4 ? 4 : 0);
            rec[1] = trace_t0;
            rec[2] = __builtin_amdgcn_s_memrealtime();
            rec[3] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) | ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32);    // HW_ID, XCC_ID
        }
    }
    if (a.lane_stats) {
        // a lane's pass counter moves only while it evaluates (nm.hpp); the lane that parks last was live in every pass of the wave
        const int mine = r.passes - passes_in;
        int sum = mine;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
        const int most = wave_max_i32(mine);
        if (lane == 0 && most > 0) {
            atomicAdd(a.lane_stats, (unsigned long long)most);
            atomicAdd(a.lane_stats + 1, (unsigned long long)sum);
        }
    }
    if (active && (lane % LPP) == 0) {
#pragma unroll
        for (int k = 0; k <= D; k++) {
#pragma unroll
            for (int i = 0; i < D; i++) a.st.sim[(size_t)(k * D + i) * a.ld + s] = nmst[(k * D + i) * NM_BLOCK + lane];
            a.st.fs[(size_t)k * a.ld + s] = nmst[((D + 1) * D + k) * NM_BLOCK + lane];
        }
        a.st.phase[s] = r.phase;
        a.st.evals[s] = r.evals;
        a.st.iters[s] = r.iters;
        a.st.passes[s] = r.passes;
        a.st.done[s] = r.done ? 1 : 0;
    }
}

template <class Cfg, int MS, class YT = double>
__global__ __launch_bounds__(NM_BLOCK) void ets_final_kernel(const FitArgs a)
{
    extern __shared__ double lds[];
    constexpr int D = Cfg::DIM;
    const int lane = threadIdx.x;
    const int s = blockIdx.x * NM_BLOCK + lane;
    const bool valid = s < a.n_series;
    const int len = valid ? a.len[s] : 0;
    int st = valid ? a.status[s] : FIT_SKIPPED;
    const bool inspect = a.insp_sel != nullptr;
    const bool active = valid && len > 0 && st == FIT_OK && (!inspect || a.insp_sel[s] == a.insp_code);

    SeriesView v;
    v.col = valid ? s : 0;
    v.yb = a.y;
    v.y = a.y + v.col;
    v.ld = a.ld;
    v.len = active ? len : 0;
    v.wave_len = wave_max_i32(v.len);
    v.wave_min_len = wave_min_i32(active ? len : 0x7fffffff);
    v.rows = a.t_rows;
    if (v.wave_len == 0) {
        if (!inspect && valid && len > 0) {
            a.aicc[s] = __builtin_huge_val(); a.evals[s] = 0; a.iters[s] = 0; a.passes[s] = 0;
            // one-spec batch: this pass is the only writer of the intervals' sd -- a wave without a fitted lane writes what an unfitted
            // lane of a live wave gets (interval_kernel reads sd only where the status is 0, i.e. never here: defined all the same)
            if (a.sd_out != nullptr) a.sd_out[s] = 0.0;
        }
        return;
    }
    if constexpr (Cfg::T == C_MUL && Cfg::D) dm_pow_tab_init();
    EtsInit in;
    in.l0 = active ? a.l0[s] : 0.0;
    in.b0 = (active && Cfg::T != C_NONE) ? a.b0[s] : 0.0;
    in.fig = a.fig ? a.fig + (valid ? s : 0) : nullptr;
    in.fig_ld = a.fig_ld;
    in.m = a.m_col ? a.m_col[(size_t)blockIdx.x * NM_BLOCK] : a.m;
    const int n_param = a.n_param - ((Cfg::S != C_NONE && a.m_col) ? a.m - in.m : 0);

    double cand[1][D], f[1];
#pragma unroll
    for (int i = 0; i < D; i++) cand[0][i] = active ? a.st.sim[(size_t)i * a.ld + s] : 0.5;
    EtsFinalOut fin;
    fin.h = a.h;
    fin.yhat = a.yhat + (size_t)(valid ? s : 0) * a.h;
    fin.sse_out = nullptr;
    fin.fitted = nullptr; fin.states = nullptr; fin.fitted_ld = 0; fin.states_ld = 0;
    double sse = 0.0, var = 0.0;
    const bool want_sd = a.sd_out != nullptr && !inspect;       // one-spec batch: this pass is the second sweep of the intervals' sd
    fin.var_out = want_sd ? &var : nullptr;
    fin.mean = (want_sd && active) ? a.mean[s] : 0.0;
    if (inspect) {
        fin.fitted = a.insp_fitted + (valid ? s : 0); fin.fitted_ld = a.ld;
        fin.states = a.insp_states + (valid ? s : 0); fin.states_ld = a.ld;
        fin.sse_out = &sse;
        fin.h = 0;                                   // the forecasts of the run stay as they are
    }
    ets_pass<Cfg, MS, 1, true, YT>(v, in, cand, f, (MS == -2) ? a.ring_scratch + (size_t)blockIdx.x * (size_t)a.m * NM_BLOCK : lds, &fin);
    if (inspect) {
        if (active && fabs(f[0]) <= 1.7976931348623157e308) {
            EtsPar par;
            ets_unpack<Cfg>(cand[0], par);
            const double dk = (double)n_param, dn = (double)len, aic = f[0] + 2.0 * dk;
            double *o = a.insp_info + s;
            o[0 * a.ld] = par.alpha;
            o[1 * a.ld] = Cfg::T != C_NONE ? par.beta : __builtin_nan("");
            o[2 * a.ld] = Cfg::S != C_NONE ? par.gamma : __builtin_nan("");
            o[3 * a.ld] = Cfg::D ? par.phi : __builtin_nan("");
            o[4 * a.ld] = aic;
            o[5 * a.ld] = aic + 2.0 * dk * (dk + 1.0) / (dn - dk - 1.0);
            o[6 * a.ld] = f[0] + dk * dm_log(dn);
            o[7 * a.ld] = sse;
        }
        return;
    }

    if (want_sd && valid && len > 0) a.sd_out[s] = active ? sqrt(var / (double)len) : 0.0;      // (a series this spec does not fit has no forecast)
    if (valid && len > 0) {
        double aicc = __builtin_huge_val();
        if (active) {
            const double lik = f[0];
            if (!(fabs(lik) <= 1.7976931348623157e308)) { st = FIT_NONFINITE; a.status[s] = st; }
            else {
                const double dk = (double)n_param, dn = (double)len;
                const double aic = lik + 2.0 * dk;
                aicc = aic + 2.0 * dk * (dk + 1.0) / (dn - dk - 1.0);
            }
        }
        a.aicc[s] = aicc;
        a.evals[s] = active ? a.st.evals[s] : 0;
        a.iters[s] = active ? a.st.iters[s] : 0;
        a.passes[s] = active ? a.st.passes[s] + 1 : 0;
    }
}

template <class Cfg, int MS, int SPEC, bool K4 = false, class YT = double>
void ets_round_launch(const FitArgs &a, hipStream_t stream)
{
    static_assert(!K4 || ((SPEC == 0 || SPEC == 3) && MS >= 0 && !Cfg::CLASSIC), "K4: the one-lane-per-problem driver of an ETS spec without a run-time ring");
    constexpr int PPB = SPEC == 0 ? NM_BLOCK : (SPEC == 1 ? NM_BLOCK / NM_K : 1);
    int grid;
    if (SPEC == 3) {
        // enough workgroups for whichever driver the count picks: sequential for any count, four lanes per problem up to
        // spec_below problems, one wave per problem up to spec2_below
        const int n = a.n_series;
        const int g_seq = (n + NM_BLOCK - 1) / NM_BLOCK;
        const int g_spec = (std::min(n, std::max(a.spec_below, 0)) + NM_BLOCK / NM_K - 1) / (NM_BLOCK / NM_K);
        const int g_spec2 = std::min(n, std::max(a.spec2_below, 0));
        grid = std::max(g_seq, std::max(g_spec, g_spec2));
    } else {
        // the one-problem-per-wave driver only ever owns a round with <= spec2_below running problems
        const int n_max = (SPEC == 2 && a.spec_below >= 0 && a.spec2_below < a.n_series) ? a.spec2_below : a.n_series;
        grid = (n_max + PPB - 1) / PPB;
    }
    if (grid <= 0) return;
    constexpr int WPB = RoundTraits<Cfg>::WPB;
    const int blocks = (grid + WPB - 1) / WPB;                  // `grid` counts one-wave workgroups; WPB of them share a real one
    if (RoundTraits<Cfg>::PARK && (a.nm_scratch == nullptr || (size_t)blocks * WPB * (size_t)nm_lds_doubles<Cfg::DIM>() > a.nm_scratch_doubles))
        throw std::runtime_error("ets_round_launch: the simplex scratch does not cover the launch");
    // the HBM ring of a long period: one area of m * 64 doubles per one-wave workgroup of THIS launch
    if ((MS == -2 || MS == -4) && (a.ring_scratch == nullptr || (size_t)blocks * WPB * (size_t)a.m * NM_BLOCK > a.ring_scratch_doubles))
        throw std::runtime_error("ets_round_launch: the seasonal-ring scratch does not cover the launch");
    size_t lds_bytes = RoundTraits<Cfg>::PARK ? 0 : sizeof(double) * (size_t)nm_lds_doubles<Cfg::DIM>();
    if (MS == -1 || MS == -3) lds_bytes += sizeof(double) * (size_t)a.m * NM_BLOCK;
    lds_bytes *= WPB;
    if (lds_bytes > 48 * 1024)
        anofox_check_attr(hipFuncSetAttribute((const void *)ets_round_kernel<Cfg, MS, SPEC, K4, YT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL((ets_round_kernel<Cfg, MS, SPEC, K4, YT>), dim3(blocks), dim3(NM_BLOCK * WPB), lds_bytes, stream, a);
}

template <class Cfg, int MS, class YT = double>
void ets_final_launch(const FitArgs &a, hipStream_t stream)
{
    const int grid = (a.n_series + NM_BLOCK - 1) / NM_BLOCK;
    if (MS == -2 && (a.ring_scratch == nullptr || (size_t)grid * (size_t)a.m * NM_BLOCK > a.ring_scratch_doubles))
        throw std::runtime_error("ets_final_launch: the seasonal-ring scratch does not cover the launch");
    size_t lds_bytes = (MS == -1) ? sizeof(double) * (size_t)a.m * NM_BLOCK : 0;
    hipLaunchKernelGGL((ets_final_kernel<Cfg, MS, YT>), dim3(grid), dim3(NM_BLOCK), lds_bytes, stream, a);
}

} // namespace anofox
