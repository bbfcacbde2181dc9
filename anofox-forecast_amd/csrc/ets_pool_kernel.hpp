// ets_pool_kernel.hpp -- fit one ETS spec for every series of the batch from a WORK POOL: persistent waves whose
// lanes pull (series, spec) problems from a queue and refill as soon as a problem converges.
//
// Why: Nelder-Mead iteration counts differ 5-10x between series.  The round schedule (ets_fit_kernel.hpp) bounds the
// waste with resumable rounds + compaction + a column gather between them; the work pool removes all three: a lane whose
// problem is done takes the next one at once, so there are no rounds, no compaction kernels, no gather traffic, no launch
// gaps, and a wave stays full until the queue runs dry.  Lanes then hold unrelated series, so the pass streams each
// lane's OWN series from a series-major copy of the block (ets_pass<..., ROWS>; 128-bit loads, every fetched sector is
// used by the lane that fetched it) instead of the wave's 64 adjacent columns.
//
// The slowest problems (hundreds of iterations, each a dependent chain of T steps) are the critical path of the whole
// batch.  A lane group = 4 adjacent lanes.  A problem that has run `promote` iterations -- or any problem once the queue
// is dry -- closes its group to new work and, as soon as it is the group's only problem, takes all four lanes: the
// speculative driver of nm.hpp (reflection, expansion, outside and inside contraction evaluated in ONE pass), one pass
// per iteration instead of ~1.7.  Both drivers walk the identical trajectory (nm.hpp), so results do not depend on when
// a problem is promoted, which lane it ran in or which problems shared its wave: they are bit-identical to the round
// schedule and to the oracle.
//
// ONE kernel per compile unit serves all of the unit's candidate specs (ets_pool_unit_kernel, fit_units.hpp): a wave walks
// the unit's spec list in priority order -- the specs with the longest critical path first -- and works on a spec until
// that spec's queue is dry and its own problems are done, then moves on.  So the slow problems of an expensive spec
// overlap with the bulk of the next ones, nothing waits for a hardware queue (25 per-spec kernels on 16 queues
// serialised), and the last spec to run is the cheapest.  The final pass of a problem (optimum -> final states -> h
// forecasts + AICc, ets_final_kernel in the round schedule) is one more pass of the lane that fitted it.
//
// One-wave workgroups; the simplex of a problem lives in the LDS slot of its home lane (a promoted problem keeps its
// slot, its three helper lanes address it).  Dominant cost as before: passes x 8 T bytes per problem; no MFMA (a scan).
#pragma once
#include "ets_fit_kernel.hpp"

namespace anofox {

enum { POOL_IDLE = 0, POOL_SEQ = 1, POOL_SPEC = 2 };
constexpr int NM_FINAL = 9;      // pool only: Nelder-Mead is done, the pass in flight is the final one (forecasts + AICc)

template <class Cfg, int MS>
__device__ __noinline__ void ets_pool_body(const FitArgs &a, double *lds)
{
    constexpr int D = Cfg::DIM;
    const int tid = threadIdx.x;
    const int sub = tid & 3, gbase = tid & ~3;
    if constexpr (Cfg::T == C_MUL && Cfg::D) dm_pow_tab_init();
    const int n_prob = a.n_active ? *a.n_active : a.n_series;
    const int maxiter = 200 * D, maxfun = 200 * D;
    double lo[D], hi[D], x0[D];
    {
        EtsModel<Cfg, MS, 1> b0;
        b0.bounds(lo, hi, x0);
    }
    double *ring = (MS == -2) ? a.ring_scratch + (size_t)blockIdx.x * (size_t)a.m * NM_BLOCK : lds + nm_lds_doubles<D>();

    // ---- lane state (a promoted problem's four lanes carry identical copies) ----
    int mode = POOL_IDLE;
    int lane = tid;                  // LDS slot of the problem this lane works on (the ANOFOX_SIM / ANOFOX_FS macros index by `lane`)
    int s = 0, len = 0;
    const double *row = a.ys;
    double l0 = 0.0, b0v = 0.0;
    const double *fig = a.fig;
    int phase = NM_ITER, evals = 0, iters = 0, passes = 0, vi = 0;
    double fxr = 0.0;
    bool q_dry = false;              // wave-uniform: the queue had nothing left for this wave
    // diagnostics (a.trace != NULL): per spec, summed over waves
    unsigned long long tr_passes = 0, tr_active = 0, tr_clk = 0, tr_spec = 0, tr_t0 = a.trace ? wall_clock64() : 0ull;

    for (;;) {
        // ================= 1. promotion and refill =================
        {
            const unsigned long long act = __ballot(mode != POOL_IDLE);
            const unsigned gact = (unsigned)(act >> gbase) & 0xFu;
            // a problem that has had its share of cheap iterations (or any problem once the queue is dry) wants four lanes
            const bool wants = mode == POOL_SEQ && (iters >= a.promote || q_dry);
            const unsigned gwant = (unsigned)(__ballot(wants) >> gbase) & 0xFu;
            const bool ready = wants && phase == NM_ITER;                  // promotion happens at an iteration boundary
            const unsigned gready = (unsigned)(__ballot(ready) >> gbase) & 0xFu;
            const bool closing = gwant != 0u;
            if (__popc(gact) == 1 && (gready & gact) != 0u) {
                // the group's only problem takes all four lanes; it keeps its LDS slot (home lane)
                const int home = gbase + (__ffs((int)gact) - 1);
                s = __shfl(s, home); len = __shfl(len, home);
                evals = __shfl(evals, home); iters = __shfl(iters, home); passes = __shfl(passes, home);
                l0 = __shfl(l0, home); b0v = __shfl(b0v, home);
                const unsigned long long rp = (unsigned long long)row, fp = (unsigned long long)fig;
                row = (const double *)(((unsigned long long)(unsigned)__shfl((int)(rp >> 32), home) << 32) | (unsigned)__shfl((int)rp, home));
                fig = (const double *)(((unsigned long long)(unsigned)__shfl((int)(fp >> 32), home) << 32) | (unsigned)__shfl((int)fp, home));
                phase = NM_ITER;
                mode = POOL_SPEC;
                lane = home;
            }
            // idle lanes of groups that are not closing take the next problems of the queue; in start_spec mode a problem
            // is taken by a whole idle group and runs speculatively from its first pass (few problems: latency matters)
            for (int attempt = 0; attempt < 3 && !q_dry; attempt++) {
                const unsigned long long act2 = __ballot(mode != POOL_IDLE);
                const unsigned gact2 = (unsigned)(act2 >> gbase) & 0xFu;
                const bool need = a.start_spec ? (gact2 == 0u && sub == 0) : (mode == POOL_IDLE && !closing);
                const unsigned long long nb = __ballot(need);
                if (nb == 0ull) break;
                int base = 0;
                if (tid == 0) base = atomicAdd(a.head, __popcll(nb));
                base = __shfl(base, 0);
                const int idx = base + __popcll(nb & ((1ull << tid) - 1ull));
                const bool got = need && idx < n_prob;
                if (__ballot(need && !got) != 0ull) q_dry = true;
                int ns = 0, nlen = 0, st = FIT_SKIPPED;
                if (got) {
                    ns = a.series_of ? a.series_of[idx] : idx;
                    nlen = a.len[ns];
                    const uint32_t fl = a.flags[ns];
                    st = FIT_OK;
                    if (nlen <= 0) st = FIT_SKIPPED;
                    else if (Cfg::S != C_NONE && nlen < 2 * a.m) st = FIT_SHORT;
                    else if (nlen < a.n_param + 2) st = FIT_SHORT;
                    else if (a.need_positive && !(fl & SF_POSITIVE)) st = FIT_NONPOSITIVE;
                    else if (a.skip_constant && (fl & SF_CONSTANT)) st = FIT_SKIPPED;
                    a.status[ns] = st;
                    if (st != FIT_OK) {
                        a.st.done[ns] = 1; a.st.passes[ns] = 0; a.st.evals[ns] = 0; a.st.iters[ns] = 0;
                        if (nlen > 0) { a.aicc[ns] = __builtin_huge_val(); a.evals[ns] = 0; a.iters[ns] = 0; a.passes[ns] = 0; }
                    }
                }
                bool take = got && st == FIT_OK;
                int src = tid;
                if (a.start_spec) {                       // the group leader popped: hand the problem to its three helpers
                    src = gbase;
                    take = __shfl((int)take, gbase) != 0;
                    ns = __shfl(ns, gbase); nlen = __shfl(nlen, gbase);
                }
                if (take) {
                    s = ns; len = nlen;
                    row = a.ys + (size_t)ns * a.tw;
                    l0 = a.l0[ns];
                    b0v = (Cfg::T != C_NONE) ? a.b0[ns] : 0.0;
                    fig = a.fig ? a.fig + ns : nullptr;
                    lane = src;
                    mode = a.start_spec ? POOL_SPEC : POOL_SEQ;
                    // initial simplex (nm_init_simplex): x0 and x0 with one coordinate scaled by 1.05, clipped
#pragma unroll
                    for (int i = 0; i < D; i++) ANOFOX_SIM(0, i) = nm_clip(x0[i], lo[i], hi[i]);
#pragma unroll
                    for (int k = 0; k < D; k++) {
#pragma unroll
                        for (int i = 0; i < D; i++) ANOFOX_SIM(k + 1, i) = ANOFOX_SIM(0, i);
                        double vv = ANOFOX_SIM(0, k);
                        vv = (vv != 0.0) ? (1.0 + 0.05) * vv : 0.00025;
                        ANOFOX_SIM(k + 1, k) = nm_clip(vv, lo[k], hi[k]);
                    }
#pragma unroll
                    for (int k = 0; k <= D; k++) ANOFOX_FS(k) = 0.0;
                    phase = NM_INIT0; evals = 0; iters = 1; passes = 0; vi = 0;
                }
            }
            if (__ballot(mode != POOL_IDLE) == 0ull) {
                if (q_dry) break;
                continue;                                  // everything popped was inadmissible: try again
            }
            if (a.trace) {
                tr_passes += 1;
                tr_active += (unsigned long long)__popcll(__ballot(mode != POOL_IDLE));
                tr_spec += (unsigned long long)__popcll(__ballot(mode == POOL_SPEC));
            }
        }

        // ================= 2. the trial point of this lane =================
        double x[D];
#pragma unroll
        for (int i = 0; i < D; i++) x[i] = 0.5;
        if (mode == POOL_SEQ) {
            if (phase == NM_INIT0) { phase = NM_SEQ_INIT; vi = 0; }
            if (phase == NM_SEQ_INIT) {
#pragma unroll
                for (int i = 0; i < D; i++) x[i] = ANOFOX_SIM(vi, i);
            } else if (phase == NM_ITER) {
                if (!(evals < maxfun && iters < maxiter) || nm_converged<D>(lds, lane)) {
                    phase = NM_FINAL;                      // one more pass, at the optimum: final states, forecasts, AICc
#pragma unroll
                    for (int i = 0; i < D; i++) x[i] = ANOFOX_SIM(0, i);
                } else {
#pragma unroll
                    for (int i = 0; i < D; i++) x[i] = nm_trial<D>(lds, lane, 0, i, lo[i], hi[i]);
                }
            } else if (phase == NM_SEQ_E || phase == NM_SEQ_OC || phase == NM_SEQ_IC) {
                const int which = phase == NM_SEQ_E ? 1 : (phase == NM_SEQ_OC ? 2 : 3);
#pragma unroll
                for (int i = 0; i < D; i++) x[i] = nm_trial<D>(lds, lane, which, i, lo[i], hi[i]);
            } else { // NM_SEQ_SHRINK
#pragma unroll
                for (int i = 0; i < D; i++) x[i] = ANOFOX_SIM(1 + vi, i);
            }
        } else if (mode == POOL_SPEC) {
            if (phase == NM_INIT0) {
                const int k = sub <= D ? sub : D;
#pragma unroll
                for (int i = 0; i < D; i++) x[i] = ANOFOX_SIM(k, i);
            } else if (phase == NM_INIT1) {
#pragma unroll
                for (int i = 0; i < D; i++) x[i] = ANOFOX_SIM(D, i);
            } else if (phase == NM_ITER) {
                if (!(evals < maxfun && iters < maxiter) || nm_converged<D>(lds, lane)) {
                    phase = NM_FINAL;
#pragma unroll
                    for (int i = 0; i < D; i++) x[i] = ANOFOX_SIM(0, i);
                } else {
#pragma unroll
                    for (int i = 0; i < D; i++) x[i] = nm_trial<D>(lds, lane, sub, i, lo[i], hi[i]);
                }
            } else { // NM_SHRINK: vertices 1..D already contracted towards the best
                const int k = sub + 1 <= D ? sub + 1 : D;
#pragma unroll
                for (int i = 0; i < D; i++) x[i] = ANOFOX_SIM(k, i);
            }
        }
        // ================= 3. one streamed pass: every lane evaluates its point on its own series =================
        const bool active = mode != POOL_IDLE;
        SeriesView v;
        v.y = nullptr; v.yb = nullptr; v.col = 0; v.ld = 0; v.rows = 0;
        v.row = active ? row : a.ys;
        v.len = active ? len : 0;
        v.wave_len = wave_max_i32(v.len);
        v.wave_min_len = wave_min_i32(active ? len : 0x7fffffff);
        EtsInit in;
        in.l0 = active ? l0 : 0.0;
        in.b0 = active ? b0v : 0.0;
        in.fig = fig;
        in.fig_ld = a.fig_ld;
        in.m = a.m;
        double c1[1][D], f1[1];
#pragma unroll
        for (int i = 0; i < D; i++) c1[0][i] = x[i];
        // a lane whose pass is the final one also writes its h forecasts (one writer per problem)
        const bool writer = active && phase == NM_FINAL && (mode == POOL_SEQ || sub == 0);
        EtsFinalOut fin;
        fin.h = writer ? a.h : 0;
        fin.yhat = a.yhat + (size_t)(writer ? s : 0) * a.h;
        fin.sse_out = nullptr; fin.fitted = nullptr; fin.states = nullptr; fin.fitted_ld = 0; fin.states_ld = 0;
        const unsigned long long c0 = a.trace ? wall_clock64() : 0ull;
        ets_pass<Cfg, MS, 1, true, true>(v, in, c1, f1, ring, &fin);
        if (a.trace) tr_clk += wall_clock64() - c0;
        const double f = f1[0];
        // the four values of a promoted problem (every lane executes the shuffles)
        double fc[NM_K];
#pragma unroll
        for (int k = 0; k < NM_K; k++) fc[k] = __shfl(f, gbase + k);
        if (active) passes += 1;

        // ================= 4. Nelder-Mead update =================
        if (active && phase == NM_FINAL) {
            // what ets_final_kernel leaves behind: AICc (or a non-finite likelihood), the counters, the optimum
            if (writer) {
                double aicc = __builtin_huge_val();
                if (!(fabs(f) <= 1.7976931348623157e308)) a.status[s] = FIT_NONFINITE;
                else {
                    const double dk = (double)a.n_param, dn = (double)len;
                    const double aic = f + 2.0 * dk;
                    aicc = aic + 2.0 * dk * (dk + 1.0) / (dn - dk - 1.0);
                }
                a.aicc[s] = aicc;
                a.evals[s] = evals; a.iters[s] = iters; a.passes[s] = passes;
#pragma unroll
                for (int i = 0; i < D; i++) a.st.sim[(size_t)i * a.ld + s] = ANOFOX_SIM(0, i);
                a.st.evals[s] = evals; a.st.iters[s] = iters; a.st.passes[s] = passes - 1; a.st.done[s] = 1;
            }
            mode = POOL_IDLE;
            lane = tid;
            len = 0;
            phase = NM_ITER;
        } else if (mode == POOL_SEQ) {
            if (phase == NM_SEQ_INIT) {
                ANOFOX_FS(vi) = f;
                vi += 1;
                evals += 1;
                if (vi == D + 1) { nm_sort_all<D>(lds, lane); phase = NM_ITER; }
            } else if (phase == NM_ITER) {
                fxr = f;
                evals += 1;
                if (fxr < ANOFOX_FS(0)) phase = NM_SEQ_E;
                else if (fxr < ANOFOX_FS(D - 1)) { nm_accept<D>(lds, lane, 0, fxr, lo, hi); iters += 1; }
                else if (fxr < ANOFOX_FS(D)) phase = NM_SEQ_OC;
                else phase = NM_SEQ_IC;
            } else if (phase == NM_SEQ_E) {
                evals += 1;
                if (f < fxr) nm_accept<D>(lds, lane, 1, f, lo, hi);
                else nm_accept<D>(lds, lane, 0, fxr, lo, hi);
                iters += 1;
                phase = NM_ITER;
            } else if (phase == NM_SEQ_OC || phase == NM_SEQ_IC) {
                evals += 1;
                const bool ok = (phase == NM_SEQ_OC) ? (f <= fxr) : (f < ANOFOX_FS(D));
                if (ok) {
                    nm_accept<D>(lds, lane, phase == NM_SEQ_OC ? 2 : 3, f, lo, hi);
                    iters += 1;
                    phase = NM_ITER;
                } else {
                    nm_shrink_vertices<D>(lds, lane, lo, hi);
                    vi = 0;
                    phase = NM_SEQ_SHRINK;
                }
            } else { // NM_SEQ_SHRINK
                ANOFOX_FS(1 + vi) = f;
                vi += 1;
                evals += 1;
                if (vi == D) { iters += 1; nm_sort_all<D>(lds, lane); phase = NM_ITER; }
            }
        } else if (mode == POOL_SPEC) {
            // the four lanes hold identical state and apply the identical update to the shared slot (same values to the
            // same addresses, in lockstep)
            if (phase == NM_INIT0) {
#pragma unroll
                for (int k = 0; k < NM_K; k++)
                    if (k <= D) ANOFOX_FS(k) = fc[k];
                evals += (D + 1 < NM_K ? D + 1 : NM_K);
                phase = (D + 1 > NM_K) ? NM_INIT1 : NM_ITER;
                if (phase == NM_ITER) nm_sort_all<D>(lds, lane);
            } else if (phase == NM_INIT1) {
                ANOFOX_FS(D) = fc[0];
                evals += 1;
                phase = NM_ITER;
                nm_sort_all<D>(lds, lane);
            } else if (phase == NM_ITER) {
                const double fr = fc[0];
                evals += 1;
                bool shrink = false;
                int which = 0;
                double fnew = fr;
                if (fr < ANOFOX_FS(0)) {
                    evals += 1;
                    if (fc[1] < fr) { which = 1; fnew = fc[1]; }
                } else if (fr < ANOFOX_FS(D - 1)) {
                    which = 0;
                } else if (fr < ANOFOX_FS(D)) {
                    evals += 1;
                    if (fc[2] <= fr) { which = 2; fnew = fc[2]; } else shrink = true;
                } else {
                    evals += 1;
                    if (fc[3] < ANOFOX_FS(D)) { which = 3; fnew = fc[3]; } else shrink = true;
                }
                if (!shrink) {
                    nm_accept<D>(lds, lane, which, fnew, lo, hi);
                    iters += 1;
                } else {
                    nm_shrink_vertices<D>(lds, lane, lo, hi);
                    phase = NM_SHRINK;
                }
            } else { // NM_SHRINK results
#pragma unroll
                for (int k = 0; k < NM_K; k++)
                    if (k + 1 <= D) ANOFOX_FS(k + 1) = fc[k];
                evals += D;
                iters += 1;
                phase = NM_ITER;
                nm_sort_all<D>(lds, lane);
            }
        }
    }
    if (a.trace && tid == 0 && tr_passes) {
        atomicAdd(a.trace + 0, tr_passes);
        atomicAdd(a.trace + 1, tr_active);
        atomicAdd(a.trace + 2, tr_clk);
        atomicAdd(a.trace + 5, tr_spec);
        atomicAdd(a.trace + 6, 1ull);                         // waves that worked on this spec
        atomicMin(a.trace + 3, tr_t0);
        atomicMax(a.trace + 4, wall_clock64());
    }
}

} // namespace anofox
