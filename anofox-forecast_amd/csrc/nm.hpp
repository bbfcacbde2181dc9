// nm.hpp -- per-problem bounded Nelder-Mead for gfx950.
//
// Semantics: scipy-style Nelder-Mead (rho 1, chi 2, psi 0.5, sigma 0.5; initial simplex
// x0 and x0 with one coordinate scaled by 1.05; bounds by clipping; stop when the simplex
// is within xatol = 1e-4 AND the values within fatol = 1e-8, or after 200*dim iterations /
// evaluations).  These constants are what the reference's known-answer vectors pin for the
// SES / Holt / HoltWinters optimisers.
//
// MI355X mapping: an objective evaluation is one streamed pass over the problem's series.  Two
// interchangeable drivers walk the SAME trajectory (they only differ in which trial points get
// evaluated when, never in the values used):
//   * nm_advance_seq  -- one lane per problem, one trial point per pass, exactly the sequential
//                        algorithm (~1.7 passes per iteration).  Least arithmetic: used while the
//                        batch still offers more problems than the chip has lanes (VALU-bound).
//   * nm_advance_spec -- the four trial points of an iteration (reflection, expansion, outside and
//                        inside contraction -- all functions of the centroid and the worst vertex only)
//                        are evaluated SPECULATIVELY in one pass, by four adjacent lanes (or by one lane
//                        holding four recursions).  One pass per iteration: used when problems are few and
//                        the critical path (passes of the slowest problem) is what is left.
// `evals` counts the evaluations of the sequential algorithm in both drivers; `passes` the streamed
// passes.  The simplex lives in LDS between passes so that the pass owns the VGPR budget.
//
// Both drivers are RESUMABLE at iteration boundaries: they run about `budget` passes and leave a state
// (simplex + NmRun) that a later launch continues bit-for-bit.  The ETS fit runs in rounds with a
// compaction of the unfinished problems in between (iteration counts differ by 5-10x between series,
// so a run-to-completion wave would idle most of its lanes).
#pragma once
#include <hip/hip_runtime.h>

namespace anofox {

constexpr int NM_K = 4;          // trial points of one iteration
constexpr int NM_BLOCK = 64;     // one wave per workgroup: all wave-uniform loops need no barrier

__device__ __forceinline__ double nm_clip(double v, double lo, double hi)
{
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}

// phases; only NM_INIT0 (nothing evaluated yet) and NM_ITER (sorted simplex) exist at round boundaries
enum { NM_INIT0 = 0, NM_INIT1 = 1, NM_ITER = 2, NM_SHRINK = 3, NM_SEQ_INIT = 4, NM_SEQ_E = 5, NM_SEQ_OC = 6, NM_SEQ_IC = 7, NM_SEQ_SHRINK = 8 };

struct NmRun { int phase; int evals; int iters; int passes; bool done; };
struct NmStats { int iters; int evals; int passes; };

// Footprint (doubles) of one wave's simplex store: element idx of lane l rests at store[idx * 64 + l].
template <int D> constexpr int nm_lds_doubles() { return ((D + 1) * D + (D + 1)) * NM_BLOCK; }

// The simplex of one lane's problem.  Round 4: it lives in REGISTERS while an iteration is decided (compile-time indices only: the
// sort is a network of compare-and-swap selects, a run-time vertex index is a select chain) and rests in a STORE between passes --
// LDS as before, or a per-workgroup global scratch (`PARK` kernels).  With the store in global memory a round kernel's LDS holds
// only the b^phi tables, so its residency is bounded by registers instead of by 12.8 KB of LDS per wave: the damped
// multiplicative-trend kernels (72 % of the 30-spec workload's instructions, two LDS lookups and a division on every step's
// critical path) go from 2 - 2.5 to 4 waves per SIMD.  The store is read twice and written once per pass (25 coalesced 512-byte
// accesses each at D = 4: 2.5 % of the pass's own y stream).
template <int D> struct NmSimplex { double x[D + 1][D]; double f[D + 1]; };

template <int D> __device__ __forceinline__ void nm_load(const double *st, int lane, NmSimplex<D> &S)
{
#pragma unroll
    for (int k = 0; k <= D; k++) {
#pragma unroll
        for (int i = 0; i < D; i++) S.x[k][i] = st[(k * D + i) * NM_BLOCK + lane];
        S.f[k] = st[((D + 1) * D + k) * NM_BLOCK + lane];
    }
}
template <int D> __device__ __forceinline__ void nm_store(double *st, int lane, const NmSimplex<D> &S)
{
#pragma unroll
    for (int k = 0; k <= D; k++) {
#pragma unroll
        for (int i = 0; i < D; i++) st[(k * D + i) * NM_BLOCK + lane] = S.x[k][i];
        st[((D + 1) * D + k) * NM_BLOCK + lane] = S.f[k];
    }
}
// after a pass: what the store holds must be READ again (the registers of the simplex belong to the pass in between)
__device__ __forceinline__ void nm_fence() { __asm__ volatile("" ::: "memory"); }

// A run-time vertex index (the INIT / SHRINK sweeps of the sequential driver, the lane roles of the two-level one) addresses the
// STORE, where it is an address computation -- on the register copy it would be a select chain over loads, which the optimiser
// turns back into an indexed access and then keeps the whole simplex in scratch memory.
template <int D> __device__ __forceinline__ double nm_store_x(const double *st, int lane, int k, int i) { return st[(k * D + i) * NM_BLOCK + lane]; }
template <int D> __device__ __forceinline__ void nm_store_set_f(double *st, int lane, int k, double f) { st[((D + 1) * D + k) * NM_BLOCK + lane] = f; }

template <int D> __device__ __forceinline__ bool nm_converged(const NmSimplex<D> &S)
{
    bool small = true;
#pragma unroll
    for (int k = 1; k <= D; k++) {
#pragma unroll
        for (int i = 0; i < D; i++)
            if (!(fabs(S.x[k][i] - S.x[0][i]) <= 1.0e-4)) small = false;
        if (!(fabs(S.f[0] - S.f[k]) <= 1.0e-8)) small = false;
    }
    return small;
}

// trial point `which` (0 reflection, 1 expansion, 2 outside, 3 inside contraction), coordinate i
template <int D> __device__ __forceinline__ double nm_trial(const NmSimplex<D> &S, int which, int i, double lo, double hi)
{
    double s = S.x[0][i];
#pragma unroll
    for (int k = 1; k < D; k++) s = s + S.x[k][i];
    const double xb = s / (double)D;
    const double xw = S.x[D][i];
    const double a = which == 0 ? 2.0 : (which == 1 ? 3.0 : (which == 2 ? 1.5 : 0.5));
    const double b = which == 0 ? 1.0 : (which == 1 ? 2.0 : 0.5);
    const double v = which == 3 ? a * xb + b * xw : a * xb - b * xw;
    return nm_clip(v, lo, hi);
}

// replace the worst vertex by trial point `which` with value fnew, stable re-insertion
template <int D> __device__ __forceinline__ void nm_accept(NmSimplex<D> &S, int which, double fnew, const double (&lo)[D], const double (&hi)[D])
{
    double xn[D];
#pragma unroll
    for (int i = 0; i < D; i++) xn[i] = nm_trial<D>(S, which, i, lo[i], hi[i]);
#pragma unroll
    for (int i = 0; i < D; i++) S.x[D][i] = xn[i];
    S.f[D] = fnew;
#pragma unroll
    for (int k = D; k >= 1; k--) {
        const bool sw = S.f[k] < S.f[k - 1];
        const double t = S.f[k];
        S.f[k] = sw ? S.f[k - 1] : t;
        S.f[k - 1] = sw ? t : S.f[k - 1];
#pragma unroll
        for (int i = 0; i < D; i++) {
            const double u = S.x[k][i];
            S.x[k][i] = sw ? S.x[k - 1][i] : u;
            S.x[k - 1][i] = sw ? u : S.x[k - 1][i];
        }
    }
}

template <int D> __device__ __forceinline__ void nm_shrink_vertices(NmSimplex<D> &S, const double (&lo)[D], const double (&hi)[D])
{
#pragma unroll
    for (int k = 1; k <= D; k++)
#pragma unroll
        for (int i = 0; i < D; i++)
            S.x[k][i] = nm_clip(S.x[0][i] + 0.5 * (S.x[k][i] - S.x[0][i]), lo[i], hi[i]);
}

template <int D> __device__ __forceinline__ void nm_sort_all(NmSimplex<D> &S)
{
#pragma unroll
    for (int k = 1; k <= D; k++) {
#pragma unroll
        for (int j = k; j >= 1; j--) {
            const bool sw = S.f[j] < S.f[j - 1];
            const double t = S.f[j];
            S.f[j] = sw ? S.f[j - 1] : t;
            S.f[j - 1] = sw ? t : S.f[j - 1];
#pragma unroll
            for (int i = 0; i < D; i++) {
                const double u = S.x[j][i];
                S.x[j][i] = sw ? S.x[j - 1][i] : u;
                S.x[j - 1][i] = sw ? u : S.x[j - 1][i];
            }
        }
    }
}

// Model concept:
//   static constexpr int DIM;
//   __device__ void bounds(double (&lo)[DIM], double (&hi)[DIM], double (&x0)[DIM]);
//   __device__ void eval(const double (&cand)[NM_K][DIM], double (&f)[NM_K]);   // one pass, four trial points
//   __device__ double eval1(const double (&x)[DIM]);                             // one pass, one point
// `st` is the wave's simplex store (nm_lds_doubles<DIM>() doubles: LDS, or global scratch).
template <class Model>
__device__ void nm_init_simplex(Model &mdl, double *st, NmRun &r, bool active)
{
    constexpr int D = Model::DIM;
    const int lane = threadIdx.x & (NM_BLOCK - 1);
    double lo[D], hi[D], x0[D];
    mdl.bounds(lo, hi, x0);
    NmSimplex<D> S;
#pragma unroll
    for (int i = 0; i < D; i++) S.x[0][i] = nm_clip(x0[i], lo[i], hi[i]);
#pragma unroll
    for (int k = 0; k < D; k++) {
#pragma unroll
        for (int i = 0; i < D; i++) S.x[k + 1][i] = S.x[0][i];
        double v = S.x[0][k];
        v = (v != 0.0) ? (1.0 + 0.05) * v : 0.00025;
        S.x[k + 1][k] = nm_clip(v, lo[k], hi[k]);
    }
#pragma unroll
    for (int k = 0; k <= D; k++) S.f[k] = 0.0;
    nm_store<D>(st, lane, S);
    r.phase = NM_INIT0;
    r.evals = 0;
    r.iters = 1;
    r.passes = 0;
    r.done = !active;
}

// ---- speculative driver: one pass per iteration --------------------------------------------------
template <class Model>
__device__ void nm_advance_spec(Model &mdl, double *st, NmRun &r, int budget)
{
    constexpr int D = Model::DIM;
    const int lane = threadIdx.x & (NM_BLOCK - 1);
    double lo[D], hi[D], x0[D];
    mdl.bounds(lo, hi, x0);
    const int maxiter = 200 * D, maxfun = 200 * D;
    int phase = r.phase, evals = r.evals, iters = r.iters, passes = r.passes;
    bool done = r.done, parked = false;
    double cand[NM_K][D], fc[NM_K];

    for (int pass = 0;; pass++) {
        if (!done && !parked) {
            NmSimplex<D> S;
            nm_load<D>(st, lane, S);
            if (phase == NM_INIT0) {
#pragma unroll
                for (int k = 0; k < NM_K; k++)
#pragma unroll
                    for (int i = 0; i < D; i++) cand[k][i] = S.x[k <= D ? k : D][i];
            } else if (phase == NM_INIT1) {
#pragma unroll
                for (int i = 0; i < D; i++) cand[0][i] = S.x[D][i];
            } else if (phase == NM_ITER) {
                if (!(evals < maxfun && iters < maxiter)) done = true;
                else if (nm_converged<D>(S)) done = true;
                if (!done) {
#pragma unroll
                    for (int k = 0; k < NM_K; k++)
#pragma unroll
                        for (int i = 0; i < D; i++) cand[k][i] = nm_trial<D>(S, k, i, lo[i], hi[i]);
                }
            } else { // NM_SHRINK: vertices 1..D already contracted towards the best
#pragma unroll
                for (int k = 0; k < NM_K; k++)
#pragma unroll
                    for (int i = 0; i < D; i++) cand[k][i] = S.x[k + 1 <= D ? k + 1 : D][i];
            }
        }
        // once the budget is used a lane parks at its next iteration boundary (INIT1 / SHRINK first finish
        // their iteration); the wave leaves when every lane is done or parked
        if (!done && pass >= budget && (phase == NM_ITER || phase == NM_INIT0)) parked = true;
        if (__all(done || parked)) break;

        mdl.eval(cand, fc);
        passes += (done || parked) ? 0 : 1;
        nm_fence();

        if (!done && !parked) {
            NmSimplex<D> S;
            nm_load<D>(st, lane, S);
            if (phase == NM_INIT0) {
#pragma unroll
                for (int k = 0; k < NM_K; k++)
                    if (k <= D) S.f[k] = fc[k];
                evals += (D + 1 < NM_K ? D + 1 : NM_K);
                phase = (D + 1 > NM_K) ? NM_INIT1 : NM_ITER;
                if (phase == NM_ITER) nm_sort_all<D>(S);
            } else if (phase == NM_INIT1) {
                S.f[D] = fc[0];
                evals += 1;
                phase = NM_ITER;
                nm_sort_all<D>(S);
            } else if (phase == NM_ITER) {
                const double fxr = fc[0];
                evals += 1;
                bool shrink = false;
                int which = 0;
                double fnew = fxr;
                if (fxr < S.f[0]) {
                    evals += 1;
                    if (fc[1] < fxr) { which = 1; fnew = fc[1]; }
                } else if (fxr < S.f[D - 1]) {
                    which = 0;
                } else if (fxr < S.f[D]) {
                    evals += 1;
                    if (fc[2] <= fxr) { which = 2; fnew = fc[2]; } else shrink = true;
                } else {
                    evals += 1;
                    if (fc[3] < S.f[D]) { which = 3; fnew = fc[3]; } else shrink = true;
                }
                if (!shrink) {
                    nm_accept<D>(S, which, fnew, lo, hi);
                    iters += 1;
                } else {
                    nm_shrink_vertices<D>(S, lo, hi);
                    phase = NM_SHRINK;
                }
            } else { // NM_SHRINK results
#pragma unroll
                for (int k = 0; k < NM_K; k++)
                    if (k + 1 <= D) S.f[k + 1] = fc[k];
                evals += D;
                iters += 1;
                phase = NM_ITER;
                nm_sort_all<D>(S);
            }
            nm_store<D>(st, lane, S);
        }
    }
    r.phase = phase; r.evals = evals; r.iters = iters; r.passes = passes; r.done = done;
}

// ---- two-level speculative driver: TWO iterations per pass, one problem per wave -------------------
// When only a few problems are left, their dependent chains of passes ARE the run time of the batch while most of the chip
// idles.  Lanes 0..3 evaluate the four trial points of the current iteration; lanes 4.. evaluate the four trial points of
// the NEXT iteration under every outcome the current one can have: the accepted point P (reflection, expansion, outside or
// inside contraction) and its rank j in the re-sorted simplex (the rank decides the order of the centroid sum, so it is
// part of the hypothesis): (E, 0), (R, 0..D-1), (OC, 0..D), (IC, 0..D) = 3 D + 3 hypotheses x 4 points = 60 lanes for D = 4.
// After the pass the first iteration is decided exactly as the sequential method decides it, the hypothesis that came true
// selects four of the speculative values, and the second iteration is decided from them -- same iterates, same evaluation
// counts, same stopping point (a shrink, or a stop after the first iteration, simply discards the second level).
// Every lane keeps an identical copy of the simplex in its own store slot, so all control flow is wave-uniform.
template <int D> __device__ __forceinline__ void nm_hypothesis(int hh, int &P, int &j)
{
    if (hh == 0) { P = 1; j = 0; }
    else if (hh <= D) { P = 0; j = hh - 1; }
    else if (hh <= 2 * D + 1) { P = 2; j = hh - D - 1; }
    else { P = 3; j = hh - 2 * D - 2; }
}

// coordinate i of trial point `which` of the simplex that results from accepting trial point P at rank j
template <int D> __device__ __forceinline__ double nm_trial2(const NmSimplex<D> &S, int P, int j, int which, int i, double lo, double hi)
{
    const double xp = nm_trial<D>(S, P, i, lo, hi);
    double s = 0.0;
#pragma unroll
    for (int pos = 0; pos < D; pos++) {
        // vertex at position pos of the hypothetical sorted simplex (the D best): the old ones with P inserted at rank j
        double v;
        if (j == D) v = S.x[pos][i];
        else {
            const double before = S.x[pos][i];
            const double after = pos >= 1 ? S.x[pos - 1][i] : before;
            v = pos < j ? before : (pos == j ? xp : after);
        }
        s = pos == 0 ? v : s + v;
    }
    const double xb = s / (double)D;
    const double xw = j == D ? xp : S.x[D - 1][i];
    const double a = which == 0 ? 2.0 : (which == 1 ? 3.0 : (which == 2 ? 1.5 : 0.5));
    const double b = which == 0 ? 1.0 : (which == 1 ? 2.0 : 0.5);
    const double t = which == 3 ? a * xb + b * xw : a * xb - b * xw;
    return nm_clip(t, lo, hi);
}

// decide one iteration from the values of its four trial points (the rules of nm_advance_spec); returns false for a shrink
template <int D> __device__ __forceinline__ bool nm_decide(const NmSimplex<D> &S, const double (&fc)[NM_K], int &evals, int &which, double &fnew)
{
    const double fxr = fc[0];
    evals += 1;
    which = 0;
    fnew = fxr;
    if (fxr < S.f[0]) {
        evals += 1;
        if (fc[1] < fxr) { which = 1; fnew = fc[1]; }
        return true;
    }
    if (fxr < S.f[D - 1]) return true;
    if (fxr < S.f[D]) {
        evals += 1;
        if (fc[2] <= fxr) { which = 2; fnew = fc[2]; return true; }
        return false;
    }
    evals += 1;
    if (fc[3] < S.f[D]) { which = 3; fnew = fc[3]; return true; }
    return false;
}

template <class Model>
__device__ void nm_advance_spec2(Model &mdl, double *st, NmRun &r, int budget)
{
    constexpr int D = Model::DIM;
    constexpr int NH = 3 * D + 3;                 // hypotheses about the first iteration
    const int lane = threadIdx.x & (NM_BLOCK - 1);
    double lo[D], hi[D], x0[D];
    mdl.bounds(lo, hi, x0);
    const int maxiter = 200 * D, maxfun = 200 * D;
    int phase = r.phase, evals = r.evals, iters = r.iters, passes = r.passes;
    bool done = r.done, parked = false;
    const int hh = lane >= NM_K ? (lane - NM_K) >> 2 : 0, w2 = (lane - NM_K) & 3;
    int hp = 0, hj = 0;
    nm_hypothesis<D>(hh < NH ? hh : 0, hp, hj);
    double x[D];

    for (int pass = 0;; pass++) {
        if (!done && !parked) {
            NmSimplex<D> S;
            nm_load<D>(st, lane, S);
            // level-1 roles are those of the 4-lane driver (lanes >= 4 repeat lane 3's point outside NM_ITER)
            const int sub = lane < NM_K ? lane : NM_K - 1;
            if (phase == NM_INIT0) {
#pragma unroll
                for (int i = 0; i < D; i++) x[i] = nm_store_x<D>(st, lane, sub <= D ? sub : D, i);
            } else if (phase == NM_INIT1) {
#pragma unroll
                for (int i = 0; i < D; i++) x[i] = S.x[D][i];
            } else if (phase == NM_ITER) {
                if (!(evals < maxfun && iters < maxiter)) done = true;
                else if (nm_converged<D>(S)) done = true;
                if (!done) {
                    if (lane < NM_K) {
#pragma unroll
                        for (int i = 0; i < D; i++) x[i] = nm_trial<D>(S, lane, i, lo[i], hi[i]);
                    } else {
#pragma unroll
                        for (int i = 0; i < D; i++) x[i] = nm_trial2<D>(S, hp, hj, w2, i, lo[i], hi[i]);
                    }
                }
            } else { // NM_SHRINK
#pragma unroll
                for (int i = 0; i < D; i++) x[i] = nm_store_x<D>(st, lane, sub + 1 <= D ? sub + 1 : D, i);
            }
        }
        if (!done && pass >= budget && (phase == NM_ITER || phase == NM_INIT0)) parked = true;
        if (done || parked) break;                          // one problem per wave: the decision is wave-uniform

        const double f = mdl.eval1(x);
        passes += 1;
        nm_fence();
        double fc[NM_K];
#pragma unroll
        for (int k = 0; k < NM_K; k++) fc[k] = __shfl(f, k);

        NmSimplex<D> S;
        nm_load<D>(st, lane, S);
        if (phase == NM_INIT0) {
#pragma unroll
            for (int k = 0; k < NM_K; k++)
                if (k <= D) S.f[k] = fc[k];
            evals += (D + 1 < NM_K ? D + 1 : NM_K);
            phase = (D + 1 > NM_K) ? NM_INIT1 : NM_ITER;
            if (phase == NM_ITER) nm_sort_all<D>(S);
        } else if (phase == NM_INIT1) {
            S.f[D] = fc[0];
            evals += 1;
            phase = NM_ITER;
            nm_sort_all<D>(S);
        } else if (phase == NM_ITER) {
            int which = 0;
            double fnew = 0.0;
            if (!nm_decide<D>(S, fc, evals, which, fnew)) {
                nm_shrink_vertices<D>(S, lo, hi);
                phase = NM_SHRINK;
            } else {
                // the rank the accepted point takes (stable insertion, nm_accept) names the hypothesis that came true
                int j = D;
#pragma unroll
                for (int k = D; k >= 1; k--)
                    if (j == k && fnew < S.f[k - 1]) j = k - 1;
                const int hstar = which == 1 ? 0 : (which == 0 ? 1 + j : (which == 2 ? D + 1 + j : 2 * D + 2 + j));
                double fc2[NM_K];
#pragma unroll
                for (int k = 0; k < NM_K; k++) fc2[k] = __shfl(f, NM_K + 4 * hstar + k);
                nm_accept<D>(S, which, fnew, lo, hi);
                iters += 1;
                // second iteration: the checks the sequential loop makes before it, then the same decision
                if (!(evals < maxfun && iters < maxiter) || nm_converged<D>(S)) done = true;
                else if (NM_K + 4 * NH <= NM_BLOCK) {
                    if (!nm_decide<D>(S, fc2, evals, which, fnew)) {
                        nm_shrink_vertices<D>(S, lo, hi);
                        phase = NM_SHRINK;
                    } else {
                        nm_accept<D>(S, which, fnew, lo, hi);
                        iters += 1;
                    }
                }
            }
        } else { // NM_SHRINK results
#pragma unroll
            for (int k = 0; k < NM_K; k++)
                if (k + 1 <= D) S.f[k + 1] = fc[k];
            evals += D;
            iters += 1;
            phase = NM_ITER;
            nm_sort_all<D>(S);
        }
        nm_store<D>(st, lane, S);
    }
    r.phase = phase; r.evals = evals; r.iters = iters; r.passes = passes; r.done = done;
}

// ---- sequential driver: one trial point per pass ---------------------------------------------------
template <class Model>
__device__ void nm_advance_seq(Model &mdl, double *st, NmRun &r, int budget)
{
    constexpr int D = Model::DIM;
    const int lane = threadIdx.x & (NM_BLOCK - 1);
    double lo[D], hi[D], x0[D];
    mdl.bounds(lo, hi, x0);
    const int maxiter = 200 * D, maxfun = 200 * D;
    int phase = r.phase, evals = r.evals, iters = r.iters, passes = r.passes;
    bool done = r.done, parked = false;
    int vi = 0;            // vertex cursor of the INIT / SHRINK sweeps
    double fxr = 0.0;      // reflection value of the running iteration
    double x[D];

    for (int pass = 0;; pass++) {
        if (!done && !parked) {
            if (phase == NM_INIT0) { phase = NM_SEQ_INIT; vi = 0; }
            if (phase == NM_SEQ_INIT || phase == NM_SEQ_SHRINK) {
                const int k = phase == NM_SEQ_INIT ? vi : 1 + vi;
#pragma unroll
                for (int i = 0; i < D; i++) x[i] = nm_store_x<D>(st, lane, k, i);
            } else {
                NmSimplex<D> S;
                nm_load<D>(st, lane, S);
                if (phase == NM_ITER) {
                    if (!(evals < maxfun && iters < maxiter)) done = true;
                    else if (nm_converged<D>(S)) done = true;
                }
                const int which = phase == NM_ITER ? 0 : (phase == NM_SEQ_E ? 1 : (phase == NM_SEQ_OC ? 2 : 3));
                if (!done) {
#pragma unroll
                    for (int i = 0; i < D; i++) x[i] = nm_trial<D>(S, which, i, lo[i], hi[i]);
                }
            }
        }
        if (!done && pass >= budget && (phase == NM_ITER || phase == NM_INIT0)) parked = true;
        if (__all(done || parked)) break;

        const double f = mdl.eval1(x);
        passes += (done || parked) ? 0 : 1;
        nm_fence();

        if (!done && !parked) {
            if (phase == NM_SEQ_INIT || phase == NM_SEQ_SHRINK) {
                const bool init = phase == NM_SEQ_INIT;
                nm_store_set_f<D>(st, lane, init ? vi : 1 + vi, f);
                vi += 1;
                evals += 1;
                if (vi == (init ? D + 1 : D)) {
                    if (!init) iters += 1;
                    nm_fence();
                    NmSimplex<D> S;
                    nm_load<D>(st, lane, S);
                    nm_sort_all<D>(S);
                    nm_store<D>(st, lane, S);
                    phase = NM_ITER;
                }
            } else {
                NmSimplex<D> S;
                nm_load<D>(st, lane, S);
                bool changed = true;
                if (phase == NM_ITER) {
                    fxr = f;
                    evals += 1;
                    if (fxr < S.f[0]) { phase = NM_SEQ_E; changed = false; }
                    else if (fxr < S.f[D - 1]) { nm_accept<D>(S, 0, fxr, lo, hi); iters += 1; }
                    else if (fxr < S.f[D]) { phase = NM_SEQ_OC; changed = false; }
                    else { phase = NM_SEQ_IC; changed = false; }
                } else if (phase == NM_SEQ_E) {
                    evals += 1;
                    if (f < fxr) nm_accept<D>(S, 1, f, lo, hi);
                    else nm_accept<D>(S, 0, fxr, lo, hi);
                    iters += 1;
                    phase = NM_ITER;
                } else { // NM_SEQ_OC / NM_SEQ_IC
                    evals += 1;
                    const bool ok = (phase == NM_SEQ_OC) ? (f <= fxr) : (f < S.f[D]);
                    if (ok) {
                        nm_accept<D>(S, phase == NM_SEQ_OC ? 2 : 3, f, lo, hi);
                        iters += 1;
                        phase = NM_ITER;
                    } else {
                        nm_shrink_vertices<D>(S, lo, hi);
                        vi = 0;
                        phase = NM_SEQ_SHRINK;
                    }
                }
                if (changed) nm_store<D>(st, lane, S);
            }
        }
    }
    r.phase = phase; r.evals = evals; r.iters = iters; r.passes = passes; r.done = done;
}

// one-shot convenience (classic models): speculative driver run to completion
template <class Model>
__device__ void nm_minimize(Model &mdl, bool active, double *st, double (&xbest)[Model::DIM], double &fbest, NmStats &stats)
{
    constexpr int D = Model::DIM;
    const int lane = threadIdx.x & (NM_BLOCK - 1);
    NmRun r;
    nm_init_simplex(mdl, st, r, active);
    nm_advance_spec(mdl, st, r, 0x7fffffff);
#pragma unroll
    for (int i = 0; i < D; i++) xbest[i] = st[(0 * D + i) * NM_BLOCK + lane];
    fbest = st[((D + 1) * D + 0) * NM_BLOCK + lane];
    stats.iters = r.iters;
    stats.evals = r.evals;
    stats.passes = r.passes;
}

} // namespace anofox
