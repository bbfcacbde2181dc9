// nm.hpp -- per-lane bounded Nelder-Mead for gfx950, one optimisation problem per lane.
//
// Semantics: scipy-style Nelder-Mead (rho 1, chi 2, psi 0.5, sigma 0.5; initial simplex
// x0 and x0 with one coordinate scaled by 1.05; bounds by clipping; stop when the simplex
// is within xatol = 1e-4 AND the values within fatol = 1e-8, or after 200*dim iterations /
// evaluations).  These constants are what the reference's known-answer vectors pin for the
// SES / Holt / HoltWinters optimisers.
//
// MI355X mapping: an objective evaluation is one streamed pass over the lane's series, so
// the four trial points of an iteration (reflection, expansion, outside and inside
// contraction) -- all functions of the centroid and the worst vertex only -- are evaluated
// SPECULATIVELY in the same pass (K = 4 candidates per lane share every y load).  The
// accept/shrink decision then uses exactly the values a sequential Nelder-Mead would have
// computed, so the trajectory is identical to the sequential algorithm; `evals` counts the
// sequential evaluations, `passes` the streamed passes.  The simplex lives in LDS between
// passes so that the pass itself owns the VGPR budget.
//
// The optimiser is RESUMABLE: nm_advance() runs at most `budget` passes and leaves a state
// (simplex in LDS + NmRun) that a later launch continues bit-for-bit.  The ETS fit uses this to
// run in rounds with compaction of the unfinished problems in between (iteration counts differ
// by 5-10x between series, so a run-to-completion wave idles most of its lanes).
#pragma once
#include <hip/hip_runtime.h>

namespace anofox {

constexpr int NM_K = 4;          // candidates evaluated per streamed pass
constexpr int NM_BLOCK = 64;     // one wave per workgroup: all wave-uniform loops need no barrier

__device__ __forceinline__ double nm_clip(double v, double lo, double hi)
{
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}

enum { NM_INIT0 = 0, NM_INIT1 = 1, NM_ITER = 2, NM_SHRINK = 3 };

struct NmRun { int phase; int evals; int iters; int passes; bool done; };
struct NmStats { int iters; int evals; int passes; };

// LDS footprint (doubles) of one wave's simplex store.
template <int D> constexpr int nm_lds_doubles() { return ((D + 1) * D + (D + 1)) * NM_BLOCK; }

#define ANOFOX_SIM(k, i) lds[((k) * D + (i)) * NM_BLOCK + lane]
#define ANOFOX_FS(k) lds[((D + 1) * D + (k)) * NM_BLOCK + lane]

// Model concept:
//   static constexpr int DIM;
//   __device__ void bounds(double (&lo)[DIM], double (&hi)[DIM], double (&x0)[DIM]);
//   __device__ void eval(const double (&cand)[NM_K][DIM], double (&f)[NM_K]);   // one streamed pass
template <class Model>
__device__ void nm_init_simplex(Model &mdl, double *lds, NmRun &r, bool active)
{
    constexpr int D = Model::DIM;
    const int lane = threadIdx.x;
    double lo[D], hi[D], x0[D];
    mdl.bounds(lo, hi, x0);
#pragma unroll
    for (int i = 0; i < D; i++) ANOFOX_SIM(0, i) = nm_clip(x0[i], lo[i], hi[i]);
#pragma unroll
    for (int k = 0; k < D; k++) {
#pragma unroll
        for (int i = 0; i < D; i++) ANOFOX_SIM(k + 1, i) = ANOFOX_SIM(0, i);
        double v = ANOFOX_SIM(0, k);
        v = (v != 0.0) ? (1.0 + 0.05) * v : 0.00025;
        ANOFOX_SIM(k + 1, k) = nm_clip(v, lo[k], hi[k]);
    }
#pragma unroll
    for (int k = 0; k <= D; k++) ANOFOX_FS(k) = 0.0;
    r.phase = NM_INIT0;
    r.evals = 0;
    r.iters = 1;
    r.passes = 0;
    r.done = !active;
}

template <class Model>
__device__ void nm_advance(Model &mdl, double *lds, NmRun &r, int budget)
{
    constexpr int D = Model::DIM;
    const int lane = threadIdx.x;
    double lo[D], hi[D], x0[D];
    mdl.bounds(lo, hi, x0);
    const int maxiter = 200 * D, maxfun = 200 * D;
    int phase = r.phase, evals = r.evals, iters = r.iters, passes = r.passes;
    bool done = r.done;
    double cand[NM_K][D], fc[NM_K];

    for (int pass = 0;; pass++) {
        if (!done) {
            if (phase == NM_INIT0) {
#pragma unroll
                for (int k = 0; k < NM_K; k++)
#pragma unroll
                    for (int i = 0; i < D; i++) cand[k][i] = ANOFOX_SIM(k <= D ? k : D, i);
            } else if (phase == NM_INIT1) {
#pragma unroll
                for (int i = 0; i < D; i++) cand[0][i] = ANOFOX_SIM(D, i);
            } else if (phase == NM_ITER) {
                if (!(evals < maxfun && iters < maxiter)) done = true;
                else {
                    bool small = true;
#pragma unroll
                    for (int k = 1; k <= D; k++) {
#pragma unroll
                        for (int i = 0; i < D; i++)
                            if (!(fabs(ANOFOX_SIM(k, i) - ANOFOX_SIM(0, i)) <= 1.0e-4)) small = false;
                        if (!(fabs(ANOFOX_FS(0) - ANOFOX_FS(k)) <= 1.0e-8)) small = false;
                    }
                    if (small) done = true;
                }
                if (!done) {
#pragma unroll
                    for (int i = 0; i < D; i++) {
                        double s = ANOFOX_SIM(0, i);
#pragma unroll
                        for (int k = 1; k < D; k++) s = s + ANOFOX_SIM(k, i);
                        double xb = s / (double)D;
                        double xw = ANOFOX_SIM(D, i);
                        cand[0][i] = nm_clip(2.0 * xb - xw, lo[i], hi[i]);        // reflection
                        cand[1][i] = nm_clip(3.0 * xb - 2.0 * xw, lo[i], hi[i]);  // expansion
                        cand[2][i] = nm_clip(1.5 * xb - 0.5 * xw, lo[i], hi[i]);  // outside contraction
                        cand[3][i] = nm_clip(0.5 * xb + 0.5 * xw, lo[i], hi[i]);  // inside contraction
                    }
                }
            } else { // NM_SHRINK: vertices 1..D already contracted towards the best
#pragma unroll
                for (int k = 0; k < NM_K; k++)
#pragma unroll
                    for (int i = 0; i < D; i++) cand[k][i] = ANOFOX_SIM(k + 1 <= D ? k + 1 : D, i);
            }
        }
        if (__all(done) || pass >= budget) break;

        mdl.eval(cand, fc);
        passes += done ? 0 : 1;

        if (!done) {
            bool need_sort = false;
            if (phase == NM_INIT0) {
#pragma unroll
                for (int k = 0; k < NM_K; k++)
                    if (k <= D) ANOFOX_FS(k) = fc[k];
                evals += (D + 1 < NM_K ? D + 1 : NM_K);
                phase = (D + 1 > NM_K) ? NM_INIT1 : NM_ITER;
                need_sort = (phase == NM_ITER);
            } else if (phase == NM_INIT1) {
                ANOFOX_FS(D) = fc[0];
                evals += 1;
                phase = NM_ITER;
                need_sort = true;
            } else if (phase == NM_ITER) {
                const double fxr = fc[0];
                evals += 1;
                bool shrink = false, take = true;
                int which = 0; // 0 reflection, 1 expansion, 2 outside, 3 inside contraction
                double fnew = fxr;
                if (fxr < ANOFOX_FS(0)) {
                    evals += 1;
                    if (fc[1] < fxr) { which = 1; fnew = fc[1]; }
                } else if (fxr < ANOFOX_FS(D - 1)) {
                    which = 0;
                } else if (fxr < ANOFOX_FS(D)) {
                    evals += 1;
                    if (fc[2] <= fxr) { which = 2; fnew = fc[2]; } else { shrink = true; take = false; }
                } else {
                    evals += 1;
                    if (fc[3] < ANOFOX_FS(D)) { which = 3; fnew = fc[3]; } else { shrink = true; take = false; }
                }
                if (take) {
                    // the trial point is recomputed (same operations, same bits) rather than kept live
                    // across the streamed pass: 32 fewer VGPRs inside the hot loop
#pragma unroll
                    for (int i = 0; i < D; i++) {
                        double s = ANOFOX_SIM(0, i);
#pragma unroll
                        for (int k = 1; k < D; k++) s = s + ANOFOX_SIM(k, i);
                        double xb = s / (double)D;
                        double xw = ANOFOX_SIM(D, i);
                        double a = which == 0 ? 2.0 : (which == 1 ? 3.0 : (which == 2 ? 1.5 : 0.5));
                        double b = which == 0 ? 1.0 : (which == 1 ? 2.0 : 0.5);
                        double v = which == 3 ? a * xb + b * xw : a * xb - b * xw;
                        ANOFOX_SIM(D, i) = nm_clip(v, lo[i], hi[i]);
                    }
                    ANOFOX_FS(D) = fnew;
                    // stable re-insertion of the last vertex
#pragma unroll
                    for (int k = D; k >= 1; k--) {
                        if (ANOFOX_FS(k) < ANOFOX_FS(k - 1)) {
                            double t = ANOFOX_FS(k); ANOFOX_FS(k) = ANOFOX_FS(k - 1); ANOFOX_FS(k - 1) = t;
#pragma unroll
                            for (int i = 0; i < D; i++) {
                                double u = ANOFOX_SIM(k, i); ANOFOX_SIM(k, i) = ANOFOX_SIM(k - 1, i); ANOFOX_SIM(k - 1, i) = u;
                            }
                        }
                    }
                    iters += 1;
                }
                if (shrink) {
#pragma unroll
                    for (int k = 1; k <= D; k++)
#pragma unroll
                        for (int i = 0; i < D; i++)
                            ANOFOX_SIM(k, i) = nm_clip(ANOFOX_SIM(0, i) + 0.5 * (ANOFOX_SIM(k, i) - ANOFOX_SIM(0, i)), lo[i], hi[i]);
                    phase = NM_SHRINK;
                }
            } else { // NM_SHRINK results
#pragma unroll
                for (int k = 0; k < NM_K; k++)
                    if (k + 1 <= D) ANOFOX_FS(k + 1) = fc[k];
                evals += D;
                iters += 1;
                phase = NM_ITER;
                need_sort = true;
            }
            if (need_sort) {
                // stable insertion sort of all vertices by value
#pragma unroll
                for (int k = 1; k <= D; k++) {
#pragma unroll
                    for (int j = k; j >= 1; j--) {
                        if (ANOFOX_FS(j) < ANOFOX_FS(j - 1)) {
                            double t = ANOFOX_FS(j); ANOFOX_FS(j) = ANOFOX_FS(j - 1); ANOFOX_FS(j - 1) = t;
#pragma unroll
                            for (int i = 0; i < D; i++) {
                                double u = ANOFOX_SIM(j, i); ANOFOX_SIM(j, i) = ANOFOX_SIM(j - 1, i); ANOFOX_SIM(j - 1, i) = u;
                            }
                        }
                    }
                }
            }
        }
    }
    r.phase = phase; r.evals = evals; r.iters = iters; r.passes = passes; r.done = done;
}

// one-shot convenience (classic models): run to completion
template <class Model>
__device__ void nm_minimize(Model &mdl, bool active, double *lds, double (&xbest)[Model::DIM], double &fbest, NmStats &stats)
{
    constexpr int D = Model::DIM;
    const int lane = threadIdx.x;
    NmRun r;
    nm_init_simplex(mdl, lds, r, active);
    nm_advance(mdl, lds, r, 0x7fffffff);
#pragma unroll
    for (int i = 0; i < D; i++) xbest[i] = ANOFOX_SIM(0, i);
    fbest = ANOFOX_FS(0);
    stats.iters = r.iters;
    stats.evals = r.evals;
    stats.passes = r.passes;
}

} // namespace anofox
