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

struct NmStats { int iters; int evals; int passes; };

// LDS footprint (doubles) of one wave's simplex store.
template <int D> constexpr int nm_lds_doubles() { return ((D + 1) * D + (D + 1)) * NM_BLOCK; }

// Model concept:
//   static constexpr int DIM;
//   __device__ void bounds(double (&lo)[DIM], double (&hi)[DIM], double (&x0)[DIM]);
//   __device__ void eval(const double (&cand)[NM_K][DIM], double (&f)[NM_K]);   // one streamed pass
template <class Model>
__device__ void nm_minimize(Model &mdl, bool active, double *lds /* nm_lds_doubles<DIM>() */,
                            double (&xbest)[Model::DIM], double &fbest, NmStats &stats)
{
    constexpr int D = Model::DIM;
    const int lane = threadIdx.x;
#define SIM(k, i) lds[((k) * D + (i)) * NM_BLOCK + lane]
#define FS(k) lds[((D + 1) * D + (k)) * NM_BLOCK + lane]

    double lo[D], hi[D], x0[D];
    mdl.bounds(lo, hi, x0);
#pragma unroll
    for (int i = 0; i < D; i++) SIM(0, i) = nm_clip(x0[i], lo[i], hi[i]);
#pragma unroll
    for (int k = 0; k < D; k++) {
#pragma unroll
        for (int i = 0; i < D; i++) SIM(k + 1, i) = SIM(0, i);
        double v = SIM(0, k);
        v = (v != 0.0) ? (1.0 + 0.05) * v : 0.00025;
        SIM(k + 1, k) = nm_clip(v, lo[k], hi[k]);
    }

    const int maxiter = 200 * D, maxfun = 200 * D;
    int evals = 0, iters = 1, passes = 0;
    enum { INIT0, INIT1, ITER, SHRINK };
    int phase = INIT0;
    bool done = !active;
    double cand[NM_K][D], fc[NM_K];

    for (;;) {
        if (!done) {
            if (phase == INIT0) {
#pragma unroll
                for (int k = 0; k < NM_K; k++)
#pragma unroll
                    for (int i = 0; i < D; i++) cand[k][i] = SIM(k <= D ? k : D, i);
            } else if (phase == INIT1) {
#pragma unroll
                for (int i = 0; i < D; i++) cand[0][i] = SIM(D, i);
            } else if (phase == ITER) {
                if (!(evals < maxfun && iters < maxiter)) done = true;
                else {
                    bool small = true;
#pragma unroll
                    for (int k = 1; k <= D; k++) {
#pragma unroll
                        for (int i = 0; i < D; i++)
                            if (!(fabs(SIM(k, i) - SIM(0, i)) <= 1.0e-4)) small = false;
                        if (!(fabs(FS(0) - FS(k)) <= 1.0e-8)) small = false;
                    }
                    if (small) done = true;
                }
                if (!done) {
#pragma unroll
                    for (int i = 0; i < D; i++) {
                        double s = SIM(0, i);
#pragma unroll
                        for (int k = 1; k < D; k++) s = s + SIM(k, i);
                        double xb = s / (double)D;
                        double xw = SIM(D, i);
                        cand[0][i] = nm_clip(2.0 * xb - xw, lo[i], hi[i]);        // reflection
                        cand[1][i] = nm_clip(3.0 * xb - 2.0 * xw, lo[i], hi[i]);  // expansion
                        cand[2][i] = nm_clip(1.5 * xb - 0.5 * xw, lo[i], hi[i]);  // outside contraction
                        cand[3][i] = nm_clip(0.5 * xb + 0.5 * xw, lo[i], hi[i]);  // inside contraction
                    }
                }
            } else { // SHRINK: vertices 1..D already contracted towards the best
#pragma unroll
                for (int k = 0; k < NM_K; k++)
#pragma unroll
                    for (int i = 0; i < D; i++) cand[k][i] = SIM(k + 1 <= D ? k + 1 : D, i);
            }
        }
        if (__all(done)) break;

        mdl.eval(cand, fc);
        passes += done ? 0 : 1;

        if (!done) {
            bool need_sort = false;
            if (phase == INIT0) {
#pragma unroll
                for (int k = 0; k < NM_K; k++)
                    if (k <= D) FS(k) = fc[k];
                evals += (D + 1 < NM_K ? D + 1 : NM_K);
                phase = (D + 1 > NM_K) ? INIT1 : ITER;
                need_sort = (phase == ITER);
            } else if (phase == INIT1) {
                FS(D) = fc[0];
                evals += 1;
                phase = ITER;
                need_sort = true;
            } else if (phase == ITER) {
                const double fxr = fc[0];
                evals += 1;
                bool shrink = false, take = true;
                int which = 0; // 0 xr, 1 xe, 2 xc, 3 xcc
                double fnew = fxr;
                if (fxr < FS(0)) {
                    evals += 1;
                    if (fc[1] < fxr) { which = 1; fnew = fc[1]; }
                } else if (fxr < FS(D - 1 >= 0 ? D - 1 : 0)) {
                    which = 0;
                } else if (fxr < FS(D)) {
                    evals += 1;
                    if (fc[2] <= fxr) { which = 2; fnew = fc[2]; } else { shrink = true; take = false; }
                } else {
                    evals += 1;
                    if (fc[3] < FS(D)) { which = 3; fnew = fc[3]; } else { shrink = true; take = false; }
                }
                if (take) {
                    // the trial points are recomputed (same operations, same bits) rather than
                    // kept live across the streamed pass: 32 fewer VGPRs inside the hot loop
#pragma unroll
                    for (int i = 0; i < D; i++) {
                        double s = SIM(0, i);
#pragma unroll
                        for (int k = 1; k < D; k++) s = s + SIM(k, i);
                        double xb = s / (double)D;
                        double xw = SIM(D, i);
                        double a = which == 0 ? 2.0 : (which == 1 ? 3.0 : (which == 2 ? 1.5 : 0.5));
                        double b = which == 0 ? 1.0 : (which == 1 ? 2.0 : 0.5);
                        double v = which == 3 ? a * xb + b * xw : a * xb - b * xw;
                        SIM(D, i) = nm_clip(v, lo[i], hi[i]);
                    }
                    FS(D) = fnew;
                    // stable re-insertion of the last vertex
#pragma unroll
                    for (int k = D; k >= 1; k--) {
                        if (FS(k) < FS(k - 1)) {
                            double t = FS(k); FS(k) = FS(k - 1); FS(k - 1) = t;
#pragma unroll
                            for (int i = 0; i < D; i++) { double u = SIM(k, i); SIM(k, i) = SIM(k - 1, i); SIM(k - 1, i) = u; }
                        }
                    }
                    iters += 1;
                }
                if (shrink) {
#pragma unroll
                    for (int k = 1; k <= D; k++)
#pragma unroll
                        for (int i = 0; i < D; i++)
                            SIM(k, i) = nm_clip(SIM(0, i) + 0.5 * (SIM(k, i) - SIM(0, i)), lo[i], hi[i]);
                    phase = SHRINK;
                }
            } else { // SHRINK results
#pragma unroll
                for (int k = 0; k < NM_K; k++)
                    if (k + 1 <= D) FS(k + 1) = fc[k];
                evals += D;
                iters += 1;
                phase = ITER;
                need_sort = true;
            }
            if (need_sort) {
                // stable insertion sort of all vertices by value
#pragma unroll
                for (int k = 1; k <= D; k++) {
#pragma unroll
                    for (int j = k; j >= 1; j--) {
                        if (FS(j) < FS(j - 1)) {
                            double t = FS(j); FS(j) = FS(j - 1); FS(j - 1) = t;
#pragma unroll
                            for (int i = 0; i < D; i++) { double u = SIM(j, i); SIM(j, i) = SIM(j - 1, i); SIM(j - 1, i) = u; }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < D; i++) xbest[i] = SIM(0, i);
    fbest = FS(0);
    stats.iters = iters;
    stats.evals = evals;
    stats.passes = passes;
#undef SIM
#undef FS
}

} // namespace anofox
