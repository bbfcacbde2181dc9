// ets_fit_kernel.hpp -- fit one ETS spec for every series of the batch.
//
// grid = ceil(n_series / 64) one-wave workgroups; lane <-> series.  Each lane runs its own
// Nelder-Mead (nm.hpp) over the smoothing parameters; every objective evaluation is a
// streamed pass over the lane's column of the time-major block (ets_device.hpp).  After
// convergence one more pass with the optimum produces the final states, the h forecasts and
// the information criteria.  Dominant cost: passes x 8 T bytes per series -> HBM/L2 stream +
// fp64 VALU recursion; no MFMA (scan, not a contraction).
#pragma once
#include "ets_device.hpp"
#include "kernels.hpp"

namespace anofox {

template <class Cfg, int MS>
__global__ __launch_bounds__(NM_BLOCK) void ets_fit_kernel(const FitArgs a)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int s = blockIdx.x * NM_BLOCK + lane;
    const int len = (s < a.n_series) ? a.len[s] : 0;

    // admissibility of this (series, spec) -- mirrors oracle ets_fit preconditions
    int st = FIT_OK;
    const uint32_t fl = (s < a.n_series) ? a.flags[s] : 0u;
    if (len <= 0) st = FIT_SKIPPED;
    else if (Cfg::S != C_NONE && len < 2 * a.m) st = FIT_SHORT;
    else if (len < a.n_param + 2) st = FIT_SHORT;
    else if (a.need_positive && !(fl & SF_POSITIVE)) st = FIT_NONPOSITIVE;
    else if (a.skip_constant && (fl & SF_CONSTANT)) st = FIT_SKIPPED;
    const bool active = (st == FIT_OK);

    SeriesView v;
    v.y = a.y + (s < a.n_series ? s : 0);
    v.ld = a.ld;
    v.len = active ? len : 0;
    v.wave_len = wave_max_i32(v.len);
    v.wave_min_len = wave_min_i32(active ? len : 0x7fffffff);
    if (v.wave_len == 0) {                  // nothing to do in this wave
        if (s < a.n_series) {
            a.status[s] = st; a.aicc[s] = __builtin_huge_val();
            a.evals[s] = 0; a.iters[s] = 0; a.passes[s] = 0;
        }
        return;
    }
    if (v.wave_min_len == 0x7fffffff) v.wave_min_len = 0;

    EtsModel<Cfg, MS> mdl;
    mdl.v = v;
    mdl.in.l0 = active ? a.l0[s] : 0.0;
    mdl.in.b0 = (active && Cfg::T != C_NONE) ? a.b0[s] : 0.0;
    mdl.in.fig = a.fig ? a.fig + (s < a.n_series ? s : 0) : nullptr;
    mdl.in.fig_ld = a.fig_ld;
    mdl.in.m = a.m;
    double *simplex = lds;
    mdl.ring = lds + nm_lds_doubles<Cfg::DIM>();

    double xbest[Cfg::DIM], fbest;
    NmStats ns;
    nm_minimize(mdl, active, simplex, xbest, fbest, ns);

    // final pass: candidate 0 = optimum; forecasts written straight from the live states
    double cand[NM_K][Cfg::DIM], f[NM_K];
#pragma unroll
    for (int k = 0; k < NM_K; k++)
#pragma unroll
        for (int i = 0; i < Cfg::DIM; i++) cand[k][i] = xbest[i];
    EtsFinalOut fin;
    fin.h = a.h;
    fin.yhat = a.yhat + (size_t)(s < a.n_series ? s : 0) * a.h;
    fin.sse_out = nullptr;
    SeriesView vf = v;                     // inactive lanes must not write
    ets_pass<Cfg, MS, NM_K, true>(vf, mdl.in, cand, f, mdl.ring, &fin);

    if (s < a.n_series) {
        double aicc = __builtin_huge_val();
        if (active) {
            const double lik = f[0];
            if (!(fabs(lik) <= 1.7976931348623157e308)) st = FIT_NONFINITE;
            else {
                const double dk = (double)a.n_param, dn = (double)len;
                const double aic = lik + 2.0 * dk;
                aicc = aic + 2.0 * dk * (dk + 1.0) / (dn - dk - 1.0);
            }
        }
        a.status[s] = st;
        a.aicc[s] = aicc;
        a.evals[s] = active ? ns.evals : 0;
        a.iters[s] = active ? ns.iters : 0;
        a.passes[s] = active ? ns.passes + 1 : 0;
    }
}

template <class Cfg, int MS>
void ets_fit_launch(const FitArgs &a, hipStream_t stream)
{
    const int grid = (a.n_series + NM_BLOCK - 1) / NM_BLOCK;
    size_t lds_bytes = sizeof(double) * (size_t)nm_lds_doubles<Cfg::DIM>();
    if (MS == -1) lds_bytes += sizeof(double) * (size_t)NM_K * (size_t)a.m * NM_BLOCK;
    if (lds_bytes > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)ets_fit_kernel<Cfg, MS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_bytes);
    hipLaunchKernelGGL((ets_fit_kernel<Cfg, MS>), dim3(grid), dim3(NM_BLOCK), lds_bytes, stream, a);
}

} // namespace anofox
