// SES / Holt / additive Holt-Winters / SeasonalES with optimised parameters on the ETS round kernels (ets_fit_kernel.hpp with
// ClassicCfg<KIND>): resumable Nelder-Mead rounds, compaction between them, the three drivers behind the device-side choice --
// instead of one lane running its Nelder-Mead to completion (classic_kernel: the slowest lane of a wave and one conditional
// row load per step set the pace: Holt-Winters on the M5 block 338 ms; 182 ms with prefetched rows).  Plus the family's final
// pass: forecasts from the optimum the rounds parked in the simplex.
#include "ets_fit_kernel.hpp"

namespace anofox {

template <int KIND, int MS> FitLaunchers classic_launchers_of()
{
    using Cfg = ClassicCfg<KIND>;
    return FitLaunchers{&ets_round_launch<Cfg, MS, 0>, &ets_round_launch<Cfg, MS, 1>, &ets_round_launch<Cfg, MS, 2>, &ets_round_launch<Cfg, MS, 3>, nullptr};
}

// MS: 0 no seasonal ring, 7 the weekly ring in registers, -1 ring of m x 64 doubles in LDS behind the simplex, -2 ring in the HBM
// scratch of the workgroup
FitLaunchers classic_fit_launcher(int kind, int m)
{
    switch (kind) {
    case CK_SES: return classic_launchers_of<CK_SES, 0>();
    case CK_HOLT: return classic_launchers_of<CK_HOLT, 0>();
    case CK_HW: return m == 7 ? classic_launchers_of<CK_HW, 7>() : (m > ETS_LDS_PERIOD ? classic_launchers_of<CK_HW, -2>() : classic_launchers_of<CK_HW, -1>());
    case CK_SEASONAL_ES:
        return m == 7 ? classic_launchers_of<CK_SEASONAL_ES, 7>() : (m > ETS_LDS_PERIOD ? classic_launchers_of<CK_SEASONAL_ES, -2>() : classic_launchers_of<CK_SEASONAL_ES, -1>());
    default: return FitLaunchers{nullptr, nullptr, nullptr, nullptr, nullptr};
    }
}

// One lane per series: the series the rounds fitted (status FIT_OK in a.status) get their forecasts from one more pass at
// the optimum (vertex 0 of the parked simplex); every series the caller's mask selects gets its status.
template <int KIND, bool GRING, bool M7 = false>
__global__ __launch_bounds__(NM_BLOCK) void classic_final_kernel(const FitArgs a, const ClassicArgs c)
{
    extern __shared__ double lds[];
    constexpr int DIM = ClassicDim<KIND>::value;
    const int s = blockIdx.x * NM_BLOCK + threadIdx.x;
    const bool valid = s < a.n_series;
    const int len = valid ? a.len[s] : 0;
    const bool selected = valid && len > 0 && (c.mask == nullptr || c.mask[s] == c.want);
    const int st = selected ? a.status[s] : FIT_SKIPPED;
    const bool active = selected && st == FIT_OK;
    const int m = a.m_col ? a.m_col[(size_t)blockIdx.x * NM_BLOCK] : a.m;

    SeriesView v;
    v.y = a.y + (valid ? s : 0);
    v.ld = a.ld;
    v.len = active ? len : 0;
    v.wave_len = wave_max_i32(v.len);
    v.wave_min_len = 0;
    if (v.wave_len == 0) {
        if (selected) c.status[s] = st;
        return;
    }
    double cand[1][DIM], f[1];
#pragma unroll
    for (int i = 0; i < DIM; i++) cand[0][i] = active ? a.st.sim[(size_t)i * a.ld + s] : 0.5;
    ClassicFinalOut fin;
    fin.h = c.h;
    fin.write = active;
    fin.yhat = c.yhat + (size_t)(valid ? s : 0) * c.h;
    double *ring = GRING ? a.ring_scratch + (size_t)blockIdx.x * (size_t)(a.m > 0 ? a.m : 1) * NM_BLOCK : lds;
    classic_pass<KIND, 1, true, M7>(v, m, cand, f, ring, &fin);
    if (selected) {
        c.status[s] = st;
        if (active) {
            c.passes[s] += a.st.passes[s] + 1;
            if (c.model_code_out) c.model_code_out[s] = c.model_code;
        }
    }
}

void launch_classic_final(int kind, const FitArgs &a, const ClassicArgs &c, hipStream_t stream)
{
    const int grid = (a.n_series + NM_BLOCK - 1) / NM_BLOCK;
    const size_t ring_bytes = sizeof(double) * (size_t)(a.m > 0 ? a.m : 1) * NM_BLOCK;
    const bool gring = a.m > ETS_LDS_PERIOD;
    switch (kind) {
    case CK_SES: hipLaunchKernelGGL((classic_final_kernel<CK_SES, false>), dim3(grid), dim3(NM_BLOCK), 0, stream, a, c); break;
    case CK_HOLT: hipLaunchKernelGGL((classic_final_kernel<CK_HOLT, false>), dim3(grid), dim3(NM_BLOCK), 0, stream, a, c); break;
    case CK_HW:
        if (a.m == 7 && !a.m_col) hipLaunchKernelGGL((classic_final_kernel<CK_HW, false, true>), dim3(grid), dim3(NM_BLOCK), 0, stream, a, c);
        else if (gring) hipLaunchKernelGGL((classic_final_kernel<CK_HW, true>), dim3(grid), dim3(NM_BLOCK), 0, stream, a, c);
        else hipLaunchKernelGGL((classic_final_kernel<CK_HW, false>), dim3(grid), dim3(NM_BLOCK), ring_bytes, stream, a, c);
        break;
    case CK_SEASONAL_ES:
        if (a.m == 7 && !a.m_col) hipLaunchKernelGGL((classic_final_kernel<CK_SEASONAL_ES, false, true>), dim3(grid), dim3(NM_BLOCK), 0, stream, a, c);
        else if (gring) hipLaunchKernelGGL((classic_final_kernel<CK_SEASONAL_ES, true>), dim3(grid), dim3(NM_BLOCK), 0, stream, a, c);
        else hipLaunchKernelGGL((classic_final_kernel<CK_SEASONAL_ES, false>), dim3(grid), dim3(NM_BLOCK), ring_bytes, stream, a, c);
        break;
    default: break;
    }
}

} // namespace anofox
