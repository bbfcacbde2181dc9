// fit_units.hpp -- helpers to instantiate the ETS kernels per (spec id, ring variant), one compile unit per model class
// (compile-time parallelism).  spec id = error*15 + trendIdx*3 + season ; trendIdx 0 N, 1 A, 2 Ad, 3 M, 4 Md ;
// season 0 N, 1 A, 2 M.  A unit .hip defines ANOFOX_UNIT_NAME, ANOFOX_UNIT_SPECS(X) and ANOFOX_UNIT_VARIANTS
// (0: no period, 1: m = 7 / run-time, 2: m = 7 / 12 / run-time) and includes fit_unit_impl.inc, which generates
//   * FitLaunchers fit_unit_<name>(spec_id, m)            -- the round schedule's per-spec kernels
//   * ets_pool_unit_<name> + launch_pool_unit_<name>      -- the work-pool kernel of the unit (ets_pool_kernel.hpp)
#pragma once
#include "ets_pool_kernel.hpp"

namespace anofox {
template <int ID> struct SpecOf {
    static constexpr int e = (ID / 15) == 0 ? C_ADD : C_MUL;
    static constexpr int ti = (ID % 15) / 3;
    static constexpr int t = ti == 0 ? C_NONE : (ti <= 2 ? C_ADD : C_MUL);
    static constexpr bool d = (ti == 2 || ti == 4);
    static constexpr int s = ID % 3;
    using Cfg = EtsCfg<e, t, d, s>;
};
template <int ID, int MS> FitLaunchers launchers_of()
{
    return FitLaunchers{&ets_round_launch<typename SpecOf<ID>::Cfg, MS, 0>, &ets_round_launch<typename SpecOf<ID>::Cfg, MS, 1>,
                        &ets_round_launch<typename SpecOf<ID>::Cfg, MS, 2>, &ets_round_launch<typename SpecOf<ID>::Cfg, MS, 3>,
                        &ets_final_launch<typename SpecOf<ID>::Cfg, MS>};
}

// one spec of the unit's work-pool kernel: the FitArgs view of the by-value arguments, then the spec's pool loop
template <int ID, int MS>
__device__ __forceinline__ void pool_run_spec(const PoolUnitArgs &u, const PoolSpec &p, double *lds)
{
    FitArgs a{};
    a.ys = u.ys; a.tw = u.tw; a.ld = u.ld; a.fig_ld = u.fig_ld; a.len = u.len; a.flags = u.flags;
    a.n_series = u.n_series; a.h = u.h; a.promote = u.promote; a.start_spec = u.start_spec; a.skip_constant = u.skip_constant;
    a.series_of = p.series_of; a.n_active = p.n_active; a.head = p.head; a.status = p.status; a.st = p.st; a.aicc = p.aicc;
    a.evals = p.evals; a.iters = p.iters; a.passes = p.passes; a.l0 = p.l0; a.b0 = p.b0; a.fig = p.fig; a.yhat = p.yhat;
    a.m = p.m; a.n_param = p.n_param; a.need_positive = p.need_positive; a.trace = p.trace;
    ets_pool_body<typename SpecOf<ID>::Cfg, MS>(a, lds);
}

FitLaunchers fit_unit_nonseasonal(int spec_id, int m);
FitLaunchers fit_unit_seasonal_add(int spec_id, int m);
FitLaunchers fit_unit_seasonal_gen_a(int spec_id, int m);
FitLaunchers fit_unit_seasonal_gen_m(int spec_id, int m);
void launch_pool_unit_nonseasonal(const PoolUnitArgs &, int grid, size_t lds_bytes, hipStream_t);
void launch_pool_unit_seasonal_add(const PoolUnitArgs &, int grid, size_t lds_bytes, hipStream_t);
void launch_pool_unit_seasonal_gen_a(const PoolUnitArgs &, int grid, size_t lds_bytes, hipStream_t);
void launch_pool_unit_seasonal_gen_m(const PoolUnitArgs &, int grid, size_t lds_bytes, hipStream_t);
} // namespace anofox
