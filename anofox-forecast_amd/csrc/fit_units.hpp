// fit_units.hpp -- helpers to instantiate the ETS round/final kernels per (spec id, ring variant).
// spec id = error*15 + trendIdx*3 + season ; trendIdx 0 N, 1 A, 2 Ad, 3 M, 4 Md ; season 0 N, 1 A, 2 M.
#pragma once
#include "ets_fit_kernel.hpp"

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
    return FitLaunchers{&ets_round_launch<typename SpecOf<ID>::Cfg, MS, false>, &ets_round_launch<typename SpecOf<ID>::Cfg, MS, true>,
                        &ets_final_launch<typename SpecOf<ID>::Cfg, MS>};
}
// seasonal spec: VGPR ring for the listed compile-time periods, LDS ring otherwise
#define ANOFOX_SEASONAL_CASE(ID)                       \
    case ID:                                           \
        if (m == 7) return launchers_of<ID, 7>();      \
        return launchers_of<ID, -1>();
#define ANOFOX_SEASONAL_CASE12(ID)                     \
    case ID:                                           \
        if (m == 7) return launchers_of<ID, 7>();      \
        if (m == 12) return launchers_of<ID, 12>();    \
        return launchers_of<ID, -1>();
#define ANOFOX_NONSEASONAL_CASE(ID) \
    case ID: return launchers_of<ID, 0>();

FitLaunchers fit_unit_nonseasonal(int spec_id, int m);
FitLaunchers fit_unit_seasonal_add(int spec_id, int m);
FitLaunchers fit_unit_seasonal_gen_a(int spec_id, int m);
FitLaunchers fit_unit_seasonal_gen_m(int spec_id, int m);
} // namespace anofox
