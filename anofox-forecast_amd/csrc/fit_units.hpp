// fit_units.hpp -- helpers to instantiate the ETS kernels per (spec id, ring variant), one compile unit per model class
// (compile-time parallelism).  spec id = error*15 + trendIdx*3 + season ; trendIdx 0 N, 1 A, 2 Ad, 3 M, 4 Md ;
// season 0 N, 1 A, 2 M.  A unit .hip defines ANOFOX_UNIT_NAME, ANOFOX_UNIT_SPECS(X) and ANOFOX_UNIT_VARIANTS
// (0: no period, 1: m = 7 / run-time, 2: m = 7 / 12 / run-time) and includes fit_unit_impl.inc, which generates
//   * FitLaunchers fit_unit_<name>(spec_id, m)            -- the round schedule's per-spec kernels
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
// the K4 forms exist for the additive class without a period or with the weekly one in registers (m = 12: four rings of 12 spill): ets_fit_kernel.hpp
template <int ID, int MS, int SPEC, class YT> FitLaunchFn k4_launcher()
{
    if constexpr (SpecOf<ID>::Cfg::ADDITIVE && (MS == 0 || MS == 7)) return &ets_round_launch<typename SpecOf<ID>::Cfg, MS, SPEC, true, YT>;
    else return nullptr;
}
// YT: what the streamed block holds (ets_device.hpp).  float exists for every period variant; uint16_t (a quarter of the bytes: worth
// another 2-3 % on the weekly M5 batch) for the two the BASELINE configurations use, no period and the weekly ring in registers --
// a batch of counts with any other period streams the float copy (host_api.hip compact_storage_decide).
constexpr bool ets_u16_variant(int ms) { return ms == 0 || ms == 7; }
template <int ID, int MS, class YT = double> FitLaunchers launchers_of()
{
    if constexpr (std::is_same_v<YT, unsigned short> && !ets_u16_variant(MS)) return FitLaunchers{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
    else {
    // the final pass sweeps the columns in their original blocks of one period each: the per-lane period variants (-3 / -4) exist for
    // the round kernels only
    constexpr int MSF = MS == -3 ? -1 : (MS == -4 ? -2 : MS);
    return FitLaunchers{&ets_round_launch<typename SpecOf<ID>::Cfg, MS, 0, false, YT>, &ets_round_launch<typename SpecOf<ID>::Cfg, MS, 1, false, YT>,
                        &ets_round_launch<typename SpecOf<ID>::Cfg, MS, 2, false, YT>, &ets_round_launch<typename SpecOf<ID>::Cfg, MS, 3, false, YT>,
                        &ets_final_launch<typename SpecOf<ID>::Cfg, MSF, YT>,
                        RoundTraits<typename SpecOf<ID>::Cfg>::PARK ? (size_t)nm_lds_doubles<SpecOf<ID>::Cfg::DIM>() : 0,
                        k4_launcher<ID, MS, 0, YT>(), k4_launcher<ID, MS, 3, YT>()};
    }
}
// run-time storage type -> the instantiation (yt: YT_F64 / YT_F32 / YT_U16)
template <int ID, int MS> FitLaunchers launchers_of_yt(int yt)
{
    if (yt == YT_F32) return launchers_of<ID, MS, float>();
    if (yt == YT_U16) return launchers_of<ID, MS, unsigned short>();
    return launchers_of<ID, MS, double>();
}

FitLaunchers fit_unit_nonseasonal(int spec_id, int m, int yt);
FitLaunchers fit_unit_seasonal_add(int spec_id, int m, int yt);
FitLaunchers fit_unit_seasonal_gen_a(int spec_id, int m, int yt);
FitLaunchers fit_unit_seasonal_gen_m(int spec_id, int m, int yt);
} // namespace anofox
