// host_semantics.hpp -- host-side restatement of the wrapper decisions of the reference's
// per-series path: model-name parsing (crates/anofox-fcst-core/src/forecast.rs:148-307), ETS
// notation / validity (forecast.rs:1255-1324), AutoETS pool names (:1524-1537), the
// seasonal_period compatibility rule (:541-565), error taxonomy (error.rs:9-61) and the z table
// of the intervals (:2571-2577).  Pure C++ (no device code): shared by the C-ABI layer.
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/anofox_fcst_hip.h"

namespace anofox {

enum ModelType {
    M_AutoETS, M_AutoARIMA, M_AutoTheta, M_AutoMFLES, M_AutoMSTL, M_AutoTBATS,
    M_Naive, M_SMA, M_SeasonalNaive, M_SES, M_SESOptimized, M_RandomWalkDrift,
    M_Holt, M_HoltWinters, M_SeasonalES, M_SeasonalESOptimized, M_SeasonalWindowAverage,
    M_Theta, M_OptimizedTheta, M_DynamicTheta, M_DynamicOptimizedTheta,
    M_ETS, M_ARIMA, M_MFLES, M_MSTL, M_TBATS,
    M_CrostonClassic, M_CrostonOptimized, M_CrostonSBA, M_ADIDA, M_IMAPA, M_TSB,
    M_Laplace, M_COUNT
};

inline const char *model_name(ModelType m)
{
    static const char *const names[M_COUNT] = {
        "AutoETS", "AutoARIMA", "AutoTheta", "AutoMFLES", "AutoMSTL", "AutoTBATS",
        "Naive", "SMA", "SeasonalNaive", "SES", "SESOptimized", "RandomWalkDrift",
        "Holt", "HoltWinters", "SeasonalES", "SeasonalESOptimized", "SeasonalWindowAverage",
        "Theta", "OptimizedTheta", "DynamicTheta", "DynamicOptimizedTheta",
        "ETS", "ARIMA", "MFLES", "MSTL", "TBATS",
        "CrostonClassic", "CrostonOptimized", "CrostonSBA", "ADIDA", "IMAPA", "TSB", "Laplace"};
    return names[m];
}

inline bool parse_model(const std::string &s, ModelType &out)
{
    for (int i = 0; i < M_COUNT; i++)
        if (s == model_name((ModelType)i)) { out = (ModelType)i; return true; }
    if (s == "RandomWalkWithDrift") { out = M_RandomWalkDrift; return true; }
    std::string l = s;
    std::transform(l.begin(), l.end(), l.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    struct A { const char *a; ModelType m; };
    static const A al[] = {
        {"autoets", M_AutoETS}, {"auto_ets", M_AutoETS}, {"autoarima", M_AutoARIMA}, {"auto_arima", M_AutoARIMA},
        {"autotheta", M_AutoTheta}, {"auto_theta", M_AutoTheta}, {"automfles", M_AutoMFLES}, {"auto_mfles", M_AutoMFLES},
        {"automstl", M_AutoMSTL}, {"auto_mstl", M_AutoMSTL}, {"autotbats", M_AutoTBATS}, {"auto_tbats", M_AutoTBATS},
        {"naive", M_Naive}, {"sma", M_SMA}, {"seasonalnaive", M_SeasonalNaive}, {"seasonal_naive", M_SeasonalNaive},
        {"snaive", M_SeasonalNaive}, {"ses", M_SES}, {"sesoptimized", M_SESOptimized}, {"ses_optimized", M_SESOptimized},
        {"randomwalkdrift", M_RandomWalkDrift}, {"random_walk_drift", M_RandomWalkDrift}, {"rwd", M_RandomWalkDrift},
        {"drift", M_RandomWalkDrift}, {"randomwalkwithdrift", M_RandomWalkDrift}, {"random_walk_with_drift", M_RandomWalkDrift},
        {"holt", M_Holt}, {"holtwinters", M_HoltWinters}, {"holt_winters", M_HoltWinters}, {"hw", M_HoltWinters},
        {"seasonales", M_SeasonalES}, {"seasonal_es", M_SeasonalES}, {"seasonalesoptimized", M_SeasonalESOptimized},
        {"seasonal_es_optimized", M_SeasonalESOptimized}, {"seasonalwindowaverage", M_SeasonalWindowAverage},
        {"seasonal_window_average", M_SeasonalWindowAverage}, {"swa", M_SeasonalWindowAverage},
        {"theta", M_Theta}, {"optimizedtheta", M_OptimizedTheta}, {"optimized_theta", M_OptimizedTheta}, {"otm", M_OptimizedTheta},
        {"dynamictheta", M_DynamicTheta}, {"dynamic_theta", M_DynamicTheta}, {"dstm", M_DynamicTheta},
        {"dynamicoptimizedtheta", M_DynamicOptimizedTheta}, {"dynamic_optimized_theta", M_DynamicOptimizedTheta},
        {"ets", M_ETS}, {"arima", M_ARIMA}, {"mfles", M_MFLES}, {"mstl", M_MSTL}, {"tbats", M_TBATS},
        {"crostonclassic", M_CrostonClassic}, {"croston_classic", M_CrostonClassic}, {"croston", M_CrostonClassic},
        {"crostonoptimized", M_CrostonOptimized}, {"croston_optimized", M_CrostonOptimized},
        {"crostonsba", M_CrostonSBA}, {"croston_sba", M_CrostonSBA}, {"sba", M_CrostonSBA},
        {"adida", M_ADIDA}, {"imapa", M_IMAPA}, {"tsb", M_TSB}, {"laplace", M_Laplace}, {"auto", M_AutoETS}};
    for (const A &a : al)
        if (l == a.a) { out = a.m; return true; }
    return false;
}

inline bool is_non_seasonal_model(ModelType m)
{
    switch (m) {
    case M_Naive: case M_SES: case M_SESOptimized: case M_Holt: case M_RandomWalkDrift: case M_ARIMA:
    case M_CrostonClassic: case M_CrostonOptimized: case M_CrostonSBA: case M_TSB: case M_ADIDA: case M_IMAPA:
        return true;
    default: return false;
    }
}

inline bool valid_ets_notation(const std::string &s)
{
    auto am = [](char c) { return c == 'A' || c == 'M'; };
    auto amn = [](char c) { return c == 'A' || c == 'M' || c == 'N'; };
    if (s.size() == 3) return am(s[0]) && amn(s[1]) && amn(s[2]);
    if (s.size() == 4) return am(s[0]) && am(s[1]) && s[2] == 'd' && amn(s[3]);
    return false;
}

// spec id = error*15 + trendIdx*3 + season ; trendIdx 0 N, 1 A, 2 Ad, 3 M, 4 Md ; season 0 N, 1 A, 2 M
inline int spec_id_from_notation(const std::string &s)
{
    auto c = [](char ch) { return ch == 'A' ? 1 : (ch == 'M' ? 2 : 0); };
    int e = c(s[0]) == 1 ? 0 : 1;
    int t = c(s[1]);
    bool d = s.size() == 4;
    int ti = t == 0 ? 0 : (t == 1 ? (d ? 2 : 1) : (d ? 4 : 3));
    int se = c(s.back());
    return e * 15 + ti * 3 + se;
}
inline int spec_error(int id) { return id / 15; }           // 0 additive, 1 multiplicative
inline int spec_trend_idx(int id) { return (id % 15) / 3; }
inline int spec_season(int id) { return id % 3; }
inline bool spec_has_mult(int id) { return spec_error(id) == 1 || spec_trend_idx(id) >= 3 || spec_season(id) == 2; }
inline bool spec_is_valid(int id) { return !(spec_error(id) == 1 && spec_season(id) == 1); }
inline int spec_dim(int id) { int ti = spec_trend_idx(id); return 1 + (ti != 0) + (spec_season(id) != 0) + ((ti == 2 || ti == 4) ? 1 : 0); }
inline int spec_n_param(int id, int m)
{
    int n_states = 1 + (spec_trend_idx(id) != 0) + (spec_season(id) != 0 ? m - 1 : 0);
    return spec_dim(id) + n_states + 1;
}

inline int parse_model_pool(const std::string &s)
{
    std::string t;
    for (char ch : s) {
        char c = (char)std::tolower((unsigned char)ch);
        if (c == '-' || c == '_') continue;
        t.push_back(c);
    }
    if (t == "complete") return 0;
    if (t == "nomultiplicativetrend") return 1;
    if (t == "dampedtrendonly") return 2;
    if (t == "matcherrorseasonal") return 3;
    if (t == "reduced") return 4;
    return -1;
}
inline bool pool_allows(int pool, int id)
{
    int ti = spec_trend_idx(id), se = spec_season(id), e = spec_error(id);
    bool tmul = ti >= 3, damped = (ti == 2 || ti == 4);
    bool match = (se == 0) || (se == 1 && e == 0) || (se == 2 && e == 1);
    switch (pool) {
    case 1: return !tmul;
    case 2: return ti == 0 || damped;
    case 3: return match;
    case 4: return !tmul && match;
    default: return true;
    }
}

inline void auto_ets_name(int spec_id, char out[64])
{
    static const char *E[2] = {"Additive", "Multiplicative"};
    static const char *T[5] = {"None", "Additive", "AdditiveDamped", "Multiplicative", "MultiplicativeDamped"};
    static const char *S[3] = {"None", "Additive", "Multiplicative"};
    std::snprintf(out, 64, "AutoETS(%s,%s,%s)", E[spec_error(spec_id)], T[spec_trend_idx(spec_id)], S[spec_season(spec_id)]);
}

// model_code of the AutoARIMA kernels: 1000000 + p*1e5 + d*1e4 + q*1e3 + P*100 + D*10 + Q  (forecast.rs:1469-1493)
inline void auto_arima_name(int code, int period, char out[64])
{
    int v = code - 1000000;
    int p = v / 100000, d = (v / 10000) % 10, q = (v / 1000) % 10, P = (v / 100) % 10, D = (v / 10) % 10, Q = v % 10;
    int s = (period > 1 && period <= 2048) ? period : 1;
    if (s > 1 && (P || D || Q)) std::snprintf(out, 64, "AutoARIMA(%d,%d,%d)(%d,%d,%d)[%d]", p, d, q, P, D, Q, s);
    else std::snprintf(out, 64, "AutoARIMA(%d,%d,%d)", p, d, q);
}

inline double z_for_confidence(double c)
{
    return c >= 0.99 ? 2.576 : c >= 0.95 ? 1.96 : c >= 0.90 ? 1.645 : c >= 0.80 ? 1.28 : 1.0;
}

inline void set_error(AnofoxError *e, int code, const std::string &msg)
{
    if (!e) return;
    e->code = (ErrorCode)code;
    size_t n = std::min<size_t>(msg.size(), 255);
    std::memcpy(e->message, msg.data(), n);
    e->message[n] = 0;
}

inline std::string cstr_field(const char *p, size_t cap)
{
    size_t n = 0;
    while (n < cap && p[n]) n++;
    return std::string(p, n);
}

} // namespace anofox
