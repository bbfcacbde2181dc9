/*
 * forecast.c -- TEST INFRASTRUCTURE (oracle).  Not part of the product.
 *
 * CPU restatement of the reference's per-series forecast path behind
 * `anofox_ts_forecast`:
 *   FFI marshalling        crates/anofox-fcst-ffi/src/lib.rs:62-87, 3344-3550
 *   forecast() wrapper     crates/anofox-fcst-core/src/forecast.rs:512-733
 *   model name parsing     forecast.rs:148-257, names :262-307
 *   NULL interpolation     crates/anofox-fcst-core/src/imputation.rs:61-114
 *   seasonality detection  crates/anofox-fcst-core/src/seasonality.rs:323-377
 *   baselines              forecast.rs:1026-1100, toy ARIMA :1391-1431
 *   ETS(spec) / fallback   forecast.rs:1255-1389
 *   AutoETS + fallback     forecast.rs:1524-1641
 *   intervals / fitted     forecast.rs:2558-2643
 *   error codes            crates/anofox-fcst-core/src/error.rs:46-61
 * The model arithmetic the reference delegates to anofox-forecast 0.15.3 is in
 * ets.c / arima.c (restated, see their headers).
 *
 * Exported with an `oracle_` prefix and the SAME structs as the product ABI
 * (include/anofox_fcst_hip.h) so tests compare field by field.
 */
#include "../include/anofox_fcst_hip.h"
#include "arima.h"
#include "det_math.h"
#include "ets.h"

#include <ctype.h>
#include <float.h>
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* model table (forecast.rs:92-307)                                           */
/* ------------------------------------------------------------------------- */

typedef enum {
    M_AutoETS, M_AutoARIMA, M_AutoTheta, M_AutoMFLES, M_AutoMSTL, M_AutoTBATS,
    M_Naive, M_SMA, M_SeasonalNaive, M_SES, M_SESOptimized, M_RandomWalkDrift,
    M_Holt, M_HoltWinters, M_SeasonalES, M_SeasonalESOptimized, M_SeasonalWindowAverage,
    M_Theta, M_OptimizedTheta, M_DynamicTheta, M_DynamicOptimizedTheta,
    M_ETS, M_ARIMA, M_MFLES, M_MSTL, M_TBATS,
    M_CrostonClassic, M_CrostonOptimized, M_CrostonSBA, M_ADIDA, M_IMAPA, M_TSB,
    M_Laplace, M_COUNT
} ModelType;

static const char *const MODEL_NAMES[M_COUNT] = {
    "AutoETS", "AutoARIMA", "AutoTheta", "AutoMFLES", "AutoMSTL", "AutoTBATS",
    "Naive", "SMA", "SeasonalNaive", "SES", "SESOptimized", "RandomWalkDrift",
    "Holt", "HoltWinters", "SeasonalES", "SeasonalESOptimized", "SeasonalWindowAverage",
    "Theta", "OptimizedTheta", "DynamicTheta", "DynamicOptimizedTheta",
    "ETS", "ARIMA", "MFLES", "MSTL", "TBATS",
    "CrostonClassic", "CrostonOptimized", "CrostonSBA", "ADIDA", "IMAPA", "TSB",
    "Laplace",
};

typedef struct { const char *alias; ModelType m; } Alias;
static const Alias ALIASES[] = {
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
    {"adida", M_ADIDA}, {"imapa", M_IMAPA}, {"tsb", M_TSB}, {"laplace", M_Laplace}, {"auto", M_AutoETS},
};

static int parse_model(const char *s, ModelType *out)
{
    for (int i = 0; i < M_COUNT; i++)
        if (strcmp(s, MODEL_NAMES[i]) == 0) { *out = (ModelType)i; return 1; }
    if (strcmp(s, "RandomWalkWithDrift") == 0) { *out = M_RandomWalkDrift; return 1; }
    char low[64];
    size_t n = strlen(s);
    if (n >= sizeof low) return 0;
    for (size_t i = 0; i <= n; i++) low[i] = (char)tolower((unsigned char)s[i]);
    for (size_t i = 0; i < sizeof ALIASES / sizeof ALIASES[0]; i++)
        if (strcmp(low, ALIASES[i].alias) == 0) { *out = ALIASES[i].m; return 1; }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* errors (error.rs:9-61, types.rs:47-55)                                     */
/* ------------------------------------------------------------------------- */

typedef struct { int code; char msg[512]; } Err;

static void set_error(AnofoxError *e, int code, const char *msg)
{
    if (!e) return;
    e->code = (ErrorCode)code;
    size_t n = strlen(msg);
    if (n > 255) n = 255;
    memcpy(e->message, msg, n);
    e->message[n] = 0;
}

#define FAIL(err, c, ...) do { (err)->code = (c); snprintf((err)->msg, sizeof (err)->msg, __VA_ARGS__); return 0; } while (0)

/* ------------------------------------------------------------------------- */
/* imputation.rs:61-114                                                       */
/* ------------------------------------------------------------------------- */

void oracle_fill_nulls_interpolate(const double *values, const uint64_t *validity, size_t n, double *out)
{
    if (n == 0) return;
#define VALID(i) (validity == NULL || ((validity[(i) / 64] >> ((i) % 64)) & 1ull))
    long first = -1, last = -1;
    for (size_t i = 0; i < n; i++) if (VALID(i)) { if (first < 0) first = (long)i; last = (long)i; }
    for (size_t i = 0; i < n; i++) out[i] = NAN;
    if (first < 0) return;
    for (long i = 0; i < first; i++) out[i] = values[first];
    for (size_t i = (size_t)last + 1; i < n; i++) out[i] = values[last];
    long prev = first;
    double pv = values[first];
    out[first] = pv;
    for (long i = first + 1; i <= last; i++) {
        if (VALID((size_t)i)) {
            double v = values[i];
            long gap = i - prev;
            if (gap > 1) {
                double slope = (v - pv) / (double)gap;
                for (long j = 1; j < gap; j++) out[prev + j] = pv + slope * (double)j;
            }
            out[i] = v;
            prev = i;
            pv = v;
        }
    }
#undef VALID
}

/* ------------------------------------------------------------------------- */
/* seasonality.rs:323-377 -- first ACF peak, or 0 when none                   */
/* ------------------------------------------------------------------------- */

int oracle_detect_seasonality_first(const double *v, size_t n)
{
    if (n < 4) return 0;
    size_t max_lag = n / 2;
    if (max_lag < 2) return 0;
    double mean = 0.0;
    for (size_t i = 0; i < n; i++) mean += v[i];
    mean /= (double)n;
    double var = 0.0;
    for (size_t i = 0; i < n; i++) { double d = v[i] - mean; var += d * d; }
    if (fabs(var) < DBL_EPSILON) return 0;
    double *acf = (double *)malloc(max_lag * sizeof(double));
    for (size_t lag = 1; lag <= max_lag; lag++) {
        double s = 0.0;
        for (size_t i = 0; i < n - lag; i++) s += (v[i] - mean) * (v[i + lag] - mean);
        acf[lag - 1] = s / var;
    }
    int best = 0;
    double best_acf = 0.0;
    for (size_t i = 1; i + 1 < max_lag; i++) {
        if (acf[i] > acf[i - 1] && acf[i] > acf[i + 1] && acf[i] > 0.1) {
            /* stable sort by ACF descending: first strictly-greatest wins */
            if (best == 0 || acf[i] > best_acf) { best = (int)(i + 1); best_acf = acf[i]; }
        }
    }
    free(acf);
    return best;
}

/* ------------------------------------------------------------------------- */
/* simple models (forecast.rs:1026-1100, 1391-1431)                           */
/* ------------------------------------------------------------------------- */

static void m_naive(const double *y, size_t n, int h, double *out)
{
    for (int i = 0; i < h; i++) out[i] = y[n - 1];
}

static void m_seasonal_naive(const double *y, size_t n, int h, size_t period, double *out)
{
    size_t p = period < 1 ? 1 : period;
    if (p > n) p = n;
    for (int i = 0; i < h; i++) out[i] = y[n - p + ((size_t)i % p)];
}

static void m_sma(const double *y, size_t n, int h, size_t window, double *out)
{
    size_t w = window < n ? window : n;
    double s = 0.0;
    for (size_t k = 0; k < w; k++) s += y[n - 1 - k];
    double v = s / (double)w;
    for (int i = 0; i < h; i++) out[i] = v;
}

static void m_drift(const double *y, size_t n, int h, double *out)
{
    double drift = (y[n - 1] - y[0]) / (double)(n - 1);
    for (int i = 1; i <= h; i++) out[i - 1] = y[n - 1] + drift * (double)i;
}

static void m_toy_arima(const double *y, size_t n, int h, double *out)
{
    if (n < 5) { m_naive(y, n, h, out); return; }
    double sd = 0.0;
    for (size_t i = 1; i < n; i++) sd += y[i] - y[i - 1];
    double mean_diff = sd / (double)(n - 1);
    double prev = y[n - 1] - y[n - 2], cum = y[n - 1];
    for (int i = 0; i < h; i++) {
        double nd = mean_diff + 0.5 * (prev - mean_diff);
        cum += nd;
        out[i] = cum;
        prev = nd;
    }
}

/* ------------------------------------------------------------------------- */
/* SES / Holt / HoltWinters / SeasonalES family.                              */
/* The reference calls SimpleExponentialSmoothing::{new(0.3),auto},           */
/* HoltLinearTrend::auto, HoltWinters::auto(p, Additive),                     */
/* SeasonalES::{new(p),optimized(p)} (forecast.rs:1102-1144, 1206-1232).      */
/* Formulations below reproduce the KATs of ts_model_distinctness.test:116,141 */
/* (SES 18.943503, SESOptimized 19.537535, SeasonalES 14.451866,              */
/*  Holt 20.330877, HoltWinters 19.953912).                                   */
/* ------------------------------------------------------------------------- */

typedef struct { const double *y; size_t n; size_t m; } SeriesCtx;

static double ses_run(const double *y, size_t n, double alpha, double *level_out)
{
    double l = y[0], sse = 0.0;
    for (size_t t = 1; t < n; t++) {
        double e = y[t] - l;
        sse = fma(e, e, sse);
        l = fma(alpha, e, l);
    }
    if (level_out) *level_out = l;
    return sse;
}
static double ses_obj(const double *x, void *c) { SeriesCtx *s = (SeriesCtx *)c; return ses_run(s->y, s->n, x[0], NULL); }

static double holt_run(const double *y, size_t n, double alpha, double beta, double *l_out, double *b_out)
{
    double l = y[0], b = y[1] - y[0], sse = 0.0;
    for (size_t t = 1; t < n; t++) {
        double f = l + b;
        double e = y[t] - f;
        sse = fma(e, e, sse);
        double ln = fma(alpha, e, f);          /* alpha y + (1-alpha)(l+b) */
        b = fma(beta, (ln - l) - b, b);        /* beta (l'-l) + (1-beta) b */
        l = ln;
    }
    if (l_out) { *l_out = l; *b_out = b; }
    return sse;
}
static double holt_obj(const double *x, void *c) { SeriesCtx *s = (SeriesCtx *)c; return holt_run(s->y, s->n, x[0], x[1], NULL, NULL); }

/* classical additive Holt-Winters; s ring indexed by phase t mod m */
static double hw_run(const double *y, size_t n, size_t m, double alpha, double beta, double gamma,
                     double *l_out, double *b_out, double *s_out)
{
    double s[ETS_MAX_PERIOD];
    double m1 = 0.0, m2 = 0.0;
    for (size_t i = 0; i < m; i++) m1 += y[i];
    for (size_t i = m; i < 2 * m; i++) m2 += y[i];
    m1 /= (double)m;
    m2 /= (double)m;
    double l = m1, b = (m2 - m1) / (double)m, sse = 0.0;
    for (size_t i = 0; i < m; i++) s[i] = y[i] - m1;
    for (size_t t = m; t < n; t++) {
        size_t j = t % m;
        double q = l + b;
        double e = y[t] - (q + s[j]);
        sse = fma(e, e, sse);
        double ln = fma(alpha, (y[t] - s[j]) - q, q);   /* alpha (y-s) + (1-alpha)(l+b) */
        b = fma(beta, (ln - l) - b, b);
        s[j] = fma(gamma, (y[t] - ln) - s[j], s[j]);    /* gamma (y-l') + (1-gamma) s */
        l = ln;
    }
    if (l_out) { *l_out = l; *b_out = b; for (size_t i = 0; i < m; i++) s_out[i] = s[i]; }
    return sse;
}
static double hw_obj(const double *x, void *c) { SeriesCtx *s = (SeriesCtx *)c; return hw_run(s->y, s->n, s->m, x[0], x[1], x[2], NULL, NULL, NULL); }

static double seasonal_es_run(const double *y, size_t n, size_t m, double alpha, double *s_out)
{
    double s[ETS_MAX_PERIOD], sse = 0.0;
    for (size_t i = 0; i < m; i++) s[i] = y[i];
    for (size_t t = m; t < n; t++) {
        size_t j = t % m;
        double e = y[t] - s[j];
        sse = fma(e, e, sse);
        s[j] = fma(alpha, e, s[j]);
    }
    if (s_out) for (size_t i = 0; i < m; i++) s_out[i] = s[i];
    return sse;
}
static double seasonal_es_obj(const double *x, void *c) { SeriesCtx *s = (SeriesCtx *)c; return seasonal_es_run(s->y, s->n, s->m, x[0], NULL); }

static const double BOX_LO[3] = { ETS_PAR_LO, ETS_PAR_LO, ETS_PAR_LO };
static const double BOX_HI[3] = { ETS_PAR_HI, ETS_PAR_HI, ETS_PAR_HI };

static int m_ses(const double *y, size_t n, int h, int optimized, double *out, Err *err)
{
    (void)err;
    double alpha = 0.3, l;
    if (optimized) {
        SeriesCtx c = { y, n, 1 };
        double x0 = 0.5;
        NmResult r;
        nm_minimize(ses_obj, &c, 1, &x0, BOX_LO, BOX_HI, &r);
        alpha = r.x[0];
    }
    ses_run(y, n, alpha, &l);
    for (int i = 0; i < h; i++) out[i] = l;
    return 1;
}

static int m_holt(const double *y, size_t n, int h, double *out, Err *err)
{
    (void)err;
    SeriesCtx c = { y, n, 1 };
    double x0[2] = { 0.3, 0.1 };
    NmResult r;
    nm_minimize(holt_obj, &c, 2, x0, BOX_LO, BOX_HI, &r);
    double l, b;
    holt_run(y, n, r.x[0], r.x[1], &l, &b);
    for (int i = 1; i <= h; i++) out[i - 1] = l + (double)i * b;
    return 1;
}

static int m_holt_winters(const double *y, size_t n, int h, size_t period, double *out, Err *err)
{
    size_t m = period < 2 ? 2 : period;
    /* fewer than two seasons: the crate's Holt-Winters falls back to Holt's linear trend (pinned by
     * test/sql/ts_forecast_exp_smoothing.test:498-503 and test/sql/ts_forecast_params.test:203-207: 10 observations,
     * seasonal_period 7, rows are returned); the caller still names the result "HoltWinters" */
    if (n < 2 * m) return m_holt(y, n, h, out, err);
    if (m > ETS_MAX_PERIOD)
        FAIL(err, COMPUTATION_ERROR, "Computation error: HoltWinters fit failed: unsupported seasonal period (periods above 2048 are not supported)");
    SeriesCtx c = { y, n, m };
    double x0[3] = { 0.3, 0.1, 0.1 };
    NmResult r;
    nm_minimize(hw_obj, &c, 3, x0, BOX_LO, BOX_HI, &r);
    double l, b, s[ETS_MAX_PERIOD];
    hw_run(y, n, m, r.x[0], r.x[1], r.x[2], &l, &b, s);
    for (int i = 1; i <= h; i++) out[i - 1] = (l + (double)i * b) + s[(n + (size_t)i - 1) % m];
    return 1;
}

static int m_seasonal_es(const double *y, size_t n, int h, size_t period, int optimized, double *out, Err *err)
{
    size_t m = period < 2 ? 2 : period;
    if (m > ETS_MAX_PERIOD || n < m)
        FAIL(err, COMPUTATION_ERROR, "Computation error: SeasonalES%s fit failed: need at least %zu observations, got %zu", optimized ? "Optimized" : "", m, n);
    double alpha = 0.1;
    if (optimized) {
        SeriesCtx c = { y, n, m };
        double x0 = 0.5;
        NmResult r;
        nm_minimize(seasonal_es_obj, &c, 1, &x0, BOX_LO, BOX_HI, &r);
        alpha = r.x[0];
    }
    double s[ETS_MAX_PERIOD];
    seasonal_es_run(y, n, m, alpha, s);
    for (int i = 0; i < h; i++) out[i] = s[(n + (size_t)i) % m];
    return 1;
}

/* ------------------------------------------------------------------------- */
/* ETS(spec) and AutoETS (forecast.rs:1255-1389, 1524-1641)                   */
/* ------------------------------------------------------------------------- */

static int valid_ets_notation(const char *s)
{
    size_t n = strlen(s);
    if (n == 3)
        return (s[0] == 'A' || s[0] == 'M') && (s[1] == 'A' || s[1] == 'M' || s[1] == 'N') &&
               (s[2] == 'A' || s[2] == 'M' || s[2] == 'N');
    if (n == 4)
        return (s[0] == 'A' || s[0] == 'M') && (s[1] == 'A' || s[1] == 'M') && s[2] == 'd' &&
               (s[3] == 'A' || s[3] == 'M' || s[3] == 'N');
    return 0;
}

static int comp(char c) { return c == 'A' ? ETS_ADD : (c == 'M' ? ETS_MUL : ETS_NONE); }

static EtsSpec spec_from_notation(const char *s)
{
    EtsSpec sp;
    size_t n = strlen(s);
    sp.error = comp(s[0]);
    sp.trend = comp(s[1]);
    sp.damped = (n == 4);
    sp.season = comp(s[n - 1]);
    sp.m = 1;
    return sp;
}

/* ETSSpec::is_valid(): multiplicative error with an additive seasonal component is rejected
 * ("MAA", "MAdA": test/sql/ts_native_param_validation.test:142-155). */
static int spec_is_valid(const EtsSpec *s) { return !(s->error == ETS_MUL && s->season == ETS_ADD); }

static const char *ets_status_text(int st)
{
    switch (st) {
    case ETS_ERR_SHORT: return "not enough observations for this model";
    case ETS_ERR_NONPOSITIVE: return "multiplicative components require strictly positive data";
    case ETS_ERR_NONFINITE: return "likelihood is not finite";
    case ETS_ERR_PERIOD: return "unsupported seasonal period (periods above 2048 are not supported)";
    default: return "unknown";
    }
}

static int m_ets_spec(const double *y, size_t n, int h, size_t period, EtsSpec spec, double *out, int *st_out)
{
    /* forecast.rs:1347-1351: period only when the spec is seasonal and period > 1; a seasonal spec
     * without a usable period degenerates to its non-seasonal counterpart. */
    if (spec.season != ETS_NONE && period > 1) spec.m = (int)period;
    else { spec.season = ETS_NONE; spec.m = 1; }
    EtsFit fit;
    double sfin[ETS_MAX_PERIOD];
    int st = ets_fit(&spec, y, (int)n, &fit, sfin);
    *st_out = st;
    if (st != ETS_OK) return 0;
    ets_forecast(&spec, (int)n, &fit, sfin, h, out);
    return 1;
}

/* pool ids: 0 complete, 1 no_multiplicative_trend, 2 damped_trend_only, 3 match_error_seasonal, 4 reduced */
static int parse_model_pool(const char *s)
{
    char t[64];
    size_t k = 0;
    for (size_t i = 0; s[i] && k + 1 < sizeof t; i++) {
        char c = (char)tolower((unsigned char)s[i]);
        if (c == '-' || c == '_') continue;
        t[k++] = c;
    }
    t[k] = 0;
    if (!strcmp(t, "complete")) return 0;
    if (!strcmp(t, "nomultiplicativetrend")) return 1;
    if (!strcmp(t, "dampedtrendonly")) return 2;
    if (!strcmp(t, "matcherrorseasonal")) return 3;
    if (!strcmp(t, "reduced")) return 4;
    return -1;
}

/* spec id = error*15 + trendIdx*3 + season ; trendIdx: 0 N, 1 A, 2 Ad, 3 M, 4 Md */
static EtsSpec spec_from_id(int id, int m)
{
    EtsSpec s;
    int e = id / 15, t = (id % 15) / 3, se = id % 3;
    s.error = e == 0 ? ETS_ADD : ETS_MUL;
    s.trend = (t == 0) ? ETS_NONE : (t <= 2 ? ETS_ADD : ETS_MUL);
    s.damped = (t == 2 || t == 4);
    s.season = se;
    s.m = (se != ETS_NONE) ? m : 1;
    return s;
}

static int pool_allows(int pool, const EtsSpec *s)
{
    switch (pool) {
    case 1: return s->trend != ETS_MUL;
    case 2: return s->trend == ETS_NONE || s->damped;
    case 3: return s->season == ETS_NONE || s->season == s->error;
    case 4: return s->trend != ETS_MUL && (s->season == ETS_NONE || s->season == s->error);
    default: return 1;
    }
}

/* Returns selected spec id (>= 0) or -1 when the search fails (caller falls back). */
int oracle_auto_ets_search(const double *y, int n, int period, int pool, int h, double *out,
                           double *aicc_out, int *evals_out, int *iters_out)
{
    int positive = 1, constant = 1;
    for (int i = 0; i < n; i++) { if (!(y[i] > 0.0)) positive = 0; if (y[i] != y[0]) constant = 0; }
    if (evals_out) *evals_out = 0;
    if (iters_out) *iters_out = 0;
    if (constant) return -1;   /* forecast.rs:1541-1542, :3110-3131: constant series -> fallback */
    int best = -1;
    double best_aicc = INFINITY;
    EtsFit best_fit;
    EtsSpec best_spec;
    double best_s[ETS_MAX_PERIOD];
    memset(&best_fit, 0, sizeof best_fit);
    memset(&best_spec, 0, sizeof best_spec);
    for (int id = 0; id < 30; id++) {
        EtsSpec sp = spec_from_id(id, period);
        if (sp.season != ETS_NONE && period <= 1) continue;
        if (!spec_is_valid(&sp) || !pool_allows(pool, &sp)) continue;
        if (!positive && (sp.error == ETS_MUL || sp.trend == ETS_MUL || sp.season == ETS_MUL)) continue;
        EtsFit fit;
        double sfin[ETS_MAX_PERIOD];
        if (ets_fit(&sp, y, n, &fit, sfin) != ETS_OK) continue;
        if (evals_out) *evals_out += fit.evals;
        if (iters_out) *iters_out += fit.iters;
        if (fit.aicc < best_aicc) {
            best_aicc = fit.aicc;
            best = id;
            best_fit = fit;
            best_spec = sp;
            if (sp.season != ETS_NONE) memcpy(best_s, sfin, sizeof(double) * (size_t)sp.m);
        }
    }
    if (best < 0) return -1;
    ets_forecast(&best_spec, n, &best_fit, best_s, h, out);
    if (aicc_out) *aicc_out = best_aicc;
    return best;
}

static const char *ERR_DBG[3] = { "", "Additive", "Multiplicative" };
static const char *SEAS_DBG[3] = { "None", "Additive", "Multiplicative" };
static const char *TREND_DBG[5] = { "None", "Additive", "AdditiveDamped", "Multiplicative", "MultiplicativeDamped" };

void oracle_auto_ets_name(int spec_id, char out[64])
{
    EtsSpec s = spec_from_id(spec_id, 2);
    snprintf(out, 64, "AutoETS(%s,%s,%s)", ERR_DBG[s.error], TREND_DBG[(spec_id % 15) / 3], SEAS_DBG[s.season]);
}

/* forecast.rs:1327-1336 -- ETS without spec, also the AutoETS fallback */
static int m_ets_default(const double *y, size_t n, int h, size_t period, double *out, Err *err)
{
    if (period > 1 && n >= 2 * period) return m_holt_winters(y, n, h, period, out, err);
    if (n >= 10) return m_holt(y, n, h, out, err);
    return m_ses(y, n, h, 0, out, err);
}

/* ------------------------------------------------------------------------- */
/* intervals / fitted (forecast.rs:2558-2643)                                 */
/* ------------------------------------------------------------------------- */

static void confidence_intervals(const double *point, int h, const double *y, size_t n, double conf,
                                 double *lo, double *hi)
{
    double s = 0.0;
    for (size_t i = 0; i < n; i++) s += y[i];
    double mean = s / (double)n;
    double v = 0.0;
    for (size_t i = 0; i < n; i++) { double d = y[i] - mean; v += d * d; }
    double sd = sqrt(v / (double)n);
    double z = conf >= 0.99 ? 2.576 : conf >= 0.95 ? 1.96 : conf >= 0.90 ? 1.645 : conf >= 0.80 ? 1.28 : 1.0;
    for (int i = 0; i < h; i++) {
        double w = z * sd * sqrt((double)(i + 1));
        lo[i] = point[i] - w;
        hi[i] = point[i] + w;
    }
}

static void fitted_values(const double *y, size_t n, ModelType model, size_t period, double *f)
{
    if (model == M_Naive) {
        f[0] = y[0];
        for (size_t i = 1; i < n; i++) f[i] = y[i - 1];
    } else if (model == M_SeasonalNaive) {
        size_t p = period < 1 ? 1 : period;
        if (p > n) p = n;
        for (size_t i = 0; i < p; i++) f[i] = y[0];
        for (size_t i = p; i < n; i++) f[i] = y[i - p];
    } else {
        /* SeasonalWindowAverage has its own rule in the reference but the model is out of scope. */
        double level = y[0];
        f[0] = level;
        for (size_t i = 1; i < n; i++) {
            f[i] = level;
            level = 0.3 * y[i] + (1.0 - 0.3) * level;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* forecast() + FFI (forecast.rs:512-733, lib.rs:3344-3550)                   */
/* ------------------------------------------------------------------------- */

static int is_non_seasonal_model(ModelType m)
{
    switch (m) {
    case M_Naive: case M_SES: case M_SESOptimized: case M_Holt: case M_RandomWalkDrift: case M_ARIMA:
    case M_CrostonClassic: case M_CrostonOptimized: case M_CrostonSBA: case M_TSB: case M_ADIDA: case M_IMAPA:
        return 1;
    default: return 0;
    }
}

static int run_forecast(const double *y, size_t n, const ForecastOptions *o, ModelType model,
                        double *point, char name[64], Err *err)
{
    const int h = o->horizon;
    size_t period;
    if (o->auto_detect_seasonality && o->seasonal_period == 0) {
        int p = oracle_detect_seasonality_first(y, n);
        period = p > 0 ? (size_t)p : 1;
    } else if (o->seasonal_period > 0) period = (size_t)o->seasonal_period;
    else period = 1;

    if (!o->auto_detect_seasonality && o->seasonal_period > 1 && is_non_seasonal_model(model))
        FAIL(err, INVALID_INPUT,
             "Invalid input: Model '%s' does not use seasonal_period (got %d). For seasonal forecasting, use: "
             "SeasonalNaive, HoltWinters, SeasonalES, AutoETS, AutoMFLES, AutoMSTL, or AutoTBATS.",
             MODEL_NAMES[model], o->seasonal_period);

    /* periods the kernels cannot hold fail loudly (never a silent non-seasonal fit): the checker states the same rule as
     * csrc/host_api.hip run_group.  The reference takes any period (forecast.rs:528-537). */
    if (period > ETS_MAX_PERIOD &&
        (model == M_AutoETS || model == M_HoltWinters || model == M_SeasonalES || model == M_SeasonalESOptimized ||
         (model == M_ETS && (!o->ets_model[0] || (valid_ets_notation(o->ets_model) && spec_from_notation(o->ets_model).season != ETS_NONE)))))
        FAIL(err, COMPUTATION_ERROR, "Computation error: %s fit failed: unsupported seasonal period (periods above %d are not supported)",
             MODEL_NAMES[model], ETS_MAX_PERIOD);

    /* AutoARIMA: a DETECTED period (the ACF heuristic of seasonality.rs:323-377 on a call without seasonal_period) goes to the
     * seasonal search exactly like an explicit one -- forecast.rs:528-537 hands it to forecast_auto_arima, :1448-1452 passes any
     * period > 1 to with_seasonal_period.  (Rounds 2-3 made detected periods above 24 non-seasonal: a product limit, not the
     * reference's behaviour.)  Above the cap it fails loudly either way. */
    if (model == M_AutoARIMA && period > ARIMA_MAX_PERIOD)
        FAIL(err, COMPUTATION_ERROR, "Computation error: AutoARIMA fit failed: unsupported seasonal period (periods above %d are not supported)",
             ARIMA_MAX_PERIOD);

    name[0] = 0;
    switch (model) {
    case M_Naive: m_naive(y, n, h, point); break;
    case M_SeasonalNaive: m_seasonal_naive(y, n, h, period, point); break;
    case M_SMA: {
        size_t w = o->window > 0 ? (size_t)o->window : (period > 3 ? period : 3);
        m_sma(y, n, h, w, point);
        break;
    }
    case M_RandomWalkDrift: m_drift(y, n, h, point); break;
    case M_ARIMA: m_toy_arima(y, n, h, point); break;
    case M_SES: if (!m_ses(y, n, h, 0, point, err)) return 0; break;
    case M_SESOptimized: if (!m_ses(y, n, h, 1, point, err)) return 0; break;
    case M_Holt: if (!m_holt(y, n, h, point, err)) return 0; break;
    case M_HoltWinters: if (!m_holt_winters(y, n, h, period, point, err)) return 0; break;
    case M_SeasonalES: if (!m_seasonal_es(y, n, h, period, 0, point, err)) return 0; break;
    case M_SeasonalESOptimized: if (!m_seasonal_es(y, n, h, period, 1, point, err)) return 0; break;
    case M_ETS: {
        if (o->ets_model[0]) {
            const char *nt = o->ets_model;
            if (!valid_ets_notation(nt))
                FAIL(err, INVALID_INPUT,
                     "Invalid input: Invalid ETS model specification '%s'. Expected 3 or 4 character notation: "
                     "Error(A/M) + Trend(A/M/N) + Seasonal(A/M/N), with optional 'd' for damped trend. "
                     "Examples: 'AAA' (additive), 'MNM' (multiplicative error, no trend), 'AAdA' (additive damped trend). "
                     "Valid characters: A=Additive, M=Multiplicative, N=None, d=Damped.", nt);
            EtsSpec sp = spec_from_notation(nt);
            if (!spec_is_valid(&sp))
                FAIL(err, INVALID_INPUT,
                     "Invalid input: ETS model '%s' is an unstable combination (multiplicative error with additive components). "
                     "Try one of: 'AAA', 'ANA', 'AAdA', 'MNM', 'MAM', 'MAdM', 'MMM', 'MMdM', or use 'AutoETS' for automatic selection.", nt);
            int st;
            if (!m_ets_spec(y, n, h, period, sp, point, &st))
                FAIL(err, COMPUTATION_ERROR, "Computation error: ETS model '%s' failed to fit: Computation error: Failed to fit ETS model: %s",
                     nt, ets_status_text(st));
            snprintf(name, 64, "ETS(%s)", nt);
        } else {
            if (!m_ets_default(y, n, h, period, point, err)) return 0;
            strcpy(name, "ETS");
        }
        break;
    }
    case M_AutoETS: {
        int pool = 0;
        if (o->model_pool[0]) {
            pool = parse_model_pool(o->model_pool);
            if (pool < 0)
                FAIL(err, INVALID_INPUT,
                     "Invalid input: Unknown model_pool '%s'. Valid options: complete, no_multiplicative_trend, "
                     "damped_trend_only, match_error_seasonal, reduced", o->model_pool);
        }
        int id = oracle_auto_ets_search(y, (int)n, (int)period, pool, h, point, NULL, NULL, NULL);
        if (id >= 0) oracle_auto_ets_name(id, name);
        else {
            if (!m_ets_default(y, n, h, period, point, err)) return 0;
            strcpy(name, "AutoETS");
        }
        break;
    }
    case M_AutoARIMA: {
        ArimaOrder ord;
        if (!oracle_auto_arima(y, (int)n, (int)period, h, point, &ord))
            FAIL(err, COMPUTATION_ERROR, "Computation error: AutoARIMA fit failed: no admissible model");
        oracle_arima_name(&ord, name);
        break;
    }
    default:
        FAIL(err, INTERNAL_ERROR, "Internal error: model '%s' is not implemented by the HIP backend", MODEL_NAMES[model]);
    }
    if (!name[0]) strcpy(name, MODEL_NAMES[model]);
    return 1;
}

bool oracle_ts_forecast(const double *values, const uint64_t *validity, size_t length,
                        const ForecastOptions *options, ForecastResult *out, AnofoxError *out_error)
{
    if (out_error) { out_error->code = SUCCESS; memset(out_error->message, 0, sizeof out_error->message); }
    if (!values || !options || !out) { set_error(out_error, NULL_POINTER, "Null pointer argument"); return false; }

    char mname[33];
    memcpy(mname, options->model, 32);
    mname[32] = 0;
    ModelType model;
    Err err = { 0, "" };
    if (!parse_model(mname, &model)) {
        snprintf(err.msg, sizeof err.msg, "Invalid model: Unknown model: '%s'", mname);
        set_error(out_error, INVALID_MODEL, err.msg);
        return false;
    }
    if (options->horizon < 0) { set_error(out_error, PANIC_CAUGHT, "Panic in Rust code"); return false; }

    double *y = (double *)malloc((length ? length : 1) * sizeof(double));
    oracle_fill_nulls_interpolate(values, validity, length, y);
    if (length == 0) {
        free(y);
        set_error(out_error, INSUFFICIENT_DATA, "Insufficient data: need at least 1 observations, got 0");
        return false;
    }
    if (length < 3) {
        free(y);
        snprintf(err.msg, sizeof err.msg, "Insufficient data: need at least 3 observations, got %zu", length);
        set_error(out_error, INSUFFICIENT_DATA, err.msg);
        return false;
    }

    const int h = options->horizon;
    double *point = (double *)malloc((size_t)(h > 0 ? h : 1) * sizeof(double));
    char name[64];
    if (!run_forecast(y, length, options, model, point, name, &err)) {
        free(y);
        free(point);
        set_error(out_error, err.code, err.msg);
        return false;
    }

    memset(out, 0, sizeof *out);
    out->n_forecasts = (size_t)h;
    if (h > 0) {
        out->point_forecasts = point;
        out->lower_bounds = (double *)malloc((size_t)h * sizeof(double));
        out->upper_bounds = (double *)malloc((size_t)h * sizeof(double));
        confidence_intervals(point, h, y, length, options->confidence_level, out->lower_bounds, out->upper_bounds);
    } else free(point);

    out->mse = NAN;
    if (options->include_fitted || options->include_residuals) {
        size_t period = 1;
        if (options->auto_detect_seasonality && options->seasonal_period == 0) {
            int p = oracle_detect_seasonality_first(y, length);
            period = p > 0 ? (size_t)p : 1;
        } else if (options->seasonal_period > 0) period = (size_t)options->seasonal_period;
        double *f = (double *)malloc(length * sizeof(double));
        fitted_values(y, length, model, period, f);
        double sse = 0.0;
        for (size_t i = 0; i < length; i++) { double d = y[i] - f[i]; sse += d * d; }
        out->mse = sse / (double)length;
        if (options->include_residuals) {
            out->residuals = (double *)malloc(length * sizeof(double));
            for (size_t i = 0; i < length; i++) out->residuals[i] = y[i] - f[i];
        }
        if (options->include_fitted) { out->fitted_values = f; out->n_fitted = length; }
        else free(f);
    }
    size_t ln = strlen(name);
    if (ln > 63) ln = 63;
    memcpy(out->model_name, name, ln);
    out->model_name[ln] = 0;
    out->aic = NAN;
    out->bic = NAN;
    free(y);
    return true;
}

void oracle_free_forecast_result(ForecastResult *r)
{
    if (!r) return;
    free(r->point_forecasts); r->point_forecasts = NULL;
    free(r->lower_bounds); r->lower_bounds = NULL;
    free(r->upper_bounds); r->upper_bounds = NULL;
    free(r->fitted_values); r->fitted_values = NULL;
    free(r->residuals); r->residuals = NULL;
}

/* test hook: fit one spec id, report optimiser effort and criteria */
int oracle_ets_fit_spec(const double *y, int n, int spec_id, int m, int *iters, int *evals, double *aicc, double *par)
{
    EtsSpec sp = spec_from_id(spec_id, m);
    EtsFit fit;
    double sfin[ETS_MAX_PERIOD];
    int st = ets_fit(&sp, y, n, &fit, sfin);
    if (iters) *iters = fit.iters;
    if (evals) *evals = fit.evals;
    if (aicc) *aicc = fit.aicc;
    if (par) for (int i = 0; i < ETS_MAX_DIM; i++) par[i] = fit.par[i];
    return st;
}

/* test hook for the inspection outputs: fit spec `spec_id` -- or, when it is negative, the spec AutoETS
 * selects from `pool` -- and report the parameters in model terms, the criteria, the final states and the one-step fitted
 * values.  Returns the spec id, -1 when no spec can be fitted. */
extern double *ets_fitted_sink;
int oracle_ets_inspect(const double *y, int n, int period, int pool, int spec_id, double *par8, double *states, double *fitted)
{
    int best = spec_id;
    if (spec_id < 0) {
        int positive = 1, constant = 1;
        for (int i = 0; i < n; i++) { if (!(y[i] > 0.0)) positive = 0; if (y[i] != y[0]) constant = 0; }
        if (constant) return -1;
        double best_aicc = INFINITY;
        for (int id = 0; id < 30; id++) {
            EtsSpec sp = spec_from_id(id, period);
            if (sp.season != ETS_NONE && period <= 1) continue;
            if (!spec_is_valid(&sp) || !pool_allows(pool, &sp)) continue;
            if (!positive && (sp.error == ETS_MUL || sp.trend == ETS_MUL || sp.season == ETS_MUL)) continue;
            EtsFit fit;
            double sfin[ETS_MAX_PERIOD];
            if (ets_fit(&sp, y, n, &fit, sfin) != ETS_OK) continue;
            if (fit.aicc < best_aicc) { best_aicc = fit.aicc; best = id; }
        }
        if (best < 0) return -1;
    }
    EtsSpec sp = spec_from_id(best, period);
    EtsFit fit;
    double sfin[ETS_MAX_PERIOD], s0[ETS_MAX_PERIOD];
    if (ets_fit(&sp, y, n, &fit, sfin) != ETS_OK) return -1;
    ets_init_states(&sp, y, n, &fit.l0, &fit.b0, s0);
    ets_fitted_sink = fitted;
    ets_lik(&sp, y, n, fit.par, fit.l0, fit.b0, s0, NULL, NULL, NULL, NULL);
    ets_fitted_sink = NULL;
    par8[0] = fit.alpha;
    par8[1] = sp.trend != ETS_NONE ? fit.alpha * fit.beta_star : NAN;
    par8[2] = sp.season != ETS_NONE ? fit.gamma_star * (1.0 - fit.alpha) : NAN;
    par8[3] = sp.damped ? fit.phi : NAN;
    par8[4] = fit.aic; par8[5] = fit.aicc; par8[6] = fit.bic; par8[7] = fit.sse;
    states[0] = fit.l; states[1] = fit.b;
    if (sp.season != ETS_NONE) for (int j = 0; j < sp.m; j++) states[2 + j] = sfin[j];
    return best;
}

/* ETS(notation) with given smoothing parameters over many series (BASELINE config 2), OpenMP over series: the checker of
 * anofox_hip_batch_set_fixed_params and the CPU baseline of bench.py --workload ets_aaa_fixed_m5.
 * status: 0 ok, else ErrorCode (6 insufficient data, 3 computation error). */
int oracle_ets_fixed_batch(const double *values, const int64_t *offsets, size_t n_series, const char *notation, int period,
                           double alpha, double beta, double gamma, double phi, int h, double conf,
                           double *yhat, double *lo, double *hi, int32_t *status, int n_threads)
{
    if (!valid_ets_notation(notation)) return -1;
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
    for (long s = 0; s < (long)n_series; s++) {
        const double *y = values + offsets[s];
        const size_t n = (size_t)(offsets[s + 1] - offsets[s]);
        if (n < 3) { status[s] = INSUFFICIENT_DATA; continue; }
        EtsSpec spec = spec_from_notation(notation);
        if (spec.season != ETS_NONE && period > 1) spec.m = period;
        else { spec.season = ETS_NONE; spec.m = 1; }
        EtsFit fit;
        double sfin[ETS_MAX_PERIOD];
        if (ets_fit_fixed(&spec, y, (int)n, alpha, beta, gamma, phi, &fit, sfin) != ETS_OK) { status[s] = COMPUTATION_ERROR; continue; }
        ets_forecast(&spec, (int)n, &fit, sfin, h, yhat + (size_t)s * h);
        confidence_intervals(yhat + (size_t)s * h, h, y, n, conf, lo + (size_t)s * h, hi + (size_t)s * h);
        status[s] = 0;
    }
    return n_threads;
}

/* test hooks */
double oracle_det_log(double x) { return det_log(x); }
double oracle_det_exp(double x) { return det_exp(x); }
double oracle_det_pow_step(double x, double y) { return det_pow_step(x, y); }
