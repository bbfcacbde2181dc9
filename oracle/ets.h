/*
 * ets.h -- TEST INFRASTRUCTURE (oracle).  Not part of the product.
 *
 * CPU restatement of the ETS arithmetic that the reference delegates to the
 * un-vendored crate anofox-forecast 0.15.3 (call sites
 * crates/anofox-fcst-core/src/forecast.rs:1104,1113,1122,1136,1213,1227 for the
 * SES/Holt/HoltWinters/SeasonalES family, :1300,1357-1367 for ETS(spec),
 * :1571-1591 for AutoETS).  The crate source is not in /root/reference, so the
 * algorithm below is the published innovations-state-space ETS (Hyndman et al.
 * 2008; the StatsForecast lineage named in THIRD_PARTY_NOTICES.md:28-48) with
 * every free choice fixed by the reference's own known-answer vectors
 * (test/sql/ts_model_distinctness.test:116,141,164) -- see DESIGN.md section 3.
 */
#ifndef ORACLE_ETS_H
#define ORACLE_ETS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ETS_NONE = 0, ETS_ADD = 1, ETS_MUL = 2 };

typedef struct EtsSpec {
    int error;   /* ETS_ADD | ETS_MUL */
    int trend;   /* ETS_NONE | ETS_ADD | ETS_MUL */
    int damped;  /* 0 | 1 (only with a trend) */
    int season;  /* ETS_NONE | ETS_ADD | ETS_MUL */
    int m;       /* seasonal period, 1 when season == ETS_NONE */
} EtsSpec;

#define ETS_MAX_PERIOD 2048
#define ETS_MAX_DIM 4

/* Bounds and starting point of the smoothing parameters (alpha, beta*, gamma*, phi). */
#define ETS_PAR_LO 1.0e-4
#define ETS_PAR_HI 0.9999
#define ETS_PHI_LO 0.8
#define ETS_PHI_HI 0.98
#define ETS_ALPHA0 0.3
#define ETS_BETA0 0.1
#define ETS_GAMMA0 0.1
#define ETS_PHI0 0.9

typedef struct EtsFit {
    int status;            /* 0 ok, else EtsStatus */
    int dim;               /* number of optimised smoothing parameters */
    double par[ETS_MAX_DIM];   /* alpha, [beta*], [gamma*], [phi] in that order */
    double alpha, beta_star, gamma_star, phi; /* unpacked (phi = 1 when undamped) */
    double l0, b0;         /* initial level / growth */
    double lik;            /* -2 log L up to a constant: n log(SSE) [+ 2 sum log|f|] */
    double sse;
    double aic, aicc, bic;
    int n_param;           /* k used by the information criteria */
    int iters;             /* Nelder-Mead iterations */
    int evals;             /* objective evaluations (sequential count) */
    double l, b;           /* final states */
    /* seasonal ring: s[j] is the state of phase j = t mod m */
} EtsFit;

enum EtsStatus {
    ETS_OK = 0,
    ETS_ERR_SHORT = 1,        /* not enough observations for this spec */
    ETS_ERR_NONPOSITIVE = 2,  /* multiplicative component on data with min <= 0 */
    ETS_ERR_NONFINITE = 3,    /* likelihood not finite at the optimum */
    ETS_ERR_PERIOD = 4,       /* seasonal spec with m < 2 or m > ETS_MAX_PERIOD */
};

/* Number of smoothing parameters / information-criterion parameter count. */
int ets_dim(const EtsSpec *spec);
int ets_n_param(const EtsSpec *spec);

/* Initial states: level, growth, seasonal ring s0[m] (phase-indexed). */
int ets_init_states(const EtsSpec *spec, const double *y, int n,
                    double *l0, double *b0, double *s0 /* [m] */);

/*
 * One likelihood pass.  Returns the objective (lik, +inf if inadmissible).
 * Optionally returns SSE and the final states (s_out[m] may be NULL).
 */
double ets_lik(const EtsSpec *spec, const double *y, int n, const double *par,
               double l0, double b0, const double *s0,
               double *sse_out, double *l_out, double *b_out, double *s_out);

/* Fit (Nelder-Mead over the smoothing parameters) + final pass. s_final[m]. */
int ets_fit(const EtsSpec *spec, const double *y, int n, EtsFit *fit, double *s_final);

/* Given smoothing parameters in the model's own terms (no optimiser): one pass + final states. */
int ets_fit_fixed(const EtsSpec *spec, const double *y, int n, double alpha, double beta, double gamma, double phi,
                  EtsFit *fit, double *s_final);

/* h-step point forecasts from final states. */
void ets_forecast(const EtsSpec *spec, int n, const EtsFit *fit, const double *s_final,
                  int h, double *out);

/* Bounded scipy-style Nelder-Mead (shared by the SES/Holt/HoltWinters family). */
typedef double (*nm_fn)(const double *x, void *ctx);
typedef struct NmResult { double x[ETS_MAX_DIM]; double f; int iters; int evals; } NmResult;
void nm_minimize(nm_fn fn, void *ctx, int n, const double *x0,
                 const double *lo, const double *hi, NmResult *res);

#ifdef __cplusplus
}
#endif
#endif
