/*
 * ets.c -- TEST INFRASTRUCTURE (oracle).  Not part of the product.
 *
 * Scalar fp64 restatement of the ETS engine (see ets.h for provenance).
 * Compile with -ffp-contract=off: every expression below is a fixed sequence of
 * IEEE binary64 operations, fma() appears only where written, and the HIP
 * kernels perform the same sequence, so oracle and GPU agree bit for bit.
 */
#include "ets.h"
#include "det_math.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ETS_TOL 1.0e-10
#define ETS_HUGEN 1.0e10
#define ETS_LN2 0.693147180559945309417232121458

/* ------------------------------------------------------------------------- */
/* Nelder-Mead: scipy.optimize._minimize_neldermead semantics (non-adaptive), */
/* bounds by clipping, xatol = 1e-4, fatol = 1e-8, maxiter = maxfev = 200 n.  */
/* Pinned by the reference KATs SESOptimized 19.537535, Holt 20.330877,       */
/* HoltWinters 19.953912 (test/sql/ts_model_distinctness.test:116,141).       */
/* ------------------------------------------------------------------------- */

#define NM_XATOL 1.0e-4
#define NM_FATOL 1.0e-8
#define NM_NONZDELT 0.05
#define NM_ZDELT 0.00025

static double clipd(double v, double lo, double hi)
{
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}

/* test hook (tools/nm_outcomes.py): when set, every iteration adds one to nm_outcome_sink[6 * (n - 1) + kind], kind = 0 expansion kept,
 * 1 reflection kept as the new best, 2 reflection kept elsewhere, 3 outside contraction, 4 inside contraction, 5 shrink; and to
 * nm_outcome_sink[24 + 8 * (n - 1) + j] for the position j the new vertex is inserted at (shrinks excluded), and to the table of
 * consecutive outcomes nm_outcome_sink[56 + 676 * (n - 1) + 26 * previous + current], outcome = 5 * kind + j (25 = shrink) */
long long *nm_outcome_sink = NULL;
/* test hook (tools/nm_spread.py): when set, every iteration appends {iteration, largest |x_k - x_0| / xatol, largest |f_k - f_0| / fatol}
 * of the sorted simplex it starts from, and the end of a run appends {-1, iterations, 0} -- how well the simplex's size predicts how many
 * iterations a problem still needs (would sorting the running problems by it make the waves of a round finish together?) */
double *nm_spread_sink = NULL;
long nm_spread_pos = 0, nm_spread_cap = 0;

void nm_minimize(nm_fn fn, void *ctx, int n, const double *x0,
                 const double *lo, const double *hi, NmResult *res)
{
    double sim[ETS_MAX_DIM + 1][ETS_MAX_DIM] = {{0}};
    double fs[ETS_MAX_DIM + 1] = {0};
    double xb[ETS_MAX_DIM] = {0}, xt[ETS_MAX_DIM] = {0}, xr[ETS_MAX_DIM] = {0};
    const int maxiter = 200 * n, maxfun = 200 * n;
    int evals = 0, iters = 1;
    int prev_outcome = -1;

    for (int i = 0; i < n; i++) sim[0][i] = clipd(x0[i], lo[i], hi[i]);
    for (int k = 0; k < n; k++) {
        for (int i = 0; i < n; i++) sim[k + 1][i] = sim[0][i];
        double v = sim[0][k];
        v = (v != 0.0) ? (1.0 + NM_NONZDELT) * v : NM_ZDELT;
        sim[k + 1][k] = clipd(v, lo[k], hi[k]);
    }
    for (int k = 0; k <= n; k++) { fs[k] = fn(sim[k], ctx); evals++; }

    /* stable insertion sort by f */
    for (int k = 1; k <= n; k++) {
        double fk = fs[k], tmp[ETS_MAX_DIM];
        memcpy(tmp, sim[k], sizeof tmp);
        int j = k;
        while (j > 0 && fk < fs[j - 1]) {
            fs[j] = fs[j - 1];
            memcpy(sim[j], sim[j - 1], sizeof tmp);
            j--;
        }
        fs[j] = fk;
        memcpy(sim[j], tmp, sizeof tmp);
    }

    while (evals < maxfun && iters < maxiter) {
        int small = 1;
        for (int k = 1; k <= n && small; k++) {
            for (int i = 0; i < n; i++)
                if (!(fabs(sim[k][i] - sim[0][i]) <= NM_XATOL)) small = 0;
            if (!(fabs(fs[0] - fs[k]) <= NM_FATOL)) small = 0;
        }
        if (nm_spread_sink && nm_spread_pos + 3 <= nm_spread_cap) {
            double dx = 0.0, df = 0.0;
            for (int k = 1; k <= n; k++) {
                for (int i = 0; i < n; i++) { const double d = fabs(sim[k][i] - sim[0][i]); if (d > dx) dx = d; }
                const double d = fabs(fs[0] - fs[k]); if (d > df) df = d;
            }
            nm_spread_sink[nm_spread_pos++] = (double)iters; nm_spread_sink[nm_spread_pos++] = dx / NM_XATOL; nm_spread_sink[nm_spread_pos++] = df / NM_FATOL;
        }
        if (small) break;

        for (int i = 0; i < n; i++) {
            double s = sim[0][i];
            for (int k = 1; k < n; k++) s = s + sim[k][i];
            xb[i] = s / (double)n;
        }
        const double *xw = sim[n];
        for (int i = 0; i < n; i++) xr[i] = clipd(2.0 * xb[i] - xw[i], lo[i], hi[i]);
        double fxr = fn(xr, ctx); evals++;
        int doshrink = 0;
        double fnew = 0.0; const double *xnew = NULL;

        int kind = 5;
        if (fxr < fs[0]) {
            for (int i = 0; i < n; i++) xt[i] = clipd(3.0 * xb[i] - 2.0 * xw[i], lo[i], hi[i]);
            double fxe = fn(xt, ctx); evals++;
            if (fxe < fxr) { xnew = xt; fnew = fxe; kind = 0; } else { xnew = xr; fnew = fxr; kind = 1; }
        } else if (fxr < fs[n - 1]) {
            xnew = xr; fnew = fxr; kind = 2;
        } else if (fxr < fs[n]) {
            for (int i = 0; i < n; i++) xt[i] = clipd(1.5 * xb[i] - 0.5 * xw[i], lo[i], hi[i]);
            double fxc = fn(xt, ctx); evals++;
            if (fxc <= fxr) { xnew = xt; fnew = fxc; kind = 3; } else doshrink = 1;
        } else {
            for (int i = 0; i < n; i++) xt[i] = clipd(0.5 * xb[i] + 0.5 * xw[i], lo[i], hi[i]);
            double fxcc = fn(xt, ctx); evals++;
            if (fxcc < fs[n]) { xnew = xt; fnew = fxcc; kind = 4; } else doshrink = 1;
        }

        if (!doshrink) {
            /* replace the worst vertex and re-insert (stable) */
            double tmp[ETS_MAX_DIM];
            for (int i = 0; i < n; i++) tmp[i] = xnew[i];
            int j = n;
            while (j > 0 && fnew < fs[j - 1]) {
                fs[j] = fs[j - 1];
                memcpy(sim[j], sim[j - 1], sizeof tmp);
                j--;
            }
            fs[j] = fnew;
            for (int i = 0; i < n; i++) sim[j][i] = tmp[i];
            if (nm_outcome_sink) {
                __atomic_fetch_add(&nm_outcome_sink[24 + 8 * (n - 1) + j], 1, __ATOMIC_RELAXED);
                const int outcome = 5 * kind + j;
                if (prev_outcome >= 0) __atomic_fetch_add(&nm_outcome_sink[56 + 676 * (n - 1) + 26 * prev_outcome + outcome], 1, __ATOMIC_RELAXED);
                prev_outcome = outcome;
            }
        } else {
            if (nm_outcome_sink) {
                if (prev_outcome >= 0) __atomic_fetch_add(&nm_outcome_sink[56 + 676 * (n - 1) + 26 * prev_outcome + 25], 1, __ATOMIC_RELAXED);
                prev_outcome = 25;
            }
            for (int k = 1; k <= n; k++) {
                for (int i = 0; i < n; i++)
                    sim[k][i] = clipd(sim[0][i] + 0.5 * (sim[k][i] - sim[0][i]), lo[i], hi[i]);
                fs[k] = fn(sim[k], ctx); evals++;
            }
            for (int k = 1; k <= n; k++) {
                double fk = fs[k], tmp[ETS_MAX_DIM];
                memcpy(tmp, sim[k], sizeof tmp);
                int j = k;
                while (j > 0 && fk < fs[j - 1]) {
                    fs[j] = fs[j - 1];
                    memcpy(sim[j], sim[j - 1], sizeof tmp);
                    j--;
                }
                fs[j] = fk;
                memcpy(sim[j], tmp, sizeof tmp);
            }
        }
        if (nm_outcome_sink) __atomic_fetch_add(&nm_outcome_sink[6 * (n - 1) + (doshrink ? 5 : kind)], 1, __ATOMIC_RELAXED);
        iters++;
    }
    if (nm_spread_sink && nm_spread_pos + 3 <= nm_spread_cap) { nm_spread_sink[nm_spread_pos++] = -1.0; nm_spread_sink[nm_spread_pos++] = (double)iters; nm_spread_sink[nm_spread_pos++] = 0.0; }
    for (int i = 0; i < n; i++) res->x[i] = sim[0][i];
    res->f = fs[0];
    res->iters = iters;
    res->evals = evals;
}

/* ------------------------------------------------------------------------- */
/* ETS                                                                        */
/* ------------------------------------------------------------------------- */

int ets_dim(const EtsSpec *s)
{
    return 1 + (s->trend != ETS_NONE) + (s->season != ETS_NONE) + (s->damped ? 1 : 0);
}

int ets_n_param(const EtsSpec *s)
{
    int n_states = 1 + (s->trend != ETS_NONE) + (s->season != ETS_NONE ? s->m - 1 : 0);
    return ets_dim(s) + n_states + 1;
}

static int ets_is_additive_class(const EtsSpec *s)
{
    return s->error == ETS_ADD && s->trend != ETS_MUL && s->season != ETS_MUL;
}

/*
 * Initial states.
 *  seasonal figure: classical decomposition (centred moving average, per-phase
 *                   mean of the detrended series, normalised to mean 0 / 1);
 *  level / growth : least squares of the seasonally adjusted series on
 *                   t = 1..n over the WHOLE sample (pinned by the AutoETS KAT
 *                   19.956521 = the full-sample trend line, distinctness test :164);
 *  level only     : mean of the first min(n, max(10, 2m)) adjusted values.
 */
int ets_init_states(const EtsSpec *spec, const double *y, int n,
                    double *l0_out, double *b0_out, double *s0)
{
    const int m = spec->m;
    double fig[ETS_MAX_PERIOD];
    if (spec->season != ETS_NONE) {
        if (m < 2 || m > ETS_MAX_PERIOD) return ETS_ERR_PERIOD;
        if (n < 2 * m) return ETS_ERR_SHORT;
        const int half = m / 2;
        const int L = (m % 2 == 0) ? m + 1 : m;
        const double w = 1.0 / (double)m;
        const double wend = (m % 2 == 0) ? 0.5 / (double)m : w;
        double sum[ETS_MAX_PERIOD];
        int cnt[ETS_MAX_PERIOD];
        for (int j = 0; j < m; j++) { sum[j] = 0.0; cnt[j] = 0; }
        for (int i = half; i < n - half; i++) {
            double acc = 0.0;
            for (int k = 0; k < L; k++) {
                double wk = (k == 0 || k == L - 1) ? wend : w;
                acc = acc + wk * y[i - half + k];
            }
            double d = (spec->season == ETS_ADD) ? (y[i] - acc) : (y[i] / acc);
            sum[i % m] = sum[i % m] + d;
            cnt[i % m]++;
        }
        double tot = 0.0;
        for (int j = 0; j < m; j++) { fig[j] = sum[j] / (double)cnt[j]; tot = tot + fig[j]; }
        double mean = tot / (double)m;
        for (int j = 0; j < m; j++) {
            if (spec->season == ETS_ADD) fig[j] = fig[j] - mean;
            else {
                fig[j] = fig[j] / mean;
                if (!(fig[j] >= 1.0e-2)) fig[j] = 1.0e-2;
            }
            s0[j] = fig[j];
        }
    } else if (s0) {
        s0[0] = 0.0;
    }

    /* seasonally adjusted value i */
#define YSA(i) (spec->season == ETS_NONE ? y[(i)] : \
               (spec->season == ETS_ADD ? y[(i)] - fig[(i) % m] : y[(i)] / fig[(i) % m]))

    double l0, b0 = 0.0;
    if (spec->trend == ETS_NONE) {
        int K = 2 * m > 10 ? 2 * m : 10;
        if (K > n) K = n;
        double s = 0.0;
        for (int i = 0; i < K; i++) s = s + YSA(i);
        l0 = s / (double)K;
    } else {
        /* sums of the least-squares line: sy = sum ysa_i, sxy = sum (i + 1) ysa_i.  Without a seasonal component they are the two
         * running sums.  With one, the adjusted value is y_i - fig[i mod m] (y_i / fig[i mod m]), and the sums are taken PER PHASE
         * first -- Sy_p = sum of y_i, Sxy_p = sum of (i + 1) y_i over the i of phase p, in time order -- and the figure applied once
         * per phase: sy = sum_p (Sy_p - n_p fig_p), sxy = sum_p (Sxy_p - Sx_p fig_p), n_p / Sx_p = the count / the sum of (i + 1) of
         * the phase (exact integers); for a multiplicative figure sum_p Sy_p / fig_p.  The same numbers up to rounding (round 5:
         * the figures are only known at the END of a sweep over the series, and this form needs no second one). */
        double sy = 0.0, sxy = 0.0;
        if (spec->season == ETS_NONE) {
            for (int i = 0; i < n; i++) {
                double v = y[i];
                sy = sy + v;
                sxy = sxy + (double)(i + 1) * v;
            }
        } else {
            double psy[ETS_MAX_PERIOD], psxy[ETS_MAX_PERIOD];
            for (int j = 0; j < m; j++) { psy[j] = 0.0; psxy[j] = 0.0; }
            for (int i = 0; i < n; i++) {
                int j = i % m;
                psy[j] = psy[j] + y[i];
                psxy[j] = psxy[j] + (double)(i + 1) * y[i];
            }
            for (int j = 0; j < m; j++) {
                if (j >= n) break;
                long long cnt = ((long long)n - j + m - 1) / m;                       /* i = j, j + m, ... < n */
                long long sx = cnt * (long long)(j + 1) + (long long)m * (cnt * (cnt - 1) / 2);
                if (spec->season == ETS_ADD) {
                    sy = sy + (psy[j] - (double)cnt * fig[j]);
                    sxy = sxy + (psxy[j] - (double)sx * fig[j]);
                } else {
                    sy = sy + psy[j] / fig[j];
                    sxy = sxy + psxy[j] / fig[j];
                }
            }
        }
        double dn = (double)n;
        double sx = dn * (dn + 1.0) / 2.0;
        double sxx = dn * (dn + 1.0) * (2.0 * dn + 1.0) / 6.0;
        double slope = (dn * sxy - sx * sy) / (dn * sxx - sx * sx);
        double icpt = (sy - slope * sx) / dn;
        if (spec->trend == ETS_ADD) {
            l0 = icpt;
            b0 = slope;
            if (fabs(l0 + b0) < 1.0e-8) { l0 = l0 * (1.0 + 1.0e-3); b0 = b0 * (1.0 - 1.0e-3); }
        } else {
            l0 = icpt + slope;
            if (fabs(l0) < 1.0e-8) l0 = 1.0e-7;
            b0 = (icpt + 2.0 * slope) / l0;
            l0 = l0 / b0;
            if (fabs(b0) > 1.0e10) b0 = (b0 < 0.0 ? -1.0e10 : 1.0e10);
            if (l0 < 1.0e-8 || b0 < 1.0e-8) {
                double y0 = YSA(0), y1 = YSA(1);
                l0 = y0 > 1.0e-3 ? y0 : 1.0e-3;
                double r = y1 / y0;
                b0 = r > 1.0e-3 ? r : 1.0e-3;
            }
        }
    }
#undef YSA
    *l0_out = l0;
    *b0_out = b0;
    return ETS_OK;
}

static void ets_unpack(const EtsSpec *spec, const double *par,
                       double *alpha, double *bstar, double *gstar, double *phi)
{
    int k = 0;
    *alpha = par[k++];
    *bstar = (spec->trend != ETS_NONE) ? par[k++] : 0.0;
    *gstar = (spec->season != ETS_NONE) ? par[k++] : 0.0;
    *phi = spec->damped ? par[k++] : 1.0;
}

/*
 * One pass of the innovations recursion.
 *
 * additive class (error A, trend N/A/Ad, season N/A) -- error-correction form
 *     q = l + phi b ; f = q + s ; e = y - f
 *     l' = q + alpha e ; b' = phi b + (alpha beta*) e ; s' = s + (gamma* (1-alpha)) e
 * every other spec -- the general form of Hyndman's etscalc (forecast R package /
 *     StatsForecast ets.py `update`), with beta/alpha = beta*, gamma = gamma*(1-alpha).
 */
/* test hook (oracle_ets_inspect): when set, every pass also stores its one-step forecasts here */
double *ets_fitted_sink = NULL;
/* test hook (tools/far_blocks.py): when set, every likelihood pass of a damped multiplicative-trend spec appends ets_far_blocks bytes
 * to the sink -- byte k = 1 when a step of time block [8 k, 8 k + 8) took the table-driven power (|b - 1| > 1/16) -- as long as room is left */
unsigned char *ets_far_sink = NULL;
long ets_far_pos = 0, ets_far_cap = 0;
int ets_far_blocks = 0;

double ets_lik(const EtsSpec *spec, const double *y, int n, const double *par,
               double l0, double b0, const double *s0,
               double *sse_out, double *l_out, double *b_out, double *s_out)
{
    const int m = spec->m;
    double alpha, bstar, gstar, phi;
    ets_unpack(spec, par, &alpha, &bstar, &gstar, &phi);
    const double beta = alpha * bstar;
    const double gamma = gstar * (1.0 - alpha);
    double sbuf[ETS_MAX_PERIOD];
    if (spec->season != ETS_NONE) for (int j = 0; j < m; j++) sbuf[j] = s0[j];
    double l = l0, b = b0, sse = 0.0;
    double mant = 1.0; long eacc = 0;
    int bad = 0;
    double powc[DET_POW_NEAR1_DEG + 1];
    if (spec->trend == ETS_MUL && spec->damped) det_pow_near1_coef(phi, powc);
    unsigned char *far_row = NULL;
    if (ets_far_sink && spec->trend == ETS_MUL && spec->damped && ets_far_pos + ets_far_blocks <= ets_far_cap) {
        far_row = ets_far_sink + ets_far_pos;
        ets_far_pos += ets_far_blocks;
        for (int k = 0; k < ets_far_blocks; k++) far_row[k] = 0;
    }

    if (ets_is_additive_class(spec)) {
        for (int t = 0; t < n; t++) {
            int j = (spec->season != ETS_NONE) ? t % m : 0;
            double phib = 0.0, q = l;
            if (spec->trend == ETS_ADD) {
                phib = spec->damped ? phi * b : b;
                q = l + phib;
            }
            double f = q;
            if (spec->season == ETS_ADD) f = q + sbuf[j];
            if (ets_fitted_sink) ets_fitted_sink[t] = f;
            double e = y[t] - f;
            sse = fma(e, e, sse);
            l = fma(alpha, e, q);
            if (spec->trend == ETS_ADD) b = fma(beta, e, phib);
            if (spec->season == ETS_ADD) sbuf[j] = fma(gamma, e, sbuf[j]);
        }
    } else if (spec->error == ETS_MUL && spec->season == ETS_ADD) {
        bad = 1;        /* never fitted: ETSSpec::is_valid() rejects it (forecast.c spec_is_valid) */
    } else {
        /*
         * Every other spec, in ERROR-CORRECTION form (round 5; Hyndman, Koehler, Ord & Snyder 2008, tables 2.2 / 2.3): the updates are
         * written on the one-step error instead of on y / s, y / q and l' / l.  Algebraically the general recursion of rounds 1-4
         * (pp - q = e / s, r - phi b = alpha e / (s l), tt - s = e / q; with a relative error eps = e / f they become q eps,
         * phib eps, s eps), but a third fewer operations and a shorter dependent chain:
         *     multiplicative error  eps = (y - f) / f :  l' = q + alpha q eps ;  b' = phib + beta q eps  (additive trend)
         *                                                b' = phib + beta phib eps (multiplicative trend) ;  s' = s + gamma s eps
         *     additive error        e = y - f          :  l' = q + alpha e / s ;  b' = phib + beta e / s  or  phib + beta e / (s l) ;
         *                                                s' = s + gamma e / q     (the quotients from ONE reciprocal of f, l or f l)
         * with beta = alpha beta*, gamma = gamma* (1 - alpha) as in the additive class.  sum log|f| is accumulated as mant * 2^eacc.
         */
        const int need_f = (spec->error == ETS_MUL) || (spec->season == ETS_MUL);
        for (int t = 0; t < n; t++) {
            int j = (spec->season != ETS_NONE) ? t % m : 0;
            double s = (spec->season != ETS_NONE) ? sbuf[j] : 0.0;
            double phib = 0.0, q = l;
            if (spec->trend == ETS_ADD) {
                phib = spec->damped ? phi * b : b;
                q = l + phib;
            } else if (spec->trend == ETS_MUL) {
                if (spec->damped) {
                    /* growth rates live next to one: the binomial series in r = b - 1 (det_math.h, round 5); anything else takes
                     * the table-driven power.  |r| <= 1/16 implies b > 0 and b in range, so the checks belong to the other branch */
                    const double r = b - 1.0;
                    if (fabs(r) <= DET_POW_NEAR1_R) phib = det_pow_near1(r, powc);
                    else {
                        /* a damped growth rate outside [2^-1000, 2^1000] (or not positive) is inadmissible: keeps phi * log(b)
                         * inside the range where exp needs no special cases, on the CPU and in the kernels alike */
                        if (!(b >= 0x1p-1000 && b <= 0x1p+1000)) { bad = 1; break; }
                        if (far_row && t / 8 < ets_far_blocks) far_row[t / 8] = 1;
                        phib = det_pow_step(b, phi);
                    }
                } else {
                    if (!(b > 0.0)) { bad = 1; break; }
                    phib = b;
                }
                q = l * phib;
            }
            double f = q;
            if (spec->season == ETS_ADD) f = q + s;
            else if (spec->season == ETS_MUL) f = q * s;
            if (ets_fitted_sink) ets_fitted_sink[t] = f;
            /* ONE reciprocal per step; its denominator must lie in [2^-1000, 2^1000] in magnitude, else the trial point is
             * inadmissible (det_math.h det_recip_ok: the domain on which the kernels' short division sequence is the correctly
             * rounded quotient).  A zero or non-finite denominator is outside that domain. */
            double lnew;
            if (spec->error == ETS_MUL) {
                if (!det_mulerr_f_ok(f)) { bad = 1; break; }      /* (det_math.h: the domain of the division AND of the log-likelihood product) */
                const double rf = 1.0 / f;
                const double eps = (y[t] - f) * rf;
                { int ex; mant = frexp(mant * fabs(f), &ex); eacc += ex; }
                sse = fma(eps, eps, sse);
                const double qe = q * eps;
                lnew = fma(alpha, qe, q);
                if (spec->trend == ETS_ADD) b = fma(beta, qe, phib);
                else if (spec->trend == ETS_MUL) b = fma(beta, phib * eps, phib);
                if (spec->season == ETS_MUL) sbuf[j] = fma(gamma, s * eps, s);
            } else {
                const double e = y[t] - f;
                sse = fma(e, e, sse);
                if (spec->season == ETS_MUL && spec->trend == ETS_MUL) {
                    const double d = f * l;                  /* = q s l : 1 / (s l) = R q, 1 / s = R q l, 1 / q = R (s l) */
                    if (!det_recip_ok(d)) { bad = 1; break; }
                    const double R = 1.0 / d;
                    const double esl = e * (R * q);
                    const double es = esl * l;
                    lnew = fma(alpha, es, q);
                    b = fma(beta, esl, phib);
                    sbuf[j] = fma(gamma, e * (R * (s * l)), s);
                } else if (need_f) {                         /* multiplicative season, trend none / additive: 1 / s = R q, 1 / q = R s */
                    if (!det_recip_ok(f)) { bad = 1; break; }
                    const double R = 1.0 / f;
                    const double es = e * (R * q);
                    lnew = fma(alpha, es, q);
                    if (spec->trend == ETS_ADD) b = fma(beta, es, phib);
                    sbuf[j] = fma(gamma, e * (R * s), s);
                } else {                                     /* multiplicative trend, season none / additive */
                    if (!det_recip_ok(l)) { bad = 1; break; }
                    const double rl = 1.0 / l;
                    lnew = fma(alpha, e, q);
                    b = fma(beta, e * rl, phib);
                    if (spec->season == ETS_ADD) sbuf[j] = fma(gamma, e, s);
                }
            }
            l = lnew;
        }
    }

    if (sse_out) *sse_out = sse;
    if (l_out) *l_out = l;
    if (b_out) *b_out = b;
    if (s_out && spec->season != ETS_NONE) for (int j = 0; j < m; j++) s_out[j] = sbuf[j];

    if (bad || !(fabs(sse) <= DBL_MAX)) return INFINITY;
    double lik = (double)n * det_log(sse);
    if (spec->error == ETS_MUL) lik = lik + 2.0 * (det_log(mant) + (double)eacc * ETS_LN2);
    if (lik != lik) return INFINITY;
    if (lik < -1.0e10) lik = -1.0e10;
    return lik;
}

typedef struct { const EtsSpec *spec; const double *y; int n; double l0, b0; const double *s0; } EtsCtx;

static double ets_objective(const double *par, void *vctx)
{
    EtsCtx *c = (EtsCtx *)vctx;
    return ets_lik(c->spec, c->y, c->n, par, c->l0, c->b0, c->s0, NULL, NULL, NULL, NULL);
}

int ets_fit(const EtsSpec *spec, const double *y, int n, EtsFit *fit, double *s_final)
{
    memset(fit, 0, sizeof *fit);
    const int dim = ets_dim(spec);
    const int k = ets_n_param(spec);
    fit->dim = dim;
    fit->n_param = k;
    if (spec->season != ETS_NONE && (spec->m < 2 || spec->m > ETS_MAX_PERIOD)) return fit->status = ETS_ERR_PERIOD;
    if (spec->season != ETS_NONE && n < 2 * spec->m) return fit->status = ETS_ERR_SHORT;
    if (n < k + 2) return fit->status = ETS_ERR_SHORT;
    if (spec->error == ETS_MUL || spec->trend == ETS_MUL || spec->season == ETS_MUL) {
        for (int i = 0; i < n; i++) if (!(y[i] > 0.0)) return fit->status = ETS_ERR_NONPOSITIVE;
    }
    double s0[ETS_MAX_PERIOD];
    int st = ets_init_states(spec, y, n, &fit->l0, &fit->b0, s0);
    if (st != ETS_OK) return fit->status = st;

    double x0[ETS_MAX_DIM], lo[ETS_MAX_DIM], hi[ETS_MAX_DIM];
    int d = 0;
    x0[d] = ETS_ALPHA0; lo[d] = ETS_PAR_LO; hi[d] = ETS_PAR_HI; d++;
    if (spec->trend != ETS_NONE) { x0[d] = ETS_BETA0; lo[d] = ETS_PAR_LO; hi[d] = ETS_PAR_HI; d++; }
    if (spec->season != ETS_NONE) { x0[d] = ETS_GAMMA0; lo[d] = ETS_PAR_LO; hi[d] = ETS_PAR_HI; d++; }
    if (spec->damped) { x0[d] = ETS_PHI0; lo[d] = ETS_PHI_LO; hi[d] = ETS_PHI_HI; d++; }

    EtsCtx ctx = { spec, y, n, fit->l0, fit->b0, s0 };
    NmResult r;
    nm_minimize(ets_objective, &ctx, dim, x0, lo, hi, &r);
    for (int i = 0; i < dim; i++) fit->par[i] = r.x[i];
    ets_unpack(spec, fit->par, &fit->alpha, &fit->beta_star, &fit->gamma_star, &fit->phi);
    fit->iters = r.iters;
    fit->evals = r.evals;
    fit->lik = ets_lik(spec, y, n, fit->par, fit->l0, fit->b0, s0, &fit->sse, &fit->l, &fit->b, s_final);
    if (!(fabs(fit->lik) <= DBL_MAX)) return fit->status = ETS_ERR_NONFINITE;
    double dk = (double)k, dn = (double)n;
    fit->aic = fit->lik + 2.0 * dk;
    fit->aicc = fit->aic + 2.0 * dk * (dk + 1.0) / (dn - dk - 1.0);
    fit->bic = fit->lik + dk * det_log(dn);
    return fit->status = ETS_OK;
}

/* Fixed smoothing parameters (BASELINE config 2): the same preconditions and initial states as ets_fit, no optimiser --
 * one likelihood pass with par = (alpha, beta / alpha, gamma / (1 - alpha), phi) restricted to the spec's coordinates. */
int ets_fit_fixed(const EtsSpec *spec, const double *y, int n, double alpha, double beta, double gamma, double phi,
                  EtsFit *fit, double *s_final)
{
    memset(fit, 0, sizeof *fit);
    const int dim = ets_dim(spec);
    const int k = ets_n_param(spec);
    fit->dim = dim;
    fit->n_param = k;
    if (spec->season != ETS_NONE && (spec->m < 2 || spec->m > ETS_MAX_PERIOD)) return fit->status = ETS_ERR_PERIOD;
    if (spec->season != ETS_NONE && n < 2 * spec->m) return fit->status = ETS_ERR_SHORT;
    if (n < k + 2) return fit->status = ETS_ERR_SHORT;
    if (spec->error == ETS_MUL || spec->trend == ETS_MUL || spec->season == ETS_MUL) {
        for (int i = 0; i < n; i++) if (!(y[i] > 0.0)) return fit->status = ETS_ERR_NONPOSITIVE;
    }
    double s0[ETS_MAX_PERIOD];
    int st = ets_init_states(spec, y, n, &fit->l0, &fit->b0, s0);
    if (st != ETS_OK) return fit->status = st;
    int d = 0;
    fit->par[d++] = alpha;
    if (spec->trend != ETS_NONE) fit->par[d++] = beta / alpha;
    if (spec->season != ETS_NONE) fit->par[d++] = gamma / (1.0 - alpha);
    if (spec->damped) fit->par[d++] = phi;
    ets_unpack(spec, fit->par, &fit->alpha, &fit->beta_star, &fit->gamma_star, &fit->phi);
    fit->lik = ets_lik(spec, y, n, fit->par, fit->l0, fit->b0, s0, &fit->sse, &fit->l, &fit->b, s_final);
    if (!(fabs(fit->lik) <= DBL_MAX)) return fit->status = ETS_ERR_NONFINITE;
    double dk = (double)k, dn = (double)n;
    fit->aic = fit->lik + 2.0 * dk;
    fit->aicc = fit->aic + 2.0 * dk * (dk + 1.0) / (dn - dk - 1.0);
    fit->bic = fit->lik + dk * det_log(dn);
    return fit->status = ETS_OK;
}

void ets_forecast(const EtsSpec *spec, int n, const EtsFit *fit, const double *s_final,
                  int h, double *out)
{
    const int m = spec->m;
    const double phi = fit->phi;
    double pp = phi, phistar = phi;
    for (int i = 0; i < h; i++) {
        double f;
        if (spec->trend == ETS_NONE) f = fit->l;
        else if (spec->trend == ETS_ADD) f = fit->l + phistar * fit->b;
        else f = (fit->b > 0.0) ? fit->l * det_pow_pos(fit->b, phistar) : NAN;
        if (spec->season != ETS_NONE) {
            double s = s_final[(n + i) % m];
            f = (spec->season == ETS_ADD) ? f + s : f * s;
        }
        out[i] = f;
        pp = pp * phi;
        phistar = phistar + pp;
    }
}
