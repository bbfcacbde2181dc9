/*
 * batch.c -- TEST INFRASTRUCTURE (oracle).  Not part of the product.
 * OpenMP driver that runs oracle_ts_forecast over many series; used by tests as
 * the batch checker and by bench.py as the same-box CPU baseline ("port").
 */
#include "../include/anofox_fcst_hip.h"
#include <omp.h>
#include <stdlib.h>
#include <string.h>

bool oracle_ts_forecast(const double *, const uint64_t *, size_t, const ForecastOptions *, ForecastResult *, AnofoxError *);
void oracle_free_forecast_result(ForecastResult *);

/* series s occupies values[offsets[s] .. offsets[s+1]); outputs are [n_series x h] row-major.
 * status[s] = ErrorCode; names = n_series x 64 chars. Returns the number of threads used. */
int oracle_forecast_batch(const double *values, const int64_t *offsets, size_t n_series,
                          const ForecastOptions *opt, double *yhat, double *lo, double *hi,
                          int32_t *status, char *names, int n_threads)
{
    const int h = opt->horizon;
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
    for (long s = 0; s < (long)n_series; s++) {
        ForecastResult r;
        AnofoxError e;
        memset(&r, 0, sizeof r);
        size_t len = (size_t)(offsets[s + 1] - offsets[s]);
        bool ok = oracle_ts_forecast(values + offsets[s], NULL, len, opt, &r, &e);
        status[s] = ok ? 0 : (int32_t)e.code;
        if (ok) {
            for (int i = 0; i < h; i++) {
                yhat[(size_t)s * h + i] = r.point_forecasts[i];
                lo[(size_t)s * h + i] = r.lower_bounds[i];
                hi[(size_t)s * h + i] = r.upper_bounds[i];
            }
            if (names) memcpy(names + (size_t)s * 64, r.model_name, 64);
            oracle_free_forecast_result(&r);
        } else if (names) {
            memset(names + (size_t)s * 64, 0, 64);
        }
    }
    return n_threads;
}
