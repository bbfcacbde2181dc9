/* Route A the way the reference's scalar binding drives it (src/scalar_functions/ts_forecast_scalar.cpp:298-523): N worker threads,
 * each calling anofox_ts_forecast once per group of its chunk, back to back.  First every series is forecast serially (the
 * expected results), then the same series from `threads` pthreads at once; every threaded result must equal the serial one
 * bit for bit (the coalescing window of the library puts concurrent calls into one multi-series batch), a series that is too
 * short must fail alone.  Prints "OK <serial us per call> <threaded us per call (wall / calls)> <mismatches>".
 * Build: gcc -std=c11 -O2 -pthread -I include tests/c_abi/concurrent.c -L anofox-forecast_amd -lanofox_fcst_hip -o concurrent
 * Usage: concurrent <model> <seasonal_period> <threads> <calls per thread> <length> */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "anofox_fcst_hip.h"

#define H 9
static int n_threads, per_thread, len;
static double *series;                 /* [n_threads * per_thread][len] */
static double *want, *got;             /* [calls][3 * H] */
static int *want_ok, *got_ok;
static struct ForecastOptions opts;

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static int one(int c, double *out)
{
    struct ForecastResult res;
    struct AnofoxError err;
    memset(&res, 0, sizeof res);
    /* every 13th series is too short: its call must fail with INSUFFICIENT_DATA and leave the neighbours alone */
    const size_t n = (c % 13 == 12) ? 2 : (size_t)(len - (c % 5) * 3);
    if (!anofox_ts_forecast(series + (size_t)c * len, NULL, n, &opts, &res, &err)) return -(int)err.code;
    for (int i = 0; i < H; i++) { out[i] = res.point_forecasts[i]; out[H + i] = res.lower_bounds[i]; out[2 * H + i] = res.upper_bounds[i]; }
    anofox_free_forecast_result(&res);
    return 1;
}

static void *worker(void *arg)
{
    const int t = (int)(size_t)arg;
    for (int k = 0; k < per_thread; k++) {
        const int c = t * per_thread + k;
        got_ok[c] = one(c, got + (size_t)c * 3 * H);
    }
    return NULL;
}

int main(int argc, char **argv)
{
    const char *model = argc > 1 ? argv[1] : "AutoETS";
    const int period = argc > 2 ? atoi(argv[2]) : 7;
    n_threads = argc > 3 ? atoi(argv[3]) : 8;
    per_thread = argc > 4 ? atoi(argv[4]) : 8;
    len = argc > 5 ? atoi(argv[5]) : 200;
    const int calls = n_threads * per_thread;
    series = malloc(sizeof(double) * (size_t)calls * len);
    want = calloc((size_t)calls * 3 * H, sizeof(double));
    got = calloc((size_t)calls * 3 * H, sizeof(double));
    want_ok = calloc(calls, sizeof(int));
    got_ok = calloc(calls, sizeof(int));
    unsigned long long s = 88172645463325252ull;
    for (int c = 0; c < calls; c++)
        for (int i = 0; i < len; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;                            /* xorshift: deterministic positive, weekly-ish series */
            series[(size_t)c * len + i] = 20.0 + 0.05 * i + ((i % 7) == (c % 7) ? 6.0 : 0.0) + (double)(s % 1000) / 250.0;
        }
    memset(&opts, 0, sizeof opts);
    strncpy(opts.model, model, sizeof opts.model - 1);
    opts.horizon = H;
    opts.confidence_level = 0.90;
    opts.seasonal_period = period;
    opts.auto_detect_seasonality = false;
    double out0[3 * H];
    (void)one(0, out0);                                                        /* runtime start-up and first allocations: untimed */
    double t0 = now_us();
    for (int c = 0; c < calls; c++) want_ok[c] = one(c, want + (size_t)c * 3 * H);
    const double serial_us = (now_us() - t0) / calls;
    pthread_t th[64];
    if (n_threads > 64) n_threads = 64;
    t0 = now_us();
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, worker, (void *)(size_t)t);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    const double threaded_us = (now_us() - t0) / calls;
    int bad = 0, failed = 0;
    for (int c = 0; c < calls; c++) {
        if (want_ok[c] != got_ok[c]) bad++;
        else if (want_ok[c] == 1 && memcmp(want + (size_t)c * 3 * H, got + (size_t)c * 3 * H, sizeof(double) * 3 * H) != 0) bad++;
        if (want_ok[c] != 1) failed++;
    }
    printf("OK %.1f %.1f %d %d\n", serial_us, threaded_us, bad, failed);
    return bad != 0;
}
