/* A plain C caller of the drop-in ABI, written the way the reference's binding fills the structs
 * (src/scalar_functions/ts_forecast_scalar.cpp:439-490): memset, strncpy, call, read, free.
 * Build: gcc -std=c11 -I include tests/c_abi/caller.c -L anofox-forecast_amd -lanofox_fcst_hip -o caller
 * Prints "OK <model_name> <first forecast>" or "ERR <code> <message>". */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "anofox_fcst_hip.h"

int main(int argc, char **argv)
{
    const char *model = argc > 1 ? argv[1] : "AutoETS";
    double y[48];
    for (int i = 0; i < 48; i++) y[i] = 20.0 + 0.5 * i + ((i % 7) == 0 ? 6.0 : 0.0);
    uint64_t validity[1] = {~0ull};
    validity[0] &= ~(1ull << 10);                       /* one NULL, interpolated by the library */
    y[10] = 0.0;
    struct ForecastOptions opts;
    memset(&opts, 0, sizeof opts);
    strncpy(opts.model, model, sizeof opts.model - 1);
    opts.horizon = 5;
    opts.confidence_level = 0.90;
    opts.seasonal_period = argc > 2 ? atoi(argv[2]) : 7;
    opts.auto_detect_seasonality = false;
    struct ForecastResult res;
    memset(&res, 0, sizeof res);
    struct AnofoxError err;
    if (!anofox_ts_forecast(y, validity, 48, &opts, &res, &err)) {
        printf("ERR %d %s\n", (int)err.code, err.message);
        return 0;
    }
    printf("OK %s %.10f %zu %d\n", res.model_name, res.point_forecasts[0], res.n_forecasts,
           res.lower_bounds[0] <= res.point_forecasts[0] && res.point_forecasts[0] <= res.upper_bounds[0]);
    anofox_free_forecast_result(&res);
    if (res.point_forecasts != NULL) { printf("ERR free did not null the pointers\n"); return 1; }
    printf("VERSION %s\n", anofox_fcst_version());
    return 0;
}
