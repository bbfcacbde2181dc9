/* arima.c -- TEST INFRASTRUCTURE (oracle). AutoARIMA restatement: see arima.h. */
#include "arima.h"
#include <stdio.h>

int oracle_auto_arima(const double *y, int n, int period, int h, double *out, ArimaOrder *ord)
{
    (void)y; (void)n; (void)period; (void)h; (void)out; (void)ord;
    return 0; /* not yet restated */
}

void oracle_arima_name(const ArimaOrder *o, char out[64])
{
    if (o->s > 1 && (o->P || o->D || o->Q))
        snprintf(out, 64, "AutoARIMA(%d,%d,%d)(%d,%d,%d)[%d]", o->p, o->d, o->q, o->P, o->D, o->Q, o->s);
    else
        snprintf(out, 64, "AutoARIMA(%d,%d,%d)", o->p, o->d, o->q);
}
