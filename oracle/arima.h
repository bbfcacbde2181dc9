/*
 * arima.h -- TEST INFRASTRUCTURE (oracle).  Not part of the product.
 * AutoARIMA restatement (reference call site crates/anofox-fcst-core/src/forecast.rs:1435-1521;
 * arithmetic in the un-vendored anofox-forecast 0.15.3 crate).
 */
#ifndef ORACLE_ARIMA_H
#define ORACLE_ARIMA_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ArimaOrder { int p, d, q, P, D, Q, s; int with_constant; } ArimaOrder;
/* returns 1 on success (h forecasts in out, selected order in ord), 0 on failure */
int oracle_auto_arima(const double *y, int n, int period, int h, double *out, ArimaOrder *ord);
void oracle_arima_name(const ArimaOrder *ord, char out[64]);
#ifdef __cplusplus
}
#endif
#endif
