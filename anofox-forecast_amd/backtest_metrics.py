"""Fold metric of the backtest caller (`_ts_backtest_native`'s ComputeMetric, ts_backtest_native.cpp:280-373).

NOT part of the fit + forecast path: the reference computes these in its metrics subsystem (out of
scope).  The backtest operator mirror (api.ts_backtest_native) needs one number per fold to fill `fold_metric_score`, so the
eight formulas live here, apart from the operator surface, as a stand-in for that subsystem.
"""
import numpy as np


def backtest_metric(metric, actual, forecast, lower, upper):
    """ComputeMetric (ts_backtest_native.cpp:280-373): sums run in row order like the reference's loops."""
    a, f = np.asarray(actual, dtype=np.float64), np.asarray(forecast, dtype=np.float64)
    n = len(a)
    if n == 0 or len(f) == 0:
        return float("nan")
    seq = lambda x: float(np.cumsum(x)[-1]) if len(x) else 0.0      # sequential accumulation, not pairwise
    with np.errstate(all="ignore"):
        if metric == "mae":
            return seq(np.abs(a - f)) / n
        if metric == "mse":
            return seq((a - f) * (a - f)) / n
        if metric == "mape":
            k = a != 0
            return seq(np.abs((a[k] - f[k]) / a[k])) / int(k.sum()) * 100.0 if k.any() else float("nan")
        if metric == "smape":
            d = np.abs(a) + np.abs(f)
            k = d > 0
            return seq(np.abs(a[k] - f[k]) / d[k]) / int(k.sum()) * 200.0 if k.any() else float("nan")
        if metric == "bias":
            return seq(f - a) / n
        if metric == "r2":
            mean = seq(a) / n
            res, tot = seq((a - f) * (a - f)), seq((a - mean) * (a - mean))
            return 1.0 - res / tot if tot > 0 else float("nan")
        if metric == "coverage":
            lo, hi = np.asarray(lower), np.asarray(upper)
            if len(lo) != n or len(hi) != n:
                return float("nan")
            return int(((a >= lo) & (a <= hi)).sum()) / n
        return float(np.sqrt(seq((a - f) * (a - f)) / n))           # "rmse" and every unknown name
