"""Differential test of AutoARIMA with the exact-likelihood refit (ANOFOX_ARIMA_CSS_ML: the two-launch refit kernel) against the
oracle in the same mode: random batches of seasonal / non-seasonal, short / long, ragged series.
python tools/fuzz_arima_ml.py [seconds] [seed] -- prints every mismatch, exits 1 if any."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import api, lib, synth
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
L = lib.load()
flag = C.c_int.in_dll(O.lib(), "oracle_arima_ml_refit")
assert L.anofox_hip_set_default_arima_method(lib.ARIMA_CSS_ML)
flag.value = 1
bad = total = 0
t_end = time.time() + budget
while time.time() < t_end:
    m = int(rng.choice([1, 1, 4, 7, 7, 12]))
    n = int(rng.integers(20, 400))
    T = int(rng.choice([40, 90, 200, 600, 1500]))
    positive = bool(rng.integers(0, 2))
    Y = synth.gen_series(synth.SEED_M5 + seed, int(rng.integers(0, 10**6)), n, T, max(m, 2), positive)
    series = []
    for s in range(n):
        L_s = int(rng.integers(max(8, T // 3), T + 1))
        y = Y[s, T - L_s:].copy()
        if rng.random() < 0.3:                      # ARMA structure on top: higher orders, long refits
            e = rng.normal(0, 1, L_s + 30)
            x = np.zeros(L_s + 30)
            a1, a2, b1 = rng.uniform(-0.6, 0.8), rng.uniform(-0.4, 0.3), rng.uniform(-0.6, 0.6)
            for t in range(2, L_s + 30):
                x[t] = a1 * x[t - 1] + a2 * x[t - 2] + e[t] + b1 * e[t - 1]
            y = y + 2.0 * x[30:]
        series.append(y)
    kw = dict(seasonal_period=m)
    got, berr = api.forecast_batch(series, lib.make_options("AutoARIMA", 6, **kw))
    assert berr["ok"], berr
    off = np.concatenate([[0], np.cumsum([len(y) for y in series])])
    ref = O.forecast_batch(np.concatenate(series), off, O.make_options("AutoARIMA", 6, **kw))
    for i in range(n):
        ok_ref = ref["status"][i] == 0
        if got[i]["ok"] != ok_ref:
            bad += 1; print("STATUS", m, T, i, got[i].get("code"), ref["status"][i]); continue
        if ok_ref and (got[i]["model_name"] != ref["names"][i] or not np.array_equal(got[i]["point"], ref["yhat"][i])):
            bad += 1; print("DIFF", m, T, i, got[i]["model_name"], ref["names"][i], float(np.max(np.abs(got[i]["point"] - ref["yhat"][i]))))
    total += n
flag.value = 0
L.anofox_hip_set_default_arima_method(lib.ARIMA_CSS)
print(f"{total} series compared (AutoARIMA, exact-likelihood refit), {bad} mismatches")
sys.exit(1 if bad else 0)
