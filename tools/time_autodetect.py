"""Cost of per-series period auto-detection (params MAP{} -> seasonal_period 0): every distinct detected period is one run of
the pipeline.  Times AutoETS / AutoARIMA / Naive on ragged long series and checks a sample against the oracle.
python tools/time_autodetect.py [n_series]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import api, lib, synth
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(3)
Y = synth.gen_series(synth.SEED_M5 + 3, 1234, n, 2000, 7, True)
lens = rng.integers(300, 2001, size=n)
series = [Y[i, 2000 - lens[i]:].copy() for i in range(n)]
for model in ("Naive", "AutoARIMA", "AutoETS", "HoltWinters", "SeasonalES", "SeasonalESOptimized"):
    opts = lib.make_options(model, 14)                       # seasonal_period 0 -> auto_detect on
    for _ in range(2):
        t0 = time.time()
        got, berr = api.forecast_batch(series, opts)
        dt = time.time() - t0
    bad = 0
    for i in range(0, n, max(1, n // 48)):
        ref = O.forecast(series[i], O.make_options(model, 14))
        if ref["ok"] != got[i]["ok"] or (ref["ok"] and (ref["model_name"] != got[i]["model_name"] or not np.array_equal(ref["point"], got[i]["point"]))):
            bad += 1
    print(f"{model}: {dt:.2f} s for {n} series with auto-detected periods, {bad} mismatches in the sample", flush=True)
