"""Latency of the drop-in single-series entry anofox_ts_forecast (route A: one call per group from every DuckDB worker,
ts_forecast_scalar.cpp:298-523): first call (creates the pooled device batch) against the following ones (re-pack + run +
fetch on a pooled batch), and 8 host threads calling concurrently.  Usage on the GPU box: python tools/time_single_call.py"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anofox_forecast_amd import api, lib, synth  # noqa: E402

Y = synth.gen_series(synth.SEED_M5, 0, 64, 1913, 7, False)
for model, kw in (("Naive", {}), ("SES", {}), ("HoltWinters", dict(seasonal_period=7)), ("AutoETS", dict(seasonal_period=7)),
                  ("AutoARIMA", dict(seasonal_period=7)), ("AutoETS (period detected)", {})):
    o = lib.make_options(model.split(" ")[0], 28, **kw)
    t0 = time.perf_counter()
    r = api.forecast_series(Y[0], o)
    first = time.perf_counter() - t0
    assert r["ok"], r
    lat = []
    for s in range(1, 33):
        t0 = time.perf_counter()
        api.forecast_series(Y[s], o)
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e3

    def worker(k, out):
        t0 = time.perf_counter()
        for s in range(8):
            api.forecast_series(Y[(k * 8 + s) % 64], o)
        out[k] = (time.perf_counter() - t0) / 8
    out = [0.0] * 8
    th = [threading.Thread(target=worker, args=(k, out)) for k in range(8)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    wall = time.perf_counter() - t0
    print(f"{model:12s} first call {first * 1e3:8.1f} ms   pooled: median {np.median(lat):7.2f} ms  p95 {np.percentile(lat, 95):7.2f} ms   "
          f"8 threads x 8 calls: {wall * 1e3 / 64:7.2f} ms per call (wall / 64)")
