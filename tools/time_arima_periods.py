"""AutoARIMA by seasonal period: the batch entry on n series x T for explicit periods (7: rings in registers, 12 / 24: in LDS, 28 and up: in the
HBM scratch ring of the wave).  python tools/time_arima_periods.py [n] [T] [periods...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import api, lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1008
periods = [int(x) for x in sys.argv[3:]] or [7, 24, 28, 52, 168]
for m in periods:
    Y = synth.gen_series(synth.SEED_M5 + 9, 777, n, T, m if m > 1 else 7, False)
    series = list(Y)
    opts = lib.make_options("AutoARIMA", 28, seasonal_period=m, auto_detect=False)
    ts = []
    for _ in range(3):
        t0 = time.time(); got, berr = api.forecast_batch(series, opts); ts.append(time.time() - t0)
        assert berr["ok"], berr
    names = {}
    for g in got:
        if g["ok"]:
            k = "seasonal" if "[" in g["model_name"] else "plain"
            names[k] = names.get(k, 0) + 1
    print(f"m={m}: {n} x {T}: " + " ".join(f"{t*1e3:.0f}" for t in ts) + f" ms; {n / min(ts):.0f} series/s; models {names}", flush=True)
