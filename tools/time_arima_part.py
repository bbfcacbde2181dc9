import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from anofox_forecast_amd import api, lib, synth
rng = np.random.default_rng(5)
for n, m in ((1703, 14), (1274, 21), (2619, 7), (282, 6), (27, 19), (23359, 0)):
    Y = synth.gen_series(synth.SEED_M5 + 5, 4321, n, 1913, 7, True)
    lens = rng.integers(400, 1914, size=n)
    series = [Y[i, 1913 - lens[i]:].copy() for i in range(n)]
    opts = lib.make_options("AutoARIMA", 28, seasonal_period=m, auto_detect=False)
    ts = []
    for _ in range(3):
        t0 = time.time(); got, berr = api.forecast_batch(series, opts); ts.append(time.time() - t0)
    print(n, m, " ".join(f"{t*1e3:.0f}" for t in ts), "ms")
