"""AutoETS with auto-detected periods on the full M5 shape (30,490 ragged series): one call of the batch entry, a sample checked
against the oracle.  python tools/time_autodetect_full.py [n_series] [model] [positive|intermittent]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import api, lib, synth
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30490
model = sys.argv[2] if len(sys.argv) > 2 else "AutoETS"
positive = (sys.argv[3] != "intermittent") if len(sys.argv) > 3 else True
rng = np.random.default_rng(5)
Y = synth.gen_series(synth.SEED_M5 + 5, 4321, n, 1913, 7, positive)
lens = rng.integers(400, 1914, size=n)
series = [Y[i, 1913 - lens[i]:].copy() for i in range(n)]
opts = lib.make_options(model, 28)
times = []
for _ in range(int(os.environ.get('REPS', '2'))):
    t0 = time.time()
    got, berr = api.forecast_batch(series, opts)
    times.append(time.time() - t0)
    assert berr["ok"], berr
dt = times[-1]
bad = 0
for i in range(0, n, max(1, n // 24)):
    ref = O.forecast(series[i], O.make_options(model, 28))
    if ref["ok"] != got[i]["ok"] or (ref["ok"] and (ref["model_name"] != got[i]["model_name"] or not np.array_equal(ref["point"], got[i]["point"]))):
        bad += 1
print("calls:", " ".join(f"{t:.2f}" for t in times))
print(f"{model}: {dt:.2f} s for {n} series with auto-detected periods = {n / dt:.0f} series/s (second call; first {times[0]:.2f} s; Python marshalling included), {bad} mismatches in the sample")
