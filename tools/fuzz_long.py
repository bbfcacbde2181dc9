"""Differential test on LONG ragged series (the fuzz of tools/fuzz_parity.py stops at 150 observations): M5-like batches
with lengths 300..2000, every series of the batch through the OpenMP oracle and through the C-ABI batch entry.
python tools/fuzz_long.py [n_series] [seed]   -- prints mismatches, exits 1 if any."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import api, lib, synth
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
bad = total = 0
for model, kw, positive, count in (("AutoETS", dict(seasonal_period=7), True, n), ("AutoETS", dict(seasonal_period=7), False, n),
                                   ("AutoETS", dict(seasonal_period=12, model_pool="reduced"), True, n // 2),
                                   ("AutoARIMA", dict(seasonal_period=7), False, n // 2), ("AutoARIMA", dict(), True, n // 4),
                                   ("ETS", dict(ets_model="MAdM", seasonal_period=7), True, n // 2), ("HoltWinters", dict(seasonal_period=7), False, n // 2),
                                   # detected periods (merged batches: per-lane periods, LDS and HBM rings), long explicit periods
                                   ("AutoETS", dict(), True, n // 2), ("AutoETS", dict(), False, n // 2), ("HoltWinters", dict(), True, n // 4),
                                   ("AutoETS", dict(seasonal_period=30), True, n // 4), ("AutoETS", dict(seasonal_period=168), True, n // 4),
                                   ("ETS", dict(ets_model="MAM", seasonal_period=365), True, n // 4)):
    Y = synth.gen_series(synth.SEED_M5 + seed, int(rng.integers(0, 100000)), count, 2000, 7, positive)
    lens = rng.integers(300, 2001, size=count)
    series = [Y[i, 2000 - lens[i]:].copy() for i in range(count)]          # ragged: the most recent lens[i] observations
    t0 = time.time()
    got, berr = api.forecast_batch(series, lib.make_options(model, 28, **kw))
    t1 = time.time()
    off = np.concatenate([[0], np.cumsum(lens)])
    ref = O.forecast_batch(np.concatenate(series), off, O.make_options(model, 28, **kw))
    t2 = time.time()
    assert berr["ok"], berr
    m = 0
    for i in range(count):
        ok_ref = ref["status"][i] == 0
        if got[i]["ok"] != ok_ref:
            m += 1; print("STATUS", model, kw, i, got[i].get("code"), ref["status"][i]); continue
        if not ok_ref:
            continue
        if got[i]["model_name"] != ref["names"][i] or not np.array_equal(got[i]["point"], ref["yhat"][i]) \
                or not np.array_equal(got[i]["lower"], ref["lower"][i]) or not np.array_equal(got[i]["upper"], ref["upper"][i]):
            m += 1
            print("DIFF", model, kw, i, got[i]["model_name"], ref["names"][i], float(np.max(np.abs(got[i]["point"] - ref["yhat"][i]))))
    bad += m; total += count
    print(f"{model} {kw} positive={positive}: {count} series, {m} mismatches (gpu path {t1 - t0:.1f} s, oracle {t2 - t1:.1f} s on {ref['threads']} threads)", flush=True)
print(f"{total} long series compared bit for bit, {bad} mismatches")
sys.exit(1 if bad else 0)
