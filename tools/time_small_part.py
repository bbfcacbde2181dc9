"""One small part of the auto-detect split: k series with an explicit period, alone and from several host threads at once.
python tools/time_small_part.py"""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import api, lib, synth

Y = synth.gen_series(synth.SEED_M5 + 3, 1234, 64, 1500, 7, True)
for k, m in ((8, 23), (8, 7), (1, 23), (32, 23)):
    series = [Y[i].copy() for i in range(k)]
    opts = lib.make_options("AutoETS", 14, seasonal_period=m)
    api.forecast_batch(series, opts)
    ts = []
    for _ in range(3):
        t0 = time.time(); api.forecast_batch(series, opts); ts.append(time.time() - t0)
    print(f"AutoETS k={k} m={m}: alone {min(ts)*1e3:.0f} ms", flush=True)
    for nthr in (4, 16):
        def work():
            for _ in range(2): api.forecast_batch(series, opts)
        thr = [threading.Thread(target=work) for _ in range(nthr)]
        t0 = time.time()
        for t in thr: t.start()
        for t in thr: t.join()
        dt = time.time() - t0
        print(f"   {nthr} threads x 2 calls: {dt*1e3:.0f} ms wall = {dt*1e3/(2*nthr):.0f} ms per call", flush=True)
