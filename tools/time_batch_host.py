"""Wall time of anofox_ts_forecast_batch over HOST buffers (pack + H2D + fit + fetch) for AutoETS on an M5-shape positive
batch, from a process whose first HIP call is made by this library (no torch.cuda use): what a C/C++ binding sees.
python tools/time_batch_host.py [n_series]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import api, lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30490
Y = synth.gen_series(synth.SEED_M5, 0, n, 1913, 7, True)
series = list(Y)
opts = lib.make_options("AutoETS", 28, seasonal_period=7)
for i in range(2):
    t0 = time.time()
    got, berr = api.forecast_batch(series, opts)
    dt = time.time() - t0
    assert berr["ok"] and all(r["ok"] for r in got)
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')!r} (as Python saw it at start): {dt*1e3:.0f} ms for {n} series, second call")
