"""Batch throughput of the SES / Holt / Holt-Winters / SeasonalES family and the closed-form baselines on the M5 shape
(device-resident block): python tools/time_classic_batch.py [n_series]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anofox_forecast_amd import lib, synth
from anofox_forecast_amd.device import DeviceBatch, pack_time_major

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30490
T, h = 1913, 28
Y = synth.gen_series(synth.SEED_M5, 0, n, T, 7, False)
for model in ("Naive", "SeasonalNaive", "SES", "SESOptimized", "Holt", "HoltWinters", "SeasonalES", "SeasonalESOptimized", "ETS:AAA"):
    kw = {"ets_model": "AAA"} if model.startswith("ETS") else {}
    seasonal = model not in ("Naive", "SES", "SESOptimized", "Holt")
    opts = lib.make_options(model.split(":")[0], h, **({"seasonal_period": 7} if seasonal else {"auto_detect": False}), **kw)
    b = DeviceBatch(n, T, opts, "cuda:0")
    y = torch.from_numpy(pack_time_major(Y, b.ld)).cuda()
    ln = torch.full((b.ld,), T, dtype=torch.int32, device="cuda"); ln[n:] = 0
    b.set_block(y, ln)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); b.run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    st = b.stats()
    print(f"{model:20s} {min(ts):9.2f} ms  {n / min(ts) * 1e3:12.0f} series/s   passes/series {st['total_passes'] / n:8.1f}", flush=True)
    b.close()
