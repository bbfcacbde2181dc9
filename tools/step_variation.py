"""Step-to-step variation of one bench workload: K steps with a device-wide wait and the library's statistics after each (wall and
device ms per step), then K steps enqueued back to back (wall per step).  Usage: python tools/step_variation.py [workload] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from anofox_forecast_amd import lib, synth
from anofox_forecast_amd.device import DeviceBatch
from anofox_forecast_amd.device import pack_time_major

name = sys.argv[1] if len(sys.argv) > 1 else "autoets_m5_positive"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wl = bench.WORKLOADS[name]
n, T, m = wl["n"], wl["T"], wl["m"]
dev = torch.device("cuda:0")
Y = synth.gen_series(wl["seed"], 0, n, T, m, wl["positive"])
opts = lib.make_options(wl["model"], 28, ets_model=wl["ets_model"], seasonal_period=m)
b = DeviceBatch(n, T, opts, dev)
y = torch.from_numpy(pack_time_major(Y, b.ld)).to(dev)
ln = torch.full((b.ld,), T, dtype=torch.int32, device=dev); ln[n:] = 0
b.set_block(y, ln)
b.run(); torch.cuda.synchronize()
wall, devms, fit = [], [], []
for _ in range(K):
    t0 = time.perf_counter(); b.run(); torch.cuda.synchronize(); wall.append((time.perf_counter() - t0) * 1e3)
    st = b.stats(); devms.append(st["total_device_ms"]); fit.append(st["fit_kernel_ms"])
print(name, "synced steps: wall ms", [round(x, 1) for x in wall])
print(name, "             device ms", [round(x, 1) for x in devms], "fit ms", [round(x, 1) for x in fit])
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(K):
        b.run()
    torch.cuda.synchronize()
    w = (time.perf_counter() - t0) * 1e3 / K
    st = b.stats()
    print(name, f"back to back x{K}: wall per step {w:.1f} ms, last step device {st['total_device_ms']:.1f} fit {st['fit_kernel_ms']:.1f}")
