"""Time the host side of the batch entry (pack + H2D) on the M5 shape: python tools/time_pack.py"""
import ctypes as C, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import lib, synth
L = lib.load()
n, T = 30490, 1913
Y = synth.gen_series(synth.SEED_M5, 0, n, T, 7)
opts = lib.make_options("AutoETS", 28, seasonal_period=7)
hb, err = C.c_void_p(), lib.AnofoxError()
assert L.anofox_hip_batch_create(n, T, C.byref(opts), C.byref(hb), C.byref(err))
vptr = (C.c_void_p * n)(*[Y[i].ctypes.data for i in range(n)])
lens = (C.c_size_t * n)(*([T] * n))
for rep in range(3):
    t0 = time.perf_counter()
    ok = L.anofox_hip_batch_pack_host(hb, vptr, None, lens, C.byref(err))
    t1 = time.perf_counter()
    assert ok, err.message
    print(f"pack_host + H2D: {(t1 - t0) * 1e3:.1f} ms  ({n * T * 8 / (t1 - t0) / 1e9:.2f} GB/s of series data)")
t0 = time.perf_counter(); assert L.anofox_hip_batch_run(hb, None, C.byref(err)); st = lib.AnofoxHipStats(); L.anofox_hip_batch_stats(hb, C.byref(st)); t1 = time.perf_counter()
print(f"run: {(t1 - t0) * 1e3:.1f} ms")
res = (lib.ForecastResult * n)(); errs = (lib.AnofoxError * n)()
t0 = time.perf_counter(); L.anofox_hip_batch_fetch(hb, res, errs); t1 = time.perf_counter()
print(f"fetch (D2H + {n} x 3 mallocs): {(t1 - t0) * 1e3:.1f} ms")
L.anofox_hip_batch_destroy(hb)
for rep in range(3):
    t0 = time.perf_counter(); hb = C.c_void_p(); assert L.anofox_hip_batch_create(n, T, C.byref(opts), C.byref(hb), C.byref(err)); t1 = time.perf_counter()
    L.anofox_hip_batch_destroy(hb); t2 = time.perf_counter()
    print(f"batch_create: {(t1 - t0) * 1e3:.1f} ms, destroy {(t2 - t1) * 1e3:.1f} ms")
