"""Where do the ~120 ms between a device-resident run (567 ms) and the run inside anofox_ts_forecast_batch (690 ms) go?
One positive M5 batch, four ways: set_block + caller's stream, set_block + the batch's own stream, pack_host + caller's stream,
pack_host + own stream; each run three times on the same batch.  python tools/time_run_variants.py"""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anofox_forecast_amd import lib, synth
from anofox_forecast_amd.device import DeviceBatch, pack_time_major

n, T, h = 30490, 1913, 28
Y = synth.gen_series(synth.SEED_M5, 0, n, T, 7, True)
opts = lib.make_options("AutoETS", h, seasonal_period=7)
L = lib.load()
vptr = (C.c_void_p * n)(*[Y.ctypes.data + s * T * 8 for s in range(n)])
lens = (C.c_size_t * n)(*([T] * n))
for packed in (False, True):
    for own in (False, True):
        b = DeviceBatch(n, T, opts, "cuda:0")
        if packed:
            err = lib.AnofoxError()
            assert L.anofox_hip_batch_pack_host(b.handle, vptr, None, lens, C.byref(err)), err.message
        else:
            y = torch.from_numpy(pack_time_major(Y, b.ld)).cuda()
            ln = torch.full((b.ld,), T, dtype=torch.int32, device="cuda"); ln[n:] = 0
            b.set_block(y, ln)
        times = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if own:
                err = lib.AnofoxError()
                assert L.anofox_hip_batch_run(b.handle, None, C.byref(err)), err.message
                b.stats()                                   # waits for the batch's stream
            else:
                b.run()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        print(f"packed={packed} own_stream={own}: " + " ".join(f"{t:.0f}" for t in times) + " ms", flush=True)
        b.close()
