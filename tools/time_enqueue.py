"""How long does the host take to enqueue one step (all launches), against the device time of the step?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anofox_forecast_amd import lib, synth
from anofox_forecast_amd.device import DeviceBatch, pack_time_major
for wl, positive in (("intermittent", False), ("positive", True)):
    n, T = 30490, 1913
    Y = synth.gen_series(synth.SEED_M5, 0, n, T, 7, positive)
    b = DeviceBatch(n, T, lib.make_options("AutoETS", 28, seasonal_period=7), "cuda:0")
    y = torch.from_numpy(pack_time_major(Y, b.ld)).cuda()
    ln = torch.full((b.ld,), T, dtype=torch.int32, device="cuda"); ln[n:] = 0
    b.set_block(y, ln)
    b.run(); torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); b.run(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{wl}: enqueue {1e3 * (t1 - t0):.1f} ms, until done {1e3 * (t2 - t0):.1f} ms, launches {b.stats()['fit_kernel_launches']}")
    b.close()
