"""Wall time of ONE fixed ETS spec on the M5-shape positive batch (30,490 x 1,913, h=28), device resident: how long the chain
of resumable rounds of that spec takes when nothing else runs.  python tools/time_single_spec.py AMdM [AAN ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anofox_forecast_amd import lib, synth
from anofox_forecast_amd.device import DeviceBatch, pack_time_major

n, T = 30490, 1913
Y = synth.gen_series(synth.SEED_M5, 0, n, T, 7, True)
for spec in sys.argv[1:]:
    opts = lib.make_options("ETS", 28, ets_model=spec, seasonal_period=7)
    b = DeviceBatch(n, T, opts)
    y_dev = torch.from_numpy(pack_time_major(Y, b.ld)).to("cuda:0")
    len_dev = torch.full((b.ld,), T, dtype=torch.int32, device="cuda:0")
    b.set_block(y_dev, len_dev)
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        b.run()
        torch.cuda.synchronize(); dt = time.time() - t0
    st = b.stats()
    print(f"{spec}: {dt*1e3:.1f} ms/step", {k: v for k, v in st.items() if "pass" in k or "eval" in k})
    b.close()
