"""How a Nelder-Mead iteration of the ETS fits ends, by parameter dimension: expansion / reflection (new best) / reflection (elsewhere) /
outside / inside contraction / shrink, and the position the new vertex is inserted at -- the outcome tree a speculative driver has to
cover.  CPU only (the oracle's nm_outcome_sink test hook).  python tools/nm_outcomes.py [n_series]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import synth
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
L = O.lib()
counts = np.zeros(56 + 4 * 676, dtype=np.int64)
sink = C.c_void_p.in_dll(L, "nm_outcome_sink")
sink.value = counts.ctypes.data
Y = synth.gen_series(synth.SEED_M5, 0, n, 1913, 7, True)
for s in range(n):
    O.forecast(Y[s], O.make_options("AutoETS", 28, seasonal_period=7))
sink.value = None
kinds = ("expansion", "reflection (new best)", "reflection (elsewhere)", "outside contraction", "inside contraction", "shrink")
for d in range(1, 5):
    c = counts[6 * (d - 1): 6 * d]
    tot = c.sum()
    if not tot: continue
    print(f"dimension {d}: {tot} iterations: " + ", ".join(f"{k} {100.0 * v / tot:.1f} %" for k, v in zip(kinds, c)))
    p = counts[24 + 8 * (d - 1): 24 + 8 * (d - 1) + d + 1]
    print("   inserted at position " + ", ".join(f"{j}: {100.0 * v / max(p.sum(), 1):.1f} %" for j, v in enumerate(p)))
    one = np.zeros(26)
    tab = counts[56 + 676 * (d - 1): 56 + 676 * d].reshape(26, 26).astype(float)
    one = tab.sum(axis=0) / max(tab.sum(), 1)
    o = np.sort(one)[::-1]
    print("   likeliest outcomes (kind x position) cover: " + ", ".join(f"top {k}: {100.0 * o[:k].sum():.0f} %" for k in (1, 2, 4, 8, 15)))
    pairs = np.sort((tab / max(tab.sum(), 1)).reshape(-1))[::-1]
    print("   likeliest PAIRS of consecutive outcomes cover: " + ", ".join(f"top {k}: {100.0 * pairs[:k].sum():.0f} %" for k in (15, 31, 47, 63, 127)))
