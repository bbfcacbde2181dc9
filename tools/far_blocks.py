"""How often a block of 8 time steps of a damped multiplicative-trend pass contains a step whose growth rate is far from one
(|b - 1| > 1/16: the table-driven power), per lane and per WAVE of 64 series in lock step -- what a block-level check with
re-execution would have to re-run.  CPU only (oracle ets_fit + the ets_far_sink hook).  python tools/far_blocks.py [n_series]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import synth
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = 1913
L = O.lib()


class EtsSpec(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("error", "trend", "damped", "season", "m")]


class EtsFit(C.Structure):
    _fields_ = [("status", C.c_int), ("dim", C.c_int), ("par", C.c_double * 4), ("alpha", C.c_double), ("beta_star", C.c_double),
                ("gamma_star", C.c_double), ("phi", C.c_double), ("l0", C.c_double), ("b0", C.c_double), ("lik", C.c_double),
                ("sse", C.c_double), ("aic", C.c_double), ("aicc", C.c_double), ("bic", C.c_double), ("n_param", C.c_int),
                ("iters", C.c_int), ("evals", C.c_int), ("l", C.c_double), ("b", C.c_double)]


L.ets_fit.restype = C.c_int
L.ets_fit.argtypes = [C.POINTER(EtsSpec), C.c_void_p, C.c_int, C.POINTER(EtsFit), C.c_void_p]
NB = (T + 7) // 8
C.c_int.in_dll(L, "ets_far_blocks").value = NB
cap = NB * 900
buf = (C.c_ubyte * cap)()
C.c_void_p.in_dll(L, "ets_far_sink").value = C.addressof(buf)
C.c_long.in_dll(L, "ets_far_cap").value = cap
Y = synth.gen_series(synth.SEED_M5, 0, n, T, 7, True)
sfin = np.zeros(2048)
for (e, s) in ((1, 0), (1, 1), (1, 2), (2, 0), (2, 2)):
    maps = []
    for i in range(n):
        C.c_long.in_dll(L, "ets_far_pos").value = 0
        y = np.ascontiguousarray(Y[i])
        fit = EtsFit()
        L.ets_fit(C.byref(EtsSpec(e, 2, 1, s, 7 if s else 1)), y.ctypes.data, T, C.byref(fit), sfin.ctypes.data)
        used = C.c_long.in_dll(L, "ets_far_pos").value
        maps.append(np.frombuffer(buf, dtype=np.uint8, count=used).reshape(-1, NB).copy())
    lane_blocks = sum(m.size for m in maps)
    lane_far = sum(int(m.sum()) for m in maps)
    lane_passes = sum(m.shape[0] for m in maps)
    lane_far_passes = sum(int((m.sum(axis=1) > 0).sum()) for m in maps)
    pos = np.zeros(NB)
    for m in maps:
        pos += m.sum(axis=0)
    # waves of 64 consecutive series, pass k of every lane side by side (a lane that has finished contributes nothing)
    wave_blocks = wave_far = 0
    for w in range(0, n, 64):
        grp = maps[w:w + 64]
        kmax = max(m.shape[0] for m in grp)
        acc = np.zeros((kmax, NB), dtype=np.uint8)
        for m in grp:
            acc[:m.shape[0]] |= m
        wave_blocks += acc.size
        wave_far += int(acc.sum())
    first = pos[:8].sum() / max(pos.sum(), 1)
    print("ETS(%s,Md,%s): passes/lane %.0f; lane-passes with a far step %.1f %%; lane-blocks far %.3f %%; WAVE-blocks (64 lanes) far %.2f %%; "
          "share of the far blocks among the first 8 blocks (64 steps) %.0f %%, first 32 blocks %.0f %%" % (
              "AM"[e - 1], "NAM"[s], lane_passes / n, 100.0 * lane_far_passes / lane_passes, 100.0 * lane_far / lane_blocks,
              100.0 * wave_far / wave_blocks, 100.0 * first, 100.0 * pos[:32].sum() / max(pos.sum(), 1)))
