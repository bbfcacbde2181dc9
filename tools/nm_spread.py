"""Would sorting the running problems of a round by the SIZE of their simplex make the waves finish together?  For the problems still
running after the rounds' cumulative budgets, the lane efficiency of the next round (active lane-iterations / (64 x the wave's longest
lane)) with the problems in arbitrary order (today: the compaction's order) and sorted by log(max(x-spread / xatol, f-spread / fatol)).
CPU only (oracle ets_fit + the nm_spread_sink hook).  python tools/nm_spread.py [n_series]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import synth
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 640
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
cap = 3 * 1000
buf = (C.c_double * cap)()
C.c_void_p.in_dll(L, "nm_spread_sink").value = C.addressof(buf)
C.c_long.in_dll(L, "nm_spread_cap").value = cap
Y = synth.gen_series(synth.SEED_M5, 0, n, T, 7, True)
sfin = np.zeros(2048)
rng = np.random.default_rng(1)
bounds = np.cumsum([24] * 6 + [48, 48, 96, 96, 192])
for name, (e, t, d, s) in (("ETS(A,Md,A)", (1, 2, 1, 1)), ("ETS(M,Ad,M)", (2, 1, 1, 2)), ("ETS(A,A,M)", (1, 1, 0, 2)), ("ETS(A,Ad,A)", (1, 1, 1, 1))):
    runs = []
    for i in range(n):
        C.c_long.in_dll(L, "nm_spread_pos").value = 0
        y = np.ascontiguousarray(Y[i])
        fit = EtsFit()
        L.ets_fit(C.byref(EtsSpec(e, t, d, s, 7 if s else 1)), y.ctypes.data, T, C.byref(fit), sfin.ctypes.data)
        used = C.c_long.in_dll(L, "nm_spread_pos").value
        a = np.frombuffer(buf, dtype=np.float64, count=used).reshape(-1, 3).copy()
        if len(a) and a[-1, 0] == -1.0:
            runs.append((a[:-1], int(a[-1, 1])))
    print(f"{name}: {len(runs)} fits, iterations mean {np.mean([r[1] for r in runs]):.0f}")
    prev = 0
    for r_i, c in enumerate(bounds[:-1]):
        budget = int(bounds[r_i + 1] - c)
        alive = [(a, tot) for a, tot in runs if tot > c]
        if len(alive) < 128:
            break
        rem = np.array([min(tot - c, budget) for a, tot in alive], dtype=float)
        key = np.array([np.log(max(a[a[:, 0] == c][0, 1], a[a[:, 0] == c][0, 2], 1.0)) if (a[:, 0] == c).any() else 0.0 for a, tot in alive])

        def eff(order):
            r = rem[order]
            nw = len(r) // 64
            r = r[: nw * 64].reshape(nw, 64)
            return r.sum() / (64.0 * r.max(axis=1).sum()), r.max(axis=1).sum()
        e_rand, w_rand = eff(rng.permutation(len(rem)))
        e_sort, w_sort = eff(np.argsort(key))
        e_best, w_best = eff(np.argsort(rem))
        cc = np.corrcoef(key, np.array([tot - c for a, tot in alive]))[0, 1]
        print(f"   after {c:4d} iterations: {len(alive):5d} running, next round {budget:3d}: lane efficiency arbitrary order {e_rand:.3f}, sorted by simplex size {e_sort:.3f}, "
              f"by the true remaining count {e_best:.3f}; wave-iterations {w_rand:.0f} -> {w_sort:.0f} ({100 * (w_sort / w_rand - 1):+.1f} %); corr(key, remaining) {cc:.2f}")
