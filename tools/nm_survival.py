"""How many of the (series, spec) problems of the 25-spec AutoETS batch still run after k Nelder-Mead iterations, by spec class -- the
thinning that decides how full the chip is round by round (DESIGN.md sections 5 and 8).  CPU only (oracle ets_fit).
python tools/nm_survival.py [n_series]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import synth
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
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
Y = synth.gen_series(synth.SEED_M5, 0, n, 1913, 7, True)
specs = []
for e in (1, 2):
    for t, d in ((0, 0), (1, 0), (1, 1), (2, 0), (2, 1)):
        for s in (0, 1, 2):
            if e == 2 and (t == 1 and s != 2 and s != 0 or False): pass
            specs.append((e, t, d, s))
# the 25 valid specs: multiplicative error with an additive season is the invalid family (forecast.rs: ETSSpec::is_valid)
specs = [sp for sp in specs if not (sp[0] == 2 and sp[3] == 1)]
assert len(specs) == 25, len(specs)
iters = {"additive": [], "general": [], "damped-M": []}
sfin = np.zeros(2048)
for i in range(n):
    y = np.ascontiguousarray(Y[i])
    for (e, t, d, s) in specs:
        sp = EtsSpec(e, t, d, s, 7 if s else 1)
        fit = EtsFit()
        if L.ets_fit(C.byref(sp), y.ctypes.data, len(y), C.byref(fit), sfin.ctypes.data) != 0:
            continue
        cls = "damped-M" if (t == 2 and d) else ("additive" if (e == 1 and t != 2 and s != 2) else "general")
        iters[cls].append(fit.iters)
bounds = np.cumsum([24] * 6 + [48, 48, 96, 96, 192, 1024])
print(f"{n} series x 25 specs; problems still running after the rounds' cumulative iteration budgets (share of the class):")
print("%-10s %7s  %s" % ("class", "fits", "  ".join("%5d" % b for b in bounds[:-1])) + "    mean / p90 / max iterations")
tot = []
for cls, v in iters.items():
    v = np.array(v); tot.append(v)
    print("%-10s %7d  %s    %.0f / %.0f / %d" % (cls, len(v), "  ".join("%5.1f" % (100.0 * np.mean(v > b)) for b in bounds[:-1]), v.mean(), np.percentile(v, 90), v.max()))
v = np.concatenate(tot)
print("%-10s %7d  %s    %.0f / %.0f / %d" % ("all", len(v), "  ".join("%5.1f" % (100.0 * np.mean(v > b)) for b in bounds[:-1]), v.mean(), np.percentile(v, 90), v.max()))
