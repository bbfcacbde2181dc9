#!/usr/bin/env python3
"""Second half of the known-answer search: fix the ESTIMATOR at what box_optimum.py found --
conditional sum of squares over RAW coefficients kept inside [-0.99, 0.99], constant on the differenced series -- and
search the SELECTION PROCEDURE instead of the optimiser budget.

    python tools/arima_kat_search/selection_search.py [--procs 6] [--quick]     # writes results/selection_*.txt / .csv

Stage 1 (fit table): every ARIMA(p,1,q), p, q <= 5 (also d = 0, 2 for the d-rule variants), x constant convention
{none, icpt_free, mean_free, mean_fixed} x conditioning {p, zero} x Nelder-Mead flavour {trial points clipped to the box
(the ETS optimiser's way), coefficients clipped inside the objective} x initial simplex {5 % relative,
+0.1, +0.25} x start {0, 0.1} x objective scale {sse, mse, half log} is fitted ONCE with a long run; the best vertex at
every iteration cap of CAPS (and at the tolerance stop) is kept, so a "recipe" = one row of that grid + one cap.

Stage 2 (selection): for every recipe, every combination of
    start set      : Hyndman-Khandakar {(2,d,2),(0,d,0),(1,d,0),(0,d,1)} | (0,d,0) only | (2,d,2) only | (1,d,1) only | exhaustive grid
    max p = max q  : 2 | 3 | 5            (order sum <= 5 or unlimited)
    neighbourhood  : HK (p+-1, q+-1, both +-1, constant toggled) | axis only (p+-1, q+-1) ; first improvement | best neighbour
    constant       : always (d <= 1) | never | searched
    criterion      : AIC | AICc | BIC ; n = len(w) | residuals used | len(y) ; k counts sigma^2 or not
    degenerate fits: kept | rejected when sigma^2 < 1e-12 ... 1e-4 | rejected when the criterion is not finite
is run as a table lookup, and the procedures that END at ARIMA(2,1,1) + constant (forecast 18.01451-18.01454) are listed.
"Natural" = no per-series constant anywhere: round caps, the published start set, one rule for all series.
"""
from __future__ import annotations

import argparse
import itertools
import math
import multiprocessing as mp
import os
import pickle
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import search as S  # noqa: E402

TARGET = S.TARGET
HERE = os.path.dirname(os.path.abspath(__file__))
CAPS = (25, 50, 100, 150, 200, 300, 500, 1000, 2000)      # iteration caps; "dim" caps are derived: 100*dim, 200*dim
BOX = 0.99


def nelder_mead_caps(f, x0, step_kind, lo, hi, maxiter, caps_wanted):
    """scipy coefficients; trial points clipped to [lo, hi] when given (None = unbounded).  Returns {cap: (x, f)} for every cap in
    caps_wanted that is reached plus 'end' (tolerance stop xatol 1e-4 & fatol 1e-8, or maxiter)."""
    n = len(x0)
    out = {}
    if n == 0:
        fx = f(np.array(x0))
        return {"end": (np.array(x0), fx, 0)}

    def clip(v):
        return np.minimum(hi, np.maximum(lo, v)) if lo is not None else v
    sim = np.tile(clip(np.array(x0, dtype=float)), (n + 1, 1))
    for k in range(n):
        if step_kind == "scipy":
            sim[k + 1, k] = sim[k + 1, k] * 1.05 if sim[k + 1, k] != 0 else 0.00025
        else:
            sim[k + 1, k] += 0.25 if step_kind == "abs025" else 0.1
        sim[k + 1] = clip(sim[k + 1])
    fs = np.array([f(v) for v in sim])
    o = np.argsort(fs, kind="stable")
    sim, fs = sim[o], fs[o]
    caps = sorted(set(caps_wanted))
    ci = 0
    it = 1
    while True:
        while ci < len(caps) and it > caps[ci]:
            ci += 1
        conv = np.max(np.abs(sim[1:] - sim[0])) <= 1e-4 and np.max(np.abs(fs[0] - fs[1:])) <= 1e-8
        if ci < len(caps) and it == caps[ci]:
            out[caps[ci]] = (sim[0].copy(), float(fs[0]), it)
        if conv or it >= maxiter:
            out["end"] = (sim[0].copy(), float(fs[0]), it)
            for c in caps:
                out.setdefault(c, out["end"]) if c >= it else None
            return out
        xb = sim[:-1].mean(axis=0)
        xr = clip(2 * xb - sim[-1])
        fr = f(xr)
        shrink = False
        if fr < fs[0]:
            xe = clip(3 * xb - 2 * sim[-1])
            fe = f(xe)
            xn, fn = (xe, fe) if fe < fr else (xr, fr)
        elif fr < fs[-2]:
            xn, fn = xr, fr
        elif fr < fs[-1]:
            xc = clip(1.5 * xb - 0.5 * sim[-1])
            fc = f(xc)
            if fc <= fr:
                xn, fn = xc, fc
            else:
                shrink = True
        else:
            xc = clip(0.5 * xb + 0.5 * sim[-1])
            fc = f(xc)
            if fc < fs[-1]:
                xn, fn = xc, fc
            else:
                shrink = True
        if shrink:
            for k in range(1, n + 1):
                sim[k] = clip(sim[0] + 0.5 * (sim[k] - sim[0]))
                fs[k] = f(sim[k])
        else:
            sim[-1], fs[-1] = xn, fn
        o = np.argsort(fs, kind="stable")
        sim, fs = sim[o], fs[o]
        it += 1


CONSTS = ("none", "icpt_free", "mean_free", "mean_fixed")
CONDS = ("p", "zero")
FLAVOURS = ("clip_trial", "clip_obj")
STEPS = ("scipy", "abs01", "abs025")
STARTS = ("zero", "tenth")
SCALES = ("sse", "mse", "halflog")


def fit_cell(job):
    d, p, q, const, cond, flav, step, start, scale = job
    m = S.Model(p, d, q, const, "clip" if flav == "clip_obj" else "raw", cond)
    f = m.objective(scale)
    x0 = m.start(start)
    if m.dim == 0:
        s, nu = m.sse(np.zeros(0))
        return job, {"end": (np.zeros(0), s, nu, m.forecast1(np.zeros(0)), 0)}
    lo = hi = None
    if flav == "clip_trial":
        lo = np.array([-BOX] * (p + q) + [-np.inf] * (m.dim - p - q))
        hi = -lo
    caps = list(CAPS) + [100 * m.dim, 200 * m.dim]
    res = nelder_mead_caps(f, x0, step, lo, hi, 2000, caps)
    out = {}
    for cap, (x, fx, it) in res.items():
        s, nu = m.sse(x)
        out[cap] = (x, s, nu, m.forecast1(x), it)
    out["dim100"] = out[100 * m.dim]
    out["dim200"] = out[200 * m.dim]
    return job, out


# ------------------------------------------------------------------------------------------------ selection
def criterion(kind, n_kind, sig_in_k, s, nu, n_w, p, q, has_c):
    n = {"w": n_w, "used": nu, "y": n_w + 1}[n_kind]
    k = p + q + (1 if has_c else 0) + (1 if sig_in_k else 0)
    s2 = s / n
    if not (s2 > 0) or not math.isfinite(s2):
        return float("-inf") if s2 == 0 else float("nan"), s2
    ll = n * math.log(s2)
    if kind == "aic":
        return ll + 2 * k, s2
    if kind == "bic":
        return ll + k * math.log(n), s2
    den = n - k - 1
    return (ll + 2 * k + 2 * k * (k + 1) / den) if den > 0 else float("inf"), s2


def min_roots(x, p, q):
    """smallest modulus of the roots of the AR polynomial 1 - phi_1 z - ... and of the MA polynomial 1 + theta_1 z + ... (raw coefficients
    as the recursion uses them: clipped to the box); inf for an empty polynomial."""
    out = []
    for coef, sign in ((np.clip(np.asarray(x[:p], dtype=float), -BOX, BOX), -1.0), (np.clip(np.asarray(x[p:p + q], dtype=float), -BOX, BOX), 1.0)):
        c = sign * coef
        while len(c) and c[-1] == 0.0:
            c = c[:-1]
        if len(c) == 0:
            out.append(float("inf"))
            continue
        r = np.roots(np.concatenate([c[::-1], [1.0]]))
        out.append(float(np.min(np.abs(r))) if len(r) else float("inf"))
    return out


def run_selection(table, d, const_kind, cap, proc):
    """table[(p,q,has_c)] -> (x, sse, nu, yhat1).  proc: dict of procedure knobs.  Returns (p, q, has_c, yhat1, n_models)."""
    maxpq, maxsum = proc["maxpq"], proc["maxsum"]
    n_w = 24 - d

    def ic(p, q, c):
        key = (p, q, c)
        if key not in table:
            return None
        x, s, nu, yh, it = table[key][:5]
        v, s2 = criterion(proc["ic"], proc["n"], proc["sigk"], s, nu, n_w, p, q, c)
        if isinstance(proc["reject"], tuple):
            # ("roots", thr): the published check of the lineage (forecast::auto.arima / StatsForecast: a model whose smallest AR or MA
            # root lies inside 1 + 1e-2 gets an infinite criterion); ("ar", thr) / ("ma", thr): one polynomial only
            ar, ma = table[key][5]
            kind, thr = proc["reject"]
            if (kind in ("roots", "ar") and ar < thr) or (kind in ("roots", "ma") and ma < thr):
                return None
            return None if math.isnan(v) else v
        if proc["reject"] == "nonfinite" and not math.isfinite(v):
            return None
        if isinstance(proc["reject"], float) and not (s2 >= proc["reject"]):
            return None
        if math.isnan(v):
            return None
        return v

    def ok(p, q):
        return 0 <= p <= maxpq and 0 <= q <= maxpq and (maxsum is None or p + q <= maxsum)
    cpol = proc["const"]
    c0 = cpol != "never"
    tried = {}

    def ev(p, q, c):
        if (p, q, c) not in tried:
            tried[(p, q, c)] = ic(p, q, c) if ok(p, q) else None
        return tried[(p, q, c)]
    if proc["start"] == "grid":
        best = None
        for p in range(maxpq + 1):
            for q in range(maxpq + 1):
                for c in ((True, False) if cpol == "search" else (c0,)):
                    v = ev(p, q, c)
                    if v is not None and (best is None or v < best[0]):
                        best = (v, p, q, c)
        if best is None:
            return None
        return best[1], best[2], best[3], table[(best[1], best[2], best[3])][3], len(tried)
    starts = {"hk": [(2, 2), (0, 0), (1, 0), (0, 1)], "00": [(0, 0)], "22": [(2, 2)], "11": [(1, 1)]}[proc["start"]]
    best = None
    for p, q in starts:
        v = ev(min(p, maxpq), min(q, maxpq), c0)
        if v is not None and (best is None or v < best[0]):
            best = (v, min(p, maxpq), min(q, maxpq), c0)
    if cpol == "search" and best is not None:
        v = ev(best[1], best[2], not c0)
        if v is not None and v < best[0]:
            best = (v, best[1], best[2], not c0)
    if best is None:
        return None
    for _ in range(100):
        v0, p, q, c = best
        if proc["nbr"] == "hk":
            cand = [(p - 1, q, c), (p + 1, q, c), (p, q - 1, c), (p, q + 1, c), (p - 1, q - 1, c), (p + 1, q + 1, c), (p - 1, q + 1, c), (p + 1, q - 1, c)]
        else:
            cand = [(p + 1, q, c), (p - 1, q, c), (p, q + 1, c), (p, q - 1, c)]
        if cpol == "search":
            cand.append((p, q, not c))
        moved = False
        nb = best
        for (pp, qq, cc) in cand:
            v = ev(pp, qq, cc)
            if v is not None and v < nb[0]:
                nb = (v, pp, qq, cc)
                moved = True
                if proc["move"] == "first":
                    break
        if not moved:
            break
        best = nb
    return best[1], best[2], best[3], table[(best[1], best[2], best[3])][3], len([t for t in tried.values() if t is not None])


_G = None


def select_recipe(recipe):
    fits, procs, caps, pq = _G
    d, cond, fl, st, sx, sc, ck = recipe
    hits5, tally, n_run = [], {}, 0
    for cap in caps:
        table = {}
        for p in pq:
            for q in pq:
                for has_c in (True, False):
                    r = fits.get((d, p, q, ck if has_c else "none", cond, fl, st, sx, sc))
                    if r is not None:
                        e = r.get(cap, r["end"])
                        table[(p, q, has_c)] = tuple(e[:5]) + (min_roots(e[0], p, q),)
        for proc in procs:
            if proc["maxpq"] > max(pq):
                continue
            res = run_selection(table, d, ck, cap, proc)
            n_run += 1
            if res is None:
                continue
            p, q, c, yh, nm = res
            tally[(p, q, c)] = tally.get((p, q, c), 0) + 1
            dist = abs(yh - TARGET)
            if dist <= 1.8e-4:
                hits5.append((dist, yh, p, q, c, d, cond, fl, st, sx, sc, ck, cap, proc, nm))
    return hits5, tally, n_run


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    os.makedirs(os.path.join(HERE, "results"), exist_ok=True)
    cache = os.path.join("/tmp", "arima_selection_fits%s.pkl" % ("_quick" if args.quick else ""))
    # d = 1 only (KPSS and the variance rules all give d = 1 on this series); the full grid keeps what the quick one showed to matter:
    # conditioning on the first p residuals (no zero-history run ever came within 1e-2), the two simplex conventions the repo's
    # optimisers use, sum of squares as the objective
    ds = (1,)
    pq = range(0, 4) if args.quick else range(0, 6)
    conds = CONDS if args.quick else ("p",)
    steps = ("scipy", "abs01")
    scales = ("sse",)
    jobs = [(d, p, q, c, cond, fl, st, sx, sc) for d in ds for p in pq for q in pq for c in CONSTS for cond in conds for fl in FLAVOURS
            for st in steps for sx in STARTS for sc in scales]
    t0 = time.time()
    if os.path.exists(cache):
        fits = pickle.load(open(cache, "rb"))
    else:
        with mp.Pool(args.procs) as pool:
            fits = dict(pool.imap_unordered(fit_cell, jobs, chunksize=8))
        pickle.dump(fits, open(cache, "wb"))
    print(f"# stage 1: {len(fits)} fits in {time.time() - t0:.0f} s", flush=True)

    caps = list(CAPS) + ["dim100", "dim200", "end"]
    procs = []
    for start, maxpq, maxsum, nbr, move, const, ic, n_kind, sigk, reject in itertools.product(
            ("hk", "00", "22", "11", "grid"), (2, 3, 5), (5, None), ("hk", "axis"), ("first", "best"), ("always", "never", "search"),
            ("aic", "aicc", "bic"), ("w", "used", "y"), (True, False), (None, "nonfinite", 1e-12, 1e-10, 1e-8, 1e-6, 1e-4, ("roots", 1.0), ("roots", 1.001), ("roots", 1.005), ("roots", 1.01),
             ("ar", 1.0), ("ar", 1.001), ("ar", 1.01), ("ma", 1.0), ("ma", 1.001), ("ma", 1.01))):
        if start == "grid" and (nbr != "hk" or move != "first"):
            continue
        procs.append(dict(start=start, maxpq=maxpq, maxsum=maxsum, nbr=nbr, move=move, const=const, ic=ic, n=n_kind, sigk=sigk, reject=reject))
    print(f"# stage 2: {len(procs)} procedures x recipes", flush=True)

    recipes = [(d, cond, fl, st, sx, sc, ck) for d in ds for cond in conds for fl in FLAVOURS for st in steps
               for sx in STARTS for sc in scales for ck in ("icpt_free", "mean_free", "mean_fixed") if d == 1]
    # (the d rule is not part of the grid: KPSS and the variance rules all give d = 1 on this series)
    global _G
    _G = (fits, procs, caps, list(pq))
    hits6, hits5, tally, n_run = [], [], {}, 0
    with mp.Pool(args.procs) as pool:
        for h5, tl, nr in pool.imap_unordered(select_recipe, recipes, chunksize=1):
            hits5 += h5
            n_run += nr
            for k, v in tl.items():
                tally[k] = tally.get(k, 0) + v
    hits6 = [r for r in hits5 if r[0] <= 5e-7]
    out = os.path.join(HERE, "results", "selection_summary%s.txt" % ("_quick" if args.quick else ""))
    with open(out, "w") as fh:
        def w(*a):
            print(*a, file=fh)
            print(*a, flush=True)
        w(f"# selection-procedure search: {n_run} (recipe, cap, procedure) runs; target {TARGET}")
        w(f"# final models (count): " + ", ".join(f"({p},1,{q}){'+c' if c else ''}: {v}" for (p, q, c), v in sorted(tally.items(), key=lambda kv: -kv[1])[:20]))
        w(f"# procedures ending within 1e-5 relative (1.8e-4): {len(hits5)}; inside the 6-decimal window: {len(hits6)}")
        # group the within-1e-5 hits by (final model, estimator recipe) to see which knobs matter
        by = {}
        for row in hits5:
            dist, yh, p, q, c, d, cond, fl, st, sx, sc, ck, cap, proc, nm = row
            by.setdefault((p, q, c, cond, fl, ck), []).append(row)
        for key, rows in sorted(by.items(), key=lambda kv: -len(kv[1])):
            w(f"## final ({key[0]},1,{key[1]}){'+c' if key[2] else ''} cond={key[3]} {key[4]} const={key[5]}: {len(rows)} procedure runs; closest {min(r[0] for r in rows):.2e}")
            # marginal counts of every knob among these
            for knob in ("start", "maxpq", "maxsum", "nbr", "move", "const", "ic", "n", "sigk", "reject"):
                cnt = {}
                for r in rows:
                    cnt[str(r[13][knob])] = cnt.get(str(r[13][knob]), 0) + 1
                w(f"     {knob:7s}: " + "  ".join(f"{k}={v}" for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])))
            for knob, idx in (("step", 8), ("start_x", 9), ("scale", 10), ("cap", 12)):
                cnt = {}
                for r in rows:
                    cnt[str(r[idx])] = cnt.get(str(r[idx]), 0) + 1
                w(f"     {knob:7s}: " + "  ".join(f"{k}={v}" for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])))
        w("## closest 40 runs")
        for row in sorted(hits5, key=lambda r: r[0])[:40]:
            dist, yh, p, q, c, d, cond, fl, st, sx, sc, ck, cap, proc, nm = row
            w(f"   |d|={dist:.3e} yhat1={yh:.7f} ({p},{d},{q}){'+c' if c else ''} cond={cond} {fl} step={st} x0={sx} obj={sc} const={ck} cap={cap} models={nm} {proc}")
    print("wrote", out)


if __name__ == "__main__":
    main()
