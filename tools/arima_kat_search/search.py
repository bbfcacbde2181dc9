#!/usr/bin/env python3
"""Known-answer search for the reference's AutoARIMA pin (test/sql/ts_model_distinctness.test:164):

    _ts_forecast(<24 literal observations>, 3, 'AutoARIMA').point[1]  ==  18.014537   (6 decimals)

The arithmetic behind that number lives in the un-vendored crate anofox-forecast 0.15.3 (Cargo.lock:50-53), so the
oracle has to *restate* it.  This tool makes the statement "no consistent estimator lands on 18.014537" checkable:
it fits every ARIMA(p,d,q) order of a grid to the literal series with every estimator / optimiser / parametrisation /
constant convention / objective scaling of a second grid, records the one-step forecast of EVERY iterate an optimiser
visits (so every stopping rule -- iteration cap, gradient or simplex tolerance -- is covered by one trajectory), and
prints, per family, how close anything gets.

    python tools/arima_kat_search/search.py [--procs 6] [--quick | --full]      # writes results/*.csv + results/summary.txt

Families
  closed   : conditional least squares (OLS) AR, Yule-Walker, Burg, Hannan-Rissanen           (no optimiser)
  nm       : Nelder-Mead (scipy coefficients, the repo's own start conventions), every iterate's best vertex
  bfgs     : scipy BFGS / L-BFGS-B (with and without +-0.99 boxes), every iterate
  lbfgs    : a two-loop L-BFGS (what the `lbfgs 0.3.0` crate of Cargo.lock:923-929 provides: a direction buffer, no
             line search of its own) + Armijo backtracking, forward / central difference gradients, every iterate
  ml       : exact Gaussian likelihood (Kalman filter, stationary start) maximised by Nelder-Mead / BFGS, every iterate

What counts as a hit: |yhat1 - 18.014537| <= 5e-7 (the printed decimals) at a point where the optimiser would actually
STOP under a rule with round constants (its converged end point, or an iteration cap in {5,10,15,20,25,30,40,50,100,
150,200,250,300,400,500,1000}).  An arbitrary mid-trajectory iterate inside the window is reported separately: with
~10^5 trajectories crossing the +-0.05 neighbourhood of 18.0 a few of them fall into a 1e-6 window by chance.
"""
from __future__ import annotations

import argparse
import csv
import itertools
import math
import multiprocessing as mp
import os
import sys
import time

import numpy as np
from scipy import linalg, optimize

Y = np.array([10, 12, 14, 11, 13, 15, 12, 14, 16, 13, 15, 17, 14, 16, 18, 15, 17, 19, 16, 18, 20, 17, 19, 21], dtype=float)
TARGET = 18.014537
ROUND_CAPS = (5, 10, 15, 20, 25, 30, 40, 50, 100, 150, 200, 250, 300, 400, 500, 1000)
HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------------ model
def difference(y, d):
    w = np.array(y, dtype=float)
    for _ in range(d):
        w = w[1:] - w[:-1]
    return w


def integrate_first(y, d, wnext):
    if d == 0:
        return wnext
    if d == 1:
        return y[-1] + wnext
    return y[-1] + (y[-1] - y[-2]) + wnext


def pacf_to_coef(u):
    """tanh -> partial autocorrelations -> coefficients of a stationary polynomial (Jones 1980)."""
    phi = []
    for j, uj in enumerate(u):
        a = math.tanh(uj)
        phi = [phi[i] - a * phi[j - 1 - i] for i in range(j)] + [a]
    return phi


class Model:
    """ARMA(p,q) on the d-times differenced series.
    const: 'none' | 'mean_fixed' (sample mean of w, not estimated) | 'mean_free' | 'icpt_free' (w_t = c + sum phi w_{t-i} + ...)
    param: 'raw' | 'tanh' (tanh-PACF on both polynomials) | 'clip' (raw, clipped to +-0.99)
    cond : 'p' (residuals start at t = p) | 'zero' (from t = 0, missing history = 0)"""

    def __init__(self, p, d, q, const, param, cond):
        self.p, self.d, self.q, self.const, self.param, self.cond = p, d, q, const, param, cond
        self.w = difference(Y, d)
        self.n = len(self.w)
        self.wmean = float(self.w.mean())
        self.dim = p + q + (1 if const in ("mean_free", "icpt_free") else 0)

    def unpack(self, x):
        p, q = self.p, self.q
        a, b = list(x[:p]), list(x[p:p + q])
        if self.param == "tanh":
            a, b = pacf_to_coef(a), pacf_to_coef(b)
        elif self.param == "clip":
            a = [min(0.99, max(-0.99, v)) for v in a]
            b = [min(0.99, max(-0.99, v)) for v in b]
        c = x[p + q] if self.dim > p + q else (self.wmean if self.const == "mean_fixed" else 0.0)
        return a, b, c

    def resid(self, x):
        a, b, c = self.unpack(x)
        p, q, w, n = self.p, self.q, self.w, self.n
        icpt = self.const == "icpt_free"
        mu = 0.0 if icpt else c
        z = w - mu
        e = np.zeros(n)
        t0 = p if self.cond == "p" else 0
        for t in range(t0, n):
            acc = z[t] - (c if icpt else 0.0)
            for i in range(p):
                if t - 1 - i >= 0:
                    acc -= a[i] * z[t - 1 - i]
            for j in range(q):
                if t - 1 - j >= 0:
                    acc -= b[j] * e[t - 1 - j]
            e[t] = acc
        return e, t0

    def sse(self, x):
        e, t0 = self.resid(x)
        s = float(np.dot(e, e))
        return s if math.isfinite(s) else 1e300, self.n - t0

    def forecast1(self, x):
        a, b, c = self.unpack(x)
        e, _ = self.resid(x)
        icpt = self.const == "icpt_free"
        mu = 0.0 if icpt else c
        z = self.w - mu
        acc = c
        for i in range(self.p):
            acc += a[i] * z[self.n - 1 - i]
        for j in range(self.q):
            acc += b[j] * e[self.n - 1 - j]
        return integrate_first(Y, self.d, acc)

    def objective(self, scale):
        def f(x):
            s, nu = self.sse(x)
            if scale == "sse":
                return s
            if scale == "mse":
                return s / nu
            v = max(s / nu, 1e-300)
            return 0.5 * math.log(v) if scale == "halflog" else 0.5 * nu * math.log(v)
        return f

    def start(self, kind):
        x = np.zeros(self.dim)
        if kind == "tenth":
            x[:self.p + self.q] = 0.1
        if self.dim > self.p + self.q:
            x[-1] = self.wmean if kind != "zeroconst" else 0.0
            if self.const == "icpt_free" and kind == "tenth":
                x[-1] = self.wmean
        return x

    # exact Gaussian likelihood of the stationary ARMA (Harvey state space, Lyapunov start), concentrated over sigma^2
    def ml_objective(self):
        def f(x):
            a, b, c = self.unpack(x)
            if self.const == "icpt_free":
                return 1e300
            r = max(self.p, self.q + 1)
            T = np.zeros((r, r))
            for i in range(self.p):
                T[i, 0] = a[i]
            for i in range(r - 1):
                T[i, i + 1] = 1.0
            R = np.zeros(r)
            R[0] = 1.0
            for j in range(self.q):
                R[j + 1] = b[j]
            try:
                if np.max(np.abs(np.linalg.eigvals(T))) >= 1.0 - 1e-9:
                    return 1e300
                P = linalg.solve_discrete_lyapunov(T, np.outer(R, R))
            except Exception:
                return 1e300
            s = np.zeros(r)
            ssq, sumlog = 0.0, 0.0
            for t in range(self.n):
                F = P[0, 0]
                if not (F > 0):
                    return 1e300
                v = (self.w[t] - c) - s[0]
                ssq += v * v / F
                sumlog += math.log(F)
                K = T @ P[:, 0] / F
                s = T @ s + K * v
                P = T @ P @ T.T + np.outer(R, R) - np.outer(K, K) * F
            return 0.5 * (math.log(max(ssq / self.n, 1e-300)) + sumlog / self.n)
        return f


# ------------------------------------------------------------------------------------------------ optimisers (with trajectories)
def nelder_mead(f, x0, step_kind, maxiter):
    """scipy coefficients (1, 2, 1/2, 1/2).  step_kind: 'scipy' (5 % / 0.00025), 'abs025' (+0.25, the repo's ARIMA start), 'abs01'.
    Yields (iteration, best vertex, converged_flag[xatol 1e-4 & fatol 1e-8])."""
    n = len(x0)
    if n == 0:
        yield 0, np.array(x0), True
        return
    sim = np.tile(np.array(x0, dtype=float), (n + 1, 1))
    for k in range(n):
        if step_kind == "scipy":
            sim[k + 1, k] = sim[k + 1, k] * 1.05 if sim[k + 1, k] != 0 else 0.00025
        else:
            sim[k + 1, k] += 0.25 if step_kind == "abs025" else 0.1
    fs = np.array([f(v) for v in sim])
    o = np.argsort(fs, kind="stable")
    sim, fs = sim[o], fs[o]
    for it in range(1, maxiter + 1):
        conv = np.max(np.abs(sim[1:] - sim[0])) <= 1e-4 and np.max(np.abs(fs[0] - fs[1:])) <= 1e-8
        yield it, sim[0].copy(), bool(conv)
        if conv:
            return
        xb = sim[:-1].mean(axis=0)
        xr = 2 * xb - sim[-1]
        fr = f(xr)
        shrink = False
        if fr < fs[0]:
            xe = 3 * xb - 2 * sim[-1]
            fe = f(xe)
            xn, fn = (xe, fe) if fe < fr else (xr, fr)
        elif fr < fs[-2]:
            xn, fn = xr, fr
        elif fr < fs[-1]:
            xc = 1.5 * xb - 0.5 * sim[-1]
            fc = f(xc)
            if fc <= fr:
                xn, fn = xc, fc
            else:
                shrink = True
        else:
            xc = 0.5 * xb + 0.5 * sim[-1]
            fc = f(xc)
            if fc < fs[-1]:
                xn, fn = xc, fc
            else:
                shrink = True
        if shrink:
            for k in range(1, n + 1):
                sim[k] = sim[0] + 0.5 * (sim[k] - sim[0])
                fs[k] = f(sim[k])
        else:
            sim[-1], fs[-1] = xn, fn
        o = np.argsort(fs, kind="stable")
        sim, fs = sim[o], fs[o]


def num_grad(f, x, fx, mode, eps):
    g = np.zeros(len(x))
    for i in range(len(x)):
        h = eps * max(1.0, abs(x[i])) if eps < 0 else eps
        h = abs(h)
        xp = x.copy()
        xp[i] += h
        if mode == "fwd":
            g[i] = (f(xp) - fx) / h
        else:
            xm = x.copy()
            xm[i] -= h
            g[i] = (f(xp) - f(xm)) / (2 * h)
    return g


def lbfgs(f, x0, mem, gmode, eps, step0, maxiter, c1=1e-4, shrink=0.5):
    """Two-loop L-BFGS + Armijo backtracking.  Yields (iteration, x, |g|_inf)."""
    x = np.array(x0, dtype=float)
    if len(x) == 0:
        yield 0, x, 0.0
        return
    fx = f(x)
    g = num_grad(f, x, fx, gmode, eps)
    S, Yk = [], []
    for it in range(1, maxiter + 1):
        yield it, x.copy(), float(np.max(np.abs(g)))
        q = g.copy()
        al = []
        for s, y in zip(reversed(S), reversed(Yk)):
            a = np.dot(s, q) / np.dot(y, s)
            al.append(a)
            q -= a * y
        if S:
            q *= np.dot(S[-1], Yk[-1]) / np.dot(Yk[-1], Yk[-1])
        for (s, y), a in zip(zip(S, Yk), reversed(al)):
            b = np.dot(y, q) / np.dot(y, s)
            q += (a - b) * s
        dirn = -q
        slope = np.dot(g, dirn)
        if not (slope < 0):
            dirn, slope = -g, -np.dot(g, g)
            S, Yk = [], []
        t = 1.0 if (S or step0 == "one") else 1.0 / max(np.linalg.norm(g), 1e-300)
        ok = False
        for _ in range(30):
            xn = x + t * dirn
            fn = f(xn)
            if math.isfinite(fn) and fn <= fx + c1 * t * slope:
                ok = True
                break
            t *= shrink
        if not ok:
            return
        gn = num_grad(f, xn, fn, gmode, eps)
        s, y = xn - x, gn - g
        if np.dot(s, y) > 1e-12 * np.dot(y, y):
            S.append(s)
            Yk.append(y)
            if len(S) > mem:
                S.pop(0)
                Yk.pop(0)
        if abs(fx - fn) <= 1e-15 * max(1.0, abs(fx)) and np.max(np.abs(s)) <= 1e-14:
            x, fx, g = xn, fn, gn
            yield it + 1, x.copy(), float(np.max(np.abs(g)))
            return
        x, fx, g = xn, fn, gn


def scipy_traj(f, x0, method, bounds):
    out = []
    if len(x0) == 0:
        return [(0, np.array(x0))]
    kw = dict(method=method, callback=lambda xk, *a: out.append((len(out) + 1, np.array(xk))), options={"maxiter": 1000})
    if bounds is not None:
        kw["bounds"] = bounds
    try:
        optimize.minimize(f, x0, **kw)
    except Exception:
        pass
    return out


# ------------------------------------------------------------------------------------------------ closed forms
def closed_forms(p, d, q, const):
    """AR estimators without an optimiser (q = 0) and Hannan-Rissanen (q > 0).  Returns [(name, yhat1)]."""
    out = []
    w = difference(Y, d)
    n = len(w)
    mu = w.mean() if const in ("mean_fixed", "mean_free") else 0.0
    z = w - mu

    def fc(a, c=None, e=None, b=()):
        acc = (mu if c is None else c)
        base = z if c is None else w
        for i in range(len(a)):
            acc += a[i] * base[n - 1 - i]
        for j in range(len(b)):
            acc += b[j] * e[n - 1 - j]
        return integrate_first(Y, d, acc)

    if q == 0 and p > 0:
        X = np.column_stack([z[p - 1 - i:n - 1 - i] for i in range(p)])
        a = np.linalg.lstsq(X, z[p:], rcond=None)[0]
        out.append(("ols", fc(a)))
        if const == "icpt_free":
            X1 = np.column_stack([np.ones(n - p)] + [w[p - 1 - i:n - 1 - i] for i in range(p)])
            co = np.linalg.lstsq(X1, w[p:], rcond=None)[0]
            out.append(("ols_icpt", fc(co[1:], c=co[0])))
        for bias in (True, False):
            r = np.array([np.dot(z[:n - k], z[k:]) / (n if bias else n - k) for k in range(p + 1)])
            try:
                a = linalg.solve_toeplitz(r[:p], r[1:p + 1])
                out.append(("yule_walker_" + ("biased" if bias else "unbiased"), fc(a)))
            except Exception:
                pass
        f_, b_ = z[1:].copy(), z[:-1].copy()
        a = np.zeros(0)
        for k in range(p):
            den = np.dot(f_, f_) + np.dot(b_, b_)
            kk = 2 * np.dot(f_, b_) / den if den > 0 else 0.0
            a = np.concatenate([a - kk * a[::-1], [kk]])
            f_, b_ = (f_ - kk * b_)[1:], (b_ - kk * f_)[:-1]
        out.append(("burg", fc(a)))
    if q > 0:
        for K in (p + q + 1, p + q + 2, 6, 8):
            if K >= n - p - q - 2 or K < 1:
                continue
            X = np.column_stack([z[K - 1 - i:n - 1 - i] for i in range(K)])
            al = np.linalg.lstsq(X, z[K:], rcond=None)[0]
            e = np.zeros(n)
            e[K:] = z[K:] - X @ al
            t0 = K + q
            cols = [z[t0 - 1 - i:n - 1 - i] for i in range(p)] + [e[t0 - 1 - j:n - 1 - j] for j in range(q)]
            co = np.linalg.lstsq(np.column_stack(cols), z[t0:], rcond=None)[0]
            out.append((f"hannan_rissanen_K{K}", fc(co[:p], e=e, b=co[p:])))
    return out


# ------------------------------------------------------------------------------------------------ one work item = one (order, const, param, cond)
def scan(model, family, variant, traj, rows):
    """traj yields (iteration, x[, extra]) -- record closest iterate, the end point and the round-cap iterates."""
    best = (9e9, None, None)
    last = None
    for rec in traj:
        it, x = rec[0], rec[1]
        try:
            yh = model.forecast1(x)
        except Exception:
            continue
        if not math.isfinite(yh):
            continue
        dist = abs(yh - TARGET)
        last = (it, yh)
        if dist < best[0]:
            best = (dist, it, yh)
        if it in ROUND_CAPS:
            rows.append((family, variant + f"|cap{it}", model.p, model.d, model.q, model.const, model.param, model.cond, it, yh, dist, "stop"))
    if last is not None:
        rows.append((family, variant + "|end", model.p, model.d, model.q, model.const, model.param, model.cond, last[0], last[1], abs(last[1] - TARGET), "stop"))
    if best[1] is not None:
        rows.append((family, variant + "|closest-iterate", model.p, model.d, model.q, model.const, model.param, model.cond, best[1], best[2], best[0], "pass"))


def work(item):
    (p, d, q, const, param, cond, quick) = item
    rows = []
    m = Model(p, d, q, const, param, cond)
    if m.dim == 0:
        yh = m.forecast1(np.zeros(0))
        rows.append(("closed", "no-parameters", p, d, q, const, param, cond, 0, yh, abs(yh - TARGET), "stop"))
        return rows
    if param == "raw" and cond == "p":
        for name, yh in closed_forms(p, d, q, const):
            if math.isfinite(yh):
                rows.append(("closed", name, p, d, q, const, param, cond, 0, yh, abs(yh - TARGET), "stop"))
    starts = ("zero", "tenth")
    scales = {0: ("sse", "mse", "halflog"), 1: ("sse", "halflog"), 2: ("sse", "mse", "halflog", "nloglik")}[quick]
    for st, sc in itertools.product(starts, scales):
        f = m.objective(sc)
        x0 = m.start(st)
        for sk in ("scipy", "abs025", "abs01"):
            scan(m, "nm", f"nm-{sk}|start-{st}|obj-{sc}", nelder_mead(f, x0, sk, 1000), rows)
        if param != "clip":
            scan(m, "bfgs", f"scipy-BFGS|start-{st}|obj-{sc}", scipy_traj(f, x0, "BFGS", None), rows)
            scan(m, "bfgs", f"scipy-L-BFGS-B|start-{st}|obj-{sc}", scipy_traj(f, x0, "L-BFGS-B", None), rows)
            if param == "raw":
                bnd = [(-0.99, 0.99)] * (p + q) + [(None, None)] * (m.dim - p - q)
                scan(m, "bfgs", f"scipy-L-BFGS-B-box0.99|start-{st}|obj-{sc}", scipy_traj(f, x0, "L-BFGS-B", bnd), rows)
        grid = itertools.product((5, 10) if quick == 2 else (5,), ("fwd", "cen"),
                                 {0: (1e-4, 1e-6, 1e-8), 1: (1e-6,), 2: (1e-4, 1e-6, 1e-8, -1.4901161193847656e-08)}[quick], ("one", "invnorm"))
        for mem, gm, eps, s0 in grid:
            scan(m, "lbfgs", f"lbfgs-m{mem}-{gm}-eps{eps:g}-step0{s0}|start-{st}|obj-{sc}", lbfgs(f, x0, mem, gm, eps, s0, 200 if quick == 1 else 1000), rows)
    if const != "icpt_free" and cond == "p" and param == "tanh":
        f = m.ml_objective()
        for st in starts:
            x0 = m.start(st)
            scan(m, "ml", f"ml-nm-abs01|start-{st}", nelder_mead(f, x0, "abs01", 600), rows)
            scan(m, "ml", f"ml-scipy-BFGS|start-{st}", scipy_traj(f, x0, "BFGS", None), rows)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=max(1, (os.cpu_count() or 2) - 2))
    ap.add_argument("--quick", action="store_true", help="~3 minutes: two objective scalings, four L-BFGS settings per start")
    ap.add_argument("--full", action="store_true", help="~1.5 hours: four objective scalings, 32 L-BFGS settings per start (default: three / twelve, ~12 minutes)")
    ap.add_argument("--out", default=os.path.join(HERE, "results"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    orders = [(p, 1, q) for p in range(4) for q in range(4) if p + q <= 5]
    orders += [(p, d, q) for d in (0, 2) for p in range(3) for q in range(3)]
    level = 1 if a.quick else (2 if a.full else 0)
    items = [(p, d, q, const, param, cond, level) for (p, d, q) in orders for const in ("none", "mean_fixed", "mean_free", "icpt_free")
             for param in ("raw", "tanh", "clip") for cond in ("p", "zero")]
    t0 = time.time()
    with mp.Pool(a.procs) as pool:
        rows = [r for rs in pool.imap_unordered(work, items, chunksize=1) for r in rs]
    rows.sort(key=lambda r: r[10])
    hdr = ("family", "variant", "p", "d", "q", "const", "param", "cond", "iteration", "yhat1", "abs_diff", "kind")
    with open(os.path.join(a.out, "closest_variants.csv"), "w", newline="") as fh:      # everything within 1e-5 relative of the target
        wr = csv.writer(fh)
        wr.writerow(hdr)
        for r in rows:
            if r[10] < 2e-4:
                wr.writerow(r[:9] + (f"{r[9]:.9f}", f"{r[10]:.3e}", r[11]))
    stops = [r for r in rows if r[11] == "stop"]
    passes = [r for r in rows if r[11] == "pass"]
    with open(os.path.join(a.out, "summary.txt"), "w") as fh:
        def P(*s):
            print(*s, file=fh)
            print(*s)
        P(f"# AutoARIMA known-answer search: target {TARGET}, {len(items)} (order, constant, parametrisation, conditioning) cells,")
        P(f"# {len(stops)} stopping points + {len(passes)} trajectories, {time.time() - t0:.0f} s on {a.procs} processes" + (" (--quick)" if a.quick else (" (--full)" if a.full else "")))
        for fam in ("closed", "nm", "bfgs", "lbfgs", "ml"):
            fs = [r for r in stops if r[0] == fam]
            if not fs:
                continue
            d = np.array([r[10] for r in fs])
            P(f"\n## family {fam}: {len(fs)} stopping points; within 5e-7 (6 decimals): {(d <= 5e-7).sum()}, within 1e-5 relative (1.8e-4): {(d <= 1.8e-4).sum()}, "
              f"exactly 18.000000 (+-5e-7): {sum(abs(r[9] - 18.0) <= 5e-7 for r in fs)}")
            P("   closest stopping points:")
            for r in fs[:12]:
                P(f"   |d|={r[10]:.3e}  yhat1={r[9]:.7f}  ARIMA({r[2]},{r[3]},{r[4]}) const={r[5]} param={r[6]} cond={r[7]} it={r[8]}  {r[1]}")
        d = np.array([r[10] for r in passes])
        P(f"\n## mid-trajectory iterates (NOT stopping points): {len(passes)} trajectories, closest iterate within 5e-7: {(d <= 5e-7).sum()}, within 1.8e-4: {(d <= 1.8e-4).sum()}")
        for r in passes[:8]:
            P(f"   |d|={r[10]:.3e}  yhat1={r[9]:.7f}  ARIMA({r[2]},{r[3]},{r[4]}) const={r[5]} param={r[6]} cond={r[7]} it={r[8]}  {r[1]}")
        hits = [r for r in stops if r[10] <= 5e-7]
        near = sum(1 for r in stops if r[10] <= 1.8e-4)
        expect = near * 5e-7 / 1.8e-4
        P(f"\n## verdict: {len(hits)} stopping point(s) inside the 6-decimal window; {near} lie within 1e-5 relative of the target, so a uniform scatter of "
          f"them puts {expect:.2f} into a window of that width BY CHANCE -- " + ("the hit count is what chance predicts: no variant is singled out" if len(hits) <= max(2, 3 * expect)
          else "more than chance predicts: inspect the hits above"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
