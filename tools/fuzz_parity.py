"""Randomised differential test: HIP path (C-ABI batch entry) against the CPU oracle.
python tools/fuzz_parity.py [seconds] [seed] -- prints every mismatch, exits 1 if any."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anofox_forecast_amd import api, lib
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MODELS = ["AutoETS", "AutoETS", "AutoETS", "AutoARIMA", "ETS", "HoltWinters", "Holt", "SES", "SESOptimized", "SeasonalES",
          "SeasonalESOptimized", "Naive", "SeasonalNaive", "SMA", "RandomWalkDrift", "ARIMA"]
SPECS = ["ANN", "AAN", "AAdN", "ANA", "AAA", "AAdA", "MNN", "MAN", "MAdN", "MMN", "MMdN", "AMN", "AMdN", "ANM", "AAM", "AAdM", "AMA",
         "AMdA", "AMM", "AMdM", "MNM", "MAM", "MAdM", "MMM", "MMdM", "MAA", "ZZZ", "AA"]
POOLS = ["", "complete", "no_multiplicative_trend", "damped_trend_only", "match_error_seasonal", "reduced"]
NO_PERIOD = {"Naive", "SES", "SESOptimized", "Holt", "RandomWalkDrift", "ARIMA"}


# FUZZ_COUNTS=1 (round 6): every series is a count series (Poisson counts, constants, rounded trends; now and then a half-integer or
# a value above 65,535) and the library is told to stream the narrowest exact copy of the block whatever the batch size
# (ANOFOX_HIP_TUNE compact=2) -- the compact-storage kernels against the oracle.  NULLs are interpolated by the packer: a batch whose
# interpolated values are not exact falls back to the wider type by itself, which is part of what is fuzzed.
COUNTS = os.environ.get("FUZZ_COUNTS") == "1"
if COUNTS:
    os.environ["ANOFOX_HIP_TUNE"] = "compact=2"


def make_series(n):
    if COUNTS:
        kind = rng.integers(0, 5)
        t = np.arange(n)
        m = int(rng.choice([1, 2, 4, 7, 12]))
        if kind == 0:
            y = rng.poisson(rng.lognormal(0, 1.2) * (1 + 0.3 * np.sin(2 * np.pi * t / max(m, 2))), n).astype(float) + float(rng.integers(0, 2))
        elif kind == 1:
            y = np.rint(20 + 0.1 * t + 5 * np.sin(2 * np.pi * t / max(m, 2)) + rng.normal(0, 1, n)) + 30.0
        elif kind == 2:
            y = np.full(n, float(rng.integers(0, 5)))
        elif kind == 3:
            y = np.rint(np.exp(0.01 * t + rng.normal(0, 0.05, n)) * (1 + 0.2 * np.sin(2 * np.pi * t / max(m, 2))) * 100.0) * (0.5 if rng.random() < 0.3 else 1.0)
        else:
            y = np.rint(np.abs(rng.normal(0, 1, n)) * 10.0 ** rng.integers(0, 6))
        return y
    kind = rng.integers(0, 6)
    t = np.arange(n)
    m = int(rng.choice([1, 2, 4, 7, 12]))
    if kind == 0:
        y = rng.poisson(rng.lognormal(0, 1.2) * (1 + 0.3 * np.sin(2 * np.pi * t / max(m, 2))), n).astype(float)
    elif kind == 1:
        y = 20 + 0.1 * t + 5 * np.sin(2 * np.pi * t / max(m, 2)) + rng.normal(0, 1, n)
    elif kind == 2:
        y = np.cumsum(rng.normal(0.1, 1, n)) + 50
    elif kind == 3:
        y = np.exp(0.01 * t + rng.normal(0, 0.05, n)) * (1 + 0.2 * np.sin(2 * np.pi * t / max(m, 2)))
    elif kind == 4:
        y = np.full(n, float(rng.integers(0, 5)))
    else:
        y = rng.normal(0, 1, n) * 10.0 ** rng.integers(-3, 6)
    return y


t_end = time.time() + budget
n_cases = n_bad = 0
seen = {}
while time.time() < t_end:
    model = str(rng.choice(MODELS))
    kw = {}
    period = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 7, 12, 24]))
    if model in NO_PERIOD:
        period = int(rng.choice([0, 0, 0, 1, 7]))
    kw["seasonal_period"] = period
    if model == "ETS" and rng.random() < 0.9:
        kw["ets_model"] = str(rng.choice(SPECS))
    if model == "AutoETS":
        kw["model_pool"] = str(rng.choice(POOLS))
    if model == "SMA":
        kw["window"] = int(rng.choice([0, 1, 3, 10]))
    kw["confidence_level"] = float(rng.choice([0.5, 0.8, 0.9, 0.95, 0.99]))
    h = int(rng.choice([1, 3, 12, 28]))
    nser = int(rng.integers(1, 70))
    series, valids = [], []
    for _ in range(nser):
        n = int(rng.choice([0, 1, 2, 3, 4, 9, 10, 15, 30, 60, 100, 150]))
        y = make_series(n)
        v = rng.random(n) > (0.1 if rng.random() < 0.3 else 0.0)
        series.append(y)
        valids.append(v)
    opts = lib.make_options(model, h, **kw)
    oo = O.make_options(model, h, **kw)
    got, berr = api.forecast_batch(series, opts, valids)
    for i, (y, v, r) in enumerate(zip(series, valids, got)):
        ref = O.forecast(y, oo, v)
        n_cases += 1
        ok = (not berr["ok"] and not ref["ok"] and berr["code"] == ref["code"]) if not berr["ok"] else (
            r["ok"] == ref["ok"] and ((not r["ok"] and r["code"] == ref["code"]) or (r["ok"] and r["model_name"] == ref["model_name"] and
            np.allclose(r["point"], ref["point"], rtol=1e-12, atol=0, equal_nan=True) and np.allclose(r["upper"], ref["upper"], rtol=1e-12, atol=0, equal_nan=True))))
        if not ok:
            n_bad += 1
            sig = (model, r.get("code"), berr["code"], ref.get("code"), r.get("ok"), ref["ok"])
            seen[sig] = seen.get(sig, 0) + 1
            if seen[sig] <= 2:
                print("MISMATCH", model, kw, "h", h, "len", len(y), "nulls", int((~v).sum()), "gpu", (r.get("ok"), r.get("code"), r.get("model_name"), berr["code"]),
                  "oracle", (ref["ok"], ref.get("code"), ref.get("model_name")), np.asarray(r.get("point", []))[:2], np.asarray(ref.get("point", []))[:2])
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
print(f"{n_cases} series compared, {n_bad} mismatches")
sys.exit(1 if n_bad else 0)
