"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on identical inputs.

Tolerance: the north star allows 1e-5 relative (fp64); oracle and kernels are written as the same
sequence of IEEE operations (fma only where stated, -ffp-contract=off), so we assert 1e-12 relative
and report the worst case.  Model names, error codes and selected specs must match exactly.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-12


def _rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return np.inf
    same = (np.isnan(a) & np.isnan(b)) | (a == b)           # both NaN, or equal (equal infinities included: inf - inf is NaN)
    with np.errstate(invalid="ignore"):
        d = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    d[same] = 0.0
    d[np.isnan(d)] = np.inf                                  # anything else that is not a number is a mismatch, not a pass
    return float(np.max(d)) if d.size else 0.0


def _compare(api, O, lib, series, model, h, valids=None, **kw):
    opts = lib.make_options(model, h, **kw)
    oo = O.make_options(model, h, **kw)
    got, berr = api.forecast_batch(series, opts, valids)
    worst = 0.0
    for s, y in enumerate(series):
        ref = O.forecast(y, oo, None if valids is None else valids[s])
        if not ref["ok"] and ref["code"] == 5:            # unknown model: rejected before any data is looked at
            assert not berr["ok"] and berr["code"] == ref["code"], (model, berr, ref)
            assert berr["message"] == ref["message"]
            return 0.0
        assert berr["ok"], (model, berr)
        if not ref["ok"] and ref["code"] == 2:            # InvalidInput is per series, after the length checks (forecast.rs:516-565)
            assert not got[s]["ok"] and got[s]["code"] == 2 and got[s]["message"] == ref["message"], (model, s, got[s], ref)
            continue
        assert got[s]["ok"] == ref["ok"], (model, s, got[s], ref)
        if not ref["ok"]:
            assert got[s]["code"] == ref["code"], (model, s, got[s], ref)
            continue
        assert got[s]["model_name"] == ref["model_name"], (model, s, got[s]["model_name"], ref["model_name"])
        for k in ("point", "lower", "upper"):
            worst = max(worst, _rel(got[s][k], ref[k]))
    assert worst <= REL_TOL, (model, kw, worst)
    return worst


@pytest.fixture(scope="module")
def env(hiplib, oracle):
    import torch
    assert torch.cuda.is_available()
    from anofox_forecast_amd import api, synth
    return api, oracle, hiplib, synth


KAT_SERIES = [10, 12, 14, 11, 13, 15, 12, 14, 16, 13, 15, 17, 14, 16, 18, 15, 17, 19, 16, 18, 20, 17, 19, 21]
KATS = {"SES": 18.943503, "SESOptimized": 19.537535, "SeasonalES": 14.451866, "Holt": 20.330877,
        "HoltWinters": 19.953912, "Naive": 21.0, "SMA": 19.0, "RandomWalkDrift": 21.478261}


def test_kernel_reciprocal_is_the_ieee_division(env):
    """The general-class step divides once (oracle/ets.c: 1.0 / d); the kernels compute that quotient with the division's own
    instruction sequence minus range scaling and fix-up (det_math.hpp dm_recip), exact on [2^-1000, 2^1000] -- the domain outside
    which both sides reject the trial point.  2^28 generated operands per seed, every exponent of the domain, both signs."""
    api, O, lib, synth = env
    import ctypes as C
    L = lib.load()
    for seed in (1, 20260101, 0xDEADBEEF):
        bad, first = C.c_uint64(123), C.c_double(0.0)
        assert L.anofox_hip_selftest_recip(1 << 28, seed, C.byref(bad), C.byref(first))
        assert bad.value == 0, (seed, bad.value, first.value.hex())


def test_lane_stats_are_consistent(env):
    """anofox_hip_batch_lane_stats: live lane-passes never exceed 64 x wave passes, every fitted spec reports, and the struct is
    versioned by size (a caller compiled against a smaller struct gets only the bytes it allocated)."""
    api, O, lib, synth = env
    import ctypes as C
    import torch
    from anofox_forecast_amd.device import DeviceBatch, pack_time_major
    n, T, m = 3000, 300, 7
    Y = synth.gen_series(synth.SEED_M5, 0, n, T, m, positive=True)
    b = DeviceBatch(n, T, lib.make_options("AutoETS", 7, seasonal_period=m), "cuda:0")
    y = torch.from_numpy(pack_time_major(Y, b.ld)).to("cuda:0")
    ln = torch.full((b.ld,), T, dtype=torch.int32, device="cuda:0"); ln[n:] = 0
    b.set_block(y, ln)
    b.run()
    ls = b.lane_stats()
    assert len(ls["by_spec"]) == 25 and all(v["wave_passes"] > 0 for v in ls["by_spec"].values())
    assert 0.3 < ls["lane_efficiency"] <= 1.0
    assert all(c["live_lane_passes"] <= 64 * c["wave_passes"] for c in ls["by_class"].values())
    st = b.stats()
    assert ls["live_lane_passes"] >= st["total_passes"] - 25 * n          # (every lane-pass of the one-lane drivers is a pass; the final passes are not counted here)
    small = (C.c_uint64 * 4)(0, 0, 0, 0xABCDEF)
    assert lib.load().anofox_hip_batch_lane_stats(b.handle, C.cast(small, C.POINTER(lib.AnofoxHipLaneStats)), 24)
    assert small[0] == 24 and small[3] == 0xABCDEF                       # 24 bytes written, the fourth word untouched
    b.close()


@pytest.mark.parametrize("scale", [1.0e-150, 1.0e-40, 1.0e30, 1.0e40, 1.0e150, 1.0e290])
def test_extreme_scales_hit_the_domain_rules_identically(env, scale):
    """Round 5 gave the general-class recursion two DOMAIN rules that oracle and kernels must apply identically: the step's reciprocal
    is defined for denominators in [2^-1000, 2^1000] (the kernels divide with the division's own sequence minus range scaling), and a
    multiplicative-error model's one-step forecast must lie in [2^-120, 2^120] (where the log-likelihood product may be renormalised
    every fourth step).  Ordinary data never meets either bound; these series do -- positive M5-shape counts scaled by 1e-150 ... 1e290:
    specs become inadmissible (their objective is +inf), the selection moves to what is left, and GPU and oracle must agree on
    every model name and every bit, whatever the scale did."""
    api, O, lib, synth = env
    Y = synth.gen_series(synth.SEED_M5, 7700, 48, 180, 7, positive=True) * scale
    series = [Y[s, : 180 - (s % 4) * 10] for s in range(48)]
    _compare(api, O, lib, series, "AutoETS", 10, seasonal_period=7)
    _compare(api, O, lib, series[:16], "AutoETS", 10, seasonal_period=1)
    for spec in ("MMdM", "AMdN", "MAdM", "AMM", "MNN"):
        _compare(api, O, lib, series[:24], "ETS", 7, ets_model=spec, seasonal_period=7)


def test_reference_kats_through_c_abi(env):
    """test/sql/ts_model_distinctness.test:116,141,180 via anofox_ts_forecast on the GPU."""
    api, O, lib, _ = env
    for model, kat in KATS.items():
        o = lib.make_options(model, 3, confidence_level=0.95, auto_detect=False, include_fitted=True, include_residuals=True)
        r = api.forecast_series(KAT_SERIES, o)
        assert r["ok"], (model, r)
        assert round(float(r["point"][0]), 6) == kat, (model, r["point"][0], kat)
    o = lib.make_options("AutoETS", 3, confidence_level=0.95, auto_detect=False)
    r = api.forecast_series(KAT_SERIES, o)
    assert r["ok"] and abs(r["point"][0] - 19.956521) / 19.956521 < 1e-5      # :164, north-star tolerance
    assert r["model_name"].startswith("AutoETS")


@pytest.mark.parametrize("model", ["Naive", "SeasonalNaive", "SMA", "RandomWalkDrift", "ARIMA", "SES", "SESOptimized", "Holt",
                                   "HoltWinters", "SeasonalES", "SeasonalESOptimized", "ETS", "AutoETS"])
def test_models_match_oracle(env, model):
    api, O, lib, synth = env
    Y = synth.gen_series(synth.SEED_M5, 0, 70, 120, 7)
    kw = {} if model in ("Naive", "SES", "SESOptimized", "Holt", "RandomWalkDrift", "ARIMA") else {"seasonal_period": 7}
    _compare(api, O, lib, list(Y), model, 14, **kw)


@pytest.mark.parametrize("spec", ["ANN", "AAN", "AAdN", "ANA", "AAA", "AAdA", "MNN", "MAN", "MAdN", "MMN", "MMdN", "AMN", "AMdN",
                                  "ANM", "AAM", "AAdM", "AMA", "AMdA", "AMM", "AMdM", "MNM", "MAM", "MAdM", "MMM", "MMdM"])
@pytest.mark.parametrize("period", [7, 5])
def test_ets_specs_match_oracle(env, spec, period):
    """Every valid spec, VGPR-ring period (7) and LDS-ring period (5), on strictly positive data."""
    api, O, lib, synth = env
    Y = synth.gen_series(synth.SEED_M5, 1000, 66, 90, period, positive=True)
    _compare(api, O, lib, list(Y), "ETS", 10, ets_model=spec, seasonal_period=period)


@pytest.mark.parametrize("period", [2, 12, 24, 64, 65, 168])
def test_autoets_other_periods(env, period):
    """The compile-time m = 12 ring (additive class only), the LDS ring at its smallest and at its largest (64), the first period
    whose ring lives in HBM scratch (65) and the hourly-data week (168, the example of the reference's own header,
    anofox_fcst_ffi.h:1075), with ragged lengths that end inside a streamed block: AutoETS over the whole grid against the oracle."""
    api, O, lib, synth = env
    T = 5 * period + 37
    Y = synth.gen_series(synth.SEED_M5, 3000 + period, 40, T, period, positive=True)
    series = [Y[s, : T - (s % 7) * 3] for s in range(40)]
    _compare(api, O, lib, series, "AutoETS", period + 3, seasonal_period=period)


@pytest.mark.parametrize("period,T", [(52, 260), (168, 1008), (365, 1100)])
def test_long_seasonal_periods(env, period, T):
    """Seasonal periods above what LDS holds (weekly-on-yearly 52 still fits; hourly 168 and yearly-on-daily 365 keep the
    seasonal ring, the decomposition window and the per-phase accumulators in HBM scratch): every seasonal model on the path
    against the oracle, bit for bit -- the reference takes any period (forecast.rs:528-537, 1347-1351)."""
    api, O, lib, synth = env
    Y = synth.gen_series(synth.SEED_STRESS, 100 + period, 24, T, period, positive=True)
    series = [Y[s, : T - (s % 5) * 7] for s in range(24)] + [Y[0, : 2 * period - 1], Y[1, : 2 * period]]     # one season short / exactly two
    for model, kw in (("HoltWinters", {}), ("SeasonalES", {}), ("SeasonalESOptimized", {}), ("ETS", dict(ets_model="AAA")),
                      ("ETS", dict(ets_model="MAdM")), ("ETS", dict(ets_model="AMdA")), ("ETS", {}), ("AutoETS", dict(model_pool="reduced"))):
        _compare(api, O, lib, series, model, 9, seasonal_period=period, **kw)
    _compare(api, O, lib, series[:6], "AutoETS", 5, seasonal_period=period)                 # the whole 30-spec grid


def test_period_above_the_cap_fails_loudly(env):
    """A period the kernels cannot hold (> 2,048) is a COMPUTATION_ERROR naming the cap for every seasonal model -- never a
    silent non-seasonal fit; models that ignore the period are unaffected."""
    api, O, lib, synth = env
    y = 50.0 + np.arange(6000.0) * 0.01 + np.sin(np.arange(6000.0) / 9.0)
    for model, kw in (("AutoETS", {}), ("HoltWinters", {}), ("SeasonalESOptimized", {}), ("ETS", dict(ets_model="AAA")), ("ETS", {})):
        r = api.forecast_series(y, lib.make_options(model, 4, seasonal_period=2500, **kw))
        ref = O.forecast(y, O.make_options(model, 4, seasonal_period=2500, **kw))
        assert not r["ok"] and r["code"] == lib.COMPUTATION_ERROR and "2048" in r["message"], (model, r)
        assert not ref["ok"] and ref["code"] == r["code"]
    _compare(api, O, lib, [y], "ETS", 4, seasonal_period=2500, ets_model="AAN")            # no seasonal component: the period is not used
    _compare(api, O, lib, [y], "SeasonalNaive", 4, seasonal_period=2500)


def test_autoets_full_grid_positive(env):
    api, O, lib, synth = env
    Y = synth.gen_series(synth.SEED_M5, 2000, 64, 100, 7, positive=True)
    _compare(api, O, lib, list(Y), "AutoETS", 28, seasonal_period=7)


def test_autoets_ragged_nulls_short_constant(env):
    """Ragged lengths, NULL runs (interpolated), too-short series, constant series (fallback chain)."""
    api, O, lib, synth = env
    rng = np.random.default_rng(7)
    Y = synth.gen_series(synth.SEED_M5, 3000, 40, 150, 7)
    series, valids = [], []
    for s in range(40):
        n = int(rng.integers(3, 150))
        y = Y[s, :n].copy()
        v = np.ones(n, bool)
        if s % 3 == 0:
            v[rng.integers(0, n, size=max(1, n // 10))] = False
        series.append(y)
        valids.append(v)
    series += [np.full(30, 42.0), np.array([1.0, 2.0]), np.array([]), np.full(8, 3.0), np.arange(9, dtype=float)]
    valids += [np.ones(len(x), bool) for x in series[40:]]
    _compare(api, O, lib, series, "AutoETS", 7, valids, seasonal_period=7)
    _compare(api, O, lib, series, "AutoETS", 7, valids)                       # auto-detected periods
    _compare(api, O, lib, series, "ETS", 7, valids, seasonal_period=7)        # default chain


def test_statement_level_errors(env):
    api, O, lib, _ = env
    y = [list(np.arange(20.0) + 1)]
    _compare(api, O, lib, y, "ETS", 3, ets_model="MAA", seasonal_period=7)
    _compare(api, O, lib, y, "ETS", 3, ets_model="XYZ")
    _compare(api, O, lib, y, "Naive", 3, seasonal_period=7)
    _compare(api, O, lib, y, "AutoETS", 3, model_pool="bogus")
    _compare(api, O, lib, y, "NoSuchModel", 3)


def test_error_isolation(env):
    """A bad series yields an error, neighbours still forecast (ts_forecast_error_isolation.test:14-60)."""
    api, O, lib, _ = env
    series = [np.arange(30.0), np.array([1.0, 2.0]), np.arange(40.0) * 2]
    got, berr = api.forecast_batch(series, lib.make_options("Holt", 5))
    assert berr["ok"] and got[0]["ok"] and got[2]["ok"] and not got[1]["ok"] and got[1]["code"] == 6


def test_ts_forecast_by_operator(env):
    api, O, lib, synth = env
    Y = synth.gen_series(synth.SEED_M5, 0, 5, 60, 7)
    grp = np.repeat(np.array(["a", "b", "c", "d", "e"], dtype=object), 60)
    ds = np.tile(np.arange("2024-01-01", "2024-03-01", dtype="datetime64[D]"), 5)
    perm = np.random.default_rng(1).permutation(len(grp))       # the operator sorts by date itself
    out = api.ts_forecast_by(grp[perm], ds[perm], Y.reshape(-1)[perm], "AutoETS", 14, "1d", {"seasonal_period": "7"})
    assert len(out["yhat"]) == 5 * 14
    assert list(out["forecast_step"][:14]) == list(range(1, 15))
    assert out["ds"][0] == np.datetime64("2024-03-01") and out["ds"][13] == np.datetime64("2024-03-14")
    assert np.all(out["yhat_lower"] <= out["yhat"]) and np.all(out["yhat"] <= out["yhat_upper"])
    oo = O.make_options("AutoETS", 14, seasonal_period=7)
    first = {}
    for g in grp[perm]:
        first.setdefault(g, len(first))
    for g, k in first.items():
        ref = O.forecast(Y["abcde".index(g)], oo)
        np.testing.assert_allclose(out["yhat"][k * 14:(k + 1) * 14], ref["point"], rtol=REL_TOL)
        assert out["model_name"][k * 14] == ref["model_name"]
    with pytest.raises(api.InvalidInputException, match="only valid when method='ETS'"):
        api.ts_forecast_by(grp, ds, Y.reshape(-1), "Naive", 3, "1d", {"model": "AAA"})
    with pytest.raises(api.InvalidInputException, match="Unknown parameter"):
        api.ts_forecast_by(grp, ds, Y.reshape(-1), "ETS", 3, "1d", {"methd": "AAA"})
    with pytest.raises(api.InvalidInputException, match="does not use seasonal_period"):
        api.ts_forecast_by(grp, ds, Y.reshape(-1), "Naive", 3, "1d", {"seasonal_period": "7"})


def test_inspect_and_explain_callers(env):
    """the fit state read back from the device (parameters, criteria, final states, one-step fitted
    values) equals the oracle's for AutoETS and for a fixed ETS spec; the caller mirrors shape it like the macros."""
    api, O, lib, synth = env
    Y = synth.gen_series(synth.SEED_M5, 900, 6, 84, 7, positive=True)
    res = api.inspect_batch(list(Y), lib.make_options("AutoETS", 5, seasonal_period=7))
    for y, r in zip(Y, res):
        ref = O.ets_inspect(y, 7)
        assert r["ok"] and ref is not None and r["model_code"] == 100 + ref["spec_id"]
        for k in ("alpha", "beta", "gamma", "phi", "aic", "aicc", "bic", "sse", "level", "trend"):
            assert _rel(np.array([r[k]]), np.array([ref[k]])) <= REL_TOL, (k, r[k], ref[k])
        assert _rel(r["fitted_values"], ref["fitted_values"]) <= REL_TOL
        if ref["gamma"] == ref["gamma"]:
            assert _rel(r["seasonal_states"], ref["seasonal_states"]) <= REL_TOL
    res = api.inspect_batch(list(Y), lib.make_options("ETS", 5, ets_model="AAdA", seasonal_period=7))
    for y, r in zip(Y, res):
        ref = O.ets_inspect(y, 7, spec_id=_spec_id_of(O, "AAdA"))
        assert _rel(r["fitted_values"], ref["fitted_values"]) <= REL_TOL and _rel(np.array([r["phi"]]), np.array([ref["phi"]])) <= REL_TOL
    # caller mirrors
    grp = np.repeat(np.array([f"g{i}" for i in range(6)], dtype=object), 84)
    ds = np.tile(np.arange(84), 6)
    insp = api.ts_forecast_inspect_by(grp, ds, Y.reshape(-1), "AutoETS", {"seasonal_period": 7})
    assert set(insp) == {f"g{i}" for i in range(6)}
    g0 = insp["g0"]
    assert g0["model_family"] == "Ets" and g0["seasonal_period"] == 7 and len(g0["fitted_values"]) == 84 and g0["order_p"] is None
    assert g0["spec"] and set(g0["spec"]) <= set("AMNd") and g0["aic"] is not None
    ar = api.ts_forecast_inspect_by(grp, ds, Y.reshape(-1), "AutoARIMA", {"seasonal_period": 7})["g0"]
    assert ar["model_family"] == "Arima" and ar["spec"] is None and ar["order_p"] is not None and ar["aic"] is not None
    with pytest.raises(api.InvalidInputException, match="does not implement Inspectable"):
        api.ts_forecast_inspect_by(grp, ds, Y.reshape(-1), "Naive", {})
    ex = api.ts_forecast_explain_by(grp, ds, Y.reshape(-1), "ETS", 12, {"model": "AAdA", "seasonal_period": 7})["g1"]
    assert ex["horizon"] == 12 and len(ex["level"]) == 12 and len(ex["trend"]) == 12 and len(ex["seasonal"]) == 12
    np.testing.assert_allclose(ex["level"] + ex["trend"] + ex["seasonal"], ex["yhat"], rtol=1e-12)
    exm = api.ts_forecast_explain_by(grp, ds, Y.reshape(-1), "ETS", 9, {"model": "MAM", "seasonal_period": 7})["g2"]
    np.testing.assert_allclose((exm["level"] + exm["trend"]) * exm["seasonal"], exm["yhat"], rtol=1e-12)
    with pytest.raises(api.InvalidInputException, match="does not implement Explainable"):
        api.ts_forecast_explain_by(grp, ds, Y.reshape(-1), "AutoETS", 3, {})


def _spec_id_of(O, notation):
    """spec id = error * 15 + trend index * 3 + season (oracle/forecast.c spec_from_id)."""
    err = {"A": 0, "M": 1}[notation[0]]
    trend = {"N": 0, "A": 1, "Ad": 2, "M": 3, "Md": 4}[notation[1:-1]]
    seas = {"N": 0, "A": 1, "M": 2}[notation[-1]]
    return err * 15 + trend * 3 + seas


def test_concurrent_single_series_calls(env):
    """Route A calls anofox_ts_forecast from every DuckDB worker thread at once (threading): concurrent calls
    from 8 host threads give the results of the serial calls."""
    import threading
    api, O, lib, synth = env
    Y = synth.gen_series(synth.SEED_M5, 500, 48, 70, 7, positive=True)
    models = [("AutoETS", dict(seasonal_period=7)), ("HoltWinters", dict(seasonal_period=7)), ("AutoARIMA", dict()), ("Naive", dict())]
    jobs = [(i, models[i % 4]) for i in range(48)]
    serial = [api.forecast_series(Y[i], lib.make_options(m, 9, **kw)) for i, (m, kw) in jobs]
    got = [None] * len(jobs)
    errs = []

    def work(tid):
        try:
            for j in range(tid, len(jobs), 8):
                i, (m, kw) = jobs[j]
                got[j] = api.forecast_series(Y[i], lib.make_options(m, 9, **kw))
        except Exception as e:          # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for a, b in zip(serial, got):
        assert a["ok"] and b["ok"] and a["model_name"] == b["model_name"]
        assert np.array_equal(a["point"], b["point"]) and np.array_equal(a["upper"], b["upper"])


@pytest.mark.parametrize("model,period", [("AutoETS", 7), ("HoltWinters", 7), ("AutoARIMA", 7), ("SES", 0)])
def test_concurrent_c_workers_coalesce_bit_identically(env, model, period):
    """tests/c_abi/concurrent.c: 8 pthreads calling anofox_ts_forecast back to back, the way the reference's scalar binding does from
    every DuckDB worker (ts_forecast_scalar.cpp:298-523).  The coalescing window of the library puts concurrent calls with an equal
    option block into one multi-series batch: every result equals its serial call bit for bit, the too-short series among them fail
    alone.  (Timings: tools/time_single_call.py and the harness itself, profiles/r03_single_call_latency.txt.)"""
    import subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "anofox-forecast_amd")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "concurrent")
        subprocess.check_call(["gcc", "-std=c11", "-O2", "-pthread", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                               os.path.join(root, "tests", "c_abi", "concurrent.c"), "-L", pkg, "-lanofox_fcst_hip", "-Wl,-rpath," + pkg, "-o", exe])
        out = subprocess.run([exe, model, str(period), "8", "8", "160"], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, GPU_MAX_HW_QUEUES="16"))
    assert out.returncode == 0 and out.stdout.startswith("OK "), (out.stdout, out.stderr[-1500:])
    serial_us, threaded_us, bad, failed = out.stdout.split()[1:5]
    assert int(bad) == 0 and int(failed) == 4, out.stdout          # calls 12, 25, 38, 51 are two observations long


def test_ts_forecast_agg_caller(env):
    """the aggregate caller -- NULL rows skipped, (timestamp, value) order, median-step forecast
    timestamps, fitted values on, a failing group reports its message instead of aborting."""
    api, O, lib, synth = env
    Y = synth.gen_series(synth.SEED_M5, 70, 3, 40, 7, positive=True)
    grp, ds, val = [], [], []
    for g in range(3):
        for t in range(40):
            grp.append(f"s{g}"); ds.append(1000 + 5 * t); val.append(Y[g, t] if (t + g) % 13 else None)   # a few NULL values
    grp += ["tiny", "tiny"]; ds += [1, 2]; val += [1.0, 2.0]                                           # too short: error row
    perm = np.random.default_rng(8).permutation(len(grp))
    out = api.ts_forecast_agg(np.array(grp, dtype=object)[perm], np.array(ds, dtype=np.int64)[perm],
                              np.array(val, dtype=object)[perm], "AutoETS", 6, {})
    assert set(out) == {"s0", "s1", "s2", "tiny"}
    assert out["tiny"]["point_forecast"] == [] and "Insufficient data" in out["tiny"]["error_message"]
    oo = O.make_options("AutoETS", 6, seasonal_period=0, confidence_level=0.90, auto_detect=False, include_fitted=True)
    for g in range(3):
        keep = [t for t in range(40) if (t + g) % 13]
        ref = O.forecast(Y[g, keep], oo)
        r = out[f"s{g}"]
        assert r["model_name"] == ref["model_name"] and r["forecast_step"] == [1, 2, 3, 4, 5, 6]
        np.testing.assert_allclose(r["point_forecast"], ref["point"], rtol=REL_TOL)
        np.testing.assert_allclose(r["lower_90"], ref["lower"], rtol=REL_TOL)
        np.testing.assert_allclose(r["insample_fitted"], ref["fitted"], rtol=REL_TOL)
        last = 1000 + 5 * keep[-1]
        assert r["forecast_timestamp"] == [last + 5 * (j + 1) for j in range(6)]                      # median step = 5


def test_columnar_ingest_feeds_the_batch(env):
    """rows appended chunk by chunk through the C-ABI ingest (block 4) give the same forecasts as
    the operator mirror that groups and sorts in Python -- NULL targets, shuffled rows, ragged groups included."""
    api, O, lib, synth = env
    Y = synth.gen_series(synth.SEED_M5, 40, 6, 80, 7, positive=True)
    rows = [(100 + g, t, Y[g, t]) for g in range(6) for t in range(80 - 5 * g)]
    perm = np.random.default_rng(4).permutation(len(rows))
    gk, dt, val = (np.array(c)[perm] for c in zip(*rows))
    vok = np.random.default_rng(5).random(len(rows)) > 0.03
    ing = api.Ingest()
    for lo in range(0, len(rows), 97):
        sl = slice(lo, lo + 97)
        ing.append(gk[sl].astype(np.int64), dt[sl].astype(np.int64), val[sl], None, vok[sl])
    ng, tmax = ing.finish()
    assert ng == 6 and tmax == 80
    got = ing.forecast(lib.make_options("AutoETS", 12, seasonal_period=7))
    ref = api.ts_forecast_by([f"k{int(k)}" for k in gk], dt.astype(np.int64), np.ma.array(val, mask=~vok), "AutoETS", 12, 1,
                             {"seasonal_period": 7})
    keys = list(dict.fromkeys(f"k{int(k)}" for k in gk))
    assert [f"k{int(k)}" for k in ing.group_keys()] == keys
    for i, k in enumerate(keys):
        assert got[i]["ok"]
        sel = [j for j, g in enumerate(ref["id"]) if g == k]
        np.testing.assert_array_equal(got[i]["point"], ref["yhat"][sel])
        assert got[i]["model_name"] == ref["model_name"][sel[0]]
    ing.close()


def _cv_folds(series_id, n, n_folds, horizon):
    """Expanding-window folds the way ts_cv_folds_by lays them out (ts_cv_forecast.test:20-37: 24 obs, 3 folds, h = 4
    -> train 12 / 16 / 20, test 4 each)."""
    rows = []
    for k in range(1, n_folds + 1):
        end = n - (n_folds - k + 1) * horizon
        rows += [(k, "train", series_id, t) for t in range(end)]
        rows += [(k, "test", series_id, t) for t in range(end, end + horizon)]
    return rows


def test_ts_cv_forecast_by_operator(env):
    """every (fold, group) pair is an independent training series whose horizon is its number
    of test rows; one batch call with per-series horizons; forecasts matched to the test rows by position."""
    api, O, lib, synth = env
    # the reference's own case (ts_cv_forecast.test): y = 10 + i, i = 1..24, Naive -> last training value
    rows = _cv_folds("A", 24, 3, 4)
    fold, split, grp, t = (np.array(c, dtype=object) for c in zip(*rows))
    ds = np.array([int(x) + 1 for x in t], dtype=np.int32)
    y = np.array([10.0 + d for d in ds])
    out = api.ts_cv_forecast_by(fold, split, grp, ds, y, "Naive", {}, group_name="series_id", date_name="ds")
    assert sorted(out.keys()) == sorted(["ds", "fold_id", "model_name", "series_id", "split", "y", "yhat", "yhat_lower", "yhat_upper"])
    assert len(out["yhat"]) == 12 and list(out["fold_id"]) == [1] * 4 + [2] * 4 + [3] * 4 and set(out["split"]) == {"test"}
    assert out["ds"].dtype == np.int32 and list(out["ds"][:4]) == [13, 14, 15, 16]
    np.testing.assert_array_equal(out["yhat"], np.repeat([22.0, 26.0, 30.0], 4))
    np.testing.assert_array_equal(out["y"], 10.0 + out["ds"])
    # AutoETS over several groups, shuffled rows, ragged horizons; against the oracle pair by pair
    Y = synth.gen_series(synth.SEED_M5, 300, 4, 90, 7, positive=True)
    rows = []
    for g in range(4):
        rows += [(k, sp, f"g{g}", tt) for (k, sp, _, tt) in _cv_folds(g, 90, 3, 5 + g)]
    perm = np.random.default_rng(3).permutation(len(rows))
    fold, split, grp, t = (np.array(c, dtype=object)[perm] for c in zip(*rows))
    ds = np.datetime64("2024-01-01") + np.array([int(x) for x in t]).astype("timedelta64[D]")
    yv = np.array([Y[int(g[1:]), int(x)] for g, x in zip(grp, t)])
    out = api.ts_cv_forecast_by(fold, split, grp, ds, yv, "AutoETS", {"seasonal_period": 7}, group_name="id", date_name="ds")
    assert len(out["yhat"]) == 3 * (5 + 6 + 7 + 8)
    pos = 0
    for k in (1, 2, 3):
        for g in range(4):
            h = 5 + g
            end = 90 - (3 - k + 1) * h
            ref = O.forecast(Y[g, :end], O.make_options("AutoETS", h, seasonal_period=7, confidence_level=0.90))
            sl = slice(pos, pos + h)
            assert list(out["fold_id"][sl]) == [k] * h and list(out["id"][sl]) == [f"g{g}"] * h
            np.testing.assert_allclose(out["yhat"][sl], ref["point"], rtol=REL_TOL)
            np.testing.assert_allclose(out["yhat_lower"][sl], ref["lower"], rtol=REL_TOL)
            np.testing.assert_array_equal(out["y"][sl], Y[g, end:end + h])
            assert out["ds"][pos] == np.datetime64("2024-01-01") + np.timedelta64(end, "D")
            assert out["model_name"][pos] == ref["model_name"]
            pos += h
    with pytest.raises(api.InvalidInputException, match="Unknown model"):
        api.ts_cv_forecast_by(fold, split, grp, ds, yv, "NoSuchModel", {})


@pytest.mark.parametrize("seq_rounds", ["0", "2", "6"])
@pytest.mark.parametrize("gather", ["0", "1"])
def test_schedule_variants_are_bit_identical(env, monkeypatch, seq_rounds, gather):
    """The sequential and the speculative Nelder-Mead drivers, with or without the column gather between
    rounds, must walk the same trajectory: every schedule reproduces the oracle bit for bit."""
    api, O, lib, synth = env
    monkeypatch.setenv("ANOFOX_HIP_TUNE", f"seq_rounds={seq_rounds};gather={gather}")
    Y = synth.gen_series(synth.SEED_M5, 5000, 130, 160, 7, positive=True)
    series = [Y[s, : 160 - (s % 5) * 9] for s in range(130)]          # ragged lengths
    _compare(api, O, lib, series, "AutoETS", 12, seasonal_period=7)
    _compare(api, O, lib, series, "ETS", 12, ets_model="MAdM", seasonal_period=7)


def _device_run(lib, Y, lens, model, h, tune, monkeypatch, **kw):
    """One device-resident run (block 3 of the header) under an ANOFOX_HIP_TUNE setting: forecasts, model codes, status, run statistics."""
    import torch
    from anofox_forecast_amd.device import DeviceBatch
    monkeypatch.setenv("ANOFOX_HIP_TUNE", tune)
    n, T = Y.shape
    batch = DeviceBatch(n, T, lib.make_options(model, h, **kw), "cuda:0")
    ld = batch.ld
    block = torch.zeros((T, ld), dtype=torch.float64, device="cuda:0")
    block[:, :n] = torch.from_numpy(np.ascontiguousarray(Y.T)).to("cuda:0")
    batch.set_block(block, torch.from_numpy(np.asarray(lens, dtype=np.int32)).to("cuda:0"))
    batch.run()
    torch.cuda.synchronize()
    r = batch.results()
    out = {k: r[k].cpu().numpy().copy() for k in ("yhat", "lower", "upper", "model_code", "status")}
    out["stats"] = batch.stats()
    return out


def test_compact_storage_is_bit_identical(env, monkeypatch):
    """Round 6: a batch whose every observation survives the round trip through float / uint16_t exactly (counts: the M5 shape) is
    streamed from a 4- / 2-byte copy of its block (ets_device.hpp YT_*, kernels.hip compact_block_kernel); the arithmetic stays
    fp64 on the same numbers.  Whatever the storage -- forced off, float at most, narrowest -- forecasts, intervals, selected models
    and pass counts are the same bits, for every period variant (none, 7, 12, LDS ring, HBM ring; a merged batch of detected periods:
    test_auto_detected_periods_merge_into_one_batch runs on counts), for a
    mixed batch (the strictly positive columns are gathered in the storage type), ragged lengths, and a one-spec fit whose final
    pass carries the intervals' sd.  ONE observation that does not survive (0.1; 70,000; -0.0 for the integer type) moves the whole
    batch to the next wider type, and the results do not move."""
    api, O, lib, synth = env
    rng = np.random.default_rng(66)

    def same(a, b, what):
        for k in ("yhat", "lower", "upper"):
            assert np.array_equal(a[k], b[k], equal_nan=True), (what, k)
        assert np.array_equal(a["model_code"], b["model_code"]) and np.array_equal(a["status"], b["status"]), what
        for k in ("total_passes", "total_evals", "total_iters"):
            assert a["stats"][k] == b["stats"][k], (what, k)

    cases = []
    Yp = synth.gen_series(synth.SEED_M5, 6600, 150, 210, 7, positive=True)
    Yi = synth.gen_series(synth.SEED_M5, 6800, 150, 210, 7)
    ragged = [210 - (s % 7) * 11 for s in range(150)]
    cases.append(("AutoETS m7 positive ragged", Yp, ragged, "AutoETS", dict(seasonal_period=7)))
    cases.append(("AutoETS m7 mixed", np.concatenate([Yp[:60], Yi[:90]]), ragged, "AutoETS", dict(seasonal_period=7)))
    cases.append(("AutoETS none", Yp, ragged, "AutoETS", dict(seasonal_period=1)))
    cases.append(("AutoETS m12", synth.gen_series(synth.SEED_M5, 6900, 80, 200, 12, positive=True), [200] * 80, "AutoETS", dict(seasonal_period=12)))
    cases.append(("AutoETS m24 (LDS ring)", synth.gen_series(synth.SEED_M5, 7000, 70, 260, 24, positive=True), [260] * 70, "AutoETS", dict(seasonal_period=24)))
    cases.append(("ETS AAA m70 (HBM ring)", synth.gen_series(synth.SEED_STRESS, 7100, 40, 300, 70, positive=True), [300] * 40, "ETS", dict(ets_model="AAA", seasonal_period=70)))
    cases.append(("ETS MAdM one spec", Yp, ragged, "ETS", dict(ets_model="MAdM", seasonal_period=7)))
    for what, Y, lens, model, kw in cases:
        base = _device_run(lib, Y, lens, model, 9, "compact=0", monkeypatch, **kw)
        f32 = _device_run(lib, Y, lens, model, 9, "compact=1", monkeypatch, **kw)
        u16 = _device_run(lib, Y, lens, model, 9, "compact=2", monkeypatch, **kw)
        # (the uint16 kernels exist without a period and for the weekly ring in registers; any other period streams the float copy)
        narrowest = 2 if kw.get("seasonal_period", 1) in (1, 7) else 1
        assert (base["stats"]["y_storage"], f32["stats"]["y_storage"], u16["stats"]["y_storage"]) == (0, 1, narrowest), what
        same(f32, base, what + " float")
        same(u16, base, what + " uint16")
        # and the fp64 run is the oracle's (so all three are)
        oo = O.make_options(model, 9, **kw)
        for s in (0, len(Y) // 2, len(Y) - 1):
            ref = O.forecast(Y[s, :lens[s]], oo)
            assert ref["ok"] and _rel(base["yhat"][s], ref["point"]) <= REL_TOL, (what, s)
    # observations that do not survive a type: the next wider one is used, the results stay
    Y = Yp.copy()
    lens = [210] * 150
    want = _device_run(lib, Y, lens, "AutoETS", 9, "compact=0", monkeypatch, seasonal_period=7)
    for value, storage in ((70000.0, 1), (2.5, 1), (16777217.0, 0), (0.1, 0)):
        Z = Y.copy()
        Z[77, 100] = value
        got = _device_run(lib, Z, lens, "AutoETS", 9, "compact=2", monkeypatch, seasonal_period=7)
        ref = _device_run(lib, Z, lens, "AutoETS", 9, "compact=0", monkeypatch, seasonal_period=7)
        assert got["stats"]["y_storage"] == storage, (value, got["stats"]["y_storage"])
        same(got, ref, f"misfit {value}")
        assert np.array_equal(np.delete(got["yhat"], 77, 0), np.delete(want["yhat"], 77, 0))      # the other series never notice
    Z = Yi.copy()
    Z[5, 50] = -0.0                                  # equal to 0 as a number, a different bit pattern: not an integer cell
    got = _device_run(lib, Z, lens, "AutoETS", 9, "compact=2", monkeypatch, seasonal_period=7)
    assert got["stats"]["y_storage"] == 1
    same(got, _device_run(lib, Z, lens, "AutoETS", 9, "compact=0", monkeypatch, seasonal_period=7), "negative zero")
    # a misfit beyond a series' length is not an observation
    Z = Yp.copy()
    Z[3, 205] = 0.3
    lens2 = list(ragged)
    assert lens2[3] <= 205
    got = _device_run(lib, Z, lens2, "AutoETS", 9, "compact=2", monkeypatch, seasonal_period=7)
    assert got["stats"]["y_storage"] == 2
    # automatic: a handful of short series stays on the fp64 block, the M5-size block does not (test_full_size_m5_properties)
    assert _device_run(lib, Yp[:20], [210] * 20, "AutoETS", 9, "", monkeypatch, seasonal_period=7)["stats"]["y_storage"] == 0
    assert _device_run(lib, np.tile(Yp, (3, 1)), [210] * 450, "AutoETS", 9, "", monkeypatch, seasonal_period=7)["stats"]["y_storage"] == 2
    # detected periods (params := MAP{}): the merged batches of several periods stream the float copy through the per-lane-period kernels
    series = []
    for p, reps in ((5, 2), (7, 3), (12, 2), (24, 2), (30, 1), (52, 2), (70, 2), (130, 1)):
        for r in range(reps):
            T = int(max(6 * p, 90) + rng.integers(0, 40))
            t = np.arange(T)
            series.append(np.rint(50.0 + 0.02 * t + (8.0 + r) * np.sin(2 * np.pi * t / p) + 3.0 * np.cos(4 * np.pi * t / p) + rng.normal(0, 0.6, T)))
    for model in ("AutoETS", "ETS"):
        opts = lib.make_options(model, 9)
        monkeypatch.setenv("ANOFOX_HIP_TUNE", "compact=0")
        want_rows, berr = api.forecast_batch(series, opts)
        assert berr["ok"]
        monkeypatch.setenv("ANOFOX_HIP_TUNE", "compact=2")
        got_rows, berr = api.forecast_batch(series, opts)
        assert berr["ok"]
        for a_row, b_row in zip(got_rows, want_rows):
            assert a_row["ok"] == b_row["ok"] and a_row.get("model_name") == b_row.get("model_name")
            if a_row["ok"]:
                for k in ("point", "lower", "upper"):
                    assert np.array_equal(a_row[k], b_row[k], equal_nan=True), (model, k)
    _compare(api, O, lib, series, "AutoETS", 9)          # (compact = 2 still set) ... and they are the oracle's
    # given smoothing parameters: one pass, no copy; the host-buffer entry takes the same route as the resident block
    monkeypatch.setenv("ANOFOX_HIP_TUNE", "compact=2")
    _compare(api, O, lib, [Yp[s, :ragged[s]] for s in range(150)], "AutoETS", 9, seasonal_period=7)
    _compare(api, O, lib, [Yi[s, :ragged[s]] for s in range(150)], "AutoETS", 9, seasonal_period=7,
             valids=None)


@pytest.mark.parametrize("k4", ["-1", "0", "1"])
def test_four_candidates_per_lane_is_bit_identical(env, monkeypatch, k4):
    """ANOFOX_HIP_TUNE k4: the additive-class specs run one lane per problem with all four trial points of an iteration evaluated by
    that lane in ONE pass (one y load feeds four recursions: the memory-bound form) instead of the sequential driver.  Same
    iterates, same forecasts as the oracle: on intermittent counts (only the additive specs are admissible: the automatic
    choice), on strictly positive data beside the general-class specs (forced), non-seasonal, ragged, and with enough series
    that the sequential-class rounds really run (SEQ_ROUNDS forced as well)."""
    api, O, lib, synth = env
    monkeypatch.setenv("ANOFOX_HIP_TUNE", f"k4={k4}")
    Yi = synth.gen_series(synth.SEED_M5, 5800, 200, 170, 7)
    series = [Yi[s, : 170 - (s % 6) * 8] for s in range(200)]
    _compare(api, O, lib, series, "AutoETS", 9, seasonal_period=7)
    _compare(api, O, lib, series, "AutoETS", 9, seasonal_period=1)
    _compare(api, O, lib, series, "ETS", 9, ets_model="AAdA", seasonal_period=7)
    Yp = synth.gen_series(synth.SEED_M5, 5900, 90, 150, 7, positive=True)
    _compare(api, O, lib, list(Yp), "AutoETS", 9, seasonal_period=7)
    monkeypatch.setenv("ANOFOX_HIP_TUNE", f"k4={k4};seq_rounds=3")
    _compare(api, O, lib, series, "AutoETS", 9, seasonal_period=7)
    _compare(api, O, lib, series[:50], "AutoETS", 9, seasonal_period=12)        # (no K4 kernels for m = 12: the sequential driver)


@pytest.mark.parametrize("tune", ["k4=1;k4_top=1;k4_top_below=60", "k4=1;k4_top=2;k4_top_below=100000", "k4=1;k4_top=6;k4_top_below=150;spec_below=40",
                                  "k4=1;k4_top_below=0"])
def test_early_four_lane_switch_of_the_top_specs_is_bit_identical(env, monkeypatch, tune):
    """ANOFOX_HIP_TUNE k4_top / k4_top_below: the most expensive additive-class spec(s) leave the four-points-per-lane driver for four
    lanes per problem at a higher count of live problems than the others (the chain that ends the step gets its parallelism
    earlier).  Whatever the threshold -- below the batch size (the switch happens between rounds, by the device-side count), above it
    (four lanes from the first round), for one, two or all specs -- the iterates are the oracle's."""
    api, O, lib, synth = env
    monkeypatch.setenv("ANOFOX_HIP_TUNE", tune)
    Yi = synth.gen_series(synth.SEED_M5, 6100, 220, 180, 7)
    series = [Yi[s, : 180 - (s % 6) * 8] for s in range(220)]
    _compare(api, O, lib, series, "AutoETS", 9, seasonal_period=7)
    _compare(api, O, lib, series[:120], "AutoETS", 9, seasonal_period=1)


@pytest.mark.parametrize("below", ["20", "100000"])
def test_two_level_speculation_is_bit_identical(env, monkeypatch, below):
    """ANOFOX_HIP_TUNE spec2_below: the last problems of a spec run one per wave, lanes 4..63 evaluating the next iteration's
    trial points under all 3 D + 3 outcomes of the current one (two Nelder-Mead iterations per pass).  Same iterates, same
    evaluation counts, same forecasts as the oracle -- for every parameter dimension (1..4: the AutoETS grid), ragged
    lengths, a mixed batch, a run-time period (ring in LDS) and a long one (ring in HBM scratch)."""
    api, O, lib, synth = env
    monkeypatch.setenv("ANOFOX_HIP_TUNE", f"spec2_below={below}")
    Y = synth.gen_series(synth.SEED_M5, 5000, 130, 160, 7, positive=True)
    series = [Y[s, : 160 - (s % 5) * 9] for s in range(130)]          # ragged lengths
    _compare(api, O, lib, series, "AutoETS", 12, seasonal_period=7)
    _compare(api, O, lib, series, "ETS", 12, ets_model="MAdM", seasonal_period=7)
    Yi = synth.gen_series(synth.SEED_M5, 5400, 130, 160, 7)
    mixed = [Yi[s] if s % 3 else Y[s] for s in range(130)] + [np.full(30, 4.0), np.arange(5.0), np.zeros(40)]
    _compare(api, O, lib, mixed, "AutoETS", 12, seasonal_period=7)
    _compare(api, O, lib, series[:40], "AutoETS", 6, seasonal_period=5)
    Yl = synth.gen_series(synth.SEED_M5, 5600, 24, 300, 70, positive=True)
    _compare(api, O, lib, list(Yl), "ETS", 5, ets_model="AAA", seasonal_period=70)


def test_two_level_speculation_counts(env, monkeypatch):
    """Same evaluation and iteration totals as the schedule without it (the speculative lanes are not evaluations of the
    method), fewer passes, bit-identical forecasts."""
    import torch
    api, O, lib, synth = env
    from anofox_forecast_amd.device import DeviceBatch, pack_time_major
    n, T, h = 300, 140, 7
    Y = synth.gen_series(synth.SEED_M5, 7300, n, T, 7, positive=True)
    opts = lib.make_options("AutoETS", h, seasonal_period=7)
    runs = []
    for below in ("0", "100000"):                              # off / every problem after the first round
        monkeypatch.setenv("ANOFOX_HIP_TUNE", f"spec2_below={below}")
        b = DeviceBatch(n, T, opts, "cuda:0")
        y = torch.from_numpy(pack_time_major(Y, b.ld)).cuda()
        ln = torch.full((b.ld,), T, dtype=torch.int32, device="cuda")
        ln[n:] = 0
        b.set_block(y, ln)
        b.run()
        torch.cuda.synchronize()
        out = b.results()
        runs.append((b.stats(), {k: out[k].cpu().numpy().copy() for k in ("yhat", "lower", "upper", "model_code")}))
        b.close()
    (s0, r0), (s1, r1) = runs
    assert s0["total_evals"] == s1["total_evals"]
    assert s1["total_passes"] < s0["total_passes"]
    for k in r0:
        np.testing.assert_array_equal(r0[k], r1[k])


@pytest.mark.parametrize("model", ["AutoETS", "HoltWinters", "SeasonalESOptimized", "ETS:AAA", "ETS:MAdM"])
def test_auto_detected_periods_merge_into_one_batch(env, monkeypatch, model):
    """params := MAP{} (no seasonal_period): every series gets its own detected period.  The batch entry runs the series of all
    periods of a ring class as ONE batch whose 64-column blocks each have their own period (prep, fit, final pass and the
    fallback chain read the period per block) instead of one tiny batch per period -- same forecasts as the oracle, which
    detects and fits series by series; the split path (ANOFOX_HIP_TUNE merge_periods=0) gives the same results."""
    api, O, lib, synth = env
    rng = np.random.default_rng(17)
    series = []
    for p, reps in ((5, 3), (7, 5), (12, 2), (24, 2), (30, 1), (52, 2), (70, 2), (130, 1), (300, 1)):
        for r in range(reps):
            T = int(max(6 * p, 90) + rng.integers(0, 40))
            t = np.arange(T)
            y = 50.0 + 0.02 * t + (8.0 + r) * np.sin(2 * np.pi * t / p) + 3.0 * np.cos(4 * np.pi * t / p) + rng.normal(0, 0.6, T)
            series.append(y)
    series += [np.full(40, 3.0), np.arange(30.0), rng.normal(10, 1, 50), np.array([1.0, 2.0]), np.array([])]
    kw = {}
    if ":" in model: model, kw = model.split(":")[0], {"ets_model": model.split(":")[1]}
    _compare(api, O, lib, series, model, 9, **kw)
    monkeypatch.setenv("ANOFOX_HIP_TUNE", "merge_periods=0")
    _compare(api, O, lib, series, model, 9, **kw)


@pytest.mark.parametrize("envset", [{"ANOFOX_HIP_CACHE_GB": "0", "ANOFOX_HIP_PINNED_CACHE_GB": "0"}, {"ANOFOX_HIP_TUNE": "prio_streams=0;budgets=32,32,64,128,256,1024"},
                                    {"ANOFOX_HIP_TUNE": "prio_streams=3", "GPU_MAX_HW_QUEUES": "8"}])
def test_process_wide_switches(envset):
    """The allocation caches and the stream priorities are decided once per process (a round schedule rides along): a fresh process
    with each of them switched off (or sized differently) reproduces the oracle like the defaults do -- three batches in a
    row, so that blocks and stream sets are handed back and taken again."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        from anofox_forecast_amd import api, lib, synth
        from oracle import oracle as O
        Y = synth.gen_series(synth.SEED_M5, 8100, 40, 120, 7, positive=True)
        for rep, (model, kw) in enumerate((("AutoETS", {"seasonal_period": 7}), ("HoltWinters", {"seasonal_period": 7}), ("AutoETS", {"seasonal_period": 7}))):
            series = [Y[s, : 120 - (s %% 4) * 7] for s in range(40 - rep)]
            got, berr = api.forecast_batch(series, lib.make_options(model, 6, **kw))
            assert berr["ok"], berr
            oo = O.make_options(model, 6, **kw)
            for s, y in enumerate(series):
                ref = O.forecast(y, oo)
                assert got[s]["ok"] and ref["ok"] and got[s]["model_name"] == ref["model_name"], (model, s)
                assert np.array_equal(np.asarray(got[s]["point"]), np.asarray(ref["point"])), (model, s)
        print("OK")
    """) % root
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **envset), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stderr[-2000:]


def test_batches_of_empty_series(env):
    """A batch whose every series is empty (t_max = 0) reports InsufficientData per series for every model -- no kernel is
    launched with an empty grid (found by tools/fuzz_parity.py after the SES / Holt family moved to the round kernels)."""
    api, O, lib, synth = env
    for model, kw in (("AutoETS", {"seasonal_period": 3}), ("HoltWinters", {"seasonal_period": 7}), ("Holt", {}), ("SESOptimized", {}),
                      ("SeasonalESOptimized", {"seasonal_period": 4}), ("ETS", {"ets_model": "AAA", "seasonal_period": 7}), ("AutoARIMA", {}), ("Naive", {})):
        for k in (1, 3):
            _compare(api, O, lib, [np.array([])] * k, model, 5, **kw)


def test_merged_periods_with_nulls_horizons_and_fitted(env):
    """The merged batch keeps the per-series extras: NULLs are interpolated before detection, per-series horizons truncate
    (the cross-validation caller), fitted values and residuals use the series' own period."""
    api, O, lib, synth = env
    rng = np.random.default_rng(23)
    series, valids, hz = [], [], []
    for p, reps in ((4, 2), (7, 3), (9, 2), (26, 2), (66, 2), (90, 1)):
        for r in range(reps):
            T = int(max(7 * p, 100) + rng.integers(0, 30))
            t = np.arange(T)
            y = 40.0 + (6.0 + r) * np.sin(2 * np.pi * t / p) + rng.normal(0, 0.5, T)
            v = rng.random(T) > 0.04
            v[0] = v[-1] = True
            series.append(np.where(v, y, 0.0)); valids.append(v); hz.append(int(rng.integers(0, 11)))
    opts = lib.make_options("AutoETS", 10, include_fitted=True, include_residuals=True)
    oo = O.make_options("AutoETS", 10, include_fitted=True, include_residuals=True)
    got, berr = api.forecast_batch(series, opts, valids, hz)
    assert berr["ok"], berr
    for s, (y, v) in enumerate(zip(series, valids)):
        ref = O.forecast(y, oo, v)
        assert got[s]["ok"] and ref["ok"] and got[s]["model_name"] == ref["model_name"], s
        assert len(got[s]["point"]) == hz[s]
        assert np.array_equal(np.asarray(got[s]["point"]), np.asarray(ref["point"])[: hz[s]]), s
        assert _rel(np.asarray(got[s]["fitted"]), np.asarray(ref["fitted"])) <= REL_TOL and _rel(np.asarray(got[s]["residuals"]), np.asarray(ref["residuals"])) <= REL_TOL, s


def test_merged_periods_many_long_periods(env):
    """Seventy-two series with as many different long periods (an auto-detected batch of M5-like series has hundreds of rare ones): the
    HBM-ring class is cut into several merged batches so that its ring scratch stays bounded (one 64-column block per period
    x the largest period x 512 B per seasonal spec had asked for 100 GB on 8,192 series and failed the whole call)."""
    api, O, lib, synth = env
    rng = np.random.default_rng(29)
    series = []
    for k in range(72):
        p = 556 + 2 * k
        T = 2 * p + 220 + int(rng.integers(0, 30))
        t = np.arange(T)
        series.append(30.0 + 9.0 * np.sin(2 * np.pi * t / p) + rng.normal(0, 0.4, T))
    _compare(api, O, lib, series, "ETS", 6, ets_model="AAA")


def test_device_resident_batch_and_stats(env):
    """Block already in HBM (torch tensor) -> anofox_hip_batch_* -> device results; counters are consistent."""
    import torch
    api, O, lib, synth = env
    from anofox_forecast_amd.device import DeviceBatch, pack_time_major
    n, T, h = 200, 96, 8
    Y = synth.gen_series(synth.SEED_M5, 7000, n, T, 7)
    opts = lib.make_options("AutoETS", h, seasonal_period=7)
    b = DeviceBatch(n, T, opts, "cuda:0")
    y = torch.from_numpy(pack_time_major(Y, b.ld)).cuda()
    ln = torch.full((b.ld,), T, dtype=torch.int32, device="cuda")
    ln[n:] = 0
    b.set_block(y, ln)
    b.run()
    torch.cuda.synchronize()
    out = b.results()
    st = b.stats()
    oo = O.make_options("AutoETS", h, seasonal_period=7)
    yhat = out["yhat"].cpu().numpy()
    codes = out["model_code"].cpu().numpy()
    for s in range(0, n, 17):
        ref = O.forecast(Y[s], oo)
        assert _rel(yhat[s], ref["point"]) <= REL_TOL
        assert b.model_name(int(codes[s])) == ref["model_name"]
    assert st["n_series"] == n and st["total_passes"] >= st["total_evals"] / 4 and st["fit_kernel_ms"] > 0
    assert st["algorithmic_bytes"] == 8 * T * st["total_passes"] + 24 * h * n
    b.close()


@pytest.mark.parametrize("cols", ["64", "192", "1024"])
def test_partial_gather_blocks_give_the_same_results(env, monkeypatch, cols):
    """A batch whose per-spec gather blocks cannot hold every column (1M series x 1,024: 25 blocks of 8.2 GB) gets blocks of fewer
    columns: the rounds index y by series until that few problems still run, then switch to the dense copy -- decided on the device
    from the running count, by the gather kernel and the round kernel alike.  Forced here on a small batch (ANOFOX_HIP_TUNE gather_cols):
    AutoETS, a fitted spec and Holt-Winters (the classic family borrows the first spec's block) agree with the oracle bit for bit."""
    api, O, lib, synth = env
    monkeypatch.setenv("ANOFOX_HIP_TUNE", f"gather_cols={cols}")
    Y = synth.gen_series(synth.SEED_M5, 9100, 300, 160, 7, positive=True)
    series = [Y[s, : 160 - (s % 5) * 9] for s in range(300)]
    _compare(api, O, lib, series, "AutoETS", 7, seasonal_period=7)
    _compare(api, O, lib, series, "ETS", 7, ets_model="MAdM", seasonal_period=7)
    _compare(api, O, lib, series, "HoltWinters", 7, seasonal_period=7)


@pytest.mark.parametrize("T", [700, 9000])
def test_period_detection_kernel_matches_oracle(env, T):
    """auto_detect_seasonality on a RESIDENT block: detect_period_kernel (one workgroup per series, the centred series in LDS, in an
    HBM scratch above 8,192 observations -- T = 9000 --, one lag per lane, every sum in the scalar loop's order) picks the lag
    seasonality.rs:323-377 picks, for every series: clean seasonal ones, noise, constants, ramps, ragged and too-short ones, huge
    values whose squares overflow, an infinity.  The forecasts then use those periods (sample against the oracle)."""
    import ctypes as C
    import torch
    api, O, lib, synth = env
    from anofox_forecast_amd.device import DeviceBatch
    rng = np.random.default_rng(T)
    series = []
    for p in (2, 3, 7, 12, 24, 52, 168, 365):
        for k in range(3):
            L = int(rng.integers(min(max(3 * p, 20), T), T + 1))
            t = np.arange(L)
            series.append(20.0 + 0.01 * t + (3.0 + k) * np.sin(2 * np.pi * t / p) + rng.normal(0, 0.5 + k, L))
    series += [rng.normal(0, 1, int(rng.integers(4, T + 1))) for _ in range(40)]                    # noise: spurious peaks
    series += [np.floor(rng.gamma(0.3, 2.0, int(rng.integers(50, T + 1)))) for _ in range(40)]      # intermittent counts
    series += [np.full(50, 3.0), np.arange(60.0), np.array([1.0, 2.0, 4.0]), np.array([1.0, 5.0, 2.0, 7.0]), np.array([1.0, 2.0]),
               np.array([]), 1e200 * rng.normal(0, 1, 80), np.r_[rng.normal(0, 1, 30), np.inf, rng.normal(0, 1, 30)], rng.normal(0, 1, T)]
    n = len(series)
    opts = lib.make_options("AutoETS", 6)
    assert opts.auto_detect_seasonality and opts.seasonal_period == 0
    b = DeviceBatch(n, T, opts, "cuda:0")
    y = np.zeros((T, b.ld))
    ln = np.zeros(b.ld, dtype=np.int32)
    for s, v in enumerate(series):
        y[: len(v), s] = v
        ln[s] = len(v)
    b.set_block(torch.from_numpy(y).cuda(), torch.from_numpy(ln).cuda())
    got = b.periods()
    L = O.lib()
    want = np.ones(n, dtype=np.int32)
    for s, v in enumerate(series):
        v = np.ascontiguousarray(v, dtype=np.float64)
        p = L.oracle_detect_seasonality_first(v.ctypes.data, len(v)) if len(v) >= 3 else 0
        want[s] = p if p > 0 else 1
    assert np.array_equal(got, want), [(s, int(got[s]), int(want[s])) for s in np.nonzero(got != want)[0][:10]]
    assert len(set(want.tolist())) > 10
    if T <= 1000:
        b.run()
        torch.cuda.synchronize()
        out = b.results()
        yhat = out["yhat"].cpu().numpy()
        status = out["status"].cpu().numpy()
        oo = O.make_options("AutoETS", 6)
        for s in range(0, n, 9):
            ref = O.forecast(series[s], oo)
            assert (status[s] == 0) == ref["ok"], (s, status[s], ref)
            if ref["ok"]:
                assert _rel(yhat[s], ref["point"]) <= REL_TOL, s
    b.close()


def _fixed_batch(lib, Y, spec, m, h, params, lens=None):
    """ETS(spec) with given parameters over a resident block; returns host copies of the device results + stats."""
    import torch
    from anofox_forecast_amd.device import DeviceBatch, pack_time_major
    n, T = Y.shape
    opts = lib.make_options("ETS", h, ets_model=spec, seasonal_period=m)
    b = DeviceBatch(n, T, opts, "cuda:0")
    ln = torch.zeros(b.ld, dtype=torch.int32, device="cuda")
    ln[:n] = torch.as_tensor(np.full(n, T) if lens is None else lens, dtype=torch.int32)
    b.set_block(torch.from_numpy(pack_time_major(Y, b.ld)).cuda(), ln)
    b.set_fixed_params(*params)
    b.run()
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy().copy() for k, v in b.results().items()}
    st = b.stats()
    b.close()
    return out, st


@pytest.mark.parametrize("spec,params", [("AAA", (0.2, 0.05, 0.1, 1.0)), ("AAdA", (0.3, 0.1, 0.2, 0.9)), ("ANN", (0.5, 0.0, 0.0, 1.0)),
                                         ("MAM", (0.15, 0.02, 0.3, 1.0)), ("MMdM", (0.25, 0.2, 0.05, 0.85)), ("AAN", (0.9, 0.9, 0.0, 1.0))])
def test_fixed_parameter_ets_matches_oracle(env, spec, params):
    """BASELINE config 2's path at test size: ETS(spec) with GIVEN smoothing parameters -- no optimiser, one streamed pass --
    equals the oracle's restatement (same initial states, same recursion) bit for bit, ragged lengths and short series
    included; exactly one pass per fitted series is counted."""
    api, O, lib, synth = env
    n, T, h, m = 300, 180, 12, 7
    Y = synth.gen_series(synth.SEED_M5, 12000, n, T, m, positive=True)
    lens = np.array([T - (s % 9) * 17 for s in range(n)], dtype=np.int64)
    lens[:4] = [2, 3, 9, 13]                               # too short for anything / for a seasonal model
    Yz = Y.copy()
    for s in range(n):
        Yz[s, lens[s]:] = 0.0
    out, st = _fixed_batch(lib, Yz, spec, m, h, params, lens)
    offs = np.concatenate([[0], np.cumsum(lens)])
    vals = np.concatenate([Y[s, :lens[s]] for s in range(n)])
    ref = O.ets_fixed_batch(vals, offs, spec, m, *params, h)
    assert np.array_equal(out["status"][:n], ref["status"])
    ok = ref["status"] == 0
    assert ok.sum() >= n - 4
    for k in ("yhat", "lower", "upper"):
        assert _rel(out[k][:n][ok], ref[k][ok]) <= REL_TOL, (spec, k)
    assert st["total_passes"] == int(ok.sum())              # one pass per fitted series: the 8 T + 24 h bytes of SURVEY 8(d) C2
    assert st["algorithmic_bytes"] == 8 * int(lens[ok].sum()) + 24 * h * n


def test_fixed_parameter_entry_rejects_bad_input(env):
    api, O, lib, synth = env
    import ctypes as C
    L = lib.load()
    err = lib.AnofoxError()
    for model, ets, args in (("AutoETS", "", (0.2, 0.05, 0.1, 1.0)), ("ETS", "", (0.2, 0.05, 0.1, 1.0)),
                             ("ETS", "AAA", (1.2, 0.05, 0.1, 1.0)), ("ETS", "AAA", (0.2, 0.3, 0.1, 1.0)),
                             ("ETS", "AAA", (0.2, 0.05, 0.9, 1.0)), ("ETS", "AAdA", (0.2, 0.05, 0.1, 1.5)),
                             ("ETS", "AAA", (float("nan"), 0.05, 0.1, 1.0))):
        h = C.c_void_p()
        o = lib.make_options(model, 4, ets_model=ets, seasonal_period=7)
        assert L.anofox_hip_batch_create(8, 32, C.byref(o), C.byref(h), C.byref(err))
        assert not L.anofox_hip_batch_set_fixed_params(h, *args, C.byref(err)) and err.code == lib.INVALID_INPUT, (model, ets, args)
        L.anofox_hip_batch_destroy(h)


def test_fixed_parameter_aaa_full_size_m5(env):
    """BASELINE.json configs[1] at its full size: ETS(A,A,A) with alpha 0.2, beta 0.05, gamma 0.1 on 30,490 x
    1,913, h = 28 -- against the oracle on EVERY series (one pass each: seconds on the host), plus the properties that hold
    without it (second run identical, intervals bracket the forecast)."""
    api, O, lib, synth = env
    n, T, h, m = 30490, 1913, 28, 7
    Y = synth.gen_series(synth.SEED_M5, 0, n, T, m, False)
    params = (0.2, 0.05, 0.1, 1.0)
    out, st = _fixed_batch(lib, Y, "AAA", m, h, params)
    again, _ = _fixed_batch(lib, Y, "AAA", m, h, params)
    for k in ("yhat", "lower", "upper", "status"):
        assert np.array_equal(out[k], again[k], equal_nan=True)
    assert np.all(out["status"][:n] == 0) and np.all(out["lower"][:n] <= out["yhat"][:n]) and np.all(out["yhat"][:n] <= out["upper"][:n])
    ref = O.ets_fixed_batch(Y.reshape(-1), np.arange(n + 1, dtype=np.int64) * T, "AAA", m, *params, h)
    assert np.all(ref["status"] == 0)
    for k in ("yhat", "lower", "upper"):
        assert _rel(out[k][:n], ref[k]) <= REL_TOL, k
    assert st["total_passes"] == n and st["algorithmic_bytes"] == n * (8 * T + 24 * h)


def test_one_sweep_prep_paths(env):
    """prep.hip streams the block once (round 5): the least-squares start of the seasonally adjusted series comes from per-phase sums
    (oracle/ets.c ets_init_states), for m = 7 in registers, for every other period from season_figures_kernel -- whose series live in LDS,
    or in an HBM scratch when 2 T + 2 m doubles do not fit (here: T = 6,500) -- and a batch with ONE candidate spec lets its final pass
    carry the intervals' sd (lower / upper compared like the forecasts).  Ragged lengths, both season types, a spec the data refuses
    (multiplicative season on a series with zeros: no forecast, and no sd needed), periods either side of the LDS ring limit."""
    api, O, lib, synth = env
    Yl = synth.gen_series(synth.SEED_M5, 9100, 6, 6500, 12, positive=True)
    long_series = [Yl[s, : 6500 - 37 * s] for s in range(6)]
    _compare(api, O, lib, long_series, "ETS", 7, ets_model="AAA", seasonal_period=12)
    _compare(api, O, lib, long_series, "ETS", 7, ets_model="MAdM", seasonal_period=12)
    _compare(api, O, lib, long_series[:3], "AutoETS", 7, seasonal_period=12, model_pool="reduced")
    Y = synth.gen_series(synth.SEED_M5, 9200, 90, 300, 7, positive=True)
    Yz = synth.gen_series(synth.SEED_M5, 9300, 90, 300, 7)
    mixed = [(Y[s] if s % 2 else Yz[s])[: 300 - (s % 7) * 11] for s in range(90)]
    for model in ("AAA", "ANA", "AAdA", "MNM", "MAM", "MMdM", "AAN", "MNN"):
        for m in (7, 5, 52, 64, 65):
            _compare(api, O, lib, mixed, "ETS", 9, ets_model=model, seasonal_period=m)


def test_stress_shape_properties(env):
    """BASELINE.json configs[4], one GPU's share (125,000 series x 1,024 observations, AutoETS, h = 28, m = 7): every series
    gets a forecast, intervals bracket it, a second run reproduces every bit, an arbitrary shuffled sub-batch reproduces
    the full batch and equals the oracle."""
    api, O, lib, synth = env
    n, T, h, m = 125000, 1024, 28, 7
    Y = synth.gen_series(synth.SEED_STRESS, 0, n, T, m, False)
    full, again, names = _run_device_batch(lib, Y, "AutoETS", h, m)
    for k in ("yhat", "lower", "upper", "model_code", "status"):
        assert np.array_equal(full[k], again[k], equal_nan=True), f"{k}: second run differs"
    assert np.all(full["status"][:n] == 0)
    yh, lo, hi = full["yhat"][:n], full["lower"][:n], full["upper"][:n]
    assert np.all(np.isfinite(yh)) and np.all(lo <= yh) and np.all(yh <= hi)
    assert all(nm.startswith("AutoETS") for nm in names)
    pick = np.random.default_rng(9).choice(n, 128, replace=False)
    sub, _, sub_names = _run_device_batch(lib, Y[pick], "AutoETS", h, m)
    for k in ("yhat", "lower", "upper", "model_code"):
        assert np.array_equal(sub[k][:128], full[k][pick], equal_nan=True), f"{k}: result depends on the batch"
    oo = O.make_options("AutoETS", h, seasonal_period=m)
    for j in range(64):
        ref = O.forecast(Y[pick[j]], oo)
        assert _rel(sub["yhat"][j], ref["point"]) <= REL_TOL and sub_names[j] == ref["model_name"]


@pytest.mark.parametrize("model", ["HoltWinters", "AutoETS"])
def test_auto_detected_big_parts_run_beside_the_small_ones(env, model):
    """params := MAP{} on a batch where one detected period has >= 2,048 series (it runs as its own batch with the compile-time kernels,
    on its own host thread) beside merged batches of the rare periods and a non-seasonal rest: every series equals the oracle."""
    api, O, lib, synth = env
    rng = np.random.default_rng(23)
    series = []
    for k in range(2300):
        p = 7 if k < 2100 else (5, 12, 20, 33, 70)[k % 5]
        T = int(6 * p + 30 + rng.integers(0, 12))
        t = np.arange(T)
        series.append(40.0 + 0.05 * t + 6.0 * np.sin(2 * np.pi * t / p) + 2.0 * np.cos(4 * np.pi * t / p) + rng.normal(0, 0.4, T))
    series += [rng.normal(10, 1, 40) for _ in range(60)] + [np.full(30, 2.0), np.array([1.0, 2.0])]
    _compare(api, O, lib, series, model, 5)


@pytest.mark.parametrize("m,T", [(700, 5800)])
def test_long_period_figures_through_the_scratch_variant(env, m, T):
    """season_figures_kernel keeps the series and its trend in LDS up to 2 T + 2 m = 12,288 doubles; above that the workgroup
    works in an HBM scratch (T = 5,800, m = 700: 13,000 doubles).  ETS(A,A,A) and AutoETS with that period equal the oracle."""
    api, O, lib, synth = env
    rng = np.random.default_rng(m)
    series = []
    for k in range(6):
        L = T - 37 * k
        t = np.arange(L)
        series.append(100.0 + 0.01 * t + (8.0 + k) * np.sin(2 * np.pi * t / m) + rng.normal(0, 0.5, L))
    _compare(api, O, lib, series, "ETS", 9, ets_model="AAA", seasonal_period=m)
    _compare(api, O, lib, series[:3], "AutoETS", 9, seasonal_period=m)


def test_host_entry_uploads_large_blocks_in_chunks(env):
    """A block above 512 MB goes to the device in column chunks through two pinned staging buffers (pitched copies while the packer
    threads fill the other buffer): 36,000 ragged series x 2,000 observations = 576 MB = two chunks (33,536 + 2,496 columns).
    SeasonalNaive makes every column checkable without the oracle: the forecast is the series' own last season, so a column that
    landed in the wrong place, or a row of the wrong chunk, shows up in exactly that series."""
    api, O, lib, synth = env
    n, T, m, h = 36000, 2000, 7, 14
    rng = np.random.default_rng(77)
    per = 5 + np.arange(n) % 40
    Y = rng.normal(0.0, 1.0, (n, T)) + np.arange(n)[:, None] * 1.0e-3 + 3.0 * np.sin(2.0 * np.pi * np.arange(T)[None, :] / per[:, None])
    lens = np.where(np.arange(n) % 9 == 0, rng.integers(20, T, n), T)
    series = [Y[s, : lens[s]] for s in range(n)]
    got, berr = api.forecast_batch(series, lib.make_options("SeasonalNaive", h, seasonal_period=m))
    assert berr["ok"], berr
    bad = 0
    for s in range(n):
        y = series[s]
        want = np.array([y[len(y) - m + (i % m)] for i in range(h)])
        if not (got[s]["ok"] and np.array_equal(got[s]["point"], want)):
            bad += 1
    assert bad == 0, bad
    # ... and the fitted path through the same upload: a sample against the oracle
    got2, berr2 = api.forecast_batch(series, lib.make_options("SES", h))
    assert berr2["ok"], berr2
    for s in (0, 9, 33535, 33536, 33537, n - 1):
        ref = O.forecast(series[s], O.make_options("SES", h))
        assert got2[s]["ok"] and np.array_equal(got2[s]["point"], ref["point"]), s
    # ... and the period detection, which packs the same series in chunks of its own (~128 MB, five here): SeasonalNaive without
    # a period repeats the last DETECTED season
    got3, berr3 = api.forecast_batch(series, lib.make_options("SeasonalNaive", h))
    assert berr3["ok"], berr3
    L = O.lib()
    seen = set()
    for s in list(range(0, n, 97)) + [8191, 8192, 16383, 16384, 33535, 33536, n - 1]:
        y = np.ascontiguousarray(series[s])
        p = L.oracle_detect_seasonality_first(y.ctypes.data, len(y)) if len(y) >= 3 else 0
        p = p if p > 0 else 1
        seen.add(p)
        want = np.array([y[len(y) - p + (i % p)] for i in range(h)])
        assert got3[s]["ok"] and np.array_equal(got3[s]["point"], want), (s, p)
    assert len(seen) > 20


@pytest.mark.parametrize("scaling", ["default", "strong", "weak", "self-launch"])
def test_bench_two_ranks_on_one_gpu(env, scaling):
    """bench.py's N > 1 path end to end -- two processes, torch.distributed rendezvous, sharded batches on the device, gather
    of the forecast chunks, max-over-ranks timing, one JSON line -- on this box's single GPU (ANOFOX_BENCH_ONE_GPU: both
    ranks use GPU 0 and the gather runs over gloo, since RCCL refuses two ranks on one device)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    envv = dict(os.environ, ANOFOX_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29300 + os.getpid() % 400), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "autoets_m5", "--n-series", "1500", "--t", "200", "--cpu-sample", "0"]
    if scaling == "self-launch":
        # the UN-WRAPPED invocation, the shape of the command the driver runs for N = 1: bench.py starts its own ranks (a child
        # torch.distributed.run, never an exec) and the line it prints says two ranks ran
        cmd = [sys.executable] + cmd[cmd.index(os.path.join(root, "bench.py")):]
        envv.pop("RANK", None); envv.pop("WORLD_SIZE", None)
        scaling = "default"
    if scaling == "default":
        scaling = "strong"          # `bench.py --gpus N` without flags IS BASELINE config 3: ONE batch, series-sharded over the ranks
    else:
        cmd += ["--scaling", scaling]
    out = subprocess.run(cmd, env=envv, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == scaling and j["value"] > 0
    assert "ranks seen: 2" in j["config"]["parallelism"]
    assert j["config"]["series_total"] == (1500 if scaling == "strong" else 3000)
    assert j["config"]["series_per_gpu"] == (750 if scaling == "strong" else 1500)


def test_auto_arima_matches_oracle(env):
    """AutoARIMA (stepwise CSS search) on the GPU walks the oracle's search bit for bit: same selected order,
    same forecasts -- seasonal (m = 7: VGPR-free generic lag polynomials), non-seasonal, ragged, short."""
    api, O, lib, synth = env
    Y = synth.gen_series(synth.SEED_M5, 9000, 70, 150, 7)
    rng = np.random.default_rng(11)
    series = [Y[s, : 150 - (s % 7) * 11] for s in range(64)]
    series += [np.cumsum(rng.normal(0.1, 1.0, 120)), 20 + 5 * np.sin(2 * np.pi * np.arange(140) / 7) + rng.normal(0, 0.3, 140),
               np.array(KAT_SERIES, dtype=float), np.arange(8.0), np.array([1.0, 2.0, 3.0]), np.full(40, 7.0)]
    _compare(api, O, lib, series, "AutoARIMA", 10, seasonal_period=7)
    _compare(api, O, lib, series, "AutoARIMA", 10)                        # auto-detected periods (host packer)
    r = api.forecast_series(KAT_SERIES, lib.make_options("AutoARIMA", 3, auto_detect=False))
    # the reference's known answer (test/sql/ts_model_distinctness.test:164), within the north star's 1e-5, and the model behind it
    assert r["ok"] and r["model_name"] == "AutoARIMA(2,1,1)" and abs(r["point"][0] - 18.014537) / 18.014537 < 1e-5


@pytest.mark.parametrize("tune", ["arima_queue_sort=0", "arima_queue_sort=2", "arima_queue_sort=3", "arima_spec_factor=0.01;arima_lookahead=0.01",
                                  "arima_spec_factor=1000;arima_lookahead=1000", "arima_shared_chunk_rounds=-0.002", "arima_prep_lanes=16",
                                  "arima_trace=2"])
def test_auto_arima_schedule_variants_are_bit_identical(env, monkeypatch, tune):
    """Round 4: the fit queues of a sweep are bucketed by (dimension, order shape) and sorted by series, the pass variant follows the
    orders of a wave's live lanes, a queue may go out in several launches when searches share the device.  None of it may move a bit:
    every queue order (as emitted / by shape / by dimension / one bucket), both drivers forced (sequential lanes only, four lanes
    only, with and without lookahead), launches of a few dozen problems each, and 16 series per prep wave reproduce the oracle --
    weekly period (register rings), no period, a period in the LDS ring class and one in the HBM-ring class."""
    api, O, lib, synth = env
    monkeypatch.setenv("ANOFOX_HIP_TUNE", tune)
    Y = synth.gen_series(synth.SEED_M5, 9300, 150, 170, 7)
    series = [Y[s, : 170 - (s % 7) * 9] for s in range(150)]
    _compare(api, O, lib, series, "AutoARIMA", 8, seasonal_period=7)
    _compare(api, O, lib, series[:60], "AutoARIMA", 8, seasonal_period=1)
    _compare(api, O, lib, series[:40], "AutoARIMA", 8, seasonal_period=12)
    _compare(api, O, lib, series[:24], "AutoARIMA", 8, seasonal_period=30)


def test_detected_period_above_24_is_seasonal(env):
    """A DETECTED period is handed to the seasonal AutoARIMA search like an explicit one (forecast.rs:528-537, 1448-1452): a series
    of period 30 called without seasonal_period comes back named ...[30] -- through the one-series entry and inside a batch whose
    other series detect other periods; equal to the explicit-period call and to the oracle bit for bit."""
    api, O, lib, synth = env
    rng = np.random.default_rng(30)
    t = np.arange(360)
    y = 50.0 + 10.0 * np.sin(2 * np.pi * t / 30) + 4.0 * np.cos(4 * np.pi * t / 30) + rng.normal(0, 0.5, t.size)
    r = api.forecast_series(y, lib.make_options("AutoARIMA", 5))
    assert r["ok"] and r["model_name"].endswith("[30]"), r
    e = api.forecast_series(y, lib.make_options("AutoARIMA", 5, seasonal_period=30))
    assert e["ok"] and e["model_name"] == r["model_name"] and np.array_equal(e["point"], r["point"])
    Y = synth.gen_series(synth.SEED_M5, 9700, 12, 200, 7)
    series = [y, y[:300]] + [Y[s] for s in range(12)] + [40.0 + 8.0 * np.sin(2 * np.pi * np.arange(400) / 52) + rng.normal(0, 0.4, 400)]
    got, berr = api.forecast_batch(series, lib.make_options("AutoARIMA", 5))
    assert berr["ok"], berr
    assert got[0]["model_name"] == r["model_name"] and np.array_equal(got[0]["point"], r["point"])
    assert got[-1]["ok"] and got[-1]["model_name"].endswith("[52]"), got[-1]
    _compare(api, O, lib, series, "AutoARIMA", 5)


@pytest.mark.parametrize("budget", ["0", "3", "25", "100"])
def test_exact_likelihood_refit_in_two_launches(env, monkeypatch, budget):
    """The exact-likelihood refit runs its sequential Nelder-Mead up to a budget of evaluations per series, parks the series that are
    still running (simplex, function values, counters) and finishes them in a second launch with four trial points per filter pass
    (eight lanes per series).  The trajectory is the sequential one whatever the budget -- 0 (one launch), 3 (everything is parked
    after the first iteration), 25, 100 (the default): every forecast equals the oracle's refit, seasonal and non-seasonal, ragged."""
    import ctypes as C
    api, O, lib, synth = env
    L = lib.load()
    flag = C.c_int.in_dll(O.lib(), "oracle_arima_ml_refit")
    monkeypatch.setenv("ANOFOX_HIP_TUNE", f"arima_refit_budget={budget}")
    rng = np.random.default_rng(41)
    Y = synth.gen_series(synth.SEED_M5, 9300, 90, 220, 7)
    series = [Y[s, : 220 - (s % 6) * 11] for s in range(90)]
    for k in range(30):                                         # richer dynamics: higher orders get selected, long Nelder-Mead runs
        T = 180 + 7 * (k % 5)
        e = rng.normal(0, 1, T + 20)
        x = np.zeros(T + 20)
        for t in range(14, T + 20):
            x[t] = 0.6 * x[t - 1] - 0.3 * x[t - 2] + 0.4 * x[t - 7] + e[t] + 0.5 * e[t - 1] + 0.3 * e[t - 7]
        series.append(50.0 + 0.05 * np.arange(T) + 3.0 * x[20:])
    try:
        assert L.anofox_hip_set_default_arima_method(lib.ARIMA_CSS_ML)
        flag.value = 1
        for kw in (dict(seasonal_period=7), dict(seasonal_period=1)):
            got, berr = api.forecast_batch(series, lib.make_options("AutoARIMA", 8, **kw))
            assert berr["ok"], berr
            names = set()
            for s, y in enumerate(series):
                ref = O.forecast(y, O.make_options("AutoARIMA", 8, **kw))
                assert got[s]["ok"] == ref["ok"], s
                if ref["ok"]:
                    assert got[s]["model_name"] == ref["model_name"] and np.array_equal(got[s]["point"], ref["point"]), (s, got[s]["model_name"])
                    names.add(ref["model_name"])
            assert len(names) >= 6, names
    finally:
        flag.value = 0
        L.anofox_hip_set_default_arima_method(lib.ARIMA_CSS)


def test_auto_arima_estimation_method_is_a_caller_choice(env):
    """ANOFOX_ARIMA_CSS (default) keeps the selected model's CSS estimates, ANOFOX_ARIMA_CSS_ML refits it on the exact Gaussian
    likelihood (the Kalman / Chandrasekhar kernel): per batch (anofox_hip_batch_set_arima_method) and as the process default
    (anofox_hip_set_default_arima_method, what the one-series and host-buffer entries use) -- each against the oracle in the
    same mode, bit for bit; the two modes select the same orders and differ in the forecasts."""
    import ctypes as C
    import torch
    api, O, lib, synth = env
    from anofox_forecast_amd.device import DeviceBatch, pack_time_major
    L = lib.load()
    flag = C.c_int.in_dll(O.lib(), "oracle_arima_ml_refit")
    n, T, h, m = 48, 150, 10, 7
    Y = synth.gen_series(synth.SEED_M5, 9100, n, T, m)
    Y[5] = 20 + 5 * np.sin(2 * np.pi * np.arange(T) / 7) + np.random.default_rng(3).normal(0, 0.3, T)
    opts = lib.make_options("AutoARIMA", h, seasonal_period=m)
    oo = O.make_options("AutoARIMA", h, seasonal_period=m)
    outs = {}
    try:
        for method in (lib.ARIMA_CSS, lib.ARIMA_CSS_ML):
            b = DeviceBatch(n, T, opts, "cuda:0")
            err = lib.AnofoxError()
            assert L.anofox_hip_batch_set_arima_method(b.handle, method, C.byref(err)), err.message
            assert not L.anofox_hip_batch_set_arima_method(b.handle, 7, C.byref(err)) and err.code == lib.INVALID_INPUT
            y = torch.from_numpy(pack_time_major(Y, b.ld)).cuda()
            ln = torch.full((b.ld,), T, dtype=torch.int32, device="cuda")
            ln[n:] = 0
            b.set_block(y, ln)
            b.run()
            torch.cuda.synchronize()
            r = b.results()
            outs[method] = (r["yhat"].cpu().numpy()[:n].copy(), r["model_code"].cpu().numpy()[:n].copy())
            b.close()
            flag.value = method
            for s in range(n):
                ref = O.forecast(Y[s], oo)
                assert ref["ok"] and np.array_equal(outs[method][0][s], ref["point"]), (method, s)
            # the process default drives the entries that only carry a ForecastOptions block
            assert L.anofox_hip_set_default_arima_method(method)
            got, berr = api.forecast_batch(list(Y[:8]), opts)
            assert berr["ok"] and all(np.array_equal(got[s]["point"], outs[method][0][s]) for s in range(8))
            one = api.forecast_series(Y[5], opts)
            assert one["ok"] and np.array_equal(one["point"], outs[method][0][5])
        assert not L.anofox_hip_set_default_arima_method(9)
    finally:
        flag.value = 0
        L.anofox_hip_set_default_arima_method(lib.ARIMA_CSS)
    assert np.array_equal(outs[0][1], outs[1][1])                      # same orders: the refit only re-estimates
    assert not np.array_equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("model,kw", [("AutoETS", {"seasonal_period": 7}), ("AutoARIMA", {"seasonal_period": 7}), ("HoltWinters", {}), ("Naive", {})])
def test_batch_entry_shards_over_the_listed_devices(env, model, kw):
    """anofox_hip_set_devices / ANOFOX_HIP_DEVICES: the batch entry cuts contiguous series ranges (ceil(N / G) each) and runs
    them on the listed devices from one host thread each.  One GPU is visible here, so it is listed twice and three times
    (two / three concurrent shard batches on it): every series' result -- forecasts, intervals, names, per-series errors,
    per-series horizons -- is bit for bit what the single-device call returns, also when periods are auto-detected."""
    api, O, lib, synth = env
    L = lib.load()
    Y = synth.gen_series(synth.SEED_M5, 8800, 150, 120, 7, positive=True)
    series = [Y[s, : 120 - (s % 6) * 9] for s in range(150)] + [np.array([1.0, 2.0]), np.array([]), np.full(30, 4.0)]
    horizons = [5 + (s % 4) for s in range(len(series))]
    opts = lib.make_options(model, 8, **kw)
    L.anofox_hip_set_min_series_per_device(16)
    try:
        lib.set_devices([])
        base, berr0 = api.forecast_batch(series, opts, horizons=horizons)
        assert berr0["ok"]
        for devs in ([0, 0], [0, 0, 0]):
            lib.set_devices(devs)
            assert L.anofox_hip_get_devices(None, 0) == len(devs)
            got, berr = api.forecast_batch(series, opts, horizons=horizons)
            assert berr["ok"] == berr0["ok"]
            for s in range(len(series)):
                assert got[s]["ok"] == base[s]["ok"] and got[s]["code"] == base[s]["code"] and got[s]["message"] == base[s]["message"], (devs, s)
                if base[s]["ok"]:
                    assert got[s]["model_name"] == base[s]["model_name"]
                    for k in ("point", "lower", "upper"):
                        assert np.array_equal(got[s][k], base[s][k]), (devs, s, k)
        # a device that is not there is refused and nothing changes; an unknown model fails the whole statement like before
        with pytest.raises(ValueError):
            lib.set_devices([0, 99])
        assert L.anofox_hip_get_devices(None, 0) == 3
        bad, berr = api.forecast_batch(series, lib.make_options("NoSuchModel", 8))
        assert not berr["ok"] and berr["code"] == lib.INVALID_MODEL
    finally:
        lib.set_devices([])
        L.anofox_hip_set_min_series_per_device(2048)


def test_release_caches_and_double_free_guard(env):
    """anofox_hip_release_caches gives the idle device blocks, pinned staging blocks, stream sets and parked one-series batches
    back (hipMemGetInfo shows the memory again) and the library keeps working afterwards -- twice over, with single-series calls
    in between, so parked batches and their stream sets are taken, handed back and destroyed."""
    import torch
    api, O, lib, synth = env
    L = lib.load()
    Y = synth.gen_series(synth.SEED_M5, 8300, 400, 300, 7, positive=True)
    opts = lib.make_options("AutoETS", 7, seasonal_period=7)
    ref, _ = api.forecast_batch(list(Y[:40]), opts)
    for rep in range(2):
        got, berr = api.forecast_batch(list(Y), opts)
        assert berr["ok"]
        one = api.forecast_series(Y[3], opts)
        assert one["ok"] and np.array_equal(one["point"], got[3]["point"])
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info()
        L.anofox_hip_release_caches()
        free_after, _ = torch.cuda.mem_get_info()
        assert free_after > free_before + 50 * 2**20, (free_before, free_after)      # the 25 gather blocks alone are 25 x 1 MB x ... > 50 MB here
        for s in range(40):
            assert np.array_equal(got[s]["point"], ref[s]["point"])
    L.anofox_hip_release_caches()                                                      # nothing idle: a no-op
    again, berr = api.forecast_batch(list(Y[:40]), opts)
    assert berr["ok"] and all(np.array_equal(again[s]["point"], ref[s]["point"]) for s in range(40))


@pytest.mark.parametrize("model,kw", [("AutoETS", dict(seasonal_period=7)), ("AutoETS", dict(seasonal_period=1)),
                                      ("AutoARIMA", dict(seasonal_period=7)), ("HoltWinters", dict(seasonal_period=7)),
                                      ("ETS", dict(ets_model="MMdM", seasonal_period=7)), ("SESOptimized", dict())])
def test_hostile_inputs_match_oracle(env, model, kw):
    """Non-finite, huge, tiny, negative, all-zero, step and alternating series, lengths around every admissibility
    threshold, horizons 0 / 5 / 200: the same error code or the same bits as the oracle, series by series."""
    api, O, lib, synth = env
    base = np.abs(np.random.default_rng(0).normal(50, 10, 120)) + 5
    cases = {"inf": np.r_[base[:60], np.inf, base[61:]], "nan": np.r_[base[:60], np.nan, base[61:]], "huge": base * 1e300,
             "tiny": base * 1e-300, "neg": base - 60, "zeros": np.zeros(80), "step": np.r_[np.full(60, 5.0), np.full(60, 9.0)],
             "len3": base[:3], "len4": base[:4], "len14": base[:14], "len15": base[:15], "alt": np.tile([1.0, 1e6], 60)}
    for h in (0, 5, 200):
        got, berr = api.forecast_batch(list(cases.values()), lib.make_options(model, h, **kw))
        assert berr["ok"]
        oo = O.make_options(model, h, **kw)
        for (name, y), r in zip(cases.items(), got):
            ref = O.forecast(y, oo)
            assert r["ok"] == ref["ok"], (name, h, r.get("code"), ref.get("code"))
            if not r["ok"]:
                assert r["code"] == ref["code"], (name, h)
            else:
                assert r["model_name"] == ref["model_name"], (name, h)
                assert np.array_equal(np.asarray(r["point"]), np.asarray(ref["point"]), equal_nan=True), (name, h)
                assert np.array_equal(np.asarray(r["lower"]), np.asarray(ref["lower"]), equal_nan=True), (name, h)


def _run_device_batch(lib, Y, model, h, m, dev="cuda:0", arima_method=None):
    import torch
    from anofox_forecast_amd.device import DeviceBatch, pack_time_major
    n, T = Y.shape
    b = DeviceBatch(n, T, lib.make_options(model, h, seasonal_period=m), dev)
    if arima_method is not None:
        b.set_arima_method(arima_method)
    y = torch.from_numpy(pack_time_major(Y, b.ld)).to(dev)
    ln = torch.full((b.ld,), T, dtype=torch.int32, device=dev)
    ln[n:] = 0
    b.set_block(y, ln)
    b.run()
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy().copy() for k, v in b.results().items()}
    b.run()                                   # a second step over the same resident block
    torch.cuda.synchronize()
    again = {k: v.cpu().numpy().copy() for k, v in b.results().items()}
    names = [b.model_name(int(c)) for c in out["model_code"][:n]]
    b.close()
    return out, again, names


@pytest.mark.parametrize("m", [2, 3, 4, 12, 24, 30, 52, 168, 365, 3000])
def test_auto_arima_other_periods(env, m):
    """Every variant of the CSS pass: per-step ring access (m = 2, 3), compile-time ring slots (m = 4, 12), the generic
    run-time ring in LDS (m = 24), the ring in HBM scratch for the long calendar periods the reference accepts like any other
    (forecast.rs:1447-1451: weekly data with a yearly period 52, hourly 168, daily 365; prep figure, fit rings and the forecast
    kernel's ring + polynomials all leave LDS) and the loud failure of an explicit period above 2,048 -- each against the oracle."""
    api, O, lib, synth = env
    rng = np.random.default_rng(100 + m)
    T = 8 * m + 40 if m <= 400 else 500
    t = np.arange(T)
    series = [10 + 3 * np.sin(2 * np.pi * t / m + k) + 0.02 * k * t + rng.normal(0, 0.5 + 0.1 * k, T) for k in range(12)]
    series += [np.cumsum(rng.normal(0.05, 1.0, T)) for _ in range(6)]
    series = [s[: T - 3 * (i % 5)] for i, s in enumerate(series)]                    # ragged
    _compare(api, O, lib, series, "AutoARIMA", 2 * m + 1, seasonal_period=m)


@pytest.mark.parametrize("model,positive", [("AutoETS", False), ("AutoETS", True), ("AutoARIMA", False), ("AutoARIMA-ML", False)])
def test_full_size_m5_properties(env, model, positive):
    """BASELINE.json's full M5 shape (30,490 series x 1,913 observations, h = 28, m = 7), checked through properties that
    do not need the oracle at that size: a second run over the resident block reproduces every bit; a series' result does
    not depend on the batch it is in nor on its position (a shuffled 96-series sub-batch reproduces the full batch bit
    for bit); that sub-batch equals the CPU oracle; intervals bracket the point forecast; every series gets a forecast and
    a model name of the right family."""
    api, O, lib, synth = env
    n, T, h, m = 30490, 1913, 28, 7
    Y = synth.gen_series(synth.SEED_M5, 0, n, T, m, positive)
    # "AutoARIMA-ML": BASELINE config 4 as written -- "(Kalman kernel)": the selected models refitted on the exact Gaussian likelihood
    # (ANOFOX_ARIMA_CSS_ML, arima_refit_kernel), the oracle switched to the same method
    method = lib.ARIMA_CSS_ML if model == "AutoARIMA-ML" else None
    model = "AutoARIMA" if method is not None else model
    import ctypes as C
    flag = C.c_int.in_dll(O.lib(), "oracle_arima_ml_refit")
    flag.value = 1 if method is not None else 0
    try:
        _full_size_checks(api, O, lib, Y, model, positive, h, m, method)
    finally:
        flag.value = 0


def _full_size_checks(api, O, lib, Y, model, positive, h, m, method):
    n = Y.shape[0]
    run = lambda YY: _run_device_batch(lib, YY, model, h, m, arima_method=method)
    full, again, names = run(Y)
    for k in ("yhat", "lower", "upper", "model_code", "status"):
        assert np.array_equal(full[k], again[k], equal_nan=True), f"{k}: second run differs"
    assert np.all(full["status"][:n] == 0)
    yh, lo, hi = full["yhat"][:n], full["lower"][:n], full["upper"][:n]
    assert np.all(np.isfinite(yh)) and np.all(lo <= yh) and np.all(yh <= hi)
    assert all(nm.startswith(model + "(") or nm == model for nm in names)
    pick = np.random.default_rng(5).choice(n, 96, replace=False)          # arbitrary series, arbitrary order
    sub, _, sub_names = run(Y[pick])
    for k in ("yhat", "lower", "upper", "model_code"):
        assert np.array_equal(sub[k][:96], full[k][pick], equal_nan=True), f"{k}: result depends on the batch"
    oo = O.make_options(model, h, seasonal_period=m)
    n_ref = 96 if model == "AutoETS" and not positive else 24              # the oracle needs seconds per series otherwise
    for j in range(n_ref):
        ref = O.forecast(Y[pick[j]], oo)
        assert _rel(sub["yhat"][j], ref["point"]) <= REL_TOL and sub_names[j] == ref["model_name"]


def test_plain_c_caller_matches_oracle(env):
    """tests/c_abi/caller.c: a C program using only include/anofox_fcst_hip.h, filling the structs the way
    the reference's binding does; its printed model name and first forecast must be the oracle's."""
    import subprocess, tempfile
    from test_abi_cpu import build_c_caller
    api, O, lib, synth = env
    y = np.array([20.0 + 0.5 * i + (6.0 if i % 7 == 0 else 0.0) for i in range(48)])
    y[10] = 0.0
    valid = np.ones(48, dtype=bool)
    valid[10] = False
    with tempfile.TemporaryDirectory() as d:
        exe = build_c_caller(d)
        for model, period in (("AutoETS", 7), ("Naive", 0), ("AutoARIMA", 7), ("HoltWinters", 12), ("Holt", 0), ("SeasonalNaive", 7)):
            out = subprocess.run([exe, model, str(period)], capture_output=True, text=True, timeout=300).stdout.split("\n")
            oo = O.make_options(model, 5, confidence_level=0.90, seasonal_period=period, auto_detect=False)
            ref = O.forecast(y, oo, valid)
            assert ref["ok"], ref
            tag, rest = out[0].split(" ", 1)
            name, first, n, inside = rest.rsplit(" ", 3)
            assert tag == "OK" and name == ref["model_name"] and int(n) == 5 and inside == "1", (model, out)
            assert abs(float(first) - ref["point"][0]) <= 1e-9 * max(1.0, abs(ref["point"][0])), (model, out, ref["point"][0])
            assert out[1].startswith("VERSION 0.1.0"), out


def test_ts_backtest_native_caller(env):
    """SURVEY section 8f rank 1, second caller (_ts_backtest_native, ts_backtest_native.cpp): position-based walk-forward
    folds over the distinct dates of the whole input; every (fold, group) pair is one training series fitted with the
    zero-initialised options of `:768-776`; one batch call; against the oracle pair by pair."""
    api, O, lib, synth = env
    # the equivalence test's data (ts_backtest_equivalence.test:11-19): 60 days, two series, 2 folds x horizon 7, Naive
    i = np.arange(60)
    grp = np.array(["A"] * 60 + ["B"] * 60, dtype=object)
    ds = np.concatenate([np.datetime64("2024-01-01") + i.astype("timedelta64[D]")] * 2)
    val = np.concatenate([100.0 + i * 2.0 + 50 + (i % 7) * 3.0, 100.0 + i * 2.0 + (i % 7) * 3.0])
    out = api.ts_backtest_native(grp, ds, val, 7, 2, {"method": "Naive"}, "mae", group_name="series_id", date_name="date")
    assert list(out.keys()) == ["fold_id", "series_id", "date", "yhat", "actual", "error", "abs_error", "yhat_lower", "yhat_upper",
                                "model_name", "fold_metric_score"]
    assert len(out["yhat"]) == 28 and list(out["fold_id"]) == [1] * 14 + [2] * 14 and list(out["series_id"][:14]) == ["A"] * 7 + ["B"] * 7
    np.testing.assert_array_equal(out["yhat"][:7], np.full(7, val[45]))          # fold 1: train ends at day 45
    np.testing.assert_array_equal(out["yhat"][21:], np.full(7, val[60 + 52]))    # fold 2, series B: train ends at day 52
    np.testing.assert_array_equal(out["actual"][:7], val[46:53])
    assert out["date"][0] == np.datetime64("2024-02-16") and out["date"].dtype == ds.dtype
    np.testing.assert_array_equal(out["error"], out["yhat"] - out["actual"])
    np.testing.assert_array_equal(out["abs_error"], np.abs(out["error"]))
    assert out["fold_metric_score"][0] == api.backtest_metric("mae", out["actual"][:14], out["yhat"][:14], [], [])
    assert set(out["model_name"]) == {"Naive"}
    # ragged groups, shuffled rows, a NULL value, AutoETS and AutoARIMA: pair by pair against the oracle
    Y = synth.gen_series(synth.SEED_M5, 310, 5, 120, 7, positive=True)
    lens = [120, 120, 104, 90, 120]
    rows = [(f"g{g}", t, Y[g, t]) for g in range(5) for t in range(lens[g])]
    perm = np.random.default_rng(5).permutation(len(rows))
    grp = np.array([rows[k][0] for k in perm], dtype=object)
    t = np.array([rows[k][1] for k in perm], dtype=np.int64)
    v = np.array([rows[k][2] for k in perm], dtype=object)
    drop = int(np.flatnonzero((grp == "g4") & (t == 17))[0])
    v[drop] = None                                                  # dropped row: g4 is one point shorter and shifted
    series = {f"g{g}": Y[g, :lens[g]] for g in range(5)}
    series["g4"] = np.delete(Y[4], 17)
    tt = {f"g{g}": np.arange(lens[g]) for g in range(5)}
    tt["g4"] = np.delete(np.arange(120), 17)
    first_seen = list(dict.fromkeys(g for k, g in enumerate(grp) if k != drop))      # group_order: first appearance (`:596-600`)
    for method, params in (("AutoETS", {}), ("AutoARIMA", {"window_type": "fixed", "min_train_size": 80, "gap": 1})):
        params = dict(params, method=method)
        out = api.ts_backtest_native(grp, t, v, 6, 3, params, "rmse")
        bounds = api.backtest_fold_bounds(120, 6, 3, params.get("window_type", "expanding"), params.get("min_train_size", 1), params.get("gap", 0))
        assert len(bounds) == (3 if method == "AutoETS" else 2)        # gap 1 pushes the third test window past the data
        pos = 0
        for (fid, tr0, tr1, te0, te1) in bounds:
            start = pos
            for g in first_seen:
                yy = series[g]
                if tr1 >= len(yy) or te0 >= len(yy):
                    continue
                ref = O.forecast(yy[tr0:tr1 + 1], O.make_options(method, 6, confidence_level=0.0, auto_detect=False))
                assert ref["ok"], ref
                n = min(6, min(te1, len(yy) - 1) - te0 + 1)
                sl = slice(pos, pos + n)
                assert list(out["fold_id"][sl]) == [fid] * n and list(out["id"][sl]) == [g] * n, (method, fid, g)
                np.testing.assert_allclose(out["yhat"][sl], ref["point"][:n], rtol=REL_TOL)
                np.testing.assert_allclose(out["yhat_lower"][sl], ref["lower"][:n], rtol=REL_TOL)
                np.testing.assert_allclose(out["yhat_upper"][sl], ref["upper"][:n], rtol=REL_TOL)
                np.testing.assert_array_equal(out["actual"][sl], yy[te0:te0 + n])
                np.testing.assert_array_equal(out["date"][sl], tt[g][te0:te0 + n])
                assert out["model_name"][pos] == ref["model_name"]
                pos += n
            want = api.backtest_metric("rmse", out["actual"][start:pos], out["yhat"][start:pos], [], [])
            assert np.all(out["fold_metric_score"][start:pos] == want)
        assert pos == len(out["yhat"]) and pos > 0
    # failing fits are skipped, not raised (`:791-794`): an unknown method gives no rows
    out = api.ts_backtest_native(grp, t, v, 6, 3, {"method": "NoSuchModel"}, "rmse")
    assert len(out["yhat"]) == 0 and out["date"].dtype == np.int64
    # the binding glues "method:model" into the model field (`:776-781`), which the core's parser does not know: no rows either
    out = api.ts_backtest_native(grp, t, v, 6, 3, {"method": "ETS", "model": "AAA"}, "rmse")
    assert len(out["yhat"]) == 0
    r = api.forecast_series(series["g0"], lib.make_options("ETS:AAA", 3))
    assert not r["ok"] and r["code"] == lib.INVALID_MODEL and "Unknown model: 'ETS:AAA'" in r["message"]


def test_refused_allocation_is_reported_and_recoverable(env):
    """A plan that cannot fit the HBM (8M series x 10k-step horizon: 640 GB per forecast block) fails with ALLOCATION_ERROR (error.rs:20-21, code 4) before
    anything is launched, leaves nothing behind, and the next ordinary call gives the oracle's numbers."""
    import ctypes as C
    api, O, lib, synth = env
    L = lib.load()
    hb, err = C.c_void_p(), lib.AnofoxError()
    opts = lib.make_options("AutoETS", 10_000, seasonal_period=7)
    ok = L.anofox_hip_batch_create(8_000_000, 64, C.byref(opts), C.byref(hb), C.byref(err))
    assert not ok and err.code == lib.ALLOCATION_ERROR and err.message.startswith(b"Allocation error:"), (err.code, err.message)
    assert not hb.value
    ok = L.anofox_hip_batch_create(40_000_000, 16, C.byref(opts), C.byref(hb), C.byref(err))      # too wide for one launch
    assert not ok and err.code == lib.INTERNAL_ERROR and b"shard it" in err.message
    Y = synth.gen_series(synth.SEED_M5, 520, 8, 60, 7, positive=True)
    assert _compare(api, O, lib, list(Y), "AutoETS", 7, seasonal_period=7) <= REL_TOL


def test_reference_sql_pins_on_the_hip_path(env):
    """The 252 pins the reference's sqllogictest files hold on `_ts_forecast(values, horizon, model)` for the models on the
    path (tests/golden/reference_sql_pins.json; model names, lengths, orderings, tolerances), replayed through
    anofox_ts_forecast with the scalar's options (ts_forecast.cpp:406-411)."""
    import sql_pins
    api, O, lib, synth = env
    o = sql_pins.PINS["options"]

    def run(values, valid, horizon, model):
        opts = lib.make_options(model, horizon, seasonal_period=o["seasonal_period"], confidence_level=o["confidence_level"],
                                auto_detect=o["auto_detect"], include_fitted=o["include_fitted"], include_residuals=o["include_residuals"])
        return api.forecast_series(values, opts, valid)
    for case in sql_pins.PINS["cases"]:
        sql_pins.check_pin(case, run)


def _model_name_tables():
    """The two groups of test/sql/ts_native_model_names.test:14-27 and its single-fold CV table (`:45-53`)."""
    i = np.arange(60)
    y = np.concatenate([10.0 + i * 0.5 + np.sin(i * 3.14159 / 7) * 3, 20.0 + i * 0.3 + np.cos(i * 3.14159 / 7) * 2])
    grp = np.array(["G1"] * 60 + ["G2"] * 60, dtype=object)
    ds = np.concatenate([np.datetime64("2024-01-01T00:00:00", "us") + i.astype("timedelta64[D]")] * 2)
    return grp, ds, y, np.concatenate([i, i])


def test_native_model_names_replay(env):
    """test/sql/ts_native_model_names.test: the model_name column of _ts_forecast_native and _ts_cv_forecast_native for
    every model on the path with an empty params MAP (so the period is auto-detected), and the unknown-model errors."""
    api, O, lib, synth = env
    grp, ds, y, i = _model_name_tables()
    exact = ["Naive", "SMA", "SeasonalNaive", "SES", "SESOptimized", "RandomWalkDrift", "Holt", "HoltWinters", "SeasonalES",
             "SeasonalESOptimized", "ETS", "ARIMA"]
    for model in exact + ["AutoETS", "AutoARIMA"]:
        out = api.ts_forecast_by(grp, ds, y, model, 3, "1d", {})
        assert len(out["yhat"]) == 6, model
        names = set(out["model_name"])
        if model in exact:
            assert names == {model}, (model, names)
        else:
            assert all(n.startswith(model) for n in names), (model, names)
        for g in ("G1", "G2"):                                       # and the oracle agrees on names and numbers
            ref = O.forecast(y[grp == g], O.make_options(model, 3))
            sel = np.array(out["id"], dtype=object) == g
            assert ref["ok"] and set(np.array(out["model_name"], dtype=object)[sel]) == {ref["model_name"]}
            np.testing.assert_allclose(np.asarray(out["yhat"])[sel], ref["point"], rtol=REL_TOL)
    keep = i < 48                                                    # train: ds < 2024-02-15 (i < 45), test: the next three days
    split = np.where(i[keep] < 45, "train", "test").astype(object)
    fold = np.ones(int(keep.sum()), dtype=np.int64)
    for model in ["Naive", "SESOptimized", "RandomWalkDrift", "HoltWinters", "SeasonalESOptimized", "ETS", "ARIMA", "AutoARIMA"]:
        out = api.ts_cv_forecast_by(fold, split, grp[keep], ds[keep], y[keep], model, {})
        assert len(out["yhat"]) == 6 and set(out["split"]) == {"test"}
        names = set(out["model_name"])
        assert names == {model} if model != "AutoARIMA" else all(n.startswith("AutoARIMA") for n in names), (model, names)
    for bad in ("AIDA", "NotAModel"):
        with pytest.raises(api.InvalidInputException, match=f"Unknown model: '{bad}'"):
            api.ts_forecast_by(grp, ds, y, bad, 3, "1d", {})
        with pytest.raises(api.InvalidInputException, match=f"Unknown model: '{bad}'"):
            api.ts_cv_forecast_by(fold, split, grp[keep], ds[keep], y[keep], bad, {})


def test_ets_model_parameter_replay(env):
    """test/sql/ts_forecast_ets_model.test: ETS with and without a `model` parameter through ts_forecast_by (two groups of 84
    days, horizon 7 -> 14 rows), the four rejected notations and the two parameter errors."""
    api, O, lib, synth = env
    i = np.arange(84)
    y = np.concatenate([100.0 + i * 0.5 + np.sin(i * 2 * 3.14159 / 7) * 10, 200.0 + i * 0.3 + np.cos(i * 2 * 3.14159 / 7) * 15])
    grp = np.array(["A"] * 84 + ["B"] * 84, dtype=object)
    ds = np.concatenate([np.datetime64("2024-01-01T00:00:00", "us") + i.astype("timedelta64[D]")] * 2)
    for params in ({}, {"model": "AAA"}, {"model": "ANN"}, {"confidence_level": "0.95"}):
        out = api.ts_forecast_by(grp, ds, y, "ETS", 7, "1d", params)
        assert len(out["yhat"]) == 14 and np.all(np.asarray(out["yhat"]) > 0), params
        assert out["ds"][0] == np.datetime64("2024-03-25T00:00:00", "us")
        for g in ("A", "B"):
            oo = O.make_options("ETS", 7, ets_model=params.get("model", ""), confidence_level=float(params.get("confidence_level", 0.90)))
            ref = O.forecast(y[grp == g], oo)
            sel = np.array(out["id"], dtype=object) == g
            np.testing.assert_allclose(np.asarray(out["yhat"])[sel], ref["point"], rtol=REL_TOL)
            np.testing.assert_allclose(np.asarray(out["yhat_upper"])[sel], ref["upper"], rtol=REL_TOL)
    for spec, msg in (("XYZ", "Invalid ETS model specification"), ("123", "Invalid ETS model specification"), ("MAA", "unstable"), ("MAdA", "unstable")):
        with pytest.raises(api.InvalidInputException, match=msg):
            api.ts_forecast_by(grp, ds, y, "ETS", 7, "1d", {"model": spec})
    with pytest.raises(api.InvalidInputException, match="only valid when method='ETS'"):
        api.ts_forecast_by(grp, ds, y, "Naive", 7, "1d", {"model": "AAA"})
    with pytest.raises(api.InvalidInputException, match="Unknown parameter"):
        api.ts_forecast_by(grp, ds, y, "ETS", 7, "1d", {"methd": "AAA"})
    agg = api.ts_forecast_agg(grp, ds, y, "ETS", 7, {})
    assert len(agg) == 2


def test_ts_forecast_by_sql_replay(env):
    """test/sql/ts_forecast_by.test for the models on the path: row counts, distinct steps and ids, model names, forecast
    dates after the data, parameter maps (typed values), frequency spellings, interval ordering, horizons 1 and 30."""
    api, O, lib, synth = env
    i = np.arange(60)
    g_grp = np.array(["A"] * 60 + ["B"] * 60, dtype=object)
    g_ds = np.concatenate([np.datetime64("2024-01-01T00:00:00", "us") + i.astype("timedelta64[D]")] * 2)
    g_y = np.concatenate([10.0 + i * 0.5 + np.sin(i * 3.14159 / 7) * 2, 20.0 + i * 0.3 + np.cos(i * 3.14159 / 7) * 3])
    j = np.arange(84)
    s_grp = np.array(["S1"] * 84 + ["S2"] * 84, dtype=object)
    s_ds = np.concatenate([np.datetime64("2024-01-01T00:00:00", "us") + j.astype("timedelta64[D]")] * 2)
    s_y = np.concatenate([100 + np.sin(j * 2 * 3.14159 / 7) * 20 + j * 0.1, 200 + np.cos(j * 2 * 3.14159 / 7) * 30 + j * 0.2])
    G, S = (g_grp, g_ds, g_y), (s_grp, s_ds, s_y)
    out = api.ts_forecast_by(*G, "Naive", 5, "1d", {})
    assert len(out["yhat"]) == 10 and list(out.keys()) == ["id", "forecast_step", "ds", "yhat", "yhat_lower", "yhat_upper", "model_name"]
    out = api.ts_forecast_by(*G, "Naive", 7, "1d", {})
    assert len(set(out["forecast_step"])) == 7
    out = api.ts_forecast_by(*G, "Naive", 3, "1d", {})
    assert len(set(out["id"])) == 2 and len(out["yhat"]) == 6
    assert int(np.sum(out["ds"] > np.datetime64("2024-02-29T00:00:00", "us"))) == 6          # forecasts lie after the data
    assert int(np.sum((out["yhat_lower"] <= out["yhat"]) & (out["yhat"] <= out["yhat_upper"]))) == 6
    for table, model, h, rows in ((G, "Naive", 3, 6), (G, "SMA", 3, 6), (S, "SeasonalNaive", 7, 14), (G, "SES", 3, 6), (G, "SESOptimized", 3, 6),
                                  (G, "RandomWalkDrift", 3, 6), (G, "Holt", 3, 6), (S, "HoltWinters", 7, 14), (S, "SeasonalES", 7, 14),
                                  (S, "SeasonalESOptimized", 7, 14), (G, "ETS", 3, 6), (G, "ARIMA", 3, 6)):
        out = api.ts_forecast_by(*table, model, h, "1d", {})
        assert len(out["yhat"]) == rows and out["model_name"][0] == model, model
    for model in ("AutoETS", "AutoARIMA"):
        out = api.ts_forecast_by(*G, model, 3, "1d", {})
        assert len(out["yhat"]) == 6 and out["model_name"][0]
    assert len(api.ts_forecast_by(*G, "Naive", 3, "1d", {"confidence_level": 0.80})["yhat"]) == 6          # typed MAP values
    assert len(api.ts_forecast_by(*S, "SeasonalNaive", 7, "1d", {"seasonal_period": 7})["yhat"]) == 14
    assert len(api.ts_forecast_by(*S, "HoltWinters", 7, "1d", {"confidence_level": 0.95, "seasonal_period": 7})["yhat"]) == 14
    for freq in ("1d", "1 day", "1w"):
        assert len(api.ts_forecast_by(*G, "Naive", 3, freq, {})["yhat"]) == 6
    out = api.ts_forecast_by(*G, "Naive", 3, "1w", {})
    assert out["ds"][0] == np.datetime64("2024-03-07T00:00:00", "us")                          # last day 2024-02-29 + one week
    out = api.ts_forecast_by(*G, "Naive", 5, "1d", {})
    assert min(out["forecast_step"]) == 1 and max(out["forecast_step"]) == 5
    assert len(api.ts_forecast_by(*G, "Naive", 1, "1d", {})["yhat"]) == 2
    assert len(api.ts_forecast_by(*G, "Naive", 30, "1d", {})["yhat"]) == 60
    assert len(api.ts_forecast_by(*G, "ETS", 5, "1d", {"model": "AAA"})["yhat"]) == 10


def test_route_a_chunks_equal_route_b(env):
    """Round 6: `ts_forecast_by` on the batch route.  The shipped macro text reaches the scalar `_ts_forecast_scalar`
    (ts_macros.cpp:576-591); binding/ts_forecast_scalar_hip.cpp hands every chunk of <= 2,048 groups to ONE anofox_ts_forecast_batch,
    binding/ts_macros_hip.cpp points the macro at `_ts_forecast_native` (one batch per statement).  Both mirrors
    (api.ts_forecast_by_scalar_route / api.ts_forecast_by) run the data of test/sql/ts_forecast_by.test and a ragged 300-group table
    through the C-ABI: the same rows, bit for bit, whatever the chunk size -- a forecast does not depend on its batch's company."""
    api, O, lib, synth = env
    i = np.arange(60)
    g_grp = np.array(["A"] * 60 + ["B"] * 60, dtype=object)
    g_ds = np.concatenate([np.datetime64("2024-01-01T00:00:00", "us") + i.astype("timedelta64[D]")] * 2)
    g_y = np.concatenate([10.0 + i * 0.5 + np.sin(i * 3.14159 / 7) * 2, 20.0 + i * 0.3 + np.cos(i * 3.14159 / 7) * 3])
    j = np.arange(84)
    s_grp = np.array(["S1"] * 84 + ["S2"] * 84, dtype=object)
    s_ds = np.concatenate([np.datetime64("2024-01-01T00:00:00", "us") + j.astype("timedelta64[D]")] * 2)
    s_y = np.concatenate([100 + np.sin(j * 2 * 3.14159 / 7) * 20 + j * 0.1, 200 + np.cos(j * 2 * 3.14159 / 7) * 30 + j * 0.2])

    def same(a, b, what):
        assert list(a.keys())[1:] == ["forecast_step", "ds", "yhat", "yhat_lower", "yhat_upper", "model_name"], what
        assert list(a["id"]) == list(b["id"]) and list(a["model_name"]) == list(b["model_name"]), what
        assert np.array_equal(a["forecast_step"], b["forecast_step"]) and np.array_equal(a["ds"], b["ds"]), what
        for c in ("yhat", "yhat_lower", "yhat_upper"):
            assert np.array_equal(a[c], b[c], equal_nan=True), (what, c)

    for table, model, h, params in (((g_grp, g_ds, g_y), "Naive", 5, {}), ((g_grp, g_ds, g_y), "AutoETS", 3, {}),
                                    ((g_grp, g_ds, g_y), "AutoARIMA", 3, {}), ((g_grp, g_ds, g_y), "ETS", 5, {"model": "AAA"}),
                                    ((s_grp, s_ds, s_y), "HoltWinters", 7, {"confidence_level": 0.95, "seasonal_period": 7}),
                                    ((s_grp, s_ds, s_y), "AutoETS", 7, {"seasonal_period": "7"}),
                                    ((s_grp, s_ds, s_y), "SeasonalNaive", 7, {"seasonal_period": 7})):
        b_rows = api.ts_forecast_by(*table, model, h, "1d", params, date_name="ds")
        for chunk in (2048, 1):
            same(api.ts_forecast_by_scalar_route(*table, model, h, "1d", params, chunk_groups=chunk), b_rows, (model, chunk))
    # a ragged table: 300 groups of 20..400 observations with NULL targets, shuffled rows, chunks of 2,048 / 128 / 7 groups
    rng = np.random.default_rng(606)
    lens = rng.integers(20, 400, 300)
    Y = synth.gen_series(synth.SEED_M5, 606, 300, 400, 7, positive=True)
    grp = np.concatenate([np.full(n, f"g{k:03d}", dtype=object) for k, n in enumerate(lens)])
    ds = np.concatenate([np.datetime64("2023-01-01", "D") + np.arange(n) for n in lens])
    y = np.ma.array(np.concatenate([Y[k, :n] for k, n in enumerate(lens)]), mask=rng.random(int(lens.sum())) < 0.02)
    perm = rng.permutation(len(grp))
    for model, params in (("AutoETS", {"seasonal_period": "7"}), ("Holt", {}), ("AutoETS", {})):
        b_rows = api.ts_forecast_by(grp[perm], ds[perm], y[perm], model, 14, "1d", params, date_name="ds")
        assert len(b_rows["yhat"]) == 300 * 14
        for chunk in (2048, 128, 7):
            same(api.ts_forecast_by_scalar_route(grp[perm], ds[perm], y[perm], model, 14, "1d", params, chunk_groups=chunk), b_rows,
                 (model, chunk))
    # per-row arguments inside one chunk (only the scalar can take them): every row equals its own single-series call
    lists_d = [ds[grp == f"g{k:03d}"] for k in range(6)]
    lists_v = [y[grp == f"g{k:03d}"] for k in range(6)]
    methods = ["AutoETS", "Holt", "AutoETS", "Naive", "Holt", "AutoETS"]
    hz = [3, 5, 7, 2, 4, 6]
    prm = [{"seasonal_period": "7"}, {}, {"seasonal_period": "7"}, {}, {"confidence_level": "0.8"}, {"seasonal_period": "7"}]
    got = api.ts_forecast_scalar(lists_d, lists_v, hz, "1d", methods, prm)
    for k in range(6):
        one = api.ts_forecast_scalar([lists_d[k]], [lists_v[k]], hz[k], "1d", methods[k], prm[k])[0]
        assert len(got[k]["yhat"]) == hz[k] and got[k]["model_name"] == one["model_name"]
        for c in ("yhat", "yhat_lower", "yhat_upper", "ds"):
            assert np.array_equal(got[k][c], one[c]), (k, c)


def test_ts_forecast_params_sql_replay(env):
    """test/sql/ts_forecast_params.test: interval shape, SeasonalNaive / auto aliases on the LIST form, parameter maps and
    every frequency spelling through ts_forecast_by on DATE columns, the aggregate with and without params, horizons 0 / 1 / 24,
    and Holt-Winters on 10 observations with seasonal_period 7 (falls back to Holt: rows are returned)."""
    api, O, lib, synth = env
    trend = np.arange(10.0, 34.0, 2.0)
    seasonal = np.array([10.0, 20.0, 30.0] * 6)
    scalar = dict(seasonal_period=0, confidence_level=0.95, auto_detect=False, include_fitted=True, include_residuals=True)
    f = lambda v, h, model: api.forecast_series(v, lib.make_options(model, h, **scalar))
    r = f(trend, 3, "Naive")
    assert len(r["lower"]) == 3 and len(r["upper"]) == 3 and r["lower"][0] < r["point"][0] < r["upper"][0]
    r = f(seasonal, 3, "SeasonalNaive")
    assert r["model_name"] == "SeasonalNaive" and np.all(np.abs(r["point"] - 30.0) < 0.01) and len(f(seasonal, 6, "SeasonalNaive")["point"]) == 6
    for model in ("auto", "AutoETS", "AutoARIMA"):
        r = f(trend, 3, model)
        assert r["ok"] and len(r["point"]) == 3 and len(r["model_name"]) > 0
    for model, v, h in (("SESOptimized", trend, 3), ("Holt", trend, 3), ("HoltWinters", seasonal, 6), ("Naive", trend, 24), ("Naive", trend, 1)):
        assert len(f(v, h, model)["point"]) == h
    r = f(trend, 0, "Naive")
    assert r["ok"] and len(r["point"]) == 0
    assert len(f(np.array([10.0, 20.0, 30.0]), 1, "Naive")["point"]) == 1
    assert len(f(np.array([10.0, 20.0, 10.0, 20.0]), 2, "SeasonalNaive")["point"]) == 2
    r = f(trend, 3, "Naive")
    assert r["mse"] >= 0
    grp = np.array(["A"] * 10 + ["B"] * 10, dtype=object)
    ds = np.concatenate([np.datetime64("2024-01-01", "D") + np.arange(10).astype("timedelta64[D]")] * 2)
    y = np.concatenate([10.0 + 2.0 * np.arange(10), 100.0 + 5.0 * np.arange(10)])
    for model, params in (("Naive", {"confidence_level": 0.80}), ("SeasonalNaive", {"seasonal_period": 7}),
                          ("HoltWinters", {"confidence_level": 0.95, "seasonal_period": 7}), ("Naive", {})):
        out = api.ts_forecast_by(grp, ds, y, model, 3, "1d", params)
        assert len(out["yhat"]) == 6 and set(out["model_name"]) == {model}, (model, params)
        assert out["ds"].dtype == ds.dtype and out["ds"][0] == np.datetime64("2024-01-11", "D")
    hw = api.ts_forecast_by(grp, ds, y, "HoltWinters", 3, "1d", {"confidence_level": 0.95, "seasonal_period": 7})
    holt = api.ts_forecast_by(grp, ds, y, "Holt", 3, "1d", {"confidence_level": 0.95})
    np.testing.assert_array_equal(hw["yhat"], holt["yhat"])          # fewer than two seasons: Holt's numbers under the HoltWinters name
    for freq in ("1d", "1 day", "1w", "1h", "30m", "1mo", "1q", "1y"):
        assert len(api.ts_forecast_by(grp, ds, y, "Naive", 3, freq, {})["yhat"]) == 6, freq
    for params in ({"confidence_level": 0.80}, {}):
        assert len(api.ts_forecast_agg(grp, ds, y, "Naive", 3, params)) == 2


def test_param_grid_sql_replay(env):
    """test/sql/ts_forecast_param_grid.test for the models on the path: every row non-NULL for SMA windows, seasonal periods,
    five ETS notations and horizons 1..20 (values arrive as strings like the SQL MAP does), and the three
    "different parameter -> different forecast" checks; each configuration also against the oracle."""
    api, O, lib, synth = env
    grp, ds, y, _ = _model_name_tables()

    def run(model, h, params):
        out = api.ts_forecast_by(grp, ds, y, model, h, "1d", params)
        assert len(out["yhat"]) == 2 * h and np.all(np.isfinite(out["yhat"])), (model, params)
        oo = O.make_options(model, h, ets_model=params.get("model", ""), seasonal_period=int(params.get("seasonal_period", 0)),
                            window=int(params.get("window", 0)))
        for g in ("G1", "G2"):
            ref = O.forecast(y[grp == g], oo)
            np.testing.assert_allclose(np.asarray(out["yhat"])[np.array(out["id"], dtype=object) == g], ref["point"], rtol=REL_TOL)
        return round(float(np.sum(out["yhat"])), 4)
    sums = {w: run("SMA", 3, {"window": str(w)}) for w in (3, 5, 12, 30)}
    assert sums[3] != sums[30]
    sums = {p: run("SeasonalNaive", 3, {"seasonal_period": str(p)}) for p in (4, 7, 12)}
    assert sums[4] != sums[12]
    for model in ("HoltWinters", "SeasonalES", "SeasonalESOptimized", "AutoETS", "AutoARIMA"):
        run(model, 3, {"seasonal_period": "7"})
    sums = {spec: run("ETS", 3, {"model": spec, "seasonal_period": "7"}) for spec in ("AAA", "ANA", "MNM", "AAdA", "MAdM")}
    assert sums["AAA"] != sums["ANA"]
    for h in (1, 3, 10, 20):
        run("Naive", h, {})


def test_ts_forecast_agg_sql_replay(env):
    """test/sql/ts_forecast_agg.test for the models on the path: one struct per group, the nine fields and their lengths,
    steps 1..h, timestamps after the data, '' as the success message, model names, interval ordering, the Naive value."""
    api, O, lib, synth = env
    i = np.arange(30)
    t0 = np.datetime64("2024-01-01T00:00:00", "us")
    sales = (np.array(["A"] * 30 + ["B"] * 30, dtype=object), np.concatenate([t0 + i.astype("timedelta64[D]")] * 2),
             np.concatenate([10.0 + i * 0.5, 20.0 + i * 0.3]))
    k = np.arange(40)
    single = (np.array(["only"] * 40, dtype=object), t0 + k.astype("timedelta64[D]"), 50.0 + k * 0.2)
    j = np.arange(56)
    s1 = (np.array(["S1"] * 56, dtype=object), t0 + j.astype("timedelta64[D]"), 100 + np.sin(j * 2 * 3.14159 / 7) * 20 + j * 0.1)
    out = api.ts_forecast_agg(*sales, "Naive", 5, {})
    assert len(out) == 2 and [len(out[g]["point_forecast"]) for g in ("A", "B")] == [5, 5]
    assert out["A"]["point_forecast"][0] != out["B"]["point_forecast"][0]
    f = api.ts_forecast_agg(*single, "Naive", 5, {})["only"]
    assert list(f.keys()) == ["forecast_step", "forecast_timestamp", "point_forecast", "lower_90", "upper_90", "model_name", "insample_fitted",
                              "date_col_name", "error_message"]
    assert [len(f[c]) for c in ("forecast_step", "forecast_timestamp", "point_forecast", "lower_90", "upper_90")] == [5] * 5
    assert f["model_name"] == "Naive" and len(f["insample_fitted"]) == 40 and f["date_col_name"] is not None and f["error_message"] == ""
    assert f["forecast_step"][0] == 1 and f["forecast_step"][4] == 5
    assert f["forecast_timestamp"][0] > (np.datetime64("2024-02-09T00:00:00", "us") - np.datetime64(0, "us")).astype(np.int64)
    assert np.all((f["lower_90"] <= f["point_forecast"]) & (f["point_forecast"] <= f["upper_90"]))
    for model in ("Naive", "SMA", "SES", "SESOptimized", "RandomWalkDrift", "Holt", "ETS", "ARIMA"):
        assert api.ts_forecast_agg(*single, model, 3, {})["only"]["model_name"] == model
    for model in ("AutoETS", "AutoARIMA"):
        assert api.ts_forecast_agg(*single, model, 3, {})["only"]["model_name"]
    assert api.ts_forecast_agg(*s1, "HoltWinters", 7, {})["S1"]["model_name"] == "HoltWinters"
    assert [api.ts_forecast_agg(*sales, "SES", 3, {})[g]["model_name"] for g in ("A", "B")] == ["SES", "SES"]
    assert [len(api.ts_forecast_agg(*sales, "SES", 3, {})[g]["insample_fitted"]) for g in ("A", "B")] == [30, 30]
    for h in (1, 12, 24):
        assert len(api.ts_forecast_agg(*single, "Naive", h, {})["only"]["point_forecast"]) == h
    assert abs(api.ts_forecast_agg(*single, "Naive", 1, {})["only"]["point_forecast"][0] - 57.8) < 0.1


def test_inspect_explain_sql_replay(env):
    """test/sql/ts_forecast_inspect_explain.test for the models on the path: the three-product monthly panel (`:13-19`), the
    inspection fields of AutoETS / AutoARIMA (incl. the pinned order_p = 0), the ETS decomposition lengths for horizons
    12 / 8, and the four "does not implement" errors."""
    api, O, lib, synth = env
    k = np.arange(60)
    one = 10.0 + 3.0 * np.sin(2 * np.pi * (k % 12) / 12.0) + 0.05 * k
    grp = np.repeat(np.array(["A", "B", "C"], dtype=object), 60)
    months = np.datetime64("2024-01", "M") + k.astype("timedelta64[M]")
    ds = np.tile(months.astype("datetime64[D]"), 3)
    y = np.tile(one, 3)
    insp = api.ts_forecast_inspect_by(grp, ds, y, "AutoETS", {"seasonal_period": 12})
    assert sorted(insp) == ["A", "B", "C"]
    for g in "ABC":
        assert insp[g]["model_family"] == "Ets" and insp[g]["seasonal_period"] == 12 and len(insp[g]["fitted_values"]) == 60
    assert insp["A"]["spec"] is not None and insp["A"]["order_p"] is None
    ar = api.ts_forecast_inspect_by(grp, ds, y, "AutoARIMA", {"seasonal_period": 12})
    for g in "ABC":
        assert ar[g]["model_family"] == "Arima" and ar[g]["aic"] is not None
    assert ar["A"]["spec"] is None and ar["A"]["order_p"] == 0
    for model, params in (("Naive", {}), ("SeasonalNaive", {"seasonal_period": 12})):
        with pytest.raises(api.InvalidInputException, match="does not implement Inspectable"):
            api.ts_forecast_inspect_by(grp, ds, y, model, params)
    for h in (12, 8):
        ex = api.ts_forecast_explain_by(grp, ds, y, "ETS", h, {"seasonal_period": 12})
        for g in "ABC":
            assert ex[g]["horizon"] == h and len(ex[g]["level"]) == h and len(ex[g]["trend"]) == h and len(ex[g]["seasonal"]) == h
    for model, params in (("AutoETS", {"seasonal_period": 12}), ("Naive", {})):
        with pytest.raises(api.InvalidInputException, match="does not implement Explainable"):
            api.ts_forecast_explain_by(grp, ds, y, model, 12, params)


def test_error_isolation_batches_sql_replay(env):
    """test/sql/ts_forecast_error_isolation.test, the table-driven parts: the 100-series mixed batch (`:218-280`: 85 structs,
    80 three-point forecasts), the 1000-series scale batch (`:287-355`: 960 five-point forecasts) and the consecutive-error
    rows (`:371-395`), each as ONE call of the batch entry with the scalar's default model and options."""
    api, O, lib, synth = env
    opts = lambda h: lib.make_options("auto", h, seasonal_period=0, confidence_level=0.95, auto_detect=False, include_fitted=True,
                                      include_residuals=True)
    series, valids = [], []
    for i in range(1, 101):
        if i % 10 == 0:
            v, ok = [], []
        elif i % 20 == 5:
            v, ok = [1.0], [True]
        elif i % 20 == 15:
            v, ok = [0.0] * 5, [False] * 5
        else:
            v, ok = [float(i + k) for k in range(10)], [True] * 10
        series.append(np.array(v)); valids.append(np.array(ok, dtype=bool))
    got, berr = api.forecast_batch(series, opts(3), valids)
    assert berr["ok"]
    assert sum(r["ok"] for r in got) == 85
    assert all(not r["ok"] and r["code"] == lib.INSUFFICIENT_DATA for r, s in zip(got, series) if len(s) <= 1)
    assert sum(r["ok"] and len(r["point"]) == 3 for r, s in zip(got, series) if len(s) == 10) == 80
    assert all(r["ok"] and np.all(np.isnan(r["point"])) for r, s in zip(got, series) if len(s) == 5)      # all-NULL: NaN forecasts, not an error
    for idx in (0, 14, 98):                                                                                # and the oracle agrees
        ref = O.forecast(series[idx], O.make_options("auto", 3, seasonal_period=0, confidence_level=0.95, auto_detect=False), valids[idx])
        assert ref["ok"] == got[idx]["ok"] and np.array_equal(ref["point"], got[idx]["point"], equal_nan=True)
    series = []
    for i in range(1, 1001):
        if i % 50 == 0:
            v = []
        elif i % 50 == 25:
            v = [42.0]
        elif i % 100 == 50:
            v = [1.0, 2.0]
        elif i % 20 == 10:
            v = [1.0, 2.0, 3.0]
        else:
            v = [float(np.sin(i + k) * 10) for k in range(10)]
        series.append(np.array(v))
    got, berr = api.forecast_batch(series, opts(5))
    assert berr["ok"] and sum(r["ok"] and len(r["point"]) == 5 for r in got) == 960
    assert got[0]["ok"] and got[998]["ok"]
    ten = np.arange(1.0, 11.0)
    rows = [ten, np.array([]), np.array([]), np.array([]), ten, np.array([1.0]), ten]
    got, berr = api.forecast_batch(rows, opts(3))
    assert [r["ok"] for r in got] == [True, False, False, False, True, False, True]
    assert [len(r["point"]) for r in got if r["ok"]] == [3, 3, 3]
    # the five-row table of `:787-793` under every model on the path (`:795-895`), and the seasonal one of `:899-906` (`:909-918`)
    table = [ten, np.array([]), ten[::-1].copy(), np.array([1.0]), np.full(10, 5.0)]
    for model in ("NAIVE", "SMA", "SES", "RandomWalkDrift", "Holt", "AutoETS", "AutoARIMA"):
        o = lib.make_options(model, 3, seasonal_period=0, confidence_level=0.95, auto_detect=False, include_fitted=True, include_residuals=True)
        got, berr = api.forecast_batch(table, o)
        assert berr["ok"] and [r["ok"] for r in got] == [True, False, True, False, True], model
    seasonal = [np.array([1.0, 2.0, 3.0, 4.0] * 3), np.array([]), np.array([4.0, 3.0, 2.0, 1.0] * 3), np.array([1.0, 2.0]), np.full(12, 5.0)]
    o = lib.make_options("HoltWinters", 4, seasonal_period=0, confidence_level=0.95, auto_detect=False, include_fitted=True, include_residuals=True)
    got, berr = api.forecast_batch(seasonal, o)
    assert berr["ok"] and [r["ok"] for r in got] == [True, False, True, False, True]
    scen = {"empty": ([], []), "single": ([1.0], [1]), "double": ([1.0, 2.0], [1, 1]), "triple": ([1.0, 2.0, 3.0], [1, 1, 1]),
            "all_null": ([0.0] * 5, [0] * 5), "mostly_null": ([0, 0, 5.0, 0, 0], [0, 0, 1, 0, 0]), "constant_zero": ([0.0] * 5, [1] * 5),
            "constant_pos": ([5.0] * 5, [1] * 5), "constant_neg": ([-5.0] * 5, [1] * 5), "valid_trend": (list(ten), [1] * 10)}      # `:741-753`
    got, berr = api.forecast_batch([np.array(v, dtype=float) for v, _ in scen.values()], opts(3), [np.array(k, dtype=bool) for _, k in scen.values()])
    assert sorted(name for name, r in zip(scen, got) if not r["ok"]) == ["double", "empty", "single"]


def test_native_param_validation_sql_replay(env):
    """test/sql/ts_native_param_validation.test for the models on the path: the bind-time and model-time errors (tested
    substrings) of _ts_forecast_native and _ts_cv_forecast_native, and the configurations that must return all 6 rows."""
    api, O, lib, synth = env
    grp, ds, y, i = _model_name_tables()
    keep = i < 48
    split = np.where(i[keep] < 45, "train", "test").astype(object)
    fold = np.ones(int(keep.sum()), dtype=np.int64)
    fb = lambda model, params: api.ts_forecast_by(grp, ds, y, model, 3, "1d", params)
    cv = lambda model, params: api.ts_cv_forecast_by(fold, split, grp[keep], ds[keep], y[keep], model, params)
    bad = [(fb, "AutoETS", {"methd": "ETS"}, "Unknown parameter"), (fb, "AutoETS", {"foo": "1", "bar": "2"}, "Unknown parameter"),
           (cv, "AutoETS", {"methd": "ETS"}, "Unknown parameter"),
           (fb, "AutoETS", {"confidence_level": "0.0"}, "Invalid confidence_level"), (fb, "AutoETS", {"confidence_level": "-0.5"}, "Invalid confidence_level"),
           (fb, "AutoETS", {"confidence_level": "1.0"}, "Invalid confidence_level"), (fb, "AutoETS", {"confidence_level": "5.0"}, "Invalid confidence_level"),
           (cv, "AutoETS", {"confidence_level": "1.5"}, "Invalid confidence_level"),
           (fb, "Naive", {"model": "AAA"}, "only valid when method='ETS'"), (cv, "Holt", {"model": "ANA"}, "only valid when method='ETS'"),
           (fb, "ETS", {"model": "XYZ"}, "Invalid ETS model specification"), (fb, "ETS", {"model": "AAAAA"}, "Invalid ETS model specification"),
           (fb, "ETS", {"model": "MAA"}, "unstable"), (fb, "ETS", {"model": "MAdA"}, "unstable"),
           (fb, "Naive", {"seasonal_period": "7"}, "does not use seasonal_period"), (fb, "SES", {"seasonal_period": "7"}, "does not use seasonal_period"),
           (fb, "SMA", {"window": "-1"}, "must be a positive integer"), (fb, "Naive", {"window": "5"}, "only valid when method='SMA'"),
           (fb, "Naive", {"seasonal_periods": "[7]"}, "only valid for multi-seasonal models")]
    for call, model, params, msg in bad:
        with pytest.raises(api.InvalidInputException, match=msg):
            call(model, params)
    for model, params in (("ETS", {"model": "AAA"}), ("ETS", {}), ("SeasonalNaive", {"seasonal_period": "7"}), ("HoltWinters", {"seasonal_period": "7"}),
                          ("AutoETS", {"confidence_level": "0.95"}), ("AutoETS", {}), ("SMA", {"window": "12"})):
        assert len(fb(model, params)["yhat"]) == 6, (model, params)


def test_cv_forecast_sql_replay_rest(env):
    """The remaining pins of test/sql/ts_cv_forecast.test: DATE dates keep their type (`:118-131`), the Drift alias names its
    model and differs from Naive (`:139-163`), two series of different length (`:169-190`), one fold, horizon 1, and
    SeasonalNaive with a STRUCT-style typed parameter (`:243`)."""
    api, O, lib, synth = env
    rows = _cv_folds("A", 24, 3, 4)
    fold, split, grp, t = (np.array(c, dtype=object) for c in zip(*rows))
    ds = np.array([int(x) + 1 for x in t], dtype=np.int32)
    y = 10.0 + ds
    naive = api.ts_cv_forecast_by(fold, split, grp, ds, y, "Naive", {}, group_name="series_id", date_name="ds")
    drift = api.ts_cv_forecast_by(fold, split, grp, ds, y, "Drift", {}, group_name="series_id", date_name="ds")
    assert set(drift["model_name"]) == {"RandomWalkDrift"} and np.any(np.abs(naive["yhat"] - drift["yhat"]) > 1e-4)
    assert np.mean(np.abs(naive["yhat"] - naive["y"])) > 0 and not np.any(np.isnan(naive["yhat"]))
    sn = api.ts_cv_forecast_by(fold, split, grp, ds, y, "SeasonalNaive", {"seasonal_period": 4})
    assert len(sn["yhat"]) == 12
    rows = _cv_folds("A", 18, 2, 3)
    fold_d, split_d, grp_d, t_d = (np.array(c, dtype=object) for c in zip(*rows))
    dates = np.datetime64("2023-01-02", "D") + np.array([int(x) for x in t_d]).astype("timedelta64[D]")
    out = api.ts_cv_forecast_by(fold_d, split_d, grp_d, dates, np.array([float(x) + 1 for x in t_d]), "Naive", {})
    assert out["date"].dtype == np.dtype("datetime64[D]") and len(out["yhat"]) == 6
    rows = _cv_folds("A", 20, 2, 3) + _cv_folds("B", 25, 2, 3)
    f2, s2, g2, t2 = (np.array(c, dtype=object) for c in zip(*rows))
    d2 = np.array([int(x) + 1 for x in t2], dtype=np.int32)
    out = api.ts_cv_forecast_by(f2, s2, g2, d2, 10.0 + d2, "Naive", {}, group_name="series_id")
    counts = {}
    for g, k in zip(out["series_id"], out["fold_id"]):
        counts[(g, int(k))] = counts.get((g, int(k)), 0) + 1
    assert counts == {("A", 1): 3, ("A", 2): 3, ("B", 1): 3, ("B", 2): 3}
    for n_folds, h, want in ((1, 6, {1: 6}), (3, 1, {1: 1, 2: 1, 3: 1})):
        rows = _cv_folds("A", 24, n_folds, h)
        f3, s3, g3, t3 = (np.array(c, dtype=object) for c in zip(*rows))
        d3 = np.array([int(x) + 1 for x in t3], dtype=np.int32)
        out = api.ts_cv_forecast_by(f3, s3, g3, d3, 10.0 + d3, "Naive", {})
        got = {}
        for k in out["fold_id"]:
            got[int(k)] = got.get(int(k), 0) + 1
        assert got == want


def test_wrapper_unit_tests_on_the_hip_path(env):
    """The assertions of the reference wrapper's own unit tests (forecast.rs `mod tests`, transcribed in
    tests/golden/reference_kats.json under "unit") through anofox_ts_forecast."""
    import json, os, sql_pins
    api, O, lib, synth = env
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")))

    def run(values, valid, horizon, model, o):
        opts = lib.make_options(model, horizon, seasonal_period=o.get("seasonal_period", 0), confidence_level=o.get("confidence_level", 0.90),
                                auto_detect=o.get("auto_detect"), include_fitted=o.get("include_fitted", False),
                                include_residuals=o.get("include_residuals", False), ets_model=o.get("ets_model", ""), window=o.get("window", 0))
        return api.forecast_series(values, opts, valid)
    assert len(gold["unit"]) >= 15
    for case in gold["unit"]:
        sql_pins.check_unit_case(case, run)


def test_frequency_varchar_and_alias_sql_replay(env):
    """test/sql/ts_integer_frequency.test:120-139 (DATE column with a gap, lower-case method, frequency given as '1d', '1 day'
    or the integer 1), ts_varchar_edge_cases.test:56-70 (VARCHAR target column: the macro's ::DOUBLE cast) and
    ts_table_macro_aliases.test:23-27 (the anofox_fcst_ aliases)."""
    api, O, lib, synth = env
    grp = np.array(["A"] * 4, dtype=object)
    ds = np.array(["2023-01-01", "2023-01-02", "2023-01-04", "2023-01-05"], dtype="datetime64[D]")
    val = np.array([10.0, 20.0, 30.0, 40.0])
    for freq in ("1d", "1 day", 1):
        out = api.ts_forecast_by(grp, ds, val, "naive", 2, freq, {})
        assert len(out["yhat"]) == 2 and list(out["ds"]) == [np.datetime64("2023-01-06"), np.datetime64("2023-01-07")] and set(out["model_name"]) == {"Naive"}
    i = np.arange(60)
    g2 = np.array(["A"] * 60 + ["B"] * 60, dtype=object)
    d2 = np.concatenate([np.datetime64("2024-01-01T00:00:00", "us") + i.astype("timedelta64[D]")] * 2)
    y_num = np.concatenate([100.0 + i * 0.5, 200.0 + i * 0.25])
    y_txt = np.array([repr(float(v)) for v in y_num], dtype=object)                      # the VARCHAR column
    a = api.ts_forecast_by(g2, d2, y_txt, "Naive", 5, "1d", {})
    b = api.ts_forecast_by(g2, d2, y_num, "Naive", 5, "1d", {})
    assert len(a["yhat"]) == 10 and a["yhat"].dtype == np.float64 and np.array_equal(a["yhat"], b["yhat"])
    assert len(api.anofox_fcst_ts_forecast_by(grp, ds, val, "Naive", 3, "1d")["yhat"]) == 3
    assert api.anofox_fcst_ts_forecast_agg is api.ts_forecast_agg and api.anofox_fcst_ts_cv_forecast_by is api.ts_cv_forecast_by
    assert api.anofox_fcst_ts_forecast_inspect_by is api.ts_forecast_inspect_by and api.anofox_fcst_ts_forecast_explain_by is api.ts_forecast_explain_by


def test_parallel_correctness_sql_replay(env, monkeypatch):
    """test/sql/ts_parallel_correctness.test:13-52: 50 series x 60 days through ts_forecast_by give the same 350 rows whatever
    the degree of parallelism.  Here: all host packer threads vs one, the whole batch vs one series at a time, and shuffled
    input rows vs ordered ones -- bit-identical for Naive and for AutoETS."""
    api, O, lib, synth = env
    day = np.tile(np.arange(60), 50)
    num = np.repeat(np.arange(1, 51), 60)
    grp = np.array([f"series_{k:03d}" for k in num], dtype=object)
    ds = np.datetime64("2024-01-01", "D") + day.astype("timedelta64[D]")
    y = 100.0 + num * 10.0 + day * 0.5 + (day % 7) * 3.0
    for model, params in (("Naive", {}), ("AutoETS", {"seasonal_period": 7})):
        monkeypatch.delenv("ANOFOX_HIP_PACK_THREADS", raising=False)
        many = api.ts_forecast_by(grp, ds, y, model, 7, "1d", params, group_name="unique_id")
        assert len(many["yhat"]) == 350 and len(set(many["unique_id"])) == 50
        monkeypatch.setenv("ANOFOX_HIP_PACK_THREADS", "1")
        one = api.ts_forecast_by(grp, ds, y, model, 7, "1d", params, group_name="unique_id")
        for c in ("yhat", "yhat_lower", "yhat_upper"):
            assert np.array_equal(many[c], one[c]), (model, c)
        assert list(many["model_name"]) == list(one["model_name"]) and np.array_equal(many["ds"], one["ds"])
        perm = np.random.default_rng(11).permutation(len(y))
        shuf = api.ts_forecast_by(grp[perm], ds[perm], y[perm], model, 7, "1d", params, group_name="unique_id")
        key = lambda o: sorted(zip(o["unique_id"], o["forecast_step"], o["yhat"], o["model_name"]))
        assert key(shuf) == key(many)
        for k in (1, 25, 50):                                                   # a series does not see its neighbours
            sel = num == k
            solo = api.ts_forecast_by(grp[sel], ds[sel], y[sel], model, 7, "1d", params, group_name="unique_id")
            assert np.array_equal(solo["yhat"], np.asarray(many["yhat"])[np.array(many["unique_id"], dtype=object) == f"series_{k:03d}"])


def test_cv_backtest_sql_replay(env):
    """test/sql/ts_cv_backtest.test: a table without fold columns is rejected with the reference's message; 3 folds x horizon 7
    over two 84-day series give 42 rows, 3 folds, 2 series per fold, for Naive and for AutoETS."""
    api, O, lib, synth = env
    i = np.arange(84)
    val = {"A": 100.0 + i * 0.5 + np.sin(i * 2 * 3.14159 / 7) * 10, "B": 200.0 + i * 0.3 + np.cos(i * 2 * 3.14159 / 7) * 5}
    t0 = np.datetime64("2024-01-01T00:00:00", "us")
    with pytest.raises(api.InvalidInputException, match="missing required columns 'fold_id' and/or 'split'"):
        api.ts_cv_forecast_by(None, None, np.array(["A"] * 84, dtype=object), t0 + i.astype("timedelta64[D]"), val["A"], "Naive", {})
    rows = _cv_folds("A", 84, 3, 7) + _cv_folds("B", 84, 3, 7)
    fold, split, grp, t = (np.array(c, dtype=object) for c in zip(*rows))
    ds = t0 + np.array([int(x) for x in t]).astype("timedelta64[D]")
    y = np.array([val[g][int(x)] for g, x in zip(grp, t)])
    for model in ("Naive", "AutoETS"):
        out = api.ts_cv_forecast_by(fold, split, grp, ds, y, model, {})
        assert len(out) == 9 and len(out["yhat"]) == 42 and not np.any(np.isnan(out["yhat"]))
        assert sorted(set(int(k) for k in out["fold_id"])) == [1, 2, 3]
        assert all(len(set(g for g, k in zip(out["id"], out["fold_id"]) if k == f)) == 2 for f in (1, 2, 3))


def test_default_chain_with_unsupported_detected_period(env):
    """Found by the fuzz: `ETS` without a spec (and the AutoETS fallback) plan Holt-Winters when two seasons fit
    (forecast.rs:1327-1336); with a detected period above the 64 the kernels supported at the time nothing ran and the series
    reported success with an empty forecast.  Long periods are fitted now (HBM ring): oracle and device agree, alone and
    inside a batch whose other periods differ."""
    api, O, lib, synth = env
    t = np.arange(150)
    long_period = 50.0 + 10.0 * np.sin(2 * np.pi * t / 70.0) + 0.01 * t            # detected period 70: two seasons fit into 150
    weekly = 20.0 + 5.0 * np.sin(2 * np.pi * t / 7.0) + 0.02 * t
    for model in ("ETS", "AutoETS"):
        opts = lib.make_options(model, 6)                                            # seasonal_period 0: detection on
        got, berr = api.forecast_batch([weekly, long_period, weekly[::-1].copy(), long_period * 2.0], opts)
        assert berr["ok"]
        for y, r in zip([weekly, long_period, weekly[::-1].copy(), long_period * 2.0], got):
            ref = O.forecast(y, O.make_options(model, 6))
            assert r["ok"] == ref["ok"], (model, r, ref)
            if ref["ok"]:
                assert r["model_name"] == ref["model_name"] and np.array_equal(r["point"], ref["point"])
            else:
                assert r["code"] == ref["code"]
        single = api.forecast_series(long_period, opts)
        ref = O.forecast(long_period, O.make_options(model, 6))
        assert single["ok"] == ref["ok"] and (single["ok"] or single["code"] == ref["code"])


@pytest.mark.gpu
def test_results_are_ordered_after_a_run_on_the_default_stream(env):
    """`DeviceBatch.run` is asynchronous and ordered on torch's current stream: a torch op enqueued there right after the run, with no
    host wait in between, reads the results of THAT run -- also when the current stream is the null stream, which cannot carry the
    run itself (the C entry reads a null handle as the batch's own non-blocking stream; device.py fences a side stream instead).
    The multi-rank step of bench.py relies on it: its gather follows the run on the default stream.  Two blocks, one batch: the
    clone taken right after the second run must hold the second block's forecasts, not the first's."""
    import torch
    from anofox_forecast_amd.device import DeviceBatch, pack_time_major
    api, O, lib, synth = env
    n, T, h, m = 4096, 400, 14, 7
    Y1 = synth.gen_series(synth.SEED_M5, 0, n, T, m, positive=True)
    Y2 = synth.gen_series(synth.SEED_M5, n, n, T, m, positive=True)
    dev = "cuda:0"
    assert torch.cuda.current_stream(torch.device(dev)).cuda_stream == 0          # the case under test
    b = DeviceBatch(n, T, lib.make_options("AutoETS", h, seasonal_period=m), dev)
    ln = torch.full((b.ld,), T, dtype=torch.int32, device=dev)
    ln[n:] = 0
    blocks = [torch.from_numpy(pack_time_major(Y, b.ld)).to(dev) for Y in (Y1, Y2)]
    seen = []
    for y in blocks:
        b.set_block(y, ln)
        b.run()
        seen.append({k: v.clone() for k, v in b.results().items()})        # enqueued behind the run: no host wait
    torch.cuda.synchronize()
    settled = {k: v.clone() for k, v in b.results().items()}               # the second block's results, after a device-wide wait
    for k in settled:
        assert torch.equal(seen[1][k], settled[k]), k
    assert not torch.equal(seen[0]["yhat"], seen[1]["yhat"])
    # and each is what a waited-for run of that block gives
    b.set_block(blocks[0], ln)
    b.run()
    torch.cuda.synchronize()
    first = b.results()
    for k in first:
        assert torch.equal(seen[0][k], first[k]), k
    b.close()
