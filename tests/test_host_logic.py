"""CPU: host logic of the operator mirror (frequency strings, forecast timestamps, bind validation,
validity masks) against the reference's binding behaviour."""
import os

import numpy as np
import pytest

from anofox_forecast_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parse_frequency():
    # src/table_functions/ts_fill_gaps_native.cpp:21-90
    P = api.parse_frequency
    assert (P("1d").seconds, P("1d").type) == (86400, "FIXED")
    assert P("2h").seconds == 7200 and P("30m").seconds == 1800 and P("15min").seconds == 900 and P("1w").seconds == 604800
    assert (P("1mo").seconds, P("1mo").type) == (1, "MONTHLY") and P("2q").type == "QUARTERLY" and P("1y").type == "YEARLY"
    assert P("1 day").seconds == 86400 and P("3 hours").seconds == 10800 and P("1 month").type == "MONTHLY"
    assert P("1 QUARTER").type == "QUARTERLY" and P("2 years").seconds == 2
    r = P(1)                       # integer literal (ts_integer_frequency.test:137)
    assert r.seconds == 1 and r.is_raw
    with pytest.raises(api.InvalidInputException, match="Invalid frequency"):
        P("fortnightly")


def test_forecast_dates():
    us = api._US_PER_DAY
    d = lambda s: int(np.datetime64(s, "D").astype(np.int64)) * us
    f = api.parse_frequency
    assert api.compute_forecast_date(d("2024-01-31"), 1, f("1mo"), "DATE") == d("2024-02-29")      # day clamp, leap year
    assert api.compute_forecast_date(d("2023-01-31"), 1, f("1mo"), "DATE") == d("2023-02-28")
    assert api.compute_forecast_date(d("2024-11-15"), 3, f("1mo"), "DATE") == d("2025-02-15")      # year rollover
    assert api.compute_forecast_date(d("2024-01-31"), 1, f("1q"), "DATE") == d("2024-04-30")
    assert api.compute_forecast_date(d("2024-02-29"), 1, f("1y"), "DATE") == d("2025-02-28")
    assert api.compute_forecast_date(d("2024-01-01"), 2, f("1d"), "TIMESTAMP") == d("2024-01-03")
    assert api.compute_forecast_date(d("2024-01-01"), 1, f("1"), "DATE") == d("2024-01-02")         # raw integer on a date column = days
    assert api.compute_forecast_date(100, 3, f("5"), "BIGINT") == 115                                # integer columns: raw units
    assert api.compute_forecast_date(100, 2, f("1d"), "INTEGER") == 100 + 2 * 86400


def test_bind_validation_messages():
    # src/table_functions/ts_forecast_native.cpp:357-399 and test/sql/ts_native_param_validation.test
    B = api.bind
    with pytest.raises(api.InvalidInputException, match="Unknown parameter"):
        B("ETS", 3, "1d", {"methd": "AAA"})
    for bad in ("1.5", "0", "-0.1", "1.0"):
        with pytest.raises(api.InvalidInputException, match="Invalid confidence_level"):
            B("AutoETS", 3, "1d", {"confidence_level": bad})
    with pytest.raises(api.InvalidInputException, match="only valid when method='ETS'"):
        B("Naive", 3, "1d", {"model": "AAA"})
    with pytest.raises(api.InvalidInputException, match="only valid when method='SMA'"):
        B("Naive", 3, "1d", {"window": "5"})
    with pytest.raises(api.InvalidInputException, match="positive integer"):
        B("SMA", 3, "1d", {"window": "-2"})
    with pytest.raises(api.InvalidInputException, match="only valid for multi-seasonal"):
        B("AutoETS", 3, "1d", {"seasonal_periods": "[7, 365]"})
    b = B("AutoETS", 28, "1d", {"seasonal_period": "7"})
    assert b.seasonal_period == 7 and b.confidence_level == 0.90 and b.horizon == 28
    o = api.options_from_bind(b)
    assert o.model == b"AutoETS" and o.seasonal_period == 7 and not o.auto_detect_seasonality and o.confidence_level == 0.90
    o = api.options_from_bind(B("AutoETS", 3, "1d", None))
    assert o.auto_detect_seasonality and o.seasonal_period == 0           # ts_forecast_scalar.cpp:450
    b = B("ETS", 3, "1d", {"model": "AAdA", "seasonal_period": 12, "confidence_level": 0.95})   # typed STRUCT values
    assert b.model_spec == "AAdA" and b.seasonal_period == 12 and b.confidence_level == 0.95


def test_validity_mask_bits():
    v = np.zeros(130, bool)
    v[[0, 63, 64, 129]] = True
    w = api.validity_mask(v)
    assert len(w) == 3 and int(w[0]) == (1 | (1 << 63)) and int(w[1]) == 1 and int(w[2]) == 2


def test_synthetic_generator_is_shard_consistent():
    from anofox_forecast_amd import synth
    a = synth.gen_series(synth.SEED_M5, 0, 2100, 64)
    b = synth.gen_series(synth.SEED_M5, 1000, 1100, 64)
    assert np.array_equal(a[1000:], b) and 0.3 < (a == 0).mean() < 0.8
    assert synth.gen_series(synth.SEED_M5, 5, 3, 64, positive=True).min() >= 1.0


def test_cv_collect_rules():
    """Collection step of _ts_cv_forecast_native (ts_cv_forecast_native.cpp:520-665): NULL fold/split/date rows dropped,
    NULL target -> 0.0, unknown split values ignored, both sides sorted by date, pairs lacking a side dropped."""
    from anofox_forecast_amd import api
    fold = np.array([1, 1, 1, 1, 1, 2, 2, None, 1, 3], dtype=object)
    split = np.array(["train", "test", "train", "valid", "train", "train", "train", "train", None, "test"], dtype=object)
    grp = np.array(["a"] * 10, dtype=object)
    ds = np.array([3, 9, 1, 5, 2, 1, 2, 7, 8, 4], dtype=np.int64)
    y = np.array([30.0, 90.0, 10.0, 50.0, None, 1.0, 2.0, 7.0, 8.0, 4.0], dtype=object)
    pairs, kind, dtype = api.cv_collect(fold, split, grp, ds, y)
    assert kind == "BIGINT" and dtype == np.int64
    assert len(pairs) == 1                                   # fold 2 has no test rows, fold 3 no train rows
    p = pairs[0]
    assert p["fold_id"] == 1 and p["group"] == "a"
    np.testing.assert_array_equal(p["train"], [10.0, 0.0, 30.0])     # sorted by date; NULL target counted as 0.0
    np.testing.assert_array_equal(p["test_us"], [9])
    np.testing.assert_array_equal(p["test_y"], [90.0])


def test_columnar_ingest_rules(hiplib):
    """anofox_hip_ingest_* (header block 4) against the collection rules of ts_forecast_native.cpp:476-610: NULL-date rows
    dropped, NULL targets kept as invalid slots, groups in first-appearance order, rows of a group stably sorted by date,
    chunked appends equivalent to one append.  Host code only: runs without a GPU."""
    from anofox_forecast_amd import api
    rng = np.random.default_rng(9)
    n = 5000
    gk = rng.integers(100, 140, n)
    dt = rng.integers(0, 400, n)                           # duplicates on purpose: equal dates keep arrival order
    val = rng.normal(size=n)
    dok = rng.random(n) > 0.02
    vok = rng.random(n) > 0.05
    ing = api.Ingest()
    for lo in range(0, n, 777):                            # DataChunk-like appends
        sl = slice(lo, min(lo + 777, n))
        ing.append(gk[sl], dt[sl], val[sl], dok[sl], vok[sl])
    ng, tmax = ing.finish()
    keep = np.nonzero(dok)[0]
    first = {}
    for i in keep:
        first.setdefault(int(gk[i]), len(first))
    assert ng == len(first) and list(ing.group_keys()) == list(first.keys())
    series = ing.series()
    lens = ing.lengths()
    assert tmax == max(lens)
    for g, key in enumerate(first):
        rows = keep[gk[keep] == key]
        rows = rows[np.argsort(dt[rows], kind="stable")]
        assert lens[g] == len(rows) and ing.last_dates()[g] == dt[rows[-1]]
        v, ok = series[g]
        np.testing.assert_array_equal(ok, vok[rows])
        np.testing.assert_array_equal(v[ok], val[rows][vok[rows]])
        assert np.all(v[~ok] == 0.0)
    ing.close()


def test_backtest_fold_bounds_and_metric():
    """ComputeFoldBoundaries / ComputeMetric (ts_backtest_native.cpp:623-711, 280-373), incl. the worked example of `:645-646`."""
    from anofox_forecast_amd import api
    assert api.backtest_fold_bounds(36, 12, 1) == [(1, 0, 23, 24, 35)]
    # 60 dates, 2 folds, horizon 7 (ts_backtest_equivalence.test:47-49): trains end at 45 and 52
    assert api.backtest_fold_bounds(60, 7, 2) == [(1, 0, 45, 46, 52), (2, 0, 52, 53, 59)]
    assert api.backtest_fold_bounds(1, 7, 2) == []
    # not enough dates: initial train of one point, folds stop at the data end
    assert api.backtest_fold_bounds(10, 4, 5) == [(1, 0, 0, 1, 4), (2, 0, 4, 5, 8)]
    assert api.backtest_fold_bounds(10, 4, 5, clip_horizon=True) == [(1, 0, 0, 1, 4), (2, 0, 4, 5, 8), (3, 0, 8, 9, 9)]
    # fixed window of 10 points, gap 2, embargo pushes the next train start past the previous test
    b = api.backtest_fold_bounds(60, 5, 3, window_type="fixed", min_train_size=10, gap=2)
    assert b == [(1, 35, 44, 47, 51), (2, 40, 49, 52, 56), (3, 45, 54, 57, 61)][:2]
    b = api.backtest_fold_bounds(60, 5, 2, window_type="expanding", embargo=3, initial_train_size=30, skip_length=10)
    assert b == [(1, 0, 29, 30, 34), (2, 38, 39, 40, 44)]
    a, f = np.array([1.0, 2.0, 0.0, 4.0]), np.array([1.5, 1.0, 1.0, 4.0])
    lo, hi = f - 0.75, f + 0.75
    assert api.backtest_metric("mae", a, f, lo, hi) == (0.5 + 1.0 + 1.0 + 0.0) / 4
    assert api.backtest_metric("mse", a, f, lo, hi) == (0.25 + 1.0 + 1.0) / 4
    assert api.backtest_metric("rmse", a, f, lo, hi) == np.sqrt((0.25 + 1.0 + 1.0) / 4)
    assert api.backtest_metric("nonsense", a, f, lo, hi) == api.backtest_metric("rmse", a, f, lo, hi)
    assert api.backtest_metric("mape", a, f, lo, hi) == (0.5 + 0.5 + 0.0) / 3 * 100.0
    assert api.backtest_metric("smape", a, f, lo, hi) == ((0.5 / 2.5 + 1.0 / 3.0) + 1.0 / 1.0 + 0.0) / 4 * 200.0
    assert api.backtest_metric("bias", a, f, lo, hi) == (0.5 - 1.0 + 1.0 + 0.0) / 4
    assert api.backtest_metric("coverage", a, f, lo, hi) == 2 / 4
    mean = 7.0 / 4
    tot = sum((x - mean) ** 2 for x in a)
    assert abs(api.backtest_metric("r2", a, f, lo, hi) - (1.0 - 2.25 / tot)) < 1e-15
    assert np.isnan(api.backtest_metric("r2", np.ones(3), np.ones(3), [], []))
    assert np.isnan(api.backtest_metric("mae", [], [], [], []))


def test_pow_tables_are_one_file_and_pow_step_is_accurate(oracle, tmp_path):
    """The b^phi tables exist twice (product / checker), byte for byte the output of tools/gen_pow_tables.py; the table
    driven power stays within 2 ulp of the exact one where growth rates live ([1/2, 2]) and 1^phi is exactly 1."""
    import ctypes as C
    import math
    import subprocess
    import sys
    from decimal import Decimal, getcontext
    a = open(os.path.join(ROOT, "oracle", "pow_tables.inc")).read()
    b = open(os.path.join(ROOT, "anofox-forecast_amd", "csrc", "pow_tables.inc")).read()
    assert a == b
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_pow_tables.py"), "--out", str(tmp_path)], stdout=subprocess.DEVNULL)
    assert open(os.path.join(str(tmp_path), "pow_tables.inc")).read() == a               # the generator reproduces the tracked files (written elsewhere: no source is touched)
    L = oracle.lib()
    L.oracle_det_pow_step.restype = C.c_double
    L.oracle_det_pow_step.argtypes = [C.c_double, C.c_double]
    getcontext().prec = 50
    rng = np.random.default_rng(3)
    worst = 0.0
    for x, y in zip(np.exp(rng.uniform(-0.69, 0.69, 1500)), rng.uniform(0.8, 0.98, 1500)):
        got = L.oracle_det_pow_step(float(x), float(y))
        ref = (Decimal(float(x)).ln() * Decimal(float(y))).exp()
        worst = max(worst, float(abs(Decimal(got) - ref) / Decimal(math.ulp(float(ref)))))
    assert worst < 2.0, worst
    assert L.oracle_det_pow_step(1.0, 0.9) == 1.0 and L.oracle_det_pow_step(4.0, 0.5) == 2.0


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2`, un-wrapped (the shape of the command the driver runs for N = 1): the process starts two ranks itself
    (a child `python -m torch.distributed.run`, never an exec) and hands their exit code on.  Without a GPU both ranks stop at the
    library's "no CPU fallback" gate -- which is the proof, on this CPU-only machine, that two ranks were started and the parent did
    not quietly run one rank with "n_gpus": 1 (missing item 1).  The GPU half of the same path:
    tests/test_gpu_parity.py::test_bench_two_ranks_on_one_gpu[self-launch]."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0                                  # the ranks' failure is the parent's exit code
    assert out.stderr.count("bench.py needs a GPU") == 2, out.stderr[-2000:]
    assert '"n_gpus"' not in out.stdout


def test_scalar_chunk_groups_rows_by_option_block(monkeypatch):
    """Route A, round 6 (binding/ts_forecast_scalar_hip.cpp, mirrored by api.ts_forecast_scalar): a chunk's rows are grouped by their
    option block and every distinct block is ONE batch call with per-row horizons; NULL / empty lists are NULL rows and never reach
    the library; the reference's error policy (ts_forecast_scalar.cpp:484-490) is applied in chunk order.  The batch entry is replaced
    by a recorder here: no GPU, no forecast is computed."""
    calls = []

    def fake_batch(series, opts, valids=None, horizons=None):
        calls.append((opts.model.decode(), opts.seasonal_period, opts.confidence_level, bool(opts.auto_detect_seasonality),
                      [list(s) for s in series], [list(v) for v in valids], list(horizons)))
        res = []
        for s, h in zip(series, horizons):
            if len(s) < 3:
                res.append({"ok": False, "code": 6, "message": "Insufficient data"})
            elif opts.model == b"Naive" and opts.seasonal_period > 1:
                res.append({"ok": False, "code": 2, "message": "Model 'Naive' does not use seasonal_period"})
            else:
                res.append({"ok": True, "code": 0, "message": "", "point": np.full(h, s[-1]), "lower": np.full(h, s[-1] - 1.0),
                            "upper": np.full(h, s[-1] + 1.0), "model_name": opts.model.decode()})
        return res, {"ok": True, "code": 0, "message": ""}
    monkeypatch.setattr(api, "forecast_batch", fake_batch)
    day = np.datetime64("2024-01-01", "D")
    d5 = day + np.arange(5)
    rows_d = [d5, d5[::-1].copy(), None, d5[:2], d5, d5[:0]]
    rows_v = [np.arange(5.0), np.ma.array([4.0, 3.0, 2.0, 1.0, 0.0], mask=[0, 0, 1, 0, 0]), np.arange(5.0), np.arange(2.0),
              np.arange(5.0) * 2, np.arange(0.0)]
    out = api.ts_forecast_scalar(rows_d, rows_v, [3, 2, 3, 3, 4, 3], "1d", "Naive", {})
    assert len(calls) == 1                                                      # one option block -> one batch call for the chunk
    model, sp, conf, auto, series, valids, hz = calls[0]
    assert (model, sp, conf, auto) == ("Naive", 0, 0.90, True) and hz == [3, 2, 3, 4]      # the NULL and the empty list never travel
    assert series[1] == [0.0, 1.0, 0.0, 3.0, 4.0] and valids[1] == [True, True, False, True, True]     # re-ordered by date; NULL slot 0.0
    assert out[2] is None and out[5] is None and out[3] is None                 # NULL list, empty list, failed series (code 6)
    assert list(out[0]["forecast_step"]) == [1, 2, 3] and out[0]["ds"][0] == day + 5 and len(out[1]["yhat"]) == 2
    assert out[4]["yhat"][0] == 8.0 and out[4]["model_name"] == ["Naive"] * 4
    # per-row method / params: two option blocks, two calls; rows keep their places
    calls.clear()
    out = api.ts_forecast_scalar([d5] * 4, [np.arange(5.0)] * 4, 2, "1d", ["Naive", "SES", "Naive", "SES"],
                                 [{}, {"confidence_level": "0.95"}, {}, {"confidence_level": 0.95}])
    assert [c[0] for c in calls] == ["Naive", "SES"] and [len(c[4]) for c in calls] == [2, 2] and calls[1][2] == 0.95
    assert [o["model_name"][0] for o in out] == ["Naive", "SES", "Naive", "SES"]
    # INVALID_INPUT aborts the statement; an unknown key is rejected before any call
    calls.clear()
    with pytest.raises(api.InvalidInputException, match="does not use seasonal_period"):
        api.ts_forecast_scalar([d5], [np.arange(5.0)], 2, "1d", "Naive", {"seasonal_period": "7"})
    with pytest.raises(api.InvalidInputException, match="Unknown parameter"):
        api.ts_forecast_scalar([d5], [np.arange(5.0)], 2, "1d", "Naive", {"sesonal_period": "7"})
    with pytest.raises(ValueError):
        api.ts_forecast_scalar([d5] * 2049, [np.arange(5.0)] * 2049, 2, "1d", "Naive", {})
    # the shipped macro text over columns: 5 groups in chunks of 2 -> 3 scalar calls, rows in group order
    calls.clear()
    grp = np.repeat(np.array(["a", "b", "c", "d", "e"], dtype=object), 4)
    ds = np.tile(day + np.arange(4), 5)
    y = np.arange(20.0)
    perm = np.random.default_rng(3).permutation(20)
    out = api.ts_forecast_by_scalar_route(grp[perm], ds[perm], y[perm], "Naive", 2, "1d", None, chunk_groups=2)
    assert len(calls) == 3 and list(out.keys()) == ["id", "forecast_step", "ds", "yhat", "yhat_lower", "yhat_upper", "model_name"]
    first = {}
    for g, v in zip(out["id"], out["yhat"]):
        first.setdefault(g, v)
    assert first == {"a": 3.0, "b": 7.0, "c": 11.0, "d": 15.0, "e": 19.0} and len(out["yhat"]) == 10
    assert out["ds"].dtype == ds.dtype and out["ds"][0] == day + 4


def test_gather_plan_layout():
    """dist.GatherPlan (round 6): a rank's chunk is ONE buffer -- three [per x h] fp64 sections, then model codes and status as int32,
    every section 8-byte aligned, sections disjoint, views typed and shaped without a copy; the root's receive buffer is world of
    them.  (The collective itself: tests/test_dist_gloo.py.)"""
    torch = pytest.importorskip("torch")
    from anofox_forecast_amd.dist import GatherPlan
    for n_total, h, world in ((30490, 28, 8), (7, 3, 2), (1000003, 1, 4), (5, 0, 2)):
        plan = GatherPlan(n_total, h, rank=0, world=world, device="cpu", dst=0)
        per = (n_total + world - 1) // world
        assert plan.per == per and plan.recv.numel() == world * plan.chunk_bytes and len(plan._parts) == world
        spans = sorted((off, off + nbytes) for off, nbytes, _, _ in plan.sections.values())
        assert all(off % 8 == 0 for off, _ in spans)
        assert all(a_end <= b_off for (_, a_end), (b_off, _) in zip(spans, spans[1:])) and spans[-1][1] <= plan.chunk_bytes
        y = plan._view(plan.send, "yhat")
        assert y.dtype == torch.float64 and tuple(y.shape) == (per, h) and (h == 0 or y.data_ptr() == plan.send.data_ptr())
        code = plan._view(plan.send, "model_code")
        assert code.dtype == torch.int32 and tuple(code.shape) == (per,)
        code.fill_(7)
        plan._view(plan.send, "status").fill_(-1)
        if h:
            plan._view(plan.send, "upper").fill_(2.5)
            assert float(plan._view(plan.send, "lower").abs().sum()) == 0.0          # a neighbour's fill does not reach it
        assert int(plan._view(plan.send, "model_code").sum()) == 7 * per
