"""Evaluator for tests/golden/reference_sql_pins.json (made by tests/golden/make_sql_pins.py): the pins the reference's
sqllogictest files hold on `_ts_forecast(values, horizon, model)`, replayed against any forecast function with the
C-ABI's result fields.  SQL semantics kept: lists are 1-indexed, an index past the end is NULL, NaN orders above
every number."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = json.load(open(os.path.join(HERE, "golden", "reference_sql_pins.json")))


def pin_id(case):
    return f'{case["calls"][0]["model"]}-{case["check"]}@{case["source"].split("/")[-1]}'


def _cmp(a, op, b):
    if a is None or b is None:
        return None
    ka, kb = (1, 0.0) if a != a else (0, a), (1, 0.0) if b != b else (0, b)          # NaN sorts last
    return {"<": ka < kb, ">": ka > kb, "<=": ka <= kb, ">=": ka >= kb, "=": ka == kb}[op]


def _at(res, field, k):
    key = {"point": "point", "lower": "lower", "upper": "upper"}[field]
    v = res[key]
    return float(v[k - 1]) if 1 <= k <= len(v) else None


def check_pin(case, run):
    """`run(values, valid, horizon, model)` -> result dict (ok, point, lower, upper, fitted, residuals, model_name, aic, bic, mse)."""
    res = []
    for c in case["calls"]:
        vals = np.array([0.0 if v is None else v for v in c["values"]], dtype=np.float64)
        valid = np.array([v is not None for v in c["values"]], dtype=bool)
        r = run(vals, None if valid.all() else valid, c["horizon"], c["model"])
        res.append(r)
    r0, kind, exp = res[0], case["check"], case["expected"]
    if kind == "struct_is_null":
        # the scalar returns a NULL struct for every failure except INVALID_INPUT (2) / INVALID_MODEL (5), which raise
        # (src/table_functions/ts_forecast.cpp:425-432)
        assert r0["ok"] or r0["code"] not in (2, 5), (case["source"], r0)
        got = (not r0["ok"]) != case["negated"]
        assert got == exp, (case["source"], kind, r0.get("code"), exp)
        return
    for r in res:
        assert r["ok"], (case["source"], r)
    if kind == "model_name":
        got = r0["model_name"]
    elif kind == "length":
        key = case["field"]
        got = float(len(r0[key])) if key in r0 else None
    elif kind == "cmp_const":
        got = _cmp(_at(r0, case["field"], case["k"]), case["op"], case["c"])
    elif kind == "near_const":
        got = _cmp(abs(_at(r0, case["field"], case["k"]) - case["c"]), case["op"], case["tol"])
    elif kind == "cmp_fields":
        got = _cmp(_at(r0, case["field"], case["k"]), case["op"], _at(res[case["rhs_call"]], case["rhs_field"], case["rhs_k"]))
    elif kind == "near_fields":
        got = _cmp(abs(_at(r0, case["field"], case["k"]) - _at(res[case["rhs_call"]], case["rhs_field"], case["rhs_k"])), case["op"], case["tol"])
    elif kind == "width_cmp":
        got = _cmp(_at(r0, "upper", case["k"]) - _at(r0, "lower", case["k_lower"]), case["op"], case["c"])
    elif kind == "is_nan":
        v = _at(r0, case["field"], case["k"])
        got = v is not None and v != v
    elif kind == "not_null":
        got = _at(r0, case["field"], case["k"]) is not None
    elif kind == "scalar_not_null":
        got = True                                   # a DOUBLE / VARCHAR struct field of a successful call is never NULL (ts_forecast.cpp:160-170)
    elif kind == "mse_not_negative":
        got = _cmp(float(r0["mse"]), ">=", 0.0)
    elif kind == "round":
        got = round(_at(r0, case["field"], case["k"]), case["digits"])
    else:
        raise AssertionError(f"unknown check {kind}")
    assert got == exp, (case["source"], kind, got, exp)


def check_unit_case(case, run):
    """One of the wrapper's own unit tests (reference_kats.json, key "unit"): `run(values, valid, horizon, model, options)`."""
    vals = np.array([0.0 if v is None else v for v in case["values"]], dtype=np.float64)
    valid = np.array([v is not None for v in case["values"]], dtype=bool)
    r = run(vals, None if valid.all() else valid, case["horizon"], case["model"], case["options"])
    e = case["expect"]
    if e.get("fails"):
        assert not r["ok"], (case["source"], r)
        return
    assert r["ok"], (case["source"], r)
    if "n_points" in e:
        assert len(r["point"]) == e["n_points"]
    if "first" in e:
        assert r["point"][0] == e["first"]
    if "name" in e:
        assert r["model_name"] == e["name"]
    if "name_prefix" in e:
        assert r["model_name"].startswith(e["name_prefix"]), r["model_name"]
    if e.get("finite"):
        assert np.all(np.isfinite(r["point"]))
    if e.get("positive"):
        assert np.all(r["point"] > 0)
    if "near" in e:
        assert np.all(np.abs(r["point"] - e["near"][0]) < e["near"][1])
    if "n_fitted" in e:
        assert len(r["fitted"]) == e["n_fitted"] and len(r["residuals"]) == e["n_residuals"] and r["mse"] == r["mse"]
    if "sma_window" in e:
        w = e["sma_window"]
        np.testing.assert_allclose(r["point"], np.full(len(r["point"]), np.mean(vals[-w:])), rtol=1e-14)
    if "same_as_model" in e:
        other = run(vals, None, case["horizon"], e["same_as_model"], case["options"])
        assert other["ok"] and np.array_equal(other["point"], r["point"])
    if e.get("interval_strict"):
        assert np.all(r["lower"] < r["point"]) and np.all(r["upper"] > r["point"])
    if e.get("interval_widens"):
        assert r["upper"][-1] - r["lower"][-1] > r["upper"][0] - r["lower"][0]
