"""Writes tests/golden/reference_sql_pins.json: the pins the reference's sqllogictest files hold on the scalar
`_ts_forecast(values, horizon, model)` (src/table_functions/ts_forecast.cpp:406-411: auto_detect = false, period from
seasonal_period = 0, confidence 0.95, fitted + residuals on) for the models on the hot path, transcribed as DATA:
inputs (values, horizon, model), a check kind with its operands, the expected result, and the source file:line.
No SQL text is kept.  Needs /root/reference (this container only); the JSON it writes is what the tests read.
Run:  python tests/golden/make_sql_pins.py
"""
import glob
import json
import os
import re
import sys

REF = "/root/reference/test/sql"
ON_PATH = {"autoets", "auto_ets", "auto", "autoarima", "auto_arima", "naive", "sma", "seasonalnaive", "seasonal_naive", "snaive", "ses",
           "sesoptimized", "ses_optimized", "randomwalkdrift", "random_walk_drift", "rwd", "drift", "randomwalkwithdrift",
           "random_walk_with_drift", "holt", "holtwinters", "holt_winters", "hw", "seasonales", "seasonal_es", "seasonalesoptimized",
           "seasonal_es_optimized", "ets", "arima"}
NUM = r"[-+]?\d+(?:\.\d*)?(?:[eE][-+]?\d+)?"


def blocks(path):
    """(first line number, query text on one line, expected lines) of every `query` block."""
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        if lines[i].startswith("query"):
            start = i + 1
            j = i + 1
            sql = []
            while j < len(lines) and lines[j].strip() != "----":
                sql.append(lines[j].strip())
                j += 1
            k = j + 1
            exp = []
            while k < len(lines) and lines[k].strip() != "":
                exp.append(lines[k].strip())
                k += 1
            yield start + 1, " ".join(sql), exp
            i = k
        else:
            i += 1


def take_calls(sql):
    """Replace every `(_ts_forecast([..], h, 'M'))` by F<i>; returns (residual, calls) or None when an argument is not a literal."""
    calls = []
    out = sql
    while True:
        p = out.find("_ts_forecast(")
        if p < 0:
            break
        depth, q = 0, p + len("_ts_forecast")
        while True:
            if out[q] == "(":
                depth += 1
            elif out[q] == ")":
                depth -= 1
                if depth == 0:
                    break
            q += 1
        args = out[p + len("_ts_forecast("):q]
        m = re.fullmatch(r"\s*\[([^\]]*)\](?:::DOUBLE\[\])?\s*,\s*(\d+)\s*(?:,\s*'([^']*)'\s*)?", args)
        if not m:
            return None
        vals = []
        for tok in ([] if not m.group(1).strip() else m.group(1).split(",")):
            tok = tok.strip().replace("::DOUBLE", "")
            if tok.upper() == "NULL":
                vals.append(None)
            elif re.fullmatch(NUM, tok):
                vals.append(float(tok))
            else:
                return None
        key = (tuple(vals), int(m.group(2)), m.group(3) if m.group(3) is not None else "auto")      # ts_forecast.cpp:380: default "auto"
        if key not in calls:
            calls.append(key)
        name = f"F{calls.index(key)}"
        lo, hi = p, q + 1
        if lo > 0 and out[lo - 1] == "(" and hi < len(out) and out[hi] == ")":      # the usual (call).field wrapping
            lo, hi = lo - 1, hi + 1
        out = out[:lo] + name + out[hi:]
    return out, calls


PATTERNS = [
    (rf"SELECT F0\.model;?", lambda m: ("model_name", {})),
    (rf"SELECT length\(F0\.(point|fitted|lower|upper|residuals)\);?", lambda m: ("length", {"field": m.group(1)})),
    (rf"SELECT F0\.(point|lower|upper)\[(\d+)\] (>|<|>=|<=) ({NUM});?", lambda m: ("cmp_const", {"field": m.group(1), "k": int(m.group(2)), "op": m.group(3), "c": float(m.group(4))})),
    (rf"SELECT ABS\(F0\.(point|lower|upper)\[(\d+)\] - \(?({NUM})\)?\) (<|>) ({NUM});?", lambda m: ("near_const", {"field": m.group(1), "k": int(m.group(2)), "c": float(m.group(3)), "op": m.group(4), "tol": float(m.group(5))})),
    (rf"SELECT F0\.(point|lower|upper)\[(\d+)\] (>|<|>=|<=|=) F([01])\.(point|lower|upper)\[(\d+)\];?",
     lambda m: ("cmp_fields", {"field": m.group(1), "k": int(m.group(2)), "op": m.group(3), "rhs_call": int(m.group(4)), "rhs_field": m.group(5), "rhs_k": int(m.group(6))})),
    (rf"SELECT ABS\(F0\.(point|lower|upper)\[(\d+)\] - F([01])\.(point|lower|upper)\[(\d+)\]\) (<|>|>=) ({NUM});?",
     lambda m: ("near_fields", {"field": m.group(1), "k": int(m.group(2)), "rhs_call": int(m.group(3)), "rhs_field": m.group(4), "rhs_k": int(m.group(5)), "op": m.group(6), "tol": float(m.group(7))})),
    (rf"SELECT F0\.(point|lower|upper)\[(\d+)\] IS NOT NULL;?", lambda m: ("not_null", {"field": m.group(1), "k": int(m.group(2))})),
    (rf"SELECT F0\.(model|aic|bic|mse) IS NOT NULL;?", lambda m: ("scalar_not_null", {"field": m.group(1)})),
    (rf"SELECT F0\.mse >= 0;?", lambda m: ("mse_not_negative", {})),
    (rf"SELECT F0 IS (NOT )?NULL;?", lambda m: ("struct_is_null", {"negated": m.group(1) is not None})),
    (rf"SELECT F0\.upper\[(\d+)\] - F0\.lower\[(\d+)\] (<|>) ({NUM});?", lambda m: ("width_cmp", {"k": int(m.group(1)), "k_lower": int(m.group(2)), "op": m.group(3), "c": float(m.group(4))})),
    (rf"SELECT isnan\(F0\.(point|lower|upper)\[(\d+)\]\);?", lambda m: ("is_nan", {"field": m.group(1), "k": int(m.group(2))})),
    (rf"SELECT ABS\(F0\.(point|lower|upper)\[(\d+)\]\) (<|>) ({NUM});?", lambda m: ("near_const", {"field": m.group(1), "k": int(m.group(2)), "c": 0.0, "op": m.group(3), "tol": float(m.group(4))})),
    (rf"SELECT ROUND\(F0\.(point|lower|upper)\[(\d+)\], (\d+)\);?", lambda m: ("round", {"field": m.group(1), "k": int(m.group(2)), "digits": int(m.group(3))})),
]


def main():
    cases, skipped = [], []
    for path in sorted(glob.glob(os.path.join(REF, "*.test"))):
        name = os.path.basename(path)
        for line, sql, exp in blocks(path):
            if "_ts_forecast(" not in sql:
                continue
            got = take_calls(sql)
            if got is None:
                skipped.append((name, line, "non-literal arguments"))
                continue
            resid, calls = got
            if any(c[2].lower() not in ON_PATH for c in calls):
                continue                                         # model outside the hot path
            resid = re.sub(r"\s+", " ", resid).strip()
            for pat, build in PATTERNS:
                m = re.fullmatch(pat, resid)
                if m and len(exp) == 1:
                    kind, operands = build(m)
                    e = exp[0]
                    expected = True if e == "true" else False if e == "false" else (float(e) if re.fullmatch(NUM, e) else e)
                    cases.append({"source": f"test/sql/{name}:{line}", "check": kind, **operands, "expected": expected,
                                  "calls": [{"values": list(c[0]), "horizon": c[1], "model": c[2]} for c in calls]})
                    break
            else:
                skipped.append((name, line, resid[:100]))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_sql_pins.json")
    json.dump({"options": {"confidence_level": 0.95, "seasonal_period": 0, "auto_detect": False, "include_fitted": True,
                           "include_residuals": True}, "cases": cases}, open(out, "w"), indent=0)
    print(len(cases), "cases written,", len(skipped), "blocks on the path not expressible as a pin:")
    for s in skipped:
        print("  ", *s)


if __name__ == "__main__":
    main()
