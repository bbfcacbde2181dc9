"""Writes tests/golden/reference_kats.json: the known-answer vectors that the reference's OWN tests
hold for the forecast path, transcribed as data (inputs + expected outputs, each with its source
file:line).  Nothing here is computed: the reference cannot be built or imported in this image
(no cargo/rustc, no DuckDB), so these literals are the only pins that exist.
Run:  python tests/golden/make_golden.py
"""
import json
import math
import os

DISTINCT = [10, 12, 14, 11, 13, 15, 12, 14, 16, 13, 15, 17, 14, 16, 18, 15, 17, 19, 16, 18, 20, 17, 19, 21]
ONE_TO_TEN = [float(i) for i in range(1, 11)]


def seasonal_data():
    # crates/anofox-fcst-ffi/tests/core_ffi_parity.rs:55-65
    out = []
    for i in range(60):
        trend = 10.0 + 0.15 * i
        season = 5.0 * math.sin(2.0 * math.pi * i / 12.0)
        noise = ((i * 7 + 3) % 11) * 0.1 - 0.5
        out.append(trend + season + noise)
    return out


def toy_arima_expected(data, h):
    # the closed form the reference test itself evaluates: core_ffi_parity.rs:680-704
    diff = [b - a for a, b in zip(data[:-1], data[1:])]
    mean_diff = sum(diff) / len(diff)
    prev, cum, out = diff[-1], data[-1], []
    for _ in range(h):
        nd = mean_diff + 0.5 * (prev - mean_diff)
        cum += nd
        out.append(cum)
        prev = nd
    return out


# options of _ts_forecast(values, horizon, model): src/table_functions/ts_forecast.cpp:406-411
SCALAR_OPTS = {"confidence_level": 0.95, "seasonal_period": 0, "auto_detect": False, "include_fitted": True, "include_residuals": True}

cases = []
for model, kat, line in [("SES", 18.943503, 116), ("SESOptimized", 19.537535, 116), ("SeasonalES", 14.451866, 116),
                         ("SeasonalESOptimized", 18.803254, 116), ("Holt", 20.330877, 141), ("HoltWinters", 19.953912, 141),
                         ("AutoETS", 19.956521, 164), ("AutoARIMA", 18.014537, 164), ("Naive", 21.0, 180), ("SMA", 19.0, 180),
                         ("RandomWalkDrift", 21.478261, 180)]:
    cases.append({"source": f"test/sql/ts_model_distinctness.test:{line}", "model": model, "values": DISTINCT, "horizon": 3,
                  "options": SCALAR_OPTS, "check": "round6_first", "expected": kat})
cases += [
    {"source": "test/sql/ts_forecast_basic_models.test:128-156", "model": "SeasonalNaive", "values": [1, 2, 3, 4] * 3, "horizon": 4,
     "options": SCALAR_OPTS, "check": "abs_all", "tol": 0.01, "expected": [4.0, 4.0, 4.0, 4.0]},
    {"source": "test/sql/ts_forecast_basic_models.test:317-346", "model": "RandomWalkDrift", "values": ONE_TO_TEN, "horizon": 3,
     "options": SCALAR_OPTS, "check": "abs_all", "tol": 0.1, "expected": [11.0, 12.0, 13.0]},
    {"source": "test/sql/ts_forecast_basic_models.test:84", "model": "SMA", "values": ONE_TO_TEN, "horizon": 3,
     "options": SCALAR_OPTS, "check": "abs_all", "tol": 0.1, "expected": [9.0, 9.0, 9.0]},
    {"source": "test/sql/ts_forecast.test:76-86", "model": "NAIVE", "values": [1, 2, 3, 4, 5], "horizon": 3,
     "options": SCALAR_OPTS, "check": "abs_all", "tol": 0.01, "expected": [5.0, 5.0, 5.0]},
    {"source": "test/sql/ts_forecast_auto.test:98", "model": "AutoETS", "values": [42.0] * 30, "horizon": 5,
     "options": SCALAR_OPTS, "check": "abs_all", "tol": 1.0, "expected": [42.0] * 5},
    # ts_forecast_by(grouped_data, ..., 'HoltWinters', 3, '1d', MAP{confidence_level: 0.95, seasonal_period: 7}) returns 3 rows per
    # group although a group has 10 observations (< 2 seasons): the fit does not fail (exp_smoothing.test:498-503 says why)
    {"source": "test/sql/ts_forecast_params.test:203-207", "model": "HoltWinters", "values": [10.0 + 2.0 * i for i in range(10)], "horizon": 3,
     "options": {"confidence_level": 0.95, "seasonal_period": 7, "auto_detect": False}, "check": "n_points", "expected": 3},
    {"source": "test/sql/ts_forecast_params.test:203-207", "model": "HoltWinters", "values": [100.0 + 5.0 * i for i in range(10)], "horizon": 3,
     "options": {"confidence_level": 0.95, "seasonal_period": 7, "auto_detect": False}, "check": "n_points", "expected": 3},
    {"source": "crates/anofox-fcst-ffi/tests/core_ffi_parity.rs:680-704", "model": "ARIMA", "values": seasonal_data(), "horizon": 5,
     "options": {"confidence_level": 0.95, "seasonal_period": 0, "auto_detect": False}, "check": "bits_all",
     "expected": toy_arima_expected(seasonal_data(), 5)},
]

# the wrapper's own unit tests (crates/anofox-fcst-core/src/forecast.rs, `mod tests`): inputs, options over the Rust default
# (`:349-366`: AutoETS, confidence 0.95, seasonal_period 0, auto_detect on) and what each test asserts
RUST_DEFAULT = {"confidence_level": 0.95, "seasonal_period": 0, "auto_detect": True}
unit = [
    {"source": "forecast.rs:2702-2713", "model": "Naive", "values": [1.0, 2.0, 3.0, 4.0, 5.0], "horizon": 3, "options": RUST_DEFAULT,
     "expect": {"n_points": 3, "first": 5.0}},
    {"source": "forecast.rs:2716-2726", "model": "SES", "values": [float(i) for i in range(20)], "horizon": 5, "options": RUST_DEFAULT,
     "expect": {"n_points": 5}},
    {"source": "forecast.rs:2729-2752", "model": "HoltWinters", "values": [100.0 + i * 0.5 + (i % 12) * 2.0 for i in range(48)], "horizon": 12,
     "options": dict(RUST_DEFAULT, seasonal_period=12, auto_detect=False), "expect": {"n_points": 12, "name": "HoltWinters", "finite": True, "positive": True}},
    {"source": "forecast.rs:2755-2772", "model": "ARIMA", "values": [100.0 + i * 2.0 + (i % 3) for i in range(30)], "horizon": 5, "options": RUST_DEFAULT,
     "expect": {"n_points": 5, "name": "ARIMA", "finite": True}},
    {"source": "forecast.rs:3095-3108", "model": "AutoETS", "values": [50.0 + (i % 7) * 3.0 for i in range(30)], "horizon": 7, "options": RUST_DEFAULT,
     "expect": {"n_points": 7, "finite": True}},
    {"source": "forecast.rs:3111-3131", "model": "AutoETS", "values": [42.0] * 30, "horizon": 5, "options": RUST_DEFAULT,
     "expect": {"n_points": 5, "finite": True, "near": [42.0, 1.0]}},
    {"source": "forecast.rs:3203-3224", "model": "Naive", "values": [1.0, 2.0, None, 4.0, 5.0, None, 7.0], "horizon": 3, "options": RUST_DEFAULT,
     "expect": {"n_points": 3, "finite": True}},
    {"source": "forecast.rs:3227-3244", "model": "SES", "values": [i * 2.0 for i in range(15)], "horizon": 3,
     "options": dict(RUST_DEFAULT, include_fitted=True, include_residuals=True), "expect": {"n_fitted": 15, "n_residuals": 15, "mse": True}},
    {"source": "forecast.rs:3247-3253", "model": "AutoETS", "values": [1.0, 2.0], "horizon": 12, "options": RUST_DEFAULT, "expect": {"fails": True}},
    {"source": "forecast.rs:3345-3362", "model": "ETS", "values": [100.0 + i for i in range(60)], "horizon": 5, "options": RUST_DEFAULT,
     "expect": {"n_points": 5}},
    {"source": "forecast.rs:3365-3395", "model": "AutoARIMA", "values": [100.0 + i * 0.5 + 10.0 * math.sin(i * 0.1) for i in range(60)], "horizon": 5,
     "options": dict(RUST_DEFAULT, seasonal_period=12), "expect": {"n_points": 5, "name_prefix": "AutoARIMA", "finite": True}},
    {"source": "forecast.rs:3398-3434", "model": "AutoETS", "values": [100.0 + i * 0.5 + 10.0 * math.sin(2.0 * math.pi * i / 12.0) for i in range(48)],
     "horizon": 5, "options": dict(RUST_DEFAULT, seasonal_period=12), "expect": {"n_points": 5, "name_prefix": "AutoETS", "finite": True}},
    {"source": "forecast.rs:3437-3468", "model": "AutoARIMA", "values": [100.0 + i * 2.0 + 5.0 * math.sin(i * 0.2) for i in range(50)], "horizon": 5,
     "options": RUST_DEFAULT, "expect": {"n_points": 5, "name_prefix": "AutoARIMA"}},
    {"source": "forecast.rs:3437-3468", "model": "ARIMA", "values": [100.0 + i * 2.0 + 5.0 * math.sin(i * 0.2) for i in range(50)], "horizon": 5,
     "options": RUST_DEFAULT, "expect": {"n_points": 5, "name": "ARIMA"}},
    # calculate_confidence_intervals (`:3156-3172`) through a model whose forecast is known: bounds strictly around, widening
    {"source": "forecast.rs:3156-3172", "model": "Naive", "values": [50.0 + i for i in range(20)], "horizon": 3, "options": RUST_DEFAULT,
     "expect": {"n_points": 3, "interval_strict": True, "interval_widens": True}},
]

# crates/anofox-fcst-ffi/tests/core_ffi_parity.rs: every configuration that file sends through the FFI on seasonal_data()
# (60 points, `:55-65`; options of `:84-100`: confidence 0.95, detection off) must succeed with HORIZON = 5 points
FFI = {"confidence_level": 0.95, "auto_detect": False}
for model, period, line in [("SES", 0, 239), ("SESOptimized", 0, 252), ("Holt", 0, 265), ("HoltWinters", 12, 278), ("SeasonalES", 12, 294),
                            ("SeasonalESOptimized", 12, 307), ("AutoETS", 12, 393), ("AutoARIMA", 12, 407), ("Naive", 0, 584),
                            ("SeasonalNaive", 12, 596), ("SMA", 12, 608), ("RandomWalkDrift", 0, 621), ("ETS", 12, 659)]:
    unit.append({"source": f"core_ffi_parity.rs:{line}", "model": model, "values": seasonal_data(), "horizon": 5,
                 "options": dict(FFI, seasonal_period=period), "expect": {"n_points": 5, "finite": True}})
for window in (3, 5, 12, 30, 60):
    unit.append({"source": "core_ffi_parity.rs:715-728", "model": "SMA", "values": seasonal_data(), "horizon": 5,
                 "options": dict(FFI, seasonal_period=0, window=window), "expect": {"n_points": 5, "sma_window": window}})
for period in (4, 6, 12):
    for model in ("HoltWinters", "SeasonalES", "SeasonalESOptimized", "SeasonalNaive"):
        unit.append({"source": "core_ffi_parity.rs:731-790", "model": model, "values": seasonal_data(), "horizon": 5,
                     "options": dict(FFI, seasonal_period=period), "expect": {"n_points": 5, "finite": True}})
for spec in ("AAA", "ANA", "MNM", "MAM", "AAdA", "MAdM"):
    unit.append({"source": "core_ffi_parity.rs:858-878", "model": "ETS", "values": seasonal_data(), "horizon": 5,
                 "options": dict(FFI, seasonal_period=12, ets_model=spec), "expect": {"n_points": 5, "finite": True}})
for horizon in (1, 3, 10, 20):
    unit.append({"source": "core_ffi_parity.rs:924-946", "model": "SES", "values": seasonal_data(), "horizon": horizon,
                 "options": dict(FFI, seasonal_period=0), "expect": {"n_points": horizon, "finite": True}})
# `:659-677`: ETS without a spec on this series IS Holt-Winters(12, additive)
unit.append({"source": "core_ffi_parity.rs:659-677", "model": "ETS", "values": seasonal_data(), "horizon": 5, "options": dict(FFI, seasonal_period=12),
             "expect": {"n_points": 5, "same_as_model": "HoltWinters"}})

errors = [
    # (source, model, options, expected code, message substring) -- test/sql/ts_native_param_validation.test:126-198
    {"source": "ts_native_param_validation.test:126-139", "model": "ETS", "options": {"ets_model": "XYZ"}, "code": 2, "substr": "Invalid ETS model specification"},
    {"source": "ts_native_param_validation.test:133-139", "model": "ETS", "options": {"ets_model": "AAAAA"}, "code": 2, "substr": "Invalid ETS model specification"},
    {"source": "ts_native_param_validation.test:142-149", "model": "ETS", "options": {"ets_model": "MAA"}, "code": 2, "substr": "unstable"},
    {"source": "ts_native_param_validation.test:150-155", "model": "ETS", "options": {"ets_model": "MAdA"}, "code": 2, "substr": "unstable"},
    {"source": "ts_native_param_validation.test:176-184", "model": "Naive", "options": {"seasonal_period": 7, "auto_detect": False}, "code": 2, "substr": "does not use seasonal_period"},
    {"source": "ts_native_param_validation.test:185-191", "model": "SES", "options": {"seasonal_period": 7, "auto_detect": False}, "code": 2, "substr": "does not use seasonal_period"},
    {"source": "crates/anofox-fcst-core/src/forecast.rs:253-255", "model": "NoSuchModel", "options": {}, "code": 5, "substr": "Unknown model"},
    {"source": "crates/anofox-fcst-core/src/forecast.rs:3313-3342", "model": "ETS", "options": {"ets_model": "AAA", "seasonal_period": 12, "auto_detect": False},
     "values": [100.0, 101.0, 102.0, 103.0, 104.0], "code": 3, "substr": "failed to fit"},
    {"source": "crates/anofox-fcst-core/src/forecast.rs:516-525", "model": "Naive", "options": {}, "values": [1.0, 2.0], "code": 6, "substr": "Insufficient data"},
]

interp = [
    {"source": "crates/anofox-fcst-core/src/imputation.rs:157-166", "values": [1.0, 0.0, 0.0, 4.0], "valid": [1, 0, 0, 1], "expected": [1.0, 2.0, 3.0, 4.0]},
]

names = {
    # forecast.rs:2855-2961: every non-auto model returns exactly its enum name; auto models a prefix
    "exact": ["Naive", "SMA", "SeasonalNaive", "SES", "SESOptimized", "RandomWalkDrift", "Holt", "HoltWinters", "SeasonalES",
              "SeasonalESOptimized", "ETS", "ARIMA"],
    "prefix": ["AutoETS"],
}

here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "reference_kats.json"), "w") as f:
    json.dump({"cases": cases, "errors": errors, "interpolation": interp, "names": names, "unit": unit}, f, indent=1)
print("wrote", len(cases), "cases,", len(errors), "error cases")
