"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE, not part of the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "build", "liboracle.so")


class AnofoxError(C.Structure):
    _fields_ = [("code", C.c_int), ("message", C.c_char * 256)]


class ForecastOptions(C.Structure):
    _fields_ = [
        ("model", C.c_char * 32),
        ("ets_model", C.c_char * 8),
        ("horizon", C.c_int),
        ("confidence_level", C.c_double),
        ("seasonal_period", C.c_int),
        ("auto_detect_seasonality", C.c_bool),
        ("include_fitted", C.c_bool),
        ("include_residuals", C.c_bool),
        ("window", C.c_int),
        ("seasonal_periods_str", C.c_char * 64),
        ("model_pool", C.c_char * 32),
        ("laplace_variant", C.c_char * 16),
        ("laplace_seasonal_batch_init", C.c_bool),
    ]


class ForecastResult(C.Structure):
    _fields_ = [
        ("point_forecasts", C.POINTER(C.c_double)),
        ("lower_bounds", C.POINTER(C.c_double)),
        ("upper_bounds", C.POINTER(C.c_double)),
        ("fitted_values", C.POINTER(C.c_double)),
        ("residuals", C.POINTER(C.c_double)),
        ("n_forecasts", C.c_size_t),
        ("n_fitted", C.c_size_t),
        ("model_name", C.c_char * 64),
        ("aic", C.c_double),
        ("bic", C.c_double),
        ("mse", C.c_double),
    ]


assert C.sizeof(ForecastOptions) == 184 and C.sizeof(ForecastResult) == 144 and C.sizeof(AnofoxError) == 260


def make_options(model, horizon, *, ets_model="", seasonal_period=0, confidence_level=0.90,
                 auto_detect=None, include_fitted=False, include_residuals=False, window=0,
                 model_pool="", seasonal_periods_str=""):
    """Build the option block the way the reference binding does
    (src/scalar_functions/ts_forecast_scalar.cpp:439-468)."""
    o = ForecastOptions()
    C.memset(C.byref(o), 0, C.sizeof(o))
    o.model = model.encode()[:31]
    o.ets_model = ets_model.encode()[:7]
    o.horizon = int(horizon)
    o.confidence_level = float(confidence_level)
    o.seasonal_period = int(seasonal_period)
    if auto_detect is None:
        auto_detect = (seasonal_period == 0 and not seasonal_periods_str)
    o.auto_detect_seasonality = bool(auto_detect)
    o.include_fitted = bool(include_fitted)
    o.include_residuals = bool(include_residuals)
    o.window = int(window)
    o.seasonal_periods_str = seasonal_periods_str.encode()[:63]
    o.model_pool = model_pool.encode()[:31]
    return o


def build(force=False):
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB)
            for f in os.listdir(_HERE) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_ts_forecast.restype = C.c_bool
        _lib.oracle_ts_forecast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(ForecastOptions),
                                           C.POINTER(ForecastResult), C.POINTER(AnofoxError)]
        _lib.oracle_free_forecast_result.argtypes = [C.POINTER(ForecastResult)]
        _lib.oracle_forecast_batch.restype = C.c_int
        _lib.oracle_forecast_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(ForecastOptions),
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.oracle_det_log.restype = C.c_double
        _lib.oracle_det_log.argtypes = [C.c_double]
        _lib.oracle_det_exp.restype = C.c_double
        _lib.oracle_det_exp.argtypes = [C.c_double]
        _lib.oracle_detect_seasonality_first.restype = C.c_int
        _lib.oracle_detect_seasonality_first.argtypes = [C.c_void_p, C.c_size_t]
        _lib.oracle_fill_nulls_interpolate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.oracle_auto_ets_search.restype = C.c_int
        _lib.oracle_auto_ets_search.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                               C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return _lib


def validity_mask(valid):
    """bool array -> DuckDB-style uint64 bitmask (bit i%64 of word i/64)."""
    valid = np.asarray(valid, dtype=bool)
    words = np.zeros((len(valid) + 63) // 64, dtype=np.uint64)
    for i, v in enumerate(valid):
        if v:
            words[i // 64] |= np.uint64(1) << np.uint64(i % 64)
    return words


def forecast(values, opts, valid=None):
    """One series through the oracle. Returns dict(ok, code, message, point, lower, upper, ...)."""
    y = np.ascontiguousarray(values, dtype=np.float64)
    res = ForecastResult()
    C.memset(C.byref(res), 0, C.sizeof(res))
    err = AnofoxError()
    mask = validity_mask(valid) if valid is not None else None
    dummy = np.zeros(1)
    ok = lib().oracle_ts_forecast(y.ctypes.data if len(y) else dummy.ctypes.data, mask.ctypes.data if mask is not None else None,
                                  len(y), C.byref(opts), C.byref(res), C.byref(err))
    out = {"ok": bool(ok), "code": int(err.code), "message": err.message.decode(errors="replace")}
    if ok:
        h = res.n_forecasts
        out["point"] = np.array(res.point_forecasts[:h], dtype=np.float64)
        out["lower"] = np.array(res.lower_bounds[:h], dtype=np.float64)
        out["upper"] = np.array(res.upper_bounds[:h], dtype=np.float64)
        out["model_name"] = res.model_name.decode()
        out["mse"] = res.mse
        out["aic"] = res.aic
        out["bic"] = res.bic
        out["n_fitted"] = res.n_fitted
        if res.fitted_values:
            out["fitted"] = np.array(res.fitted_values[:res.n_fitted])
        if res.residuals:
            out["residuals"] = np.array(res.residuals[:len(y)])
        lib().oracle_free_forecast_result(C.byref(res))
    return out


def forecast_batch(values_concat, offsets, opts, n_threads=0):
    """Many series (ragged, concatenated) through the oracle with OpenMP."""
    v = np.ascontiguousarray(values_concat, dtype=np.float64)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(off) - 1
    h = opts.horizon
    yhat = np.full((n, h), np.nan)
    lo = np.full((n, h), np.nan)
    hi = np.full((n, h), np.nan)
    status = np.zeros(n, dtype=np.int32)
    names = np.zeros((n, 64), dtype=np.uint8)
    used = lib().oracle_forecast_batch(v.ctypes.data, off.ctypes.data, n, C.byref(opts), yhat.ctypes.data,
                                       lo.ctypes.data, hi.ctypes.data, status.ctypes.data, names.ctypes.data,
                                       int(n_threads))
    name_list = [bytes(r).split(b"\0", 1)[0].decode() for r in names]
    return {"yhat": yhat, "lower": lo, "upper": hi, "status": status, "names": name_list, "threads": used}


def ets_inspect(values, period, spec_id=-1, pool=0):
    """Test hook: parameters, criteria, final states and one-step fitted values of spec `spec_id`, or of
    the spec AutoETS selects when it is negative.  Returns None when nothing can be fitted."""
    y = np.ascontiguousarray(values, dtype=np.float64)
    L = lib()
    L.oracle_ets_inspect.restype = C.c_int
    L.oracle_ets_inspect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    par = np.zeros(8)
    states = np.full(2 + 64, np.nan)
    fitted = np.full(len(y), np.nan)
    sid = L.oracle_ets_inspect(y.ctypes.data, len(y), int(period), int(pool), int(spec_id), par.ctypes.data, states.ctypes.data, fitted.ctypes.data)
    if sid < 0:
        return None
    keys = ("alpha", "beta", "gamma", "phi", "aic", "aicc", "bic", "sse")
    out = dict(zip(keys, par))
    out.update(spec_id=sid, level=states[0], trend=states[1], seasonal_states=states[2:2 + max(period, 1)], fitted_values=fitted)
    return out


def ets_fixed_batch(values_concat, offsets, notation, period, alpha, beta, gamma, phi, h, conf=0.90, n_threads=0):
    """ETS(notation) with GIVEN smoothing parameters over many series (BASELINE config 2): initial states as in the fitted
    path, one pass, forecasts + intervals.  The checker of anofox_hip_batch_set_fixed_params."""
    v = np.ascontiguousarray(values_concat, dtype=np.float64)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(off) - 1
    L = lib()
    L.oracle_ets_fixed_batch.restype = C.c_int
    L.oracle_ets_fixed_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_int, C.c_double, C.c_double, C.c_double,
                                         C.c_double, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    yhat = np.full((n, h), np.nan)
    lo = np.full((n, h), np.nan)
    hi = np.full((n, h), np.nan)
    status = np.zeros(n, dtype=np.int32)
    used = L.oracle_ets_fixed_batch(v.ctypes.data, off.ctypes.data, n, notation.encode(), int(period), float(alpha), float(beta),
                                    float(gamma), float(phi), int(h), float(conf), yhat.ctypes.data, lo.ctypes.data, hi.ctypes.data,
                                    status.ctypes.data, int(n_threads))
    if used < 0:
        raise ValueError(f"bad ETS notation {notation!r}")
    return {"yhat": yhat, "lower": lo, "upper": hi, "status": status, "threads": used}
