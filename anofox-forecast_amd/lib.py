"""ctypes binding of libanofox_fcst_hip.so (the C-ABI of include/anofox_fcst_hip.h).

The product path is the HIP library and nothing else: importing this module without the built
library, or calling it without a GPU, fails loudly.  The structs mirror the reference's
cbindgen header (src/include/anofox_fcst_ffi.h:72-84, 269-272, 1036-1145).
"""
from __future__ import annotations

import ctypes as C
import os

# The batch path runs the candidate ETS specs on concurrent HIP streams; ROCm maps streams onto 4
# hardware queues by default, which serialises them.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # measured best of 4/8/16/24/32 on MI355X

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ANOFOX_HIP_LIB", os.path.join(_HERE, "libanofox_fcst_hip.so"))   # override: A/B builds only

SUCCESS, NULL_POINTER, INVALID_INPUT, COMPUTATION_ERROR, ALLOCATION_ERROR, INVALID_MODEL = 0, 1, 2, 3, 4, 5
INSUFFICIENT_DATA, INVALID_DATE_FORMAT, INVALID_FREQUENCY, PANIC_CAUGHT, INTERNAL_ERROR = 6, 7, 8, 9, 10


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


class AnofoxHipStats(C.Structure):
    _fields_ = [
        ("n_series", C.c_uint64),
        ("t_max", C.c_uint64),
        ("n_problems", C.c_uint64),
        ("total_passes", C.c_uint64),
        ("max_passes", C.c_uint64),
        ("total_evals", C.c_uint64),
        ("algorithmic_bytes", C.c_uint64),
        ("fit_kernel_ms", C.c_double),
        ("total_device_ms", C.c_double),
        ("fit_kernel_launches", C.c_uint32),
        ("y_storage", C.c_uint32),
        ("total_iters", C.c_uint64),
        ("min_pass_bytes", C.c_uint64),
    ]


class AnofoxHipLaneStats(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint64),
        ("wave_passes", C.c_uint64 * 3),
        ("live_lane_passes", C.c_uint64 * 3),
        ("n_slots", C.c_uint32),
        ("reserved", C.c_uint32),
        ("slot_spec_id", C.c_int32 * 30),
        ("slot_wave_passes", C.c_uint64 * 30),
        ("slot_live_lane_passes", C.c_uint64 * 30),
    ]


assert C.sizeof(ForecastOptions) == 184 and C.sizeof(ForecastResult) == 144 and C.sizeof(AnofoxError) == 260

# every symbol include/anofox_fcst_hip.h declares
class AnofoxHipInspection(C.Structure):
    _fields_ = [("model_code", C.c_int32), ("status", C.c_int32), ("seasonal_period", C.c_int32), ("reserved", C.c_int32),
                ("alpha", C.c_double), ("beta", C.c_double), ("gamma", C.c_double), ("phi", C.c_double),
                ("aic", C.c_double), ("aicc", C.c_double), ("bic", C.c_double), ("sse", C.c_double),
                ("level", C.c_double), ("trend", C.c_double)]


EXPORTED_SYMBOLS = [
    "anofox_ts_forecast", "anofox_free_forecast_result", "anofox_fcst_version", "anofox_ts_forecast_batch",
    "anofox_hip_device_count", "anofox_hip_set_device", "anofox_hip_batch_create", "anofox_hip_batch_destroy",
    "anofox_hip_batch_ld", "anofox_hip_batch_pack_host", "anofox_hip_batch_set_device_block", "anofox_hip_batch_run",
    "anofox_hip_batch_stats", "anofox_hip_batch_device_results", "anofox_hip_batch_fetch", "anofox_hip_model_name", "anofox_hip_batch_model_name",
    "anofox_hip_ingest_create", "anofox_hip_ingest_destroy", "anofox_hip_ingest_append", "anofox_hip_ingest_finish",
    "anofox_hip_ingest_group_keys", "anofox_hip_ingest_last_dates", "anofox_hip_ingest_lengths", "anofox_hip_ingest_values",
    "anofox_hip_ingest_validity", "anofox_hip_batch_pack_ingest", "anofox_hip_batch_inspect",
    "anofox_hip_batch_n_series", "anofox_hip_batch_periods", "anofox_hip_batch_set_fixed_params",
    "anofox_hip_set_devices", "anofox_hip_get_devices", "anofox_hip_set_min_series_per_device", "anofox_hip_shard_range",
    "anofox_hip_set_default_arima_method", "anofox_hip_batch_set_arima_method", "anofox_hip_release_caches",
    "anofox_hip_batch_run_many", "anofox_hip_batch_lane_stats", "anofox_hip_selftest_recip",
]

ARIMA_CSS, ARIMA_CSS_ML = 0, 1     # include/anofox_fcst_hip.h: ANOFOX_ARIMA_CSS / ANOFOX_ARIMA_CSS_ML

_lib = None


def load():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C anofox-forecast_amd/csrc` "
                           "(or __graft_entry__.build()); the backend has no CPU fallback")
    try:
        # PyTorch-ROCm bundles its own HIP runtime; load it first so that this library binds to the
        # same one (two HIP runtimes in one process do not share devices, streams or allocations).
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.anofox_ts_forecast.restype = C.c_bool
    L.anofox_ts_forecast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, P(ForecastOptions), P(ForecastResult), P(AnofoxError)]
    L.anofox_free_forecast_result.argtypes = [P(ForecastResult)]
    L.anofox_fcst_version.restype = C.c_char_p
    L.anofox_ts_forecast_batch.restype = C.c_bool
    L.anofox_ts_forecast_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, P(ForecastOptions), C.c_void_p,
                                          C.c_void_p, C.c_void_p, P(AnofoxError)]
    L.anofox_hip_device_count.restype = C.c_int
    L.anofox_hip_set_device.restype = C.c_int
    L.anofox_hip_set_device.argtypes = [C.c_int]
    L.anofox_hip_batch_create.restype = C.c_bool
    L.anofox_hip_batch_create.argtypes = [C.c_size_t, C.c_size_t, P(ForecastOptions), P(C.c_void_p), P(AnofoxError)]
    L.anofox_hip_batch_destroy.argtypes = [C.c_void_p]
    L.anofox_hip_batch_ld.restype = C.c_size_t
    L.anofox_hip_batch_ld.argtypes = [C.c_void_p]
    L.anofox_hip_batch_n_series.restype = C.c_size_t
    L.anofox_hip_batch_n_series.argtypes = [C.c_void_p]
    L.anofox_hip_batch_periods.restype = C.c_bool
    L.anofox_hip_batch_periods.argtypes = [C.c_void_p, C.c_void_p]
    L.anofox_hip_batch_set_fixed_params.restype = C.c_bool
    L.anofox_hip_batch_set_fixed_params.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, P(AnofoxError)]
    L.anofox_hip_batch_pack_host.restype = C.c_bool
    L.anofox_hip_batch_pack_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, P(AnofoxError)]
    L.anofox_hip_batch_set_device_block.restype = C.c_bool
    L.anofox_hip_batch_set_device_block.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, P(AnofoxError)]
    L.anofox_hip_batch_run.restype = C.c_bool
    L.anofox_hip_batch_run.argtypes = [C.c_void_p, C.c_void_p, P(AnofoxError)]
    L.anofox_hip_batch_stats.restype = C.c_bool
    L.anofox_hip_batch_stats.argtypes = [C.c_void_p, P(AnofoxHipStats)]
    L.anofox_hip_selftest_recip.restype = C.c_bool
    L.anofox_hip_selftest_recip.argtypes = [C.c_uint64, C.c_uint64, P(C.c_uint64), P(C.c_double)]
    L.anofox_hip_batch_lane_stats.restype = C.c_bool
    L.anofox_hip_batch_lane_stats.argtypes = [C.c_void_p, P(AnofoxHipLaneStats), C.c_size_t]
    L.anofox_hip_batch_device_results.restype = C.c_bool
    L.anofox_hip_batch_device_results.argtypes = [C.c_void_p] + [P(C.c_void_p)] * 5
    L.anofox_hip_batch_fetch.restype = C.c_bool
    L.anofox_hip_batch_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.anofox_hip_model_name.argtypes = [P(ForecastOptions), C.c_int32, C.c_char * 64]
    L.anofox_hip_batch_model_name.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_char * 64]
    L.anofox_hip_batch_model_name.restype = None
    L.anofox_hip_batch_inspect.restype = C.c_bool
    L.anofox_hip_batch_inspect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, P(AnofoxError)]
    L.anofox_hip_set_devices.restype = C.c_bool
    L.anofox_hip_set_devices.argtypes = [C.c_void_p, C.c_size_t]
    L.anofox_hip_get_devices.restype = C.c_size_t
    L.anofox_hip_get_devices.argtypes = [C.c_void_p, C.c_size_t]
    L.anofox_hip_set_min_series_per_device.argtypes = [C.c_size_t]
    L.anofox_hip_shard_range.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, P(C.c_size_t), P(C.c_size_t)]
    L.anofox_hip_set_default_arima_method.restype = C.c_bool
    L.anofox_hip_set_default_arima_method.argtypes = [C.c_int]
    L.anofox_hip_batch_set_arima_method.restype = C.c_bool
    L.anofox_hip_batch_set_arima_method.argtypes = [C.c_void_p, C.c_int, P(AnofoxError)]
    L.anofox_hip_release_caches.argtypes = []
    L.anofox_hip_batch_run_many.restype = C.c_bool
    L.anofox_hip_batch_run_many.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    # block 4: columnar ingest (host side only; usable without a GPU up to pack_ingest)
    L.anofox_hip_ingest_create.restype = C.c_void_p
    L.anofox_hip_ingest_destroy.argtypes = [C.c_void_p]
    L.anofox_hip_ingest_append.restype = C.c_bool
    L.anofox_hip_ingest_append.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, P(AnofoxError)]
    L.anofox_hip_ingest_finish.restype = C.c_bool
    L.anofox_hip_ingest_finish.argtypes = [C.c_void_p, P(C.c_size_t), P(C.c_size_t), P(AnofoxError)]
    for fn, rt in (("group_keys", P(C.c_int64)), ("last_dates", P(C.c_int64)), ("lengths", P(C.c_size_t)),
                   ("values", P(P(C.c_double))), ("validity", P(P(C.c_uint64)))):
        f = getattr(L, "anofox_hip_ingest_" + fn)
        f.restype = rt
        f.argtypes = [C.c_void_p]
    L.anofox_hip_batch_pack_ingest.restype = C.c_bool
    L.anofox_hip_batch_pack_ingest.argtypes = [C.c_void_p, C.c_void_p, P(AnofoxError)]
    _lib = L
    return L


def make_options(model, horizon, *, ets_model="", seasonal_period=0, confidence_level=0.90, auto_detect=None,
                 include_fitted=False, include_residuals=False, window=0, model_pool="", seasonal_periods_str=""):
    """Option block as the reference binding fills it (src/scalar_functions/ts_forecast_scalar.cpp:439-468,
    src/table_functions/ts_forecast_native.cpp:612-645): memset 0, strncpy, defaults conf 0.90,
    auto_detect = (seasonal_period == 0 and no seasonal_periods)."""
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


def set_devices(devices):
    """Devices the batch entry shards series ranges over (include/anofox_fcst_hip.h block 2); [] = the current device only."""
    L = load()
    arr = (C.c_int * len(devices))(*devices) if devices else None
    if not L.anofox_hip_set_devices(arr, len(devices)):
        raise ValueError(f"not visible devices: {devices!r}")


def shard_range(n_series, n_shards, shard):
    lo, hi = C.c_size_t(), C.c_size_t()
    load().anofox_hip_shard_range(n_series, n_shards, shard, C.byref(lo), C.byref(hi))
    return lo.value, hi.value
