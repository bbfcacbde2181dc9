"""anofox-forecast_amd: MI355X-native batch forecasting behind the reference's `ts_forecast_by`.

Only the hot path: csrc/ (HIP kernels + C-ABI), lib (ctypes binding), api (operator mirror),
device (HBM-resident batches), dist (series sharding + gather), synth (benchmark inputs).
"""
from . import lib  # noqa: F401
from .api import InvalidInputException, forecast_batch, forecast_series, ts_forecast_by  # noqa: F401
