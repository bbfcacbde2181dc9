"""Device-resident batches: the series block lives in HBM as a torch tensor (plumbing only:
allocation, streams, torch.distributed); all arithmetic is in libanofox_fcst_hip.so.

Layout: time-major fp64 block Y[t, s] of shape [t_max, ld], ld = n_series rounded up to 64, so
the 64 lanes of a wave read one 512-byte row segment per time step.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib as _lib


def pack_time_major(series_major: np.ndarray, ld: int | None = None) -> np.ndarray:
    """[n, T] series-major host array -> [T, ld] time-major (zero padded)."""
    n, T = series_major.shape
    ld = ld or (n + 63) // 64 * 64
    out = np.zeros((T, ld), dtype=np.float64)
    out[:, :n] = series_major.T
    return out


class DeviceBatch:
    """anofox_hip_batch_* over torch-owned HBM."""

    def __init__(self, n_series: int, t_max: int, opts: _lib.ForecastOptions, device: torch.device | str = "cuda:0"):
        self.L = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceBatch needs a HIP device (no CPU fallback)")
        idx = self.device.index or 0
        if self.L.anofox_hip_set_device(idx) != 0:
            raise RuntimeError(f"hipSetDevice({idx}) failed")
        torch.cuda.set_device(idx)
        self.n, self.t_max, self.opts = int(n_series), int(t_max), opts
        self.h = int(opts.horizon)
        handle = C.c_void_p()
        err = _lib.AnofoxError()
        if not self.L.anofox_hip_batch_create(self.n, self.t_max, C.byref(opts), C.byref(handle), C.byref(err)):
            raise RuntimeError(f"anofox_hip_batch_create failed: [{err.code}] {err.message.decode()}")
        self.handle = handle
        self.ld = int(self.L.anofox_hip_batch_ld(handle))
        self._y = self._len = None
        self._side = None            # launch stream of runs asked for on torch's null stream (run)

    def set_block(self, y_time_major: torch.Tensor, lengths: torch.Tensor):
        assert y_time_major.dtype == torch.float64 and y_time_major.is_cuda and y_time_major.is_contiguous()
        assert tuple(y_time_major.shape) == (self.t_max, self.ld), (tuple(y_time_major.shape), (self.t_max, self.ld))
        assert lengths.dtype == torch.int32 and lengths.is_cuda and lengths.numel() >= self.n
        self._y, self._len = y_time_major, lengths
        err = _lib.AnofoxError()
        if not self.L.anofox_hip_batch_set_device_block(self.handle, y_time_major.data_ptr(), self.ld, lengths.data_ptr(), C.byref(err)):
            raise RuntimeError(f"set_device_block failed: [{err.code}] {err.message.decode()}")

    def set_fixed_params(self, alpha: float, beta: float = 0.0, gamma: float = 0.0, phi: float = 1.0):
        """ETS(spec) with given smoothing parameters: one streamed pass per series, no optimiser (BASELINE config 2)."""
        err = _lib.AnofoxError()
        if not self.L.anofox_hip_batch_set_fixed_params(self.handle, float(alpha), float(beta), float(gamma), float(phi), C.byref(err)):
            raise RuntimeError(f"set_fixed_params failed: [{err.code}] {err.message.decode()}")

    def set_arima_method(self, method: int):
        """AutoARIMA estimation method of this batch: lib.ARIMA_CSS (default) or lib.ARIMA_CSS_ML (exact-likelihood refit)."""
        err = _lib.AnofoxError()
        if not self.L.anofox_hip_batch_set_arima_method(self.handle, int(method), C.byref(err)):
            raise RuntimeError(f"set_arima_method failed: [{err.code}] {err.message.decode()}")

    def periods(self) -> np.ndarray:
        """The seasonal period every series runs with (auto-detected on the device when the options ask for it)."""
        out = np.zeros(self.n, dtype=np.int32)
        if not self.L.anofox_hip_batch_periods(self.handle, out.ctypes.data):
            raise RuntimeError("anofox_hip_batch_periods: no block set")
        return out

    def run(self, stream: torch.cuda.Stream | None = None):
        """One fit + forecast of the block, asynchronous and ORDERED on `stream` (torch's current stream by default): what was
        enqueued there before is visible to the run, what is enqueued there afterwards -- torch ops on results(), a collective -- sees
        the run's results.  The null stream cannot carry the run itself (the C entry reads a null handle as "the batch's own stream", a
        non-blocking one the null stream does not wait for: until the last day of round 6 a consumer on torch's default stream, the
        multi-rank gather of bench.py among them, could read the result arrays while the closing kernels were still writing them),
        so a run asked for on the null stream goes to a side stream of this batch, fenced against it on both ends."""
        cur = stream if stream is not None else torch.cuda.current_stream(self.device)
        st = cur
        if cur.cuda_stream == 0:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            self._side.wait_stream(cur)
            st = self._side
        err = _lib.AnofoxError()
        ok = self.L.anofox_hip_batch_run(self.handle, C.c_void_p(st.cuda_stream), C.byref(err))
        if st is not cur:
            cur.wait_stream(st)
        if not ok:
            raise RuntimeError(f"anofox_hip_batch_run failed: [{err.code}] {err.message.decode()}")

    def stats(self) -> dict:
        s = _lib.AnofoxHipStats()
        if not self.L.anofox_hip_batch_stats(self.handle, C.byref(s)):
            raise RuntimeError("anofox_hip_batch_stats failed")
        return {f: getattr(s, f) for f, _ in s._fields_}

    SPEC_CLASSES = ("additive", "general", "damped_mul_trend")

    def lane_stats(self) -> dict:
        """Lane-level efficiency of the round kernels of the last run (include/anofox_fcst_hip.h AnofoxHipLaneStats): per spec class and
        per spec, live lane-passes / (64 x wave passes)."""
        s = _lib.AnofoxHipLaneStats()
        if not self.L.anofox_hip_batch_lane_stats(self.handle, C.byref(s), C.sizeof(s)):
            raise RuntimeError("anofox_hip_batch_lane_stats failed")
        eff = lambda live, waves: round(live / (64.0 * waves), 4) if waves else None
        E, T, S = "AM", ("N", "A", "Ad", "M", "Md"), "NAM"
        out = {"by_class": {}, "by_spec": {}}
        tw = tl = 0
        for c, name in enumerate(self.SPEC_CLASSES):
            w, l = int(s.wave_passes[c]), int(s.live_lane_passes[c])
            tw += w; tl += l
            out["by_class"][name] = {"wave_passes": w, "live_lane_passes": l, "lane_efficiency": eff(l, w)}
        out["wave_passes"], out["live_lane_passes"], out["lane_efficiency"] = tw, tl, eff(tl, tw)
        for k in range(int(s.n_slots)):
            sid = int(s.slot_spec_id[k])
            name = E[sid // 15] + T[(sid % 15) // 3] + S[sid % 3]
            out["by_spec"][name] = {"wave_passes": int(s.slot_wave_passes[k]), "lane_efficiency": eff(int(s.slot_live_lane_passes[k]), int(s.slot_wave_passes[k]))}
        return out

    def results(self) -> dict:
        """Zero-copy torch views of the device result arrays."""
        ptrs = [C.c_void_p() for _ in range(5)]
        self.L.anofox_hip_batch_device_results(self.handle, *[C.byref(p) for p in ptrs])

        def view(ptr, shape, dtype, itemsize):
            n = int(np.prod(shape))
            if n == 0:
                return torch.empty(shape, dtype=dtype, device=self.device)
            iface = {"shape": tuple(shape), "typestr": {8: "<f8", 4: "<i4"}[itemsize], "data": (ptr.value, False), "version": 2}
            holder = type("H", (), {"__cuda_array_interface__": iface})()
            return torch.as_tensor(holder, device=self.device)

        return {
            "yhat": view(ptrs[0], (self.n, self.h), torch.float64, 8),
            "lower": view(ptrs[1], (self.n, self.h), torch.float64, 8),
            "upper": view(ptrs[2], (self.n, self.h), torch.float64, 8),
            "model_code": view(ptrs[3], (self.n,), torch.int32, 4),
            "status": view(ptrs[4], (self.n,), torch.int32, 4),
        }

    def model_name(self, code: int, series: int | None = None) -> str:
        """Name of a model code; with `series`, the name anofox_hip_batch_fetch gives that series (an AutoARIMA code carries the
        period the series was fitted with -- detected periods included)."""
        buf = (C.c_char * 64)()
        if series is None:
            self.L.anofox_hip_model_name(C.byref(self.opts), int(code), buf)
        else:
            self.L.anofox_hip_batch_model_name(self.handle, int(series), int(code), buf)
        return buf.value.decode()

    def close(self):
        if getattr(self, "handle", None):
            self.L.anofox_hip_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
