"""Multi-GPU: series-id ranges shard across ranks with no data-path collective; the only exchange
is one gather of the forecast chunks to rank 0 (RCCL over xGMI when the backend is "nccl").

Sizes per rank: n_local * h * 3 fp64 (2.6 MB for the M5 shape) + model codes / status, so the
gather is a point-to-point fan-in over the root's seven xGMI links, not a ring.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous series-id range [lo, hi) of `rank` (ceil(N/G) per rank)."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def shard_ranges_balanced(lengths, world: int) -> list[tuple[int, int]]:
    """Contiguous series-id ranges with (nearly) equal total length per rank -- the cost-aware variant of SURVEY.md 8e for
    ragged batches (the fit streams 8 T_s bytes per pass, so a rank's work is proportional to the sum of its lengths).
    Boundaries sit where the running sum of lengths crosses k / world of the total; every rank gets a (possibly empty) range."""
    import numpy as np
    lens = np.asarray(lengths, dtype=np.int64)
    n = len(lens)
    csum = np.concatenate([[0], np.cumsum(lens)])
    total = int(csum[-1])
    cuts = [0]
    for k in range(1, world):
        target = total * k / world
        i = int(np.searchsorted(csum, target, side="left"))
        i = min(max(i, cuts[-1]), n)
        if i > cuts[-1] and abs(csum[i - 1] - target) <= abs(csum[i] - target):
            i -= 1 if i - 1 >= cuts[-1] else 0
        cuts.append(i)
    cuts.append(n)
    return [(cuts[k], cuts[k + 1]) for k in range(world)]


class GatherPlan:
    """The step's one exchange, allocation-free: every rank owns ONE send buffer and the root ONE receive buffer, made once and reused
    by every step -- forecasts, intervals, model codes and status travel as one message per rank (round 6; rounds 1-5 issued five
    collectives per step, each with its own padded temporary and a torch.cat on the root: nothing at 2.6 MB per rank, the wrong shape
    for the 84 MB-per-rank gather of the 1M-series configuration).  Layout of a rank's chunk, `per` = ceil(N / G) rows:
    [yhat per x h f64][lower per x h f64][upper per x h f64][model_code per i32][status per i32] -- every section 8-byte aligned."""

    def __init__(self, n_total: int, h: int, rank: int, world: int, device, dst: int = 0):
        self.n_total, self.h, self.rank, self.world, self.dst = int(n_total), int(h), int(rank), int(world), int(dst)
        self.per = (self.n_total + self.world - 1) // self.world
        per, hh = self.per, max(self.h, 0)
        self.sections = {"yhat": (0, per * hh * 8, torch.float64, (per, hh)),
                         "lower": (per * hh * 8, per * hh * 8, torch.float64, (per, hh)),
                         "upper": (2 * per * hh * 8, per * hh * 8, torch.float64, (per, hh)),
                         "model_code": (3 * per * hh * 8, per * 4, torch.int32, (per,)),
                         "status": (3 * per * hh * 8 + ((per * 4 + 7) // 8) * 8, per * 4, torch.int32, (per,))}
        self.chunk_bytes = 3 * per * hh * 8 + 2 * (((per * 4 + 7) // 8) * 8)
        self.device = torch.device(device)
        self.send = torch.zeros(self.chunk_bytes, dtype=torch.uint8, device=self.device)
        self.recv = torch.zeros(self.world * self.chunk_bytes, dtype=torch.uint8, device=self.device) if self.rank == self.dst else None
        self._parts = [self.recv[k * self.chunk_bytes:(k + 1) * self.chunk_bytes] for k in range(self.world)] if self.recv is not None else None

    def _view(self, buf, key):
        off, nbytes, dtype, shape = self.sections[key]
        return buf[off:off + nbytes].view(dtype).view(shape)

    def gather(self, local: dict):
        """One collective.  Returns on the root {'yhat', 'lower', 'upper' [N, h] f64, 'model_code', 'status' [N] i32} -- rank-major row
        order = series order (contiguous ranges) -- as fresh tensors assembled from the receive buffer's sections; None elsewhere."""
        for key in self.sections:
            x = local[key]
            if x.device != self.device:
                x = x.to(self.device)
            self._view(self.send, key)[: x.shape[0]].copy_(x)
        if self.rank == self.dst:
            dist.gather(self.send, self._parts, dst=self.dst)
            out = {}
            for key, (_, _, dtype, shape) in self.sections.items():
                rows = torch.empty((self.world * self.per,) + tuple(shape[1:]), dtype=dtype, device=self.device)
                for k in range(self.world):
                    rows[k * self.per:(k + 1) * self.per].copy_(self._view(self._parts[k], key))
                out[key] = rows[: self.n_total]
            return out
        dist.gather(self.send, None, dst=self.dst)
        return None


_plans: dict = {}


def gather_forecasts(local: dict, n_total: int, rank: int, world: int, dst: int = 0):
    """Gather {'yhat','lower','upper' [n_local,h] f64, 'model_code','status' [n_local] i32} to `dst`: ONE collective over one
    pre-allocated buffer per rank (GatherPlan; the plan of a shape is made once and kept).  Chunks are padded to ceil(N/G) rows;
    returns the assembled dict on `dst`, None elsewhere.  With world == 1 nothing is communicated."""
    if world == 1:
        return local
    host_only = dist.get_backend() == "gloo"          # gloo gathers host tensors only (CPU tests, the one-GPU debug mode)
    h = int(local["yhat"].shape[1]) if local["yhat"].dim() > 1 else 0
    device = torch.device("cpu") if host_only else local["yhat"].device
    key = (int(n_total), h, int(rank), int(world), int(dst), str(device))
    plan = _plans.get(key)
    if plan is None:
        plan = _plans[key] = GatherPlan(n_total, h, rank, world, device, dst)
    return plan.gather(local)
