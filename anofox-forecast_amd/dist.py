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


def gather_forecasts(local: dict, n_total: int, rank: int, world: int, dst: int = 0):
    """Gather {'yhat','lower','upper' [n_local,h] f64, 'model_code','status' [n_local] i32} to `dst`.

    Chunks are padded to ceil(N/G) rows so that one all-equal-size gather suffices; returns the
    assembled dict on `dst`, None elsewhere.  With world == 1 nothing is communicated.
    """
    if world == 1:
        return local
    per = (n_total + world - 1) // world
    out = {}
    host_only = dist.get_backend() == "gloo"          # gloo gathers host tensors only (CPU tests, the one-GPU debug mode)
    for key in ("yhat", "lower", "upper", "model_code", "status"):
        x = local[key].cpu() if host_only else local[key]
        pad_shape = (per,) + tuple(x.shape[1:])
        buf = torch.zeros(pad_shape, dtype=x.dtype, device=x.device)
        buf[: x.shape[0]] = x
        if rank == dst:
            parts = [torch.empty_like(buf) for _ in range(world)]
            dist.gather(buf, parts, dst=dst)
            out[key] = torch.cat(parts, dim=0)[:n_total]
        else:
            dist.gather(buf, None, dst=dst)
    return out if rank == dst else None
