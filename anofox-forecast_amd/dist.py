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


def gather_forecasts(local: dict, n_total: int, rank: int, world: int, dst: int = 0):
    """Gather {'yhat','lower','upper' [n_local,h] f64, 'model_code','status' [n_local] i32} to `dst`.

    Chunks are padded to ceil(N/G) rows so that one all-equal-size gather suffices; returns the
    assembled dict on `dst`, None elsewhere.  With world == 1 nothing is communicated.
    """
    if world == 1:
        return local
    per = (n_total + world - 1) // world
    out = {}
    for key in ("yhat", "lower", "upper", "model_code", "status"):
        x = local[key]
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
