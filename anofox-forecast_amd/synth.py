"""Deterministic synthetic M5-shape batches.

Counter-based: series block b (1024 series) is drawn from Philox keyed by (seed, b), so any
shard regenerates exactly its slice.  Intermittent retail demand: log-normal level, weekly
profile, slow trend, Poisson counts, leading zeros.  `positive=True` adds 1 so that the
multiplicative ETS specs are admissible ("full grid" variant).
"""
from __future__ import annotations

import numpy as np

BLOCK = 1024
SEED_M5 = 20260101
SEED_STRESS = 20260102


def gen_block(seed: int, block: int, T: int, m: int = 7, positive: bool = False) -> np.ndarray:
    """Return [BLOCK, T] float64 series of block `block`."""
    rng = np.random.Generator(np.random.Philox(key=[seed, block]))
    level = np.exp(rng.normal(0.0, 1.2, size=BLOCK))
    phase = rng.uniform(0.0, 2.0 * np.pi, size=BLOCK)
    tau = rng.uniform(-1.0, 1.0, size=BLOCK)
    lead = rng.integers(0, T // 3 + 1, size=BLOCK)
    t = np.arange(T, dtype=np.float64)
    prof = 1.0 + 0.3 * np.sin(2.0 * np.pi * (np.arange(T) % m)[None, :] / m + phase[:, None])
    trend = 1.0 + 0.0002 * t[None, :] * tau[:, None]
    lam = np.maximum(level[:, None] * prof * trend, 0.0)
    y = rng.poisson(lam).astype(np.float64)
    y[t[None, :] < lead[:, None]] = 0.0
    if positive:
        y += 1.0
    return y


def gen_series(seed: int, start: int, count: int, T: int, m: int = 7, positive: bool = False) -> np.ndarray:
    """Series [start, start+count) as a [count, T] float64 array (series-major)."""
    out = np.empty((count, T), dtype=np.float64)
    s = start
    while s < start + count:
        b, off = divmod(s, BLOCK)
        take = min(BLOCK - off, start + count - s)
        out[s - start:s - start + take] = gen_block(seed, b, T, m, positive)[off:off + take]
        s += take
    return out
