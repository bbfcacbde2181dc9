"""CPU: the multi-GPU path (series-range sharding + the gather of forecast chunks) with world_size 2
over gloo -- the same code bench.py runs over RCCL."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, n_total, h, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from anofox_forecast_amd.dist import gather_forecasts, shard_range
    lo, hi = shard_range(n_total, rank, world)
    ids = torch.arange(lo, hi, dtype=torch.float64)
    local = {"yhat": ids[:, None] * 10 + torch.arange(h, dtype=torch.float64)[None, :],
             "lower": ids[:, None] - 1 + torch.zeros(h, dtype=torch.float64)[None, :],
             "upper": ids[:, None] + 1 + torch.zeros(h, dtype=torch.float64)[None, :],
             "model_code": (100 + ids % 30).to(torch.int32), "status": torch.zeros(hi - lo, dtype=torch.int32)}
    out = gather_forecasts(local, n_total, rank, world)
    if rank == 0:
        ret["yhat"] = out["yhat"].numpy()
        ret["code"] = out["model_code"].numpy()
    else:
        assert out is None
    dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    from anofox_forecast_amd.dist import shard_range
    for n, g in [(30490, 8), (30490, 1), (7, 4), (1000000, 8), (3, 8)]:
        r = [shard_range(n, k, g) for k in range(g)]
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert max(hi - lo for lo, hi in r) == -(-n // g)


def test_gather_world2_gloo():
    n_total, h, world = 37, 5, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29517 + os.getpid() % 500, n_total, h, ret), nprocs=world, join=True)
    ids = np.arange(n_total, dtype=np.float64)
    np.testing.assert_array_equal(ret["yhat"], ids[:, None] * 10 + np.arange(h)[None, :])
    np.testing.assert_array_equal(ret["code"], (100 + ids % 30).astype(np.int32))


def test_balanced_shards_cover_and_balance():
    import numpy as np
    from anofox_forecast_amd.dist import shard_ranges_balanced
    rng = np.random.default_rng(2)
    lens = rng.integers(3, 2000, 5000)
    for world in (1, 2, 3, 8):
        r = shard_ranges_balanced(lens, world)
        assert r[0][0] == 0 and r[-1][1] == len(lens) and all(r[k][1] == r[k + 1][0] for k in range(world - 1))
        work = np.array([lens[a:b].sum() for a, b in r], dtype=float)
        assert work.max() <= work.mean() * 1.01 + 2000            # within one series of the ideal share
    assert shard_ranges_balanced([5, 5], 4) and sum(b - a for a, b in shard_ranges_balanced([5, 5], 4)) == 2


def _pipeline_worker(rank, world, port, n_total, T, h, scaling, ret):
    """What bench.py does per rank, with the oracle standing in for the device (no GPU here): regenerate this rank's series-id
    range, forecast it, gather the chunks to rank 0."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from anofox_forecast_amd import synth
    from anofox_forecast_amd.dist import gather_forecasts, shard_range
    from oracle import oracle as O
    if scaling == "strong":
        total = n_total
        lo, hi = shard_range(total, rank, world)
    else:
        total = n_total * world
        lo, hi = rank * n_total, (rank + 1) * n_total
    Y = synth.gen_series(synth.SEED_M5, lo, hi - lo, T, 7, False)
    r = O.ets_fixed_batch(Y.reshape(-1), np.arange(hi - lo + 1, dtype=np.int64) * T, "AAA", 7, 0.2, 0.05, 0.1, 1.0, h, n_threads=2)
    local = {"yhat": torch.from_numpy(r["yhat"]), "lower": torch.from_numpy(r["lower"]), "upper": torch.from_numpy(r["upper"]),
             "model_code": torch.zeros(hi - lo, dtype=torch.int32), "status": torch.from_numpy(r["status"])}
    out = gather_forecasts(local, total, rank, world)
    if rank == 0:
        ret["yhat"] = out["yhat"].numpy()
        ret["status"] = out["status"].numpy()
    dist.destroy_process_group()


def test_sharded_pipeline_world2_gloo():
    """Two ranks, both scaling modes of bench.py: the gathered forecasts equal one process over the whole series-id range
    (series regenerate from their ids, shards are contiguous, chunks arrive in rank order, uneven last shard included)."""
    from anofox_forecast_amd import synth
    from oracle import oracle as O
    T, h, world = 60, 6, 2
    for scaling, n_arg in (("strong", 1031), ("weak", 300)):
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_pipeline_worker, args=(world, 29117 + os.getpid() % 500, n_arg, T, h, scaling, ret), nprocs=world, join=True)
        total = n_arg if scaling == "strong" else n_arg * world
        Y = synth.gen_series(synth.SEED_M5, 0, total, T, 7, False)
        ref = O.ets_fixed_batch(Y.reshape(-1), np.arange(total + 1, dtype=np.int64) * T, "AAA", 7, 0.2, 0.05, 0.1, 1.0, h, n_threads=2)
        np.testing.assert_array_equal(ret["yhat"], ref["yhat"])
        np.testing.assert_array_equal(ret["status"], ref["status"])
