#!/usr/bin/env python3
"""bench.py -- series/sec fit+forecast, AutoETS h=28 on M5-shape batches (BASELINE.json metric).

A "step" is one full pass of the hot path (prep -> every candidate ETS spec fitted per series by
per-lane Nelder-Mead -> AICc selection -> fallback chain -> intervals) over one synthetic batch that
is already resident in HBM when the timed region starts.  One process per GPU; series-id ranges
shard across ranks with no data-path collective (weak scaling: every rank owns a full batch) and the
only exchange is the gather of the forecast chunks to rank 0, inside the timed region.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads (--workload):
    autoets_m5_positive   (default) 30,490 x 1,913, h=28, m=7, strictly positive counts: all 25
                          valid specs of the 30-model grid are fitted for every series
    autoets_m5            same shape, raw intermittent counts (zeros -> only the 6 additive specs admissible)
    ets_aaa_m5            ETS(A,A,A) single spec on the same batch
    autoets_stress        n x 1,024 AutoETS (use --n-series; the 1M config shards over 8 GPUs)
    autoarima_m5          AutoARIMA stepwise search, m = 7, on the M5-shape batch (BASELINE config 4)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before HIP initialises: the spec streams need hardware queues

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="autoets_m5_positive")
    ap.add_argument("--n-series", type=int, default=0, help="series per GPU (0 = workload default)")
    ap.add_argument("--t", type=int, default=0, help="observations per series (0 = workload default)")
    ap.add_argument("--horizon", type=int, default=28)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="series for the CPU baseline (-1 auto, 0 skip)")
    return ap.parse_args()


WORKLOADS = {
    # name: (model, ets_model, n, T, m, positive, seed, cpu_sample)
    "autoets_m5_positive": ("AutoETS", "", 30490, 1913, 7, True, 20260101, 384),
    "autoets_m5": ("AutoETS", "", 30490, 1913, 7, False, 20260101, 8192),
    "ets_aaa_m5": ("ETS", "AAA", 30490, 1913, 7, False, 20260101, 8192),
    "autoets_stress": ("AutoETS", "", 125000, 1024, 7, False, 20260102, 8192),
    "autoarima_m5": ("AutoARIMA", "", 30490, 1913, 7, False, 20260101, 512),
    # single-spec probes (kernel efficiency without cross-kernel effects)
    "ets_amdn_stress": ("ETS", "AMdN", 125000, 1024, 7, True, 20260102, 256),
    "ets_mam_stress": ("ETS", "MAM", 125000, 1024, 7, True, 20260102, 256),
}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from anofox_forecast_amd import lib, synth
    from anofox_forecast_amd.device import DeviceBatch, pack_time_major
    from anofox_forecast_amd.dist import gather_forecasts

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    # ANOFOX_BENCH_ONE_GPU=1 (debug only): every rank uses GPU 0 and the gather runs over gloo -- exercises the N > 1 code
    # path on a single-GPU box; RCCL refuses two ranks on one device
    one_gpu = os.environ.get("ANOFOX_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)     # "nccl" is RCCL on ROCm

    model, ets_model, n_def, T_def, m, positive, seed, cpu_def = WORKLOADS[args.workload]
    n = args.n_series or n_def
    T = args.t or T_def
    h = args.horizon

    # ---- synthetic batch, resident in HBM before anything is timed ------------------------------
    t0 = time.time()
    Y = synth.gen_series(seed, rank * n, n, T, m, positive)                 # this rank's series-id range
    opts = lib.make_options(model, h, ets_model=ets_model, seasonal_period=m)
    batch = DeviceBatch(n, T, opts, dev)
    y_dev = torch.from_numpy(pack_time_major(Y, batch.ld)).to(dev)
    len_dev = torch.full((batch.ld,), T, dtype=torch.int32, device=dev)
    len_dev[n:] = 0
    batch.set_block(y_dev, len_dev)
    torch.cuda.synchronize()
    gen_s = time.time() - t0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        batch.run()                                   # async on torch's current stream
        local = batch.results()
        if world > 1:
            gather_forecasts(local, n * world, rank, world)   # the only exchange: forecast chunks -> rank 0
        return local

    for _ in range(args.warmup):
        step()
    barrier()
    fit_ms, dev_ms, alg_bytes, st = [], [], [], None
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
        # per-step kernel timing comes from HIP events recorded on the launch stream inside the library;
        # reading them waits for that step only (steps are sequential on one stream anyway)
        st = batch.stats()
        fit_ms.append(st["fit_kernel_ms"])
        dev_ms.append(st["total_device_ms"])
        alg_bytes.append(st["algorithmic_bytes"])
    barrier()
    elapsed = time.perf_counter() - t_start
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    res = batch.results()
    n_ok = int((res["status"] == 0).sum().item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n * world * args.steps / elapsed
        fit_ms_avg = float(np.mean(fit_ms))
        achieved = float(np.mean(alg_bytes)) / (fit_ms_avg * 1e-3) / 1e9 if fit_ms_avg > 0 else 0.0
        out = {
            "metric": f"series/sec fit+forecast, {'AutoETS' if model != 'AutoARIMA' else 'AutoARIMA'} h={h} on M5-shape batches",
            "value": round(value, 1), "unit": "series/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "model": model + (f"({ets_model})" if ets_model else ""),
                       "series_per_gpu": n, "T": T, "horizon": h, "seasonal_period": m, "positive": positive,
                       "parallelism": f"series-sharded x{world}, gather of yhat chunks to rank 0",
                       "series_ok": n_ok, "mean_passes_per_series": round(st["total_passes"] / max(n, 1), 1),
                       "max_passes_per_series": st["max_passes"], "mean_evals_per_series": round(st["total_evals"] / max(n, 1), 1),
                       "problems": st["n_problems"], "fit_kernel_launches": st["fit_kernel_launches"],
                       "device_ms_per_step": round(float(np.mean(dev_ms)), 3), "datagen_s": round(gen_s, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": ("arima_fit_kernel + arima_fit_spec_kernel (all sweeps of one step)" if model == "AutoARIMA" else
                                    "ets_round_kernel<spec,period,driver> + ets_final_kernel (all spec launches of one step, concurrent streams)"),
                         "kernel_ms": round(fit_ms_avg, 3), "algorithmic_bytes": int(np.mean(alg_bytes))},
        }
        # HBM traffic of the fit kernels comes from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE,
        # corrected as MI355X_MICROARCH.md prescribes); the committed summary is quoted when it was taken on
        # this workload and shape, else the field stays null.
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if tr.get("workload") == args.workload and n == n_def and T == T_def:
                out["roofline"]["traffic"] = int(tr["ets_round_kernel_traffic_bytes_per_step"])
                out["roofline"]["traffic_source"] = "profiles/r01_pmc_traffic.json (PMC passes, per step)"
        except (OSError, ValueError, KeyError):
            pass
        # ---- CPU baseline: the oracle ("port") on a bounded sample of the same workload -------------
        sample = cpu_def if args.cpu_sample < 0 else args.cpu_sample
        if sample > 0:
            from oracle import oracle as O
            sample = min(sample, n)
            oo = O.make_options(model, h, ets_model=ets_model, seasonal_period=m)
            vals = np.ascontiguousarray(Y[:sample]).reshape(-1)
            offs = np.arange(sample + 1, dtype=np.int64) * T
            c0 = time.perf_counter()
            cres = O.forecast_batch(vals, offs, oo, 0)
            cdt = time.perf_counter() - c0
            got = res["yhat"][:sample].cpu().numpy()
            okm = cres["status"] == 0
            rel = np.abs(got[okm] - cres["yhat"][okm]) / np.maximum(1.0, np.abs(cres["yhat"][okm]))
            out["cpu_baseline"] = {"value": round(sample / cdt, 2), "unit": "series/s", "cores": int(cres["threads"]),
                                   "kind": "port", "sample": f"first {sample} series of the same batch, oracle (C, OpenMP), {cdt:.1f} s",
                                   "max_rel_diff_vs_gpu": float(np.max(rel)) if rel.size else 0.0}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    batch.close()


if __name__ == "__main__":
    main()
