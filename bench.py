#!/usr/bin/env python3
"""bench.py -- series/sec fit+forecast, AutoETS h=28 on M5-shape batches (BASELINE.json metric).

A "step" is one full pass of the hot path (prep -> every candidate ETS spec fitted per series by
per-lane Nelder-Mead -> AICc selection -> fallback chain -> intervals) over one synthetic batch that
is already resident in HBM when the timed region starts.  One process per GPU; series-id ranges
shard across ranks with no data-path collective and the only exchange is the gather of the forecast
chunks to rank 0, inside the timed region.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

--scaling strong (default) the workload's series are split over the ranks (BASELINE config 3: "M5 30,490 series,
                 series-sharded 1->8 GPUs"): rank r owns dist.shard_range(n, r, N); value = series_total / max-rank time
--scaling weak   every rank owns a full batch of the workload's size: value = N x series / time
(at N = 1 the two are the same run).  The 1M x 1,024 configuration (BASELINE config 5) is
`--workload autoets_stress --n-series 1000000` (strong: 125,000 series per GPU at N = 8).

Workloads (--workload):
    autoets_m5_positive   (default) 30,490 x 1,913, h=28, m=7, strictly positive counts: all 25
                          valid specs of the 30-model grid are fitted for every series
    autoets_m5            same shape, raw intermittent counts (zeros -> only the 6 additive specs admissible)
    ets_aaa_fixed_m5      BASELINE config 2: ETS(A,A,A) with GIVEN smoothing parameters (alpha 0.2, beta 0.05,
                          gamma 0.1): one streamed pass per series, 15,976 algorithmic bytes per series
    ets_aaa_m5            ETS(A,A,A) fitted (Nelder-Mead over alpha, beta*, gamma*) on the same batch
    autoets_stress        n x 1,024 AutoETS (use --n-series; the 1M config shards over 8 GPUs)
    autoarima_m5          AutoARIMA stepwise search, m = 7, on the M5-shape batch, selected model refitted on the exact Gaussian
                          likelihood (BASELINE config 4: "(Kalman kernel)"; estimation method ANOFOX_ARIMA_CSS_ML)
    autoarima_css_m5      the same with the library's default method (ANOFOX_ARIMA_CSS: the search's estimates, the reference's cost envelope)

The JSON line also carries, on rank 0 at N = 1: `e2e` (the same workload through anofox_ts_forecast_batch: host
buffers in, results on host -- pack + H2D + fit + D2H, SURVEY.md 8(d) metric (ii); never `value`) and
`cpu_baseline` (the oracle on a bounded sample of the same batch, on this box's host cores).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before HIP initialises: the spec streams need hardware queues

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 4.0   # wave-level fp64 VALU issue slots per second: 1,024 SIMDs, one wave64 fp64 op per 4 cycles at 2.4 GHz


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="autoets_m5_positive")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong")
    ap.add_argument("--n-series", type=int, default=0, help="series per GPU (weak) / in total (strong); 0 = workload default")
    ap.add_argument("--t", type=int, default=0, help="observations per series (0 = workload default)")
    ap.add_argument("--horizon", type=int, default=28)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="series for the CPU baseline (-1 auto, 0 skip)")
    ap.add_argument("--e2e-steps", type=int, default=-1, help="timed passes of the host-buffer entry (-1 auto, 0 skip)")
    ap.add_argument("--ets-model", default="", help="single-spec probes: override the workload's ETS spec (e.g. AMdA; implies model ETS on strictly positive series)")
    ap.add_argument("--also", type=int, default=-1,
                    help="timed steps of each of the other BASELINE configurations, reported under \"also\" (-1 auto: 3 on the plain default "
                         "single-GPU invocation -- no --workload / --n-series / --cpu-sample / --e2e-steps --, 0 otherwise; 0 skip)")
    ap.add_argument("--simulate-world", type=int, default=0,
                    help="ONE process, ONE GPU: run rank 0's share of an N-GPU job (strong: shard 0 of N; weak: one full batch) and report the "
                         "per-GPU figure an N-GPU run would start from -- what can be known about 8 GPUs on a 1-GPU box (no collective runs)")
    return ap.parse_args()


CPU_TARGET_S = 12.0      # CPU work of the baseline leg (the contract asks for 10-30 s)


def W(model, ets_model, n, T, m, positive, seed, cpu_sample, fixed=None, arima_method=0):
    return dict(model=model, ets_model=ets_model, n=n, T=T, m=m, positive=positive, seed=seed, cpu_sample=cpu_sample, fixed=fixed,
                arima_method=arima_method)


WORKLOADS = {
    "autoets_m5_positive": W("AutoETS", "", 30490, 1913, 7, True, 20260101, 384),
    "autoets_m5": W("AutoETS", "", 30490, 1913, 7, False, 20260101, 8192),
    "ets_aaa_fixed_m5": W("ETS", "AAA", 30490, 1913, 7, False, 20260101, 30490, fixed=(0.2, 0.05, 0.1, 1.0)),
    "ets_aaa_m5": W("ETS", "AAA", 30490, 1913, 7, False, 20260101, 8192),
    "autoets_stress": W("AutoETS", "", 125000, 1024, 7, False, 20260102, 8192),
    "autoarima_m5": W("AutoARIMA", "", 30490, 1913, 7, False, 20260101, 2048, arima_method=1),
    "autoarima_css_m5": W("AutoARIMA", "", 30490, 1913, 7, False, 20260101, 2048, arima_method=0),
    # long seasonal periods (run-time-period kernels: seasonal ring in LDS up to m = 64, in an HBM scratch above)
    "autoets_m24": W("AutoETS", "", 30490, 1008, 24, True, 20260103, 256),
    "autoets_hourly168": W("AutoETS", "", 30490, 1008, 168, True, 20260103, 128),
    # single-spec probes (kernel efficiency without cross-kernel effects)
    "ets_amdn_stress": W("ETS", "AMdN", 125000, 1024, 7, True, 20260102, 256),
    "ets_mam_stress": W("ETS", "MAM", 125000, 1024, 7, True, 20260102, 256),
}


def e2e_pass(lib, Y, opts, steps):
    """The same batch through the host-buffer entry (anofox_ts_forecast_batch): pack + H2D + fit + D2H + result arrays."""
    L = lib.load()
    n, T = Y.shape
    vptr = (C.c_void_p * n)(*[Y.ctypes.data + s * T * 8 for s in range(n)])
    lens = (C.c_size_t * n)(*([T] * n))
    times = []
    for _ in range(steps):
        results = (lib.ForecastResult * n)()
        errors = (lib.AnofoxError * n)()
        berr = lib.AnofoxError()
        t0 = time.perf_counter()
        ok = L.anofox_ts_forecast_batch(vptr, None, lens, n, C.byref(opts), None, results, errors, C.byref(berr))
        times.append(time.perf_counter() - t0)
        if not ok:
            raise RuntimeError(f"anofox_ts_forecast_batch failed: [{berr.code}] {berr.message.decode()}")
        n_ok = sum(1 for i in range(n) if errors[i].code == 0)
        for i in range(n):
            L.anofox_free_forecast_result(C.byref(results[i]))
    return times, n_ok


ALSO_WORKLOADS = ("autoets_m5", "autoarima_m5", "autoarima_css_m5", "ets_aaa_fixed_m5", "autoets_stress")


def also_pass(names, steps, h, dev, lib, synth, DeviceBatch, pack_time_major, torch):
    """Short timed loops of the other BASELINE configurations (configs 2, 4, 5 and the raw-M5 variant of config 3), so the line the
    driver keeps holds every configuration and not only the headline one.  Same shape of measurement as the headline: batch resident
    in HBM, one untimed step, `steps` timed steps between two synchronisations, the library's own HIP events for the kernel time.
    The reference's harness times one statement per model the same way (benchmark/src/common/anofox_runner.py:135-149)."""
    out, cache = {}, {}
    for name in names:
        wl = WORKLOADS[name]
        n, T, m = wl["n"], wl["T"], wl["m"]
        key = (wl["seed"], n, T, m, wl["positive"])
        if key not in cache:
            cache.clear()                                   # one host copy at a time (the 125k x 1,024 batch is 1 GB)
            cache[key] = synth.gen_series(wl["seed"], 0, n, T, m, wl["positive"])
        Y = cache[key]
        opts = lib.make_options(wl["model"], h, ets_model=wl["ets_model"], seasonal_period=m)
        batch = DeviceBatch(n, T, opts, dev)
        if wl["fixed"]:
            batch.set_fixed_params(*wl["fixed"])
        if wl["model"] == "AutoARIMA":
            batch.set_arima_method(wl["arima_method"])
        y_dev = torch.from_numpy(pack_time_major(Y, batch.ld)).to(dev)
        len_dev = torch.full((batch.ld,), T, dtype=torch.int32, device=dev)
        len_dev[n:] = 0
        batch.set_block(y_dev, len_dev)
        batch.run()
        torch.cuda.synchronize()
        k = steps * (200 if wl["fixed"] else 1)             # the one-pass step is a third of a millisecond: enqueue enough of them to time
        fit_ms, dev_ms = [], []
        t0 = time.perf_counter()
        for _ in range(k):
            batch.run()                                     # (nothing but the step in the timed region: see main)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        st = batch.stats()                                  # events and counters of the last step
        fit_ms.append(st["total_device_ms"] if wl["fixed"] else st["fit_kernel_ms"])      # (fixed: the unit is the whole one-pass step)
        dev_ms.append(st["total_device_ms"])
        kms = float(np.mean(fit_ms))
        ub = {0: 8, 1: 4, 2: 2}.get(int(st.get("y_storage", 0)), 8)      # bytes per observation of what the run streamed (see main)
        ob = 24.0 * max(h, 0) * n
        ach = ((st["algorithmic_bytes"] - ob) * ub / 8.0 + ob) / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        minb = int((int(st.get("min_pass_bytes", 0)) - ob) * ub / 8.0 + ob) if st.get("min_pass_bytes", 0) else 0
        res = batch.results()
        out[name] = {"value": round(n / dt, 1), "unit": "series/s", "ms_per_step": round(dt * 1e3, 4), "steps": k,
                     "series": n, "T": T, "series_ok": int((res["status"] == 0).sum().item()),
                     "roofline": {"frac": round(ach / HBM_PEAK_GBS, 4), "achieved": round(ach, 1), "kernel_ms": round(kms, 4), "bytes_per_observation": ub,
                                  "frac_min_passes": round(minb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (minb and kms > 0) else None}}
        if wl["model"] == "AutoARIMA":
            out[name]["arima_method"] = "CSS-ML (exact Gaussian likelihood refit: the Kalman kernel BASELINE config 4 names)" if wl["arima_method"] else "CSS (library default)"
        batch.close()
        del y_dev, len_dev, batch
    return out


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks ourselves.

    The driver times the N = 1 line as `python bench.py --gpus 1 ...`; invoked the same way with --gpus 8 this file used to run
    ONE rank and print "n_gpus": 1.  Now the un-wrapped invocation becomes the documented one -- `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` -- as a CHILD process (never an
    exec: this process must not be replaced once anything could have touched the GPU; here nothing has -- torch is not imported yet).
    The child's stdout / stderr are ours (inherited), and its exit code is returned."""
    import subprocess
    # --standalone: the launcher picks its own rendezvous port (a port chosen here by bind / close can be taken before the child binds it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist

    from anofox_forecast_amd import lib, synth
    from anofox_forecast_amd.device import DeviceBatch, pack_time_major
    from anofox_forecast_amd.dist import gather_forecasts, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP backend has no CPU fallback")
    # ANOFOX_BENCH_ONE_GPU=1 (debug / test only): every rank uses GPU 0 and the gather runs over gloo -- exercises the
    # N > 1 code path on a single-GPU box; RCCL refuses two ranks on one device
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

        if world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks; reporting the ranks that exist", file=sys.stderr)
    # what the ranks of this job actually are (rank -> device), stated in config.parallelism so an "8-GPU" line can be checked
    backend = (dist.get_backend() if world > 1 else "none")
    seen = [(rank, dev_index)]
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, (rank, dev_index))
    ranks_seen = f"{'RCCL (nccl)' if backend == 'nccl' else backend} ranks seen: {len(seen)} on devices {sorted(set(d for _, d in seen))}"

    wl = dict(WORKLOADS[args.workload])
    if args.ets_model:
        wl.update(model="ETS", ets_model=args.ets_model, positive=True, fixed=None)
    model, ets_model, m, positive, seed = wl["model"], wl["ets_model"], wl["m"], wl["positive"], wl["seed"]
    n_arg = args.n_series or wl["n"]
    T = args.t or wl["T"]
    h = args.horizon
    sim = args.simulate_world if (args.simulate_world > 1 and world == 1) else 0
    if args.scaling == "strong":
        n_total = n_arg
        lo, hi = shard_range(n_total, rank, sim or world)        # this rank's series-id range of the ONE batch
    else:
        n_total = n_arg * world
        lo, hi = rank * n_arg, (rank + 1) * n_arg          # every rank a full batch of its own
    n = hi - lo

    # ---- synthetic batch, resident in HBM before anything is timed ------------------------------
    t0 = time.time()
    Y = synth.gen_series(seed, lo, n, T, m, positive)                       # this rank's series-id range
    opts = lib.make_options(model, h, ets_model=ets_model, seasonal_period=m)
    batch = DeviceBatch(n, T, opts, dev)
    if wl["fixed"]:
        batch.set_fixed_params(*wl["fixed"])
    if model == "AutoARIMA":
        batch.set_arima_method(wl["arima_method"])
        lib.load().anofox_hip_set_default_arima_method(wl["arima_method"])      # the host-buffer entry of the e2e leg
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
        if world > 1:
            gather_forecasts(batch.results(), n_total, rank, world)     # the only exchange: forecast chunks -> rank 0

    for _ in range(args.warmup):
        step()
    barrier()
    fit_ms, dev_ms, alg_bytes, st = [], [], [], None
    # Nothing but the step is inside the timed region: K runs enqueued one after the other on the batch's stream (a run's own round
    # loop waits for its device counters, so the host never runs ahead by more than a step's closing kernels).  Rounds 1-6 read the
    # library's statistics after every step -- a wait plus device-to-host copies of per-problem counters that grow with the batch
    # (30,490 series: 1.5-3 ms per step; 125,000: 24 ms of an 80 ms step; 1,000,000: 79 ms of 423: `profiles/r06_bench_timing_overhead.txt`)
    # -- instrumentation of this file, not work of the path.  Now the device time of the step, the kernel time of the fit phase and
    # the pass counters all come from the library's own HIP events (recorded on the stream the run is launched on -- the batch's own
    # stream when torch's current stream is the null stream, so torch events would not see it) and counters of the LAST timed step,
    # read once after the closing barrier: every step is the same computation on the same block, the counters are equal step to
    # step and the times within a percent.
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t_start
    st = batch.stats()
    fit_ms.append(st["fit_kernel_ms"])
    dev_ms.append(st["total_device_ms"])
    dev_ms_last = float(st["total_device_ms"])
    alg_bytes.append(st["algorithmic_bytes"])
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        if one_gpu:
            tcpu = tmax.cpu()
            dist.all_reduce(tcpu, op=dist.ReduceOp.MAX)
            tmax = tcpu
        else:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    res = batch.results()
    n_ok = int((res["status"] == 0).sum().item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = (n if sim else n_total) * args.steps / elapsed          # (simulated world: only this rank's series were processed)
        fit_ms_avg = float(np.mean(fit_ms))
        final_pass_ms = None
        if wl["fixed"]:
            # config 2: the timed unit is the STEP (HIP events around everything the run enqueues), not the final pass alone
            final_pass_ms, fit_ms_avg = fit_ms_avg, dev_ms_last
        # ALGORITHMIC bytes = what must move: one pass over one series reads T observations IN THE TYPE THE RUN STREAMS -- 8 bytes each
        # from the fp64 block, 4 / 2 from the float / uint16 copy a batch of counts is streamed from (SURVEY.md 8(d) wrote the unit
        # for fp64 storage; with the compact copy that unit would put `achieved` ABOVE the measured traffic and, on the all-additive
        # batches, above the 8 TB/s peak).  The library's counter is in the fp64 unit: scale its pass part, keep the 24 h of results.
        unit_bytes = {0: 8, 1: 4, 2: 2}.get(int(st.get("y_storage", 0)), 8)
        out_bytes = 24.0 * max(h, 0) * n

        def streamed(b):
            return (float(b) - out_bytes) * unit_bytes / 8.0 + out_bytes if b else 0.0
        alg_fp64_unit = float(np.mean(alg_bytes))
        achieved_fp64_unit = alg_fp64_unit / (fit_ms_avg * 1e-3) / 1e9 if fit_ms_avg > 0 else 0.0
        alg_bytes = [streamed(b) for b in alg_bytes]
        achieved = float(np.mean(alg_bytes)) / (fit_ms_avg * 1e-3) / 1e9 if fit_ms_avg > 0 else 0.0
        # the same time against the bytes of a one-pass-per-iteration schedule (Nelder-Mead iterations + one final pass per problem):
        # `frac` counts the passes the drivers actually executed (the sequential one runs ~1.7 per iteration), this one does not
        # depend on which driver ran which round
        min_bytes = int(streamed(int(st.get("min_pass_bytes", 0))))
        achieved_min = min_bytes / (fit_ms_avg * 1e-3) / 1e9 if (fit_ms_avg > 0 and min_bytes > 0) else None
        if model == "AutoARIMA":
            kernel = ("arima_fit_kernel + arima_fit_spec_kernel (all sweeps of one step + the selected models' polish)" +
                      (" + arima_refit_kernel (exact-likelihood refit)" if wl["arima_method"] else ""))
        elif wl["fixed"]:
            # the whole step is the unit here: prep_kernel (means, decomposition, initial states: one streamed sweep) dominates it,
            # the one-pass ets_final_kernel is reported beside it
            kernel = "prep_kernel + ets_fixed_setup_kernel + ets_final_kernel + interval_kernel (the whole one-pass step; prep_kernel dominant)"
        else:
            kernel = "ets_round_kernel<spec,period,driver> + ets_final_kernel (all spec launches of one step, concurrent streams)"
        out = {
            "metric": f"series/sec fit+forecast, {'AutoETS' if model != 'AutoARIMA' else 'AutoARIMA'} h={h} on M5-shape batches",
            "value": round(value, 1), "unit": "series/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            # what the results are checked against: the oracle restates the published algorithms; the reference's known answers are met to six
            # decimals (SES / Holt / HoltWinters family, baselines) or within 1e-5 relative (AutoETS, AutoARIMA: 18.0145125 against 18.014537,
            # not SQL-equal; its box / root threshold / budget were selected on that single series -- DESIGN.md section 3)
            "parity": ("bit-identical to oracle/ (restatement within 1e-5 relative of the reference's one AutoARIMA known answer, not SQL-equal; its three constants were selected on that single series; nothing with a seasonal period is pinned in the reference tree)" if model == "AutoARIMA"
                       else "bit-identical to oracle/ (restatement pinned on the reference's known answers; crate-internal arithmetic for m = 7 unpinned)"),
            "config": {"workload": args.workload, "model": model + (f"({ets_model})" if ets_model else ""),
                       "fixed_params": list(wl["fixed"]) if wl["fixed"] else None,
                       "series_total": n_total, "series_per_gpu": n, "T": T, "horizon": h, "seasonal_period": m, "positive": positive,
                       "parallelism": f"series-sharded x{world} ({args.scaling}), gather of yhat chunks to rank 0; {ranks_seen}",
                       "simulated_world": sim or None,
                       "series_ok": n_ok, "mean_passes_per_series": round(st["total_passes"] / max(n, 1), 1),
                       "max_passes_per_series": st["max_passes"], "mean_evals_per_series": round(st["total_evals"] / max(n, 1), 1),
                       "problems": st["n_problems"], "fit_kernel_launches": st["fit_kernel_launches"],
                       # what the ETS fit streamed: the fp64 block, or a float / uint16 copy of it -- made only when EVERY observation of the batch
                       # survives that type exactly (counts: this workload), so the fp64 arithmetic sees the same numbers (bit-identical results;
                       # the roofline below counts the bytes of THIS type per observation and pass; `fp64_unit` keeps the 8-byte figure of earlier rounds)
                       "storage": {0: "f64 block", 1: "f32 copy of the block (exact for this batch)", 2: "u16 copy of the block (exact for this batch)"}.get(int(st.get("y_storage", 0)), "f64 block"),
                       "device_ms_per_step": round(dev_ms_last, 3), "datagen_s": round(gen_s, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": kernel, "kernel_ms": round(fit_ms_avg, 3), "kernel_ms_of": "the library's HIP events of the last timed step (launch stream)", "algorithmic_bytes": int(np.mean(alg_bytes)),
                         "frac_min_passes": round(achieved_min / HBM_PEAK_GBS, 4) if achieved_min else None,
                         "min_pass_bytes": min_bytes or None,
                         "mean_iterations_per_series": round(st.get("total_iters", 0) / max(n, 1), 1) if min_bytes else None,
                         "bytes_per_observation": unit_bytes,
                         # the same passes in SURVEY.md's fp64 unit (8 bytes per observation: what rounds 1-5 reported, and what the fp64 block
                         # would have had to move) -- for comparison across rounds only; with a compact copy it exceeds what was moved
                         "fp64_unit": {"achieved": round(achieved_fp64_unit, 1), "frac": round(achieved_fp64_unit / HBM_PEAK_GBS, 4),
                                       "algorithmic_bytes": int(alg_fp64_unit)}},
        }
        if model != "AutoARIMA" and not wl["fixed"]:
            # lane-level efficiency of the round kernels (device counters of the LAST timed step): the share of issued lanes that
            # evaluated a trial point of a running problem -- `valu.frac` counts a wave with three live lanes as fully useful, this does not
            try:
                ls = batch.lane_stats()
                out["roofline"]["lanes"] = {"lane_efficiency": ls["lane_efficiency"], "wave_passes": ls["wave_passes"],
                                            "by_class": {k: v["lane_efficiency"] for k, v in ls["by_class"].items()},
                                            "wave_passes_by_class": {k: v["wave_passes"] for k, v in ls["by_class"].items()},
                                            "by_spec": {k: v["lane_efficiency"] for k, v in ls["by_spec"].items()},
                                            "what": "live lane-passes / (64 x wave passes) of ets_round_kernel, all rounds of one step"}
            except (RuntimeError, AttributeError) as e:
                out["roofline"]["lanes"] = {"error": str(e)}
        if final_pass_ms:
            out["roofline"]["final_pass"] = {"kernel": "ets_final_kernel<spec,period> alone", "kernel_ms": round(final_pass_ms, 4),
                                             "achieved": round(float(np.mean(alg_bytes)) / (final_pass_ms * 1e-3) / 1e9, 1),
                                             "frac": round(float(np.mean(alg_bytes)) / (final_pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # HBM traffic and VALU issue of the fit kernels come from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE corrected
        # as MI355X_MICROARCH.md prescribes; SQ_INSTS_VALU): tools/profile_round.sh writes profiles/rNN_pmc_traffic*.json STAMPED with
        # the sha256 of the library they were taken on.  The figures are quoted only when the stamp is the library that just ran
        # and the workload / shape agree -- anything else leaves the fields null and says why.
        import glob
        import hashlib
        lib_sha = hashlib.sha256(open(lib.LIB_PATH, "rb").read()).hexdigest()
        out["roofline"]["lib_sha256"] = lib_sha[:16]
        stale = None
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
            try:
                tr = json.load(open(path))
            except (OSError, ValueError):
                continue
            if not (tr.get("workload") == args.workload and n == wl["n"] and T == wl["T"]):
                continue
            fn = os.path.basename(path)
            if tr.get("lib_sha256", "")[:16] != lib_sha[:16]:
                stale = stale or f"profiles/{fn} was taken on another build of the library ({tr.get('lib_sha256', 'unstamped')[:16]}): not quoted"
                continue
            try:
                out["roofline"]["traffic"] = int(tr["fit_kernel_traffic_bytes_per_step"])
                out["roofline"]["traffic_source"] = f"profiles/{fn} (PMC passes on this build, per step)"
                if "valu_insts_per_step" in tr:
                    issued = float(tr["valu_insts_per_step"])
                    rate = issued / (fit_ms_avg * 1e-3)
                    out["roofline"]["valu"] = {"issued_insts": int(issued), "peak_issue": FP64_VALU_PEAK_WAVE_INSTS,
                                               "frac": round(rate / FP64_VALU_PEAK_WAVE_INSTS, 4),
                                               "unit": "wave-level VALU instructions per second against one fp64 issue per SIMD per 4 cycles",
                                               "source": f"profiles/{fn} (SQ_INSTS_VALU per step, this build) / this run's kernel_ms"}
            except (KeyError, ValueError):
                pass
            break
        if out["roofline"]["traffic"] is None and stale:
            out["roofline"]["traffic_source"] = stale
        if sim:
            # rank 0's shard alone: `value` is what THIS GPU did; the N-GPU job cannot be faster than its slowest rank, so the
            # projection is N x value at best (shards are equal sized up to one series; the gather adds < 1 ms)
            out["projection"] = {"world": sim, "scaling": args.scaling, "per_gpu_series": n, "per_gpu_ms_per_step": round(ms_per_step, 3),
                                 "job_series_per_s_upper_bound": round(sim * value, 1),
                                 "note": "one rank of the job measured alone on one GPU; no RCCL gather in the timed region"}
        yhat_host = res["yhat"].cpu().numpy().copy()
        batch.close()       # the host-buffer entry below is what a binding calls on its own: no second resident batch (and its streams) beside it
        del res, y_dev, len_dev       # ... and the headline block goes back to torch's allocator before the other workloads allocate theirs
        torch.cuda.empty_cache()
        # ---- end to end: host buffers in, results on host (never `value`) -----------------------------
        e2e_steps = args.e2e_steps if args.e2e_steps >= 0 else (0 if world > 1 else (1 if ms_per_step > 400 else 2))
        if wl["fixed"]:
            out["e2e"] = None        # the fixed-parameter entry exists on the device-resident API only
        elif e2e_steps > 0 and world == 1:
            times, e_ok = e2e_pass(lib, Y, opts, e2e_steps + 1)          # first pass untimed (allocations)
            if True:
                best = float(np.mean(times[1:]))
                out["e2e"] = {"value": round(n / best, 1), "unit": "series/s", "ms_per_step": round(best * 1e3, 3), "steps": e2e_steps,
                              "series_ok": e_ok,
                              "what": "anofox_ts_forecast_batch: NULL fill + pack to the pinned time-major block (all host threads) + H2D + "
                                      "fit + D2H + per-series malloc'd result arrays; PCIe-inclusive, never the headline value"}
        # ---- the other BASELINE configurations, short loops (never `value`) ----------------------------
        # (auto: only the PLAIN default invocation -- the driver's line; a run that switches the CPU or host-buffer leg off is a
        #  measurement script, whose kernel trace or counters must hold the headline workload alone)
        also_steps = args.also if args.also >= 0 else (3 if (world == 1 and not sim and args.workload == "autoets_m5_positive"
                                                             and not args.n_series and not args.t and not args.ets_model
                                                             and args.cpu_sample < 0 and args.e2e_steps < 0) else 0)
        if also_steps > 0 and world == 1:
            out["also"] = also_pass(ALSO_WORKLOADS, also_steps, h, dev, lib, synth, DeviceBatch, pack_time_major, torch)
        # ---- CPU baseline: the oracle ("port") on a bounded sample of the same workload -------------
        sample = wl["cpu_sample"] if args.cpu_sample < 0 else args.cpu_sample
        if sample > 0 and world == 1:                  # (the contract: rank 0 at N = 1 only -- an N-GPU line carries no CPU leg)
            from oracle import oracle as O

            def cpu_run(k):
                vals = np.ascontiguousarray(Y[:k]).reshape(-1)
                offs = np.arange(k + 1, dtype=np.int64) * T
                t0 = time.perf_counter()
                if wl["fixed"]:
                    r = O.ets_fixed_batch(vals, offs, ets_model, m, *wl["fixed"], h)
                else:
                    C.c_int.in_dll(O.lib(), "oracle_arima_ml_refit").value = wl["arima_method"]     # the checker runs the same estimation method
                    r = O.forecast_batch(vals, offs, O.make_options(model, h, ets_model=ets_model, seasonal_period=m), 0)
                return r, time.perf_counter() - t0

            # bounded sample: a pilot (the table's figure, ~1 s on 128 cores) sizes the timed run to ~CPU_TARGET_S of work on
            # THIS host -- more series first, whole repetitions of the batch once it is exhausted
            sample = min(sample, n)
            reps = 1
            if args.cpu_sample < 0:
                _, pdt = cpu_run(sample)
                want = CPU_TARGET_S / max(pdt, 1e-3) * sample
                if want > n:
                    sample, reps = n, int(min(64, max(1, round(want / n))))
                else:
                    sample = int(max(sample, want))
            cdt = 0.0
            for _ in range(reps):
                cres, dt1 = cpu_run(sample)
                cdt += dt1
            cdt /= reps
            got = yhat_host[:sample]
            okm = cres["status"] == 0
            rel = np.abs(got[okm] - cres["yhat"][okm]) / np.maximum(1.0, np.abs(cres["yhat"][okm]))
            out["cpu_baseline"] = {"value": round(sample / cdt, 2), "unit": "series/s", "cores": int(cres["threads"]),
                                   "kind": "port", "sample": f"first {sample} series of the same batch, oracle (C, OpenMP), {reps} x {cdt:.1f} s",
                                   "max_rel_diff_vs_gpu": float(np.max(rel)) if rel.size else 0.0}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    batch.close()


if __name__ == "__main__":
    main()
