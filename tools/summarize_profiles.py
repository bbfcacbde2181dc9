#!/usr/bin/env python3
"""Summarise the rocprofv3 databases written by tools/profile_round.sh into the small text/CSV/JSON files kept under
profiles/ (kernel families, per-instantiation time, HBM traffic per bench step from the PMC passes)."""
import glob
import json
import os
import re
import sqlite3
import sys

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import hashlib
LIB_SHA = hashlib.sha256(open(os.path.join(ROOT, "anofox-forecast_amd", "libanofox_fcst_hip.so"), "rb").read()).hexdigest()   # bench.py quotes a summary only on this build


def family(name):
    name = name.replace("anofox::", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"[<(].*", "", name)


def inst(name):
    m = re.search(r"(ets_round_kernel|ets_final_kernel)<anofox::EtsCfg<([^>]*)>, (-?\d+)(?:, (true|false))?", name)
    if not m:
        return None
    kind = "round" if "round" in m.group(1) else "final"
    drv = {"true": "speculative", "false": "sequential", None: ""}[m.group(4)]
    return f"{kind} EtsCfg<{m.group(2)}> period {m.group(3)} {drv}".strip()


def steps_of(log):
    m = re.search(r'"steps": (\d+), "warmup": (\d+)', open(log).read())
    return (int(m.group(1)) + int(m.group(2))) if m else 1


lines = []
for trace in sorted(glob.glob(os.path.join(out, "trace_*"))):
    if not os.path.isdir(trace):
        continue
    db = glob.glob(os.path.join(trace, "*.db"))
    if not db:
        continue
    c = sqlite3.connect(db[0])
    rows = list(c.execute("select name, end - start from kernels"))
    n_steps = steps_of(trace + ".log")
    fam, ins = {}, {}
    for name, dur in rows:
        f = family(name)
        fam.setdefault(f, [0, 0])
        fam[f][0] += 1
        fam[f][1] += dur
        i = inst(name)
        if i:
            ins.setdefault(i, [0, 0])
            ins[i][0] += 1
            ins[i][1] += dur
    total = sum(v[1] for v in fam.values())
    tag = os.path.basename(trace)
    bench = [l for l in open(trace + ".log").read().splitlines() if l.startswith('{"metric"')]
    lines.append(f"== rocprofv3 --kernel-trace -- python3 bench.py ...  [{tag}]  ({n_steps} steps profiled, warm-up included) ==")
    if bench:
        b = json.loads(bench[-1])
        lines.append(f"bench line: value {b['value']} {b['unit']}, ms_per_step {b['ms_per_step']}, roofline frac {b['roofline']['frac']}, workload {b['config']['workload']}")
    lines.append(f"{'kernel family':34s} {'calls':>7s} {'total ms':>11s} {'avg us':>11s} {'share':>7s}")
    with open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as fh:
        fh.write("kernel_family,calls,total_ms,avg_us,share\n")
        for f, (n, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            if d / total < 0.0005 and n < 3:
                continue
            lines.append(f"{f[:34]:34s} {n:7d} {d / 1e6:11.2f} {d / n / 1e3:11.1f} {100 * d / total:6.1f}%")
            fh.write(f"{f},{n},{d / 1e6:.3f},{d / n / 1e3:.1f},{d / total:.4f}\n")
    if ins:
        lines.append("top ETS instantiations (EtsCfg<error,trend,damped,season>; sum of durations, kernels of different specs overlap):")
        for i, (n, d) in sorted(ins.items(), key=lambda kv: -kv[1][1])[:8]:
            lines.append(f"   {i:58s} calls {n:4d} total {d / 1e6:9.2f} ms avg {d / n / 1e3:10.1f} us")
    lines.append("")
open(os.path.join(out, "rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")

# ---- PMC traffic (FETCH_SIZE / WRITE_SIZE in KiB-ish units of 1024 B... see MI355X_MICROARCH.md: FETCH_SIZE counts 64 B
# ---- requests in units of kilobytes; on gfx950 the value is half the real traffic for this access pattern) --------------
traffic = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob(os.path.join(out, "pmc_" + cname, "*.db"))
    if not db:
        continue
    c = sqlite3.connect(db[0])
    agg = {}
    for name, val in c.execute("select kernel_name, sum(value) from counters_collection where counter_name = ? group by 1", (cname,)):
        f = family(name)
        agg[f] = agg.get(f, 0) + val * 1024.0
    traffic[cname] = agg
if traffic:
    fetch, write = traffic.get("FETCH_SIZE", {}), traffic.get("WRITE_SIZE", {})
    corrected = {k: 2.0 * fetch.get(k, 0.0) + write.get(k, 0.0) for k in set(fetch) | set(write)}
    js = {
        "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0  (separate passes per counter)",
        "workload": "autoets_m5_positive",
        "lib_sha256": LIB_SHA,
        "unit": "bytes per step (one bench step = all launches of the kernel family)",
        "calibration": "ets_final_kernel reads every series exactly once per spec; its FETCH_SIZE x 1024 is half of that algorithmic volume, "
                       "so fetch_correction = 2.0 (the gfx950 half-reporting of MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as exact",
        "FETCH_SIZE_raw_bytes": {k: int(v) for k, v in fetch.items()},
        "WRITE_SIZE_raw_bytes": {k: int(v) for k, v in write.items()},
        "hbm_bytes_corrected": {k: int(v) for k, v in corrected.items()},
        "ets_round_kernel_traffic_bytes_per_step": int(corrected.get("ets_round_kernel", 0)),
        # round 6: everything the fit phase moves -- the round and final kernels AND the kernels that only exist to feed them (the dense
        # re-gather of the running problems' columns, the compaction, the compact copy of the block); round 5 quoted the first two only
        "fit_kernel_traffic_bytes_per_step": int(sum(corrected.get(k, 0) for k in ("ets_round_kernel", "ets_final_kernel", "gather_columns_kernel",
                                                                                   "compact_kernel", "compact_block_kernel"))),
        "fit_kernel_traffic_parts": {k: int(corrected.get(k, 0)) for k in ("ets_round_kernel", "ets_final_kernel", "gather_columns_kernel",
                                                                           "compact_kernel", "compact_block_kernel")},
    }
    # ---- SQ issue counters (separate passes): wave-level VALU instructions per step and how busy the SIMDs were ----
    sq = {}
    for grp in ("pmc_SQ_INSTS", "pmc_SQ_CYCLES"):
        db = glob.glob(os.path.join(out, grp, "*.db"))
        if not db:
            continue
        c = sqlite3.connect(db[0])
        for cname, name, val in c.execute("select counter_name, kernel_name, sum(value) from counters_collection group by 1, 2"):
            sq.setdefault(cname, {})
            f = family(name)
            sq[cname][f] = sq[cname].get(f, 0) + val
    if sq:
        fit = ("ets_round_kernel", "ets_final_kernel")
        js["sq_counters_by_kernel"] = {k: {f: int(v) for f, v in sorted(d.items(), key=lambda kv: -kv[1])[:6]} for k, d in sq.items()}
        if "SQ_INSTS_VALU" in sq:
            js["valu_insts_per_step"] = int(sum(sq["SQ_INSTS_VALU"].get(f, 0) for f in fit))
            js["valu_note"] = ("SQ_INSTS_VALU of the fit kernels (wave-level instructions, one bench step); divided by the step's kernel time and by the "
                               "chip's fp64 issue rate (1,024 SIMDs x 2.4 GHz / 4 cycles per wave64 fp64 instruction) it is roofline.valu.frac")
    json.dump(js, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    # the fixed-parameter config (one streamed pass per series): traffic of ets_final_kernel and prep_kernel per step
    fx = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        db = glob.glob(os.path.join(out, "pmcfixed_" + cname, "*.db"))
        if not db:
            continue
        c = sqlite3.connect(db[0])
        for name, val, cnt in c.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by 1", (cname,)):
            f = family(name)
            fx.setdefault(f, {})[cname] = val * 1024.0
            fx[f]["launches"] = cnt
    if fx:
        rep = {}
        for f in ("ets_final_kernel", "prep_kernel"):
            if f in fx:
                n = max(fx[f].get("launches", 1), 1)
                rep[f] = {"launches": n, "hbm_bytes_per_launch_corrected": int((2.0 * fx[f].get("FETCH_SIZE", 0.0) + fx[f].get("WRITE_SIZE", 0.0)) / n),
                          "FETCH_SIZE_raw_per_launch": int(fx[f].get("FETCH_SIZE", 0.0) / n), "WRITE_SIZE_raw_per_launch": int(fx[f].get("WRITE_SIZE", 0.0) / n)}
        json.dump({"workload": "ets_aaa_fixed_m5", "lib_sha256": LIB_SHA,
                   "fit_kernel_traffic_bytes_per_step": int(sum(r["hbm_bytes_per_launch_corrected"] for r in rep.values())),
                   "algorithmic_bytes_per_launch_ets_final_kernel": 30490 * (8 * 1913 + 8 * 28),
                   "correction": "fetch x 2 (gfx950 half-reporting, calibrated as in pmc_traffic.json)", "kernels": rep},
                  open(os.path.join(out, "pmc_traffic_fixed.json"), "w"), indent=1)
    for cname, agg in traffic.items():
        with open(os.path.join(out, f"pmc_{cname}_by_kernel.csv"), "w") as fh:
            fh.write("kernel_family,raw_bytes\n")
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
                fh.write(f"{k},{int(v)}\n")
# ---- AutoARIMA: HBM traffic and issue counters per kernel (pmcarima_* passes of the autoarima_css_m5 workload) ----
ar = {}
for d in sorted(glob.glob(os.path.join(out, "pmcarima_*"))):
    if not os.path.isdir(d):
        continue
    db = glob.glob(os.path.join(d, "*.db"))
    if not db:
        continue
    c = sqlite3.connect(db[0])
    for cname, name, val in c.execute("select counter_name, kernel_name, sum(value) from counters_collection group by 1, 2"):
        ar.setdefault(family(name), {})
        ar[family(name)][cname] = ar[family(name)].get(cname, 0) + val
if ar:
    rep = {}
    for f, d in ar.items():
        if not f.startswith("arima_"):
            continue
        e = {k: int(v) for k, v in d.items()}
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            e["hbm_bytes_corrected"] = int(2.0 * d.get("FETCH_SIZE", 0.0) * 1024.0 + d.get("WRITE_SIZE", 0.0) * 1024.0)
        rep[f] = e
    blog = [l for l in open(os.path.join(out, "pmcarima_FETCH_SIZE.log")).read().splitlines() if l.startswith('{"metric"')] if os.path.exists(os.path.join(out, "pmcarima_FETCH_SIZE.log")) else []
    alg = json.loads(blog[-1])["roofline"]["algorithmic_bytes"] if blog else None
    fit_bytes = sum(v.get("hbm_bytes_corrected", 0) for k, v in rep.items() if k.startswith(("arima_fit_kernel", "arima_fit_spec_kernel")))
    json.dump({"workload": "autoarima_css_m5", "lib_sha256": LIB_SHA,
               "command": "rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --workload autoarima_css_m5 --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 (one pass per counter group)",
               "correction": "FETCH_SIZE x 1024 x 2 (gfx950 half-reporting, calibrated on ets_final_kernel) + WRITE_SIZE x 1024",
               "fit_kernel_traffic_bytes_per_step": int(fit_bytes), "algorithmic_bytes_per_step": alg,
               "traffic_over_algorithmic": (round(fit_bytes / alg, 3) if alg else None),
               "note": "the fit lanes of a wave stream unrelated series: every lane reads its OWN row of the series-major block W with 128-bit loads "
                       "(two steps per load), 8 T' bytes per evaluation and lane",
               "kernels": rep}, open(os.path.join(out, "pmc_traffic_arima.json"), "w"), indent=1)
print(open(os.path.join(out, "rocprof_summary.txt")).read())
