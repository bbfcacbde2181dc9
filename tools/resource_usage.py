#!/usr/bin/env python3
"""Registers, scratch (spills), LDS and residency of every kernel of a compile unit, from hipcc's own remarks.

    python tools/resource_usage.py [unit ...] [--filter substring] [-D...]      # default units: the four ETS fit units

Compiles the unit to /tmp/ru with -Rpass-analysis=kernel-resource-usage and prints one line per kernel (demangled).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "anofox-forecast_amd", "csrc")
OUT = "/tmp/ru"


def usage(unit, defs):
    os.makedirs(OUT, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
           "-Wno-unused-function", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage",
           "-c", os.path.join(CSRC, unit + ".hip"), "-o", os.path.join(OUT, unit + ".o")] + defs
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    rows = []
    for b in blocks:
        name = b.split("\n")[0].split(" [-R")[0].strip()

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((name, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    return [(d.replace("anofox::", "").replace("void ", ""),) + r[1:] for r, d in zip(rows, dem)]


def main():
    args = sys.argv[1:]
    flt = None
    defs = [a for a in args if a.startswith("-D")]
    args = [a for a in args if not a.startswith("-D")]
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    units = args or ["fit_nonseasonal", "fit_seasonal_add", "fit_seasonal_gen_a", "fit_seasonal_gen_m"]
    for u in units:
        for name, vgpr, agpr, scratch, occ, lds in usage(u, defs):
            if flt and flt not in name:
                continue
            print(f"{name[:110]:110s} vgpr {vgpr:4d} agpr {agpr:3d} scratch {scratch:4d} waves/SIMD {occ} static-lds {lds}")


if __name__ == "__main__":
    main()
