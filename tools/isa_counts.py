#!/usr/bin/env python3
"""Instruction counts of the hot loops of the round kernels, from the compiler's assembly (hipcc --save-temps).

    python tools/isa_counts.py [unit ...]      # default: fit_nonseasonal fit_seasonal_gen_a (compiles them to /tmp/isa)

For every ets_round_kernel instantiation of the unit (sequential driver, SPEC = 0) the largest loop body -- the unpredicated block of
S time steps -- is located (a backward branch whose body holds the most fp64 instructions) and its instructions are counted per
class and divided by the block's step count S: VALU fp64 / VALU other / SALU / LDS / buffer loads / waits per time step.  These are
the "executed instructions per 8-byte load" figures of DESIGN.md section 4.3 and the per-spec-class table of profiles/r03_isa_counts.txt.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "anofox-forecast_amd", "csrc")
OUT = "/tmp/isa"


def compile_unit(unit):
    os.makedirs(OUT, exist_ok=True)
    s = os.path.join(OUT, f"{unit}-hip-amdgcn-amd-amdhsa-gfx950.s")
    src = os.path.join(CSRC, unit + ".hip")
    if not os.path.exists(s) or os.path.getmtime(s) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".inc"))):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                               "-Wno-unused-function", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "--save-temps", "-c", src, "-o", os.path.join(OUT, unit + ".o")],
                              cwd=OUT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return s


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def classify(ins):
    op = ins.split()[0]
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_", )):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu_f64" if ("_f64" in op or op.startswith(("v_rcp_f64", "v_rndne_f64", "v_ldexp_f64", "v_frexp"))) else "valu_other"
    return "other"


def loops_of(body):
    """body: list of (label-or-None, instruction).  Returns [(start, end)] index ranges of backward branches."""
    labels = {}
    for i, (lab, _) in enumerate(body):
        if lab:
            labels[lab] = i
    out = []
    for i, (_, ins) in enumerate(body):
        m = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", ins or "")
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] <= i:
                out.append((labels[tgt], i))
    return out


def analyse(path, want=lambda name: True):
    txt = open(path).read().split("\n")
    funcs, cur, name = {}, None, None
    for line in txt:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, cur = m.group(1), []
            funcs[name] = cur
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^(\.L\w+):", line)
        if m:
            cur.append((m.group(1), None))
            continue
        ins = line.strip()
        if ins and not ins.startswith((";", ".")):
            cur.append((None, ins.split(";")[0].strip()))
    dm = demangle(list(funcs))
    rows = []
    for mangled, body in funcs.items():
        nm = dm.get(mangled, mangled)
        if "ets_round_kernel" not in nm or not want(nm):
            continue
        # merge labels into the following instruction
        merged, pend = [], None
        for lab, ins in body:
            if ins is None:
                pend = lab
                continue
            merged.append((pend, ins))
            pend = None
        # The unpredicated main loop of ets_pass: one iteration = one block of S steps (two blocks for the additive class).  Round 5: a damped
        # multiplicative-trend step holds a branch over its rarely needed table path, so the iteration is no longer ONE basic block --
        # count the whole loop body instead, leaving out what an s_cbranch_execz skips (the path a wave takes when no lane needs the block).
        labels = {lab: i for i, (lab, _) in enumerate(merged) if lab}

        def count(a, b):
            c, n, i = {}, 0, a
            while i <= b:
                ins = merged[i][1]
                k = classify(ins)
                c[k] = c.get(k, 0) + 1
                n += 1
                m = re.match(r"s_cbranch_execz\s+(\S+)", ins)
                if m and m.group(1) in labels and i < labels[m.group(1)] <= b:
                    i = labels[m.group(1)]
                    continue
                i += 1
            return c, n
        cand = []
        for a, b in sorted(set(loops_of(merged))):
            if any(a <= a2 and b2 <= b and (a2, b2) != (a, b) for a2, b2 in loops_of(merged)):
                continue                                    # not innermost
            c, n = count(a, b)
            cand.append((a, c, n))
        best = None
        if cand:
            top = max(c.get("valu_f64", 0) for _, c, _ in cand)
            for a, c, n in cand:                            # the first (main, unpredicated) of the loops that hold the recursion
                if c.get("valu_f64", 0) >= 0.6 * top:
                    best = (c, n)
                    break
        if best:
            rows.append((nm, best[0], best[1]))
    return rows


def steps_of(name):
    # block length S of ets_pass: 32 additive class, 8 damped multiplicative trend, 16 otherwise; rounded to the period when compile-time
    m = re.search(r"EtsCfg<(\d+), (\d+), (true|false), (\d+)>, (-?\d+), (\d+)", name)
    if not m:
        return None
    e, t, d, s, ms, spec = int(m.group(1)), int(m.group(2)), m.group(3) == "true", int(m.group(4)), int(m.group(5)), int(m.group(6))
    additive = e == 1 and t != 2 and s != 2
    target = 32 if additive else (8 if (t == 2 and d) else 16)      # ANOFOX_S_DM / ANOFOX_S_GEN of ets_device.hpp
    S = (max(target // ms, 1) * ms) if ms > 0 else target
    if additive and ms >= 0:
        S *= 2          # two blocks per iteration on alternating buffers
    return S, spec


def main():
    units = sys.argv[1:] or ["fit_nonseasonal", "fit_seasonal_gen_a"]
    comp = {1: "A", 2: "M", 0: "N"}
    print(f"{'kernel (error,trend,damped,season | period | driver)':58s} {'S':>3s} {'fp64':>6s} {'valu':>6s} {'salu':>6s} {'lds':>5s} {'vmem':>5s} {'wait':>5s} {'total':>6s}   per time step")
    for u in units:
        for nm, cnt, n in sorted(analyse(compile_unit(u))):
            st = steps_of(nm)
            if not st or st[1] != 0:
                continue
            S = st[0]
            m = re.search(r"EtsCfg<(\d+), (\d+), (true|false), (\d+)>, (-?\d+)", nm)
            label = f"E={comp[int(m.group(1))]} T={comp[int(m.group(2))]}{'d' if m.group(3) == 'true' else ''} S={comp[int(m.group(4))]} | m={m.group(5)} | seq"
            g = lambda k: cnt.get(k, 0) / S
            print(f"{label:58s} {S:3d} {g('valu_f64'):6.1f} {g('valu_other'):6.1f} {g('salu'):6.1f} {g('lds'):5.1f} {g('vmem'):5.1f} {g('wait'):5.1f} {n / S:6.1f}")


if __name__ == "__main__":
    main()
