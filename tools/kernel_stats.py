#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace run (rocpd sqlite database): calls, total / average / max duration, start-to-end span.

    python tools/kernel_stats.py <dir-or-db> [--timeline N]     # --timeline: also the first N dispatches in start order
"""
import glob
import os
import re
import sqlite3
import sys

arg = sys.argv[1]
dbs = [arg] if arg.endswith(".db") else glob.glob(os.path.join(arg, "**", "*.db"), recursive=True)
for db in dbs:
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = con.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    if not rows:
        continue

    def short(n):
        n = re.sub(r"^void ", "", n.replace("anofox::", "").replace("(anonymous namespace)::", ""))
        return re.sub(r"\(.*$", "", n)[:110]
    agg = {}
    for n, a, b in rows:
        k = short(n)
        c = agg.setdefault(k, [0, 0, 0])
        c[0] += 1
        c[1] += b - a
        c[2] = max(c[2], b - a)
    tot = sum(c[1] for c in agg.values())
    print(f"# {db}: {len(rows)} dispatches, kernel time {tot / 1e6:.3f} ms, span {(rows[-1][2] - rows[0][1]) / 1e6:.3f} ms")
    print(f"{'kernel':112s} {'calls':>7s} {'total ms':>10s} {'avg us':>10s} {'max us':>10s} {'%':>6s}")
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:112s} {c[0]:7d} {c[1] / 1e6:10.3f} {c[1] / c[0] / 1e3:10.2f} {c[2] / 1e3:10.2f} {100.0 * c[1] / tot:6.1f}")
    if "--timeline" in sys.argv:
        n = int(sys.argv[sys.argv.index("--timeline") + 1])
        t0 = rows[-n][1] if n < len(rows) else rows[0][1]
        print("# last dispatches: start us, duration us, kernel")
        for nme, a, b in rows[-n:]:
            print(f"{(a - t0) / 1e3:12.1f} {(b - a) / 1e3:10.1f}  {short(nme)}")
