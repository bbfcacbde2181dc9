"""Per-spec timeline of one AutoETS step from a rocprofv3 kernel trace (the rocpd .db or …_kernel_trace.csv): for every ETS spec, when its
first round kernel starts and its last one ends, relative to the step's first kernel; plus the number of kernels running
in each 25 ms window.  Usage: python tools/timeline.py <kernel_trace.csv> [step_index]"""
import csv, re, sys
from collections import defaultdict

step = int(sys.argv[2]) if len(sys.argv) > 2 else -1
if sys.argv[1].endswith(".db"):                      # rocprofv3's default rocpd database
    import sqlite3
    ks = [(int(a), int(b), n) for n, a, b in sqlite3.connect(sys.argv[1]).execute("select name, start, end from kernels")]
else:
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ks.sort()
# steps are separated by prep_kernel launches
starts = [i for i, k in enumerate(ks) if "prep_kernel" in k[2] and "arima" not in k[2]]
lo = starts[step]
hi = starts[step + 1] if step != -1 and step + 1 < len(starts) else len(ks)
sel = ks[lo:hi]
t0 = sel[0][0]
spec = defaultdict(lambda: [1e18, 0, 0, 0.0])
for s, e, name in sel:
    m = re.search(r"EtsCfg<(\d+), (\d+), (true|false), (\d+)>, (-?\d+)", name)
    if not m or "round" not in name:
        continue
    key = f"E{m.group(1)} T{m.group(2)}{'d' if m.group(3) == 'true' else ' '} S{m.group(4)} m{m.group(5)}"
    v = spec[key]
    v[0] = min(v[0], s); v[1] = max(v[1], e); v[2] += 1; v[3] += (e - s) / 1e6
print("step wall %.1f ms, %d kernels" % ((max(e for _, e, _ in sel) - t0) / 1e6, len(sel)))
for key, v in sorted(spec.items(), key=lambda kv: kv[1][1]):
    print("%-18s first start %7.1f ms   last end %7.1f ms   launches %3d   busy %7.1f ms" % (key, (v[0] - t0) / 1e6, (v[1] - t0) / 1e6, v[2], v[3]))
W = 25e6
n = int((max(e for _, e, _ in sel) - t0) / W) + 1
for w in range(n):
    a, b = t0 + w * W, t0 + (w + 1) * W
    act = [name for s, e, name in sel if s < b and e > a and "round" in name]
    md = sum(1 for x in act if re.search(r"EtsCfg<\d+, 2, true", x))
    print("%4d-%4d ms: %3d round kernels active (%d damped-M-trend)" % (w * 25, (w + 1) * 25, len(act), md))
