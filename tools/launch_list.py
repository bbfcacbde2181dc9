"""Every kernel launch of the LAST step of a rocprofv3 kernel trace (rocpd .db), in start order: start (ms from the step's first
kernel), duration, grid (workgroups), short name.  Usage: python tools/launch_list.py <trace.db> [substring filter] [anchor]
(anchor: the kernel whose last FIRST launch of a burst marks the step's start; default prep_kernel, "detect_period_kernel" for the
default call shape, whose step holds one prep_kernel per part)"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
q = "select name, start, end" + (f", {gx}" if gx else ", 0") + (f", {wx}" if wx else ", 1") + " from kernels order by start"
ks = list(con.execute(q))
anchor = sys.argv[3] if len(sys.argv) > 3 else "prep_kernel"
starts = [i for i, k in enumerate(ks) if anchor in k[0] and "arima" not in k[0]]
while len(starts) > 1 and ks[starts[-1]][1] - ks[starts[-2]][1] < 100e6:      # launches of one burst (within 100 ms): keep the first
    starts.pop()
sel = ks[starts[-1]:] if starts else ks
t0 = sel[0][1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for name, s, e, g, w in sel:
    if flt and flt not in name:
        continue
    m = re.search(r"(\w+)<anofox::EtsCfg<(\d+), (\d+), (true|false), (\d+)>, (-?\d+)(?:, (\d+))?(?:, (true|false))?", name)
    short = (f"{m.group(1)} E{m.group(2)}T{m.group(3)}{'d' if m.group(4) == 'true' else ''}S{m.group(5)} m{m.group(6)} drv{m.group(7)} k4={m.group(8)}" if m
             else re.sub(r"\(.*", "", name)[:60])
    print("%9.3f ms  %9.3f ms  wg %6d  %s" % ((s - t0) / 1e6, (e - s) / 1e6, (g // max(w, 1)) if g else 0, short))
print("cols:", cols)
