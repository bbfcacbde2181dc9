"""Who is resident when: the per-wave records of the ETS round kernels (developer instrument, FitArgs::wave_trace; written by
anofox_hip_batch_lane_stats when ANOFOX_HIP_TUNE holds wave_trace=<file>).

    ANOFOX_HIP_TUNE="wave_trace=/root/repo/gpurun_out/x/wt.bin" python bench.py --steps 1 --warmup 1 --cpu-sample 0 --e2e-steps 0 --also 0
    python tools/wave_trace.py gpurun_out/x/wt.bin [bin_ms]

Prints (1) resident waves per time bin by spec class (capacity: 256 CUs x 4 SIMDs x 2 waves of these kernels = 2,048), (2) per spec and
round: driver, waves, first start, last end, the waves' median / max duration and how long after the launch's first wave the median /
last wave started (the wait for a SIMD with room).  Clock: s_memrealtime, 100 MHz."""
import sys

import numpy as np

TICK_MS = 1.0e-5          # 100 MHz


def spec_name(sid):
    # spec id layout of host_semantics.hpp: error (0 A, 1 M) * 15 + trend index (N, A, Ad, M, Md) * 3 + season (N, A, M)
    e, rest = divmod(int(sid), 15)
    t, s = divmod(rest, 3)
    return "ETS(%s,%s,%s)" % ("AM"[e], ["N", "A", "Ad", "M", "Md"][t], "NAM"[s])


def spec_class(sid):
    e, rest = divmod(int(sid), 15)
    t, s = divmod(rest, 3)
    if t == 4:
        return "damped-M"
    return "additive" if (e == 0 and t < 3 and s < 2) else "general"


def main():
    raw = np.fromfile(sys.argv[1], dtype=np.uint64)
    used, cap = int(raw[0]), int(raw[1])
    n = min(used, cap)
    rec = raw[4:4 + 4 * n].reshape(n, 4)
    tag, t0, t1, hw = rec[:, 0], rec[:, 1].astype(np.int64), rec[:, 2].astype(np.int64), rec[:, 3]
    sid = (tag >> np.uint64(32)).astype(np.int64)
    rnd = ((tag >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64)
    mode = ((tag >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64)
    k4 = (tag & np.uint64(4)) != 0
    base = t0.min()
    s_ms, e_ms = (t0 - base) * TICK_MS, (t1 - base) * TICK_MS
    print(f"{n} wave records ({used} written, capacity {cap}); span {e_ms.max():.1f} ms")
    bin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    edges = np.arange(0.0, e_ms.max() + bin_ms, bin_ms)
    classes = ["damped-M", "general", "additive"]
    cls = np.array([spec_class(x) for x in sid])
    print("\nresident waves (time-average per bin; capacity 2,048 at two waves per SIMD)")
    print("%12s  %9s %9s %9s %9s" % ("ms", *classes, "all"))
    for lo in edges[:-1]:
        hi = lo + bin_ms
        row = []
        for c in classes:
            m = cls == c
            ov = np.clip(np.minimum(e_ms[m], hi) - np.maximum(s_ms[m], lo), 0.0, None)
            row.append(ov.sum() / bin_ms)
        print("%5.0f-%-6.0f  %9.0f %9.0f %9.0f %9.0f" % (lo, hi, *row, sum(row)))
    # SIMD-level view: how many distinct SIMDs hold at least one traced wave, per bin (HW_ID bits: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13; XCC in the high word)
    simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
    cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
    sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64)
    se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
    xcc = ((hw >> np.uint64(32)) & np.uint64(15)).astype(np.int64)
    where = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    print(f"\ndistinct SIMDs seen: {len(np.unique(where))}")
    print("\nper launch: spec, round, driver (0 one lane, 1 four lanes, 2 one wave per problem; k4), waves, first start, last end, wave duration median / max,"
          " start delay after the launch's first wave median / p90 / max (ms)")
    keys = sorted(set(zip(sid.tolist(), rnd.tolist())), key=lambda k: (s_ms[(sid == k[0]) & (rnd == k[1])].min()))
    for k in keys:
        m = (sid == k[0]) & (rnd == k[1])
        d = e_ms[m] - s_ms[m]
        w = s_ms[m] - s_ms[m].min()
        print("%-14s r%-2d drv%d%s  waves %5d  start %7.1f  end %7.1f  dur %6.2f / %6.2f   delay %6.2f / %6.2f / %6.2f" % (
            spec_name(k[0]), k[1], int(np.bincount(mode[m]).argmax()), " k4" if k4[m].any() else "   ", m.sum(), s_ms[m].min(), e_ms[m].max(),
            np.median(d), d.max(), np.median(w), np.percentile(w, 90), w.max()))


if __name__ == "__main__":
    main()
