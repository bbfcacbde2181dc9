#!/bin/bash
# Wave cycles and VALU instructions of the round kernels for one bench workload (two --pmc passes, kernel trace only).
# Usage (GPU box): bash tools/pmc_workload.sh autoets_hourly168
W=${1:?workload}
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_$W
mkdir -p $OUT
i=0
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $g --kernel-trace -d $OUT/g$i -o p -- python3 /root/repo/bench.py --workload $W --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 --also 0 > $OUT/g$i.log 2>&1
done
cd /root/repo
python3 - $OUT <<'PY'
import sqlite3, glob, sys, re, collections
tab = collections.defaultdict(dict)
for db in glob.glob(sys.argv[1] + "/g*/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    for kname, cname, val in con.execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"):
        m = re.search(r"ets_round_kernel<anofox::EtsCfg<(\d+), (\d+), (true|false), (\d+)>, (-?\d+)", kname)
        if not m: continue
        cls = "m%s %s" % (m.group(5), "damped-M" if (m.group(2) == "2" and m.group(3) == "true") else ("additive" if (m.group(1) == "1" and m.group(2) != "2" and m.group(4) != "2") else "general"))
        tab[cls][cname] = tab[cls].get(cname, 0) + val
names = sorted({c for v in tab.values() for c in v})
print("%-22s" % "round kernels" + "".join("%20s" % n for n in names) + "   wave-cycles per VALU instruction")
for cls in sorted(tab):
    r = tab[cls]
    print("%-22s" % cls + "".join("%20.4g" % r.get(n, float("nan")) for n in names) + "   %.1f" % (4.0 * r.get("SQ_WAVE_CYCLES", 0) / max(r.get("SQ_INSTS_VALU", 1), 1)))
PY
find $OUT -name "*.db" -delete
