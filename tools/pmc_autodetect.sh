#!/bin/bash
# VALU instructions and wave cycles of the default call shape (periods detected per series), by kind of round kernel.
# Usage (GPU box): bash tools/pmc_autodetect.sh [AutoETS]
M=${1:-AutoETS}
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_autodetect_$M
mkdir -p $OUT
i=0
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  REPS=1 rocprofv3 --pmc $g --kernel-trace -d $OUT/g$i -o p -- python3 /root/repo/tools/time_autodetect_full.py 30490 $M > $OUT/g$i.log 2>&1
done
cd /root/repo
python3 - $OUT <<'PY'
import sqlite3, glob, sys, re, collections
out = sys.argv[1]
tab = collections.defaultdict(dict)
for db in glob.glob(out + "/g*/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    for kname, cname, val in con.execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"):
        m = re.search(r"ets_round_kernel<anofox::EtsCfg<(\d+), (\d+), (true|false), (\d+)>, (-?\d+)", kname)
        if m:
            cls = "round m%s %s" % (m.group(5), "damped-M" if (m.group(2) == "2" and m.group(3) == "true") else ("additive" if (m.group(1) == "1" and m.group(2) != "2" and m.group(4) != "2") else "general"))
        else:
            cls = re.sub(r"<.*", "", kname.replace("void ", "").replace("anofox::", ""))[:40]
        tab[cls][cname] = tab[cls].get(cname, 0) + val
names = sorted({c for v in tab.values() for c in v})
print("%-34s" % "kernel kind" + "".join("%20s" % n for n in names))
tot = collections.Counter()
for cls in sorted(tab, key=lambda c: -tab[c].get("SQ_INSTS_VALU", 0)):
    print("%-34s" % cls + "".join("%20.4g" % tab[cls].get(n, float("nan")) for n in names))
    for n in names: tot[n] += tab[cls].get(n, 0)
print("%-34s" % "total" + "".join("%20.4g" % tot[n] for n in names))
PY
find $OUT -name "*.db" -delete
