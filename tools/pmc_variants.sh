#!/bin/bash
# PMC counter groups of the round kernels for several builds of the library (abtest/lib<V>.so) on one workload:
#   gpurun -- "VARIANTS='N P4' BENCH_ARGS='--workload ets_amdn_stress' bash tools/pmc_variants.sh"
cd /tmp && export TMPDIR=/tmp
# (the variant is selected with ANOFOX_HIP_LIB, exported before rocprofv3 starts the program: the product library is never overwritten)
G1="SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
G2="SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
G3="SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS_F64"
G4="SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
for v in ${VARIANTS}; do
  export ANOFOX_HIP_LIB=/root/repo/abtest/lib$v.so
  i=0
  for g in "$G1" "$G2" "$G3" "$G4"; do
    i=$((i+1))
    rocprofv3 --pmc $g --kernel-trace -d /root/repo/gpurun_out/pmcv_${v}_$i -o p -- python3 /root/repo/bench.py ${BENCH_ARGS} --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 > /root/repo/gpurun_out/pmcv_${v}_$i.log 2>&1
  done
done
unset ANOFOX_HIP_LIB
cd /root/repo
python3 - <<'PY'
import sqlite3, glob, os
vs = os.environ.get("VARIANTS", "").split()
tab = {}
for v in vs:
    for db in glob.glob(f"gpurun_out/pmcv_{v}_*/**/*.db", recursive=True):
        con = sqlite3.connect(db)
        for name, val, dur in con.execute("select counter_name, sum(value), sum(end-start) from counters_collection where kernel_name like '%ets_round_kernel%' group by counter_name"):
            tab.setdefault(name, {})[v] = val
        tab.setdefault("kernel_ns(sum of round kernels, last group)", {})[v] = con.execute("select sum(end-start) from (select distinct dispatch_id, start, end from counters_collection where kernel_name like '%ets_round_kernel%')").fetchone()[0]
print("%-34s" % "counter (ets_round_kernel)" + "".join("%18s" % v for v in vs))
for name in sorted(tab):
    print("%-34s" % name[:34] + "".join("%18.4g" % tab[name].get(v, float('nan')) for v in vs))
PY
find /root/repo/gpurun_out -name "*.db" -path "*pmcv_*" -delete
