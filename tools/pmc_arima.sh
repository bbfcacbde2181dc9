#!/bin/bash
# PMC counter passes over the AutoARIMA kernels of one bench step (one counter group per run, kernel trace only); prints per-kernel sums.
# Usage: bash tools/pmc_arima.sh [workload] [n_series]      (results also in gpurun_out/pmc_arima/)
W=${1:-autoarima_css_m5}
N=${2:-30490}
OUT=/root/repo/gpurun_out/pmc_arima
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INST_CYCLES_VMEM"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$n -o p -- python3 /root/repo/bench.py --workload $W --n-series $N --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 > $OUT/$n.log 2>&1
done
python3 - <<'P'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('/root/repo/gpurun_out/pmc_arima/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('anofox::', '').replace('void ', ''))[:60]
        if 'arima' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(agg.items()):
    print(k)
    for c, x in sorted(v.items()): print(f'    {c:28s} {x:18.0f}')
P
find $OUT -name "*.db" -delete
