#!/bin/bash
# Kernel totals of the default call shape (periods detected per series) for one model.  Usage: bash tools/trace_autodetect.sh AutoARIMA
M=${1:-AutoARIMA}
OUT=/root/repo/gpurun_out/trace_autodetect_$M
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/t -o t -- python3 /root/repo/tools/time_autodetect_full.py 30490 $M > $OUT/run.log 2>&1
grep -a "s for" $OUT/run.log
python3 /root/repo/tools/kernel_stats.py $OUT/t | cut -c1-190 | head -40
rm -rf $OUT/t
