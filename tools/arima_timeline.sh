#!/bin/bash
# Per-launch picture of one AutoARIMA step on the GPU box: the queue length of every sweep (library trace) next to the duration of every
# dispatch in start order (rocprofv3 kernel trace).  Usage: bash tools/arima_timeline.sh [workload] > gpurun_out/arima_timeline.txt
W=${1:-autoarima_css_m5}
OUT=/root/repo/gpurun_out/arima_tl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ANOFOX_HIP_TUNE="arima_trace=1"
rocprofv3 --kernel-trace -d $OUT/t -o t -- python3 /root/repo/bench.py --workload $W --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 > $OUT/bench.log 2>&1
grep -a "AutoARIMA\|value" $OUT/bench.log | tail -40
python3 /root/repo/tools/kernel_stats.py $OUT/t --timeline 75
rm -rf $OUT/t
