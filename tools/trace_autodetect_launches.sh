#!/bin/bash
# Per-launch durations of the long-period (period per lane, HBM ring) round kernels in the default call shape.
# Usage (GPU box): bash tools/trace_autodetect_launches.sh [AutoETS]
M=${1:-AutoETS}
OUT=/root/repo/gpurun_out/ad_launches_$M
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/t -o t -- python3 /root/repo/tools/time_autodetect_full.py 30490 $M > $OUT/run.log 2>&1
DB=$(find $OUT/t -name "*.db" | head -1)
python3 /root/repo/tools/launch_list.py $DB "" detect_period_kernel > $OUT/launches.txt
grep -c . $OUT/launches.txt
