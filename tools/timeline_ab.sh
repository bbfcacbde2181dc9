#!/bin/bash
# Per-spec timeline (tools/timeline.py) of the default bench line for several builds: VARIANTS='A H' bash tools/timeline_ab.sh
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/timeline_ab; rm -rf $OUT; mkdir -p $OUT
for v in ${VARIANTS}; do
  export ANOFOX_HIP_LIB=/root/repo/abtest/lib$v.so
  rocprofv3 --kernel-trace -d $OUT/t_$v -o t -- python3 /root/repo/bench.py ${BENCH_ARGS} --steps 1 --warmup 1 --cpu-sample 0 --e2e-steps 0 --also 0 > $OUT/log_$v.txt 2>&1
  db=$(find $OUT/t_$v -name "*.db" | head -1)
  python3 /root/repo/tools/timeline.py $db > $OUT/timeline_$v.txt 2>&1
done
find $OUT -name "*.db" -delete
