#!/bin/bash
# First-round kernel time (a fixed number of passes: the comparable unit when experiment builds change the iterates) of one workload for
# several builds of the library (abtest/lib<V>.so) and batch sizes:  VARIANTS='A X1' SIZES='62500 125000' WL=ets_amdn_stress bash tools/round1_time.sh
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/round1; rm -rf $OUT; mkdir -p $OUT
for v in ${VARIANTS}; do for n in ${SIZES:-125000}; do
  export ANOFOX_HIP_LIB=/root/repo/abtest/lib$v.so
  rocprofv3 --kernel-trace -d $OUT/t_${v}_$n -o t -- python3 /root/repo/bench.py --workload ${WL:-ets_amdn_stress} --n-series $n --steps 1 --warmup 1 --cpu-sample 0 --e2e-steps 0 --also 0 > $OUT/log_${v}_$n.txt 2>&1
  db=$(find $OUT/t_${v}_$n -name "*.db" | head -1)
  echo "$v n=$n: $(python3 /root/repo/tools/launch_list.py $db round | head -2 | awk '{printf "%s ms (wg %s)  ", $3, $6}')  step: $(tail -1 $OUT/log_${v}_$n.txt | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>/dev/null)"
done; done
find $OUT -name "*.db" -delete
