#!/bin/bash
# rounds vs work pool on single-spec workloads (kernel efficiency without cross-spec effects) -> gpurun_out/ab_sched.txt
OUT=/root/repo/gpurun_out/ab_sched.txt
: > $OUT
for wl in ${WLS:-ets_aaa_m5 ets_amdn_stress ets_mam_stress autoets_m5}; do
  for sc in rounds pool; do
    line=$(ANOFOX_HIP_SCHED=$sc timeout 300 python /root/repo/bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 2>&1 | tail -1)
    echo "$wl $sc $(echo $line | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["config"]["mean_passes_per_series"], j["roofline"]["frac"])' 2>/dev/null || echo FAIL)" >> $OUT
  done
done
cat $OUT
