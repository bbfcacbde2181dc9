#!/bin/bash
# Sweep the work-pool knobs on the default bench workload; one line per configuration -> gpurun_out/pool_sweep.txt
OUT=/root/repo/gpurun_out/pool_sweep.txt
: > $OUT
run() { # label, env...
  local label=$1; shift
  local line=$(env "$@" timeout 300 python /root/repo/bench.py --steps 2 --warmup 1 --cpu-sample 0 --e2e-steps 0 ${BENCH_ARGS} 2>&1 | tail -1)
  echo "$label $(echo $line | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["config"]["mean_passes_per_series"])' 2>/dev/null || echo FAIL $line | cut -c1-300)" >> $OUT
}
for w in ${WAVES:-1024 2048 3072 4096}; do run "waves=$w" ANOFOX_HIP_POOL_WAVES=$w; done
for p in ${PROMOTES:-48 96 192 100000}; do run "promote=$p" ANOFOX_HIP_PROMOTE=$p; done
run "rounds" ANOFOX_HIP_SCHED=rounds
cat $OUT
