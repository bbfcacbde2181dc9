#!/bin/bash
# Sweep the round-schedule knobs on the default bench workload -> gpurun_out/rounds_sweep.txt
OUT=/root/repo/gpurun_out/rounds_sweep.txt
: > $OUT
run() { # label, env...
  local label=$1; shift
  local line=$(env "$@" timeout 300 python /root/repo/bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 ${BENCH_ARGS} 2>&1 | tail -1)
  echo "$label $(echo $line | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["config"]["mean_passes_per_series"])' 2>/dev/null || echo FAIL)" >> $OUT
}
run base X=1
for v in 2048 4096 16384 32768; do run "spec_below_md=$v" ANOFOX_HIP_TUNE=spec_below_md=$v; done
for v in 2048 4096 16384 32768; do run "spec_below=$v(all)" ANOFOX_HIP_TUNE=spec_below=$v; done
run "budgets=16x8,32,32,64,64,128,1024" ANOFOX_HIP_TUNE=budgets=16,16,16,16,16,16,16,16,32,32,64,64,128,1024
run "budgets=32x4,64,64,128,1024" ANOFOX_HIP_TUNE=budgets=32,32,32,32,64,64,128,1024
run "budgets=24x8,48,96,192,1024" ANOFOX_HIP_TUNE=budgets=24,24,24,24,24,24,24,24,48,96,192,1024
run "budgets=24x6,48,48,96,96,96,96,1024" ANOFOX_HIP_TUNE=budgets=24,24,24,24,24,24,48,48,96,96,96,96,1024
run "seq_rounds=6" ANOFOX_HIP_TUNE=seq_rounds=6
run "seq_rounds=8" ANOFOX_HIP_TUNE=seq_rounds=8
cat $OUT
