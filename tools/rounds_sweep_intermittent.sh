#!/bin/bash
OUT=/root/repo/gpurun_out/rounds_sweep_int.txt
: > $OUT
run() { # label, workload, env...
  local label=$1; shift
  local wl=$1; shift
  local line=$(env "$@" timeout 200 python /root/repo/bench.py --workload $wl --steps 4 --warmup 1 --cpu-sample 0 --e2e-steps 0 2>&1 | tail -1)
  echo "$wl $label $(echo $line | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["config"]["mean_passes_per_series"])' 2>/dev/null || echo FAIL)" >> $OUT
}
for wl in autoets_m5 ets_aaa_m5; do
run base $wl X=1
run "48,48,96,192,1024" $wl ANOFOX_HIP_TUNE=budgets=48,48,96,192,1024
run "32,32,64,128,256,1024" $wl ANOFOX_HIP_TUNE=budgets=32,32,64,128,256,1024
run "32,64,128,1024" $wl ANOFOX_HIP_TUNE=budgets=32,64,128,1024
run "64,128,1024" $wl ANOFOX_HIP_TUNE=budgets=64,128,1024
run "24,24,48,96,192,1024" $wl ANOFOX_HIP_TUNE=budgets=24,24,48,96,192,1024
run "16,16,32,64,128,256,1024" $wl ANOFOX_HIP_TUNE=budgets=16,16,32,64,128,256,1024
done
cat $OUT
