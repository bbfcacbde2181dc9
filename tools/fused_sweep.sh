#!/bin/bash
# Compaction + gather fused into the end of every round kernel (ANOFOX_HIP_FUSED=1) against the separate kernels -> gpurun_out/fused_sweep.txt
OUT=/root/repo/gpurun_out/fused_sweep.txt
: > $OUT
run() { # label, workload, env...
  local label=$1; shift
  local wl=$1; shift
  local line=$(env "$@" timeout 300 python /root/repo/bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 2>&1 | tail -1)
  echo "$wl $label $(echo $line | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["config"]["mean_passes_per_series"])' 2>/dev/null || echo FAIL)" >> $OUT
}
for wl in autoets_m5 autoets_m5_positive ets_aaa_m5 autoets_stress; do
  run separate $wl ANOFOX_HIP_FUSED=0
  run fused $wl ANOFOX_HIP_FUSED=1
  run "fused,6 rounds" $wl ANOFOX_HIP_FUSED=1 ANOFOX_HIP_BUDGETS=48,48,48,96,192,1024
  run "separate,6 rounds" $wl ANOFOX_HIP_FUSED=0 ANOFOX_HIP_BUDGETS=48,48,48,96,192,1024
done
cat $OUT
