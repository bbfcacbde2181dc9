#!/bin/bash
# Skip the dense re-gather of a round while more than a share of the series is still running -> gpurun_out/gather_sweep.txt
OUT=/root/repo/gpurun_out/gather_sweep.txt
: > $OUT
run() { # label, workload, env...
  local label=$1; shift
  local wl=$1; shift
  local line=$(env "$@" timeout 300 python /root/repo/bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 2>&1 | tail -1)
  echo "$wl $label $(echo $line | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["config"]["mean_passes_per_series"])' 2>/dev/null || echo FAIL)" >> $OUT
}
for wl in autoets_m5 autoets_m5_positive autoets_stress ets_aaa_m5; do
  for v in 1.0 0.9 0.75 0.5 0.25; do run "gather_max_frac=$v" $wl ANOFOX_HIP_GATHER_MAX_FRAC=$v; done
done
cat $OUT
