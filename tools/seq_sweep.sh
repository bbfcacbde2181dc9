#!/bin/bash
# Which driver for the early rounds of a batch that does not fill the chip eight times over (the intermittent M5 batch: 183k
# live problems)?  -> gpurun_out/seq_sweep.txt
OUT=/root/repo/gpurun_out/seq_sweep.txt
: > $OUT
run() { # label, workload, env...
  local label=$1; shift
  local wl=$1; shift
  local line=$(env "$@" timeout 300 python /root/repo/bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 2>&1 | tail -1)
  echo "$wl $label $(echo $line | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["config"]["mean_passes_per_series"])' 2>/dev/null || echo FAIL)" >> $OUT
}
for wl in autoets_m5 ets_aaa_m5 autoets_stress; do
  run default $wl X=1
  for v in 1 2 3 4 6; do run "seq_rounds=$v" $wl ANOFOX_HIP_TUNE=seq_rounds=$v; done
done
cat $OUT
