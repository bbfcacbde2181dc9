#!/bin/bash
# Sweep the two-level speculation thresholds on the default bench workload (and the intermittent batch) -> gpurun_out/spec2_sweep.txt
OUT=/root/repo/gpurun_out/spec2_sweep.txt
: > $OUT
run() { # label, workload, env...
  local label=$1; shift
  local wl=$1; shift
  local line=$(env "$@" timeout 300 python /root/repo/bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 2>&1 | tail -1)
  echo "$wl $label $(echo $line | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["config"]["mean_passes_per_series"])' 2>/dev/null || echo FAIL)" >> $OUT
}
run base autoets_m5_positive X=1
for v in 2048 4096 8192; do run "spec2_md=$v" autoets_m5_positive ANOFOX_HIP_TUNE=spec2_below_md=$v; done
for v in 1024 2048 4096; do run "spec2_all=$v" autoets_m5_positive ANOFOX_HIP_TUNE=spec2_below=$v; done
run "spec2_md=4096,other=1024" autoets_m5_positive "ANOFOX_HIP_TUNE=spec2_below=1024;spec2_below_md=4096"
run base autoets_m5 X=1
for v in 64 256 1024 4096 16384; do run "spec2_all=$v" autoets_m5 ANOFOX_HIP_TUNE=spec2_below=$v; done
run base ets_aaa_m5 X=1
for v in 2048 4096 8192 30490; do run "spec2_all=$v" ets_aaa_m5 ANOFOX_HIP_TUNE=spec2_below=$v; done
run base autoets_stress X=1
for v in 1024 4096; do run "spec2_all=$v" autoets_stress ANOFOX_HIP_TUNE=spec2_below=$v; done
cat $OUT
