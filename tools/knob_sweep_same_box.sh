#!/bin/bash
# Same-box sweep of the round-schedule knobs on the default bench line (box-to-box spread is 2-3 %, larger than most effects):
# every configuration runs twice, interleaved with the base configuration.  -> gpurun_out/knob_sweep.txt
OUT=/root/repo/gpurun_out/knob_sweep.txt
: > $OUT
run() { # label, env...
  local label=$1; shift
  local line=$(env "$@" timeout 300 python /root/repo/bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 ${BENCH_ARGS} 2>&1 | tail -1)
  echo "$label $(echo $line | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"])' 2>/dev/null || echo FAIL)" | tee -a $OUT
}
for rep in 1 2; do
run base X=1
run "budgets=24x6,48,96,192,1024" ANOFOX_HIP_TUNE=budgets=24,24,24,24,24,24,48,96,192,1024
run "budgets=24x4,48,48,96,96,192,1024" ANOFOX_HIP_TUNE=budgets=24,24,24,24,48,48,96,96,192,1024
run "budgets=32x5,64,64,128,256,1024" ANOFOX_HIP_TUNE=budgets=32,32,32,32,32,64,64,128,256,1024
run base X=1
run "budgets=24x8,48,48,96,96,192,1024" ANOFOX_HIP_TUNE=budgets=24,24,24,24,24,24,24,24,48,48,96,96,192,1024
run "budgets=16x6,32,32,64,64,128,256,1024" ANOFOX_HIP_TUNE=budgets=16,16,16,16,16,16,32,32,64,64,128,256,1024
run "spec2_md=4096" ANOFOX_HIP_TUNE=spec2_below_md=4096
run base X=1
run "spec2_md=1024" ANOFOX_HIP_TUNE=spec2_below_md=1024
run "spec_below=2x" ANOFOX_HIP_TUNE=spec_below=16384
run "spec_below_md=2x" ANOFOX_HIP_TUNE=spec_below_md=32768
run "prio=0" ANOFOX_HIP_TUNE=prio_streams=0
done
sort $OUT | awk '{k=$1; for(i=2;i<NF;i++) k=k" "$i; s[k]+=$NF; n[k]++} END {for (k in s) printf "%-48s %.1f ms (n=%d)\n", k, s[k]/n[k], n[k]}' | sort -k2 -t' ' | tee -a $OUT
