#!/bin/bash
# Same-box sweep of ANOFOX_HIP_TUNE settings on a bench workload, every setting interleaved with the base: TUNES="k4=1 spec_below_md=4096" bash tools/tune_sweep.sh
cd /root/repo
OUT=gpurun_out/tune_sweep.txt; : > $OUT
run() { local t="$1"; local line=$(ANOFOX_HIP_TUNE="$t" timeout 300 python bench.py --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 --e2e-steps 0 --also 0 ${BENCH_ARGS} 2>/dev/null | tail -1)
  echo "${t:-base} $(echo $line | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"])' 2>/dev/null || echo FAIL)" | tee -a $OUT; }
for rep in $(seq 1 ${ROUNDS:-2}); do run ""; for t in $TUNES; do run "$t"; done; done
awk '{s[$1]+=$2; n[$1]++} END {for (k in s) printf "%-60s %.1f ms (n=%d)\n", k, s[k]/n[k], n[k]}' $OUT | sort -k2 -n
