#!/bin/bash
# One library (ANOFOX_HIP_LIB), several ANOFOX_HIP_TUNE settings, alternating on one box.
# Usage: LIB=abtest/libB.so TUNES="arima_queue_sort=1 arima_queue_sort=4" BENCH_ARGS="--workload autoarima_css_m5" bash tools/ab_tune.sh
cd /root/repo
export ANOFOX_HIP_LIB=$PWD/${LIB:-anofox-forecast_amd/libanofox_fcst_hip.so}
for i in $(seq 1 ${ROUNDS:-3}); do for t in $TUNES; do
  export ANOFOX_HIP_TUNE="$t"
  echo "$t: $(python3 bench.py $BENCH_ARGS --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 --e2e-steps 0 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])')"
done; done
