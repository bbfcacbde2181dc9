#!/bin/bash
# AutoARIMA queue order (tune arima_queue_sort) on the M5 batch, same box.  Usage: bash tools/arima_sort_sweep.sh
cd /root/repo
for k in 0 1 2 3; do
  export ANOFOX_HIP_TUNE="arima_queue_sort=$k"
  echo "queue_sort=$k: $(python3 bench.py --workload autoarima_css_m5 --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])')"
done
