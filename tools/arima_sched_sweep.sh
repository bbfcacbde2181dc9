#!/bin/bash
# AutoARIMA schedule knobs on the M5 batch, same box: four-lane threshold (arima_spec_factor) x lookahead factor x depth.
# Usage: bash tools/arima_sched_sweep.sh [workload] ["spec factors"] ["lookahead factors"] ["depths"]
cd /root/repo
W=${1:-autoarima_css_m5}
for sf in ${2:-3 4 5 6}; do
  for la in ${3:-1 2 4 6}; do
    for dp in ${4:-2}; do
      export ANOFOX_HIP_TUNE="arima_spec_factor=$sf;arima_lookahead=$la;arima_lookahead_depth=$dp"
      echo "spec_factor=$sf lookahead=$la depth=$dp: $(python3 bench.py --workload $W --steps 2 --warmup 1 --cpu-sample 0 --e2e-steps 0 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])')"
    done
  done
done
