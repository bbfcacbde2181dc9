#!/bin/bash
# Priority streams (tune prio_streams) on one box.  Usage: bash tools/prio_sweep.sh workload "0 1 2 3 7"
cd /root/repo
echo "GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-unset}"
for r in 1 2; do for p in ${2:-0 1 2 3 7}; do
  export ANOFOX_HIP_TUNE="prio_streams=$p"
  echo "prio_streams=$p: $(python3 bench.py --workload ${1:-autoets_m5} --steps 6 --warmup 2 --cpu-sample 0 --e2e-steps 0 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["roofline"]["frac"])')"
done; done
