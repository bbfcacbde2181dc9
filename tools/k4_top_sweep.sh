#!/bin/bash
# Four-lane threshold of the K4 (additive-class) specs: tune k4_top (how many of the most expensive specs) x k4_top_below (live problems).
# Usage: bash tools/k4_top_sweep.sh [workload] ; same box, two rounds
cd /root/repo
W=${1:-autoets_m5}
for r in 1 2; do for cfg in "k4_top_below=0" "k4_top=6;k4_top_below=10240" "k4_top=6;k4_top_below=12288" "k4_top=6;k4_top_below=14336" "k4_top=6;k4_top_below=16384" "k4_top=6;k4_top_below=18432" "k4_top=6;k4_top_below=20480" "k4_top=6;k4_top_below=24576" "k4_top=1;k4_top_below=16384" "k4_top=3;k4_top_below=16384"; do
  export ANOFOX_HIP_TUNE="$cfg"
  echo "$cfg: $(python3 bench.py --workload $W --steps 6 --warmup 2 --cpu-sample 0 --e2e-steps 0 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["roofline"]["frac"])')"
done; done
