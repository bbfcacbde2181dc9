#!/bin/bash
# Driver thresholds of the damped multiplicative-trend specs on the default bench line, same box, two rounds.
cd /root/repo
for r in 1 2; do for cfg in "spec_below_md=8192" "spec_below_md=6144" "spec_below_md=10240" "spec_below_md=12288" "spec2_below_md=1536" "spec2_below_md=3072" "spec_below=6144" "spec_below=10240"; do
  export ANOFOX_HIP_TUNE="$cfg"
  echo "$cfg: $(python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["roofline"]["frac"])')"
done; done
