cd /root/repo
for r in 1 2; do for cfg in "k4_top_below=0" "k4_top=1;k4_top_below=20480" "k4_top=1;k4_top_below=65536" "k4_top=1;k4_top_below=98304"; do
  export ANOFOX_HIP_TUNE="$cfg"
  echo "stress $cfg: $(python3 bench.py --workload autoets_stress --steps 4 --warmup 1 --cpu-sample 0 --e2e-steps 0 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["roofline"]["frac"])')"
done; done
for cfg in "k4_top_below=0" "k4_top=1;k4_top_below=20480"; do
  export ANOFOX_HIP_TUNE="$cfg"
  echo "positive $cfg: $(python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["roofline"]["frac"])')"
  echo "shard $cfg: $(python3 bench.py --workload autoets_m5 --simulate-world 8 --steps 6 --warmup 2 --cpu-sample 0 --e2e-steps 0 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["roofline"]["frac"])')"
done
