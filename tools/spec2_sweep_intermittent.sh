cd /root/repo
for r in 1 2; do for cfg in "spec2_below=1024" "spec2_below=2048" "spec2_below=4096" "spec2_below=512" "k4_top=1;k4_top_below=24576"; do
  export ANOFOX_HIP_TUNE="$cfg"
  echo "$cfg: $(python3 bench.py --workload autoets_m5 --steps 6 --warmup 2 --cpu-sample 0 --e2e-steps 0 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["roofline"]["frac"])')"
done; done
