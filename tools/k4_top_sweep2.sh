cd /root/repo
for r in 1 2; do for cfg in "k4_top_below=0" "k4_top=1;k4_top_below=12288" "k4_top=1;k4_top_below=14336" "k4_top=1;k4_top_below=16384" "k4_top=1;k4_top_below=18432" "k4_top=1;k4_top_below=20480" "k4_top=1;k4_top_below=24576" "k4_top=2;k4_top_below=16384" "k4_top=2;k4_top_below=14336"; do
  export ANOFOX_HIP_TUNE="$cfg"
  echo "$cfg: $(python3 bench.py --workload autoets_m5 --steps 6 --warmup 2 --cpu-sample 0 --e2e-steps 0 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["roofline"]["frac"])')"
done; done
