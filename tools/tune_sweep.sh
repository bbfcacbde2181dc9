#!/bin/bash
# One-box sweep of ANOFOX_HIP_TUNE settings on a bench workload: SETTINGS="a=1 b=2;c=3 ..." (space separated), each run twice, alternating.
#   gpurun -- "SETTINGS='k4=0 k4=1' BENCH_ARGS='--workload autoets_m5_positive' bash tools/tune_sweep.sh"
cd /root/repo
for i in $(seq 1 ${ROUNDS:-2}); do for v in base ${SETTINGS}; do
  t=$v; [ $v = base ] && t=""
  echo -n "$v "
  ANOFOX_HIP_TUNE="$t" timeout 300 python bench.py ${BENCH_ARGS} --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 --e2e-steps 0 --also 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['mean_passes_per_series'])"
done; done
