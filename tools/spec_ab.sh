#!/bin/bash
# Single-spec fits on the M5 shape (30,490 x 1,913, m = 7, strictly positive) for several builds: SPECS='AMdA MMdM' VARIANTS='A H' bash tools/spec_ab.sh
cd /root/repo
for sp in ${SPECS}; do for v in ${VARIANTS}; do
  echo -n "$sp $v "
  ANOFOX_HIP_LIB=$PWD/abtest/lib$v.so python bench.py --workload ets_aaa_m5 --ets-model $sp --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 --e2e-steps 0 --also 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['mean_passes_per_series'], d['config']['max_passes_per_series'])"
done; done
