#!/bin/bash
# A/B/... of several builds of the library on ONE box (box-to-box spread is 2-3 %): put the builds in abtest/lib<V>.so
# (make -C anofox-forecast_amd/csrc BUILD=/tmp/bV OUT=$PWD/abtest/libV.so EXTRA=-D...), then
#   gpurun -- "VARIANTS='A B C' BENCH_ARGS='--workload W' bash tools/ab_lib.sh"
# alternates runs of the bench line.  The variant is selected with ANOFOX_HIP_LIB (lib.py): the product library in the tree is
# never touched (round 4's version copied the variants over it and a timeout left an experiment build in its place).
cd /root/repo
for i in $(seq 1 ${ROUNDS:-3}); do for v in ${VARIANTS:-A B}; do
  echo -n "$v "
  ANOFOX_HIP_LIB=$PWD/abtest/lib$v.so timeout 300 python bench.py ${BENCH_ARGS} --steps ${STEPS:-4} --warmup 1 --cpu-sample 0 --e2e-steps 0 --also 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['mean_passes_per_series'])"
done; done
