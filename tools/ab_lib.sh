#!/bin/bash
# A/B/... of several builds of the library on ONE box (box-to-box spread is 2-3 %): put the builds in abtest/lib<V>.so
# (make -C anofox-forecast_amd/csrc BUILD=/tmp/bV OUT=$PWD/abtest/libV.so EXTRA=-D...), then
#   gpurun -- "VARIANTS='A B C' BENCH_ARGS='--workload W' bash tools/ab_lib.sh"
# alternates runs of the bench line.  The real build is restored afterwards.
cd /root/repo
cp anofox-forecast_amd/libanofox_fcst_hip.so /tmp/lib_real.so
for i in $(seq 1 ${ROUNDS:-3}); do for v in ${VARIANTS:-A B}; do cp abtest/lib$v.so anofox-forecast_amd/libanofox_fcst_hip.so; echo -n "$v "; timeout 300 python bench.py ${BENCH_ARGS} --steps ${STEPS:-4} --warmup 1 --cpu-sample 0 --e2e-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['mean_passes_per_series'])"; done; done
cp /tmp/lib_real.so anofox-forecast_amd/libanofox_fcst_hip.so
