#!/bin/bash
# A/B of two builds of the library on ONE box (box-to-box spread is 2-3 %): put the builds in abtest/libA.so and abtest/libB.so,
# then  gpurun -- "BENCH_ARGS=\"--workload W\" bash tools/ab_lib.sh"; alternates runs of the bench line.  Restore the real build afterwards.
cd /root/repo
for i in 1 2 3; do for v in A B; do cp abtest/lib$v.so anofox-forecast_amd/libanofox_fcst_hip.so; echo -n "$v "; timeout 200 python bench.py ${BENCH_ARGS} --steps 4 --warmup 1 --cpu-sample 0 --e2e-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
