#!/bin/bash
# A/B of run-time knobs on ONE box with ONE build:  gpurun -- "KNOB=ANOFOX_HIP_TUNE VALUES="k4=0 k4=1" BENCH_ARGS='--workload autoets_m5' bash tools/ab_env.sh"
cd /root/repo
for i in $(seq 1 ${ROUNDS:-3}); do for v in ${VALUES}; do echo -n "$KNOB=$v "; env $KNOB=$v timeout 300 python bench.py ${BENCH_ARGS} --steps ${STEPS:-4} --warmup 1 --cpu-sample 0 --e2e-steps 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_min_passes'), d['config']['mean_passes_per_series'])"; done; done
