#!/bin/bash
# The round's bench lines on the GPU box -> gpurun_out/bench_round/*.json (copy to profiles/<round>_bench_*.json).
# Every line is the plain `python bench.py --workload W` (default steps, CPU baseline and host-buffer leg included); the strong-scaling
# shards are rank 0 of an 8-rank job measured alone (--scaling strong --simulate-world 8).
OUT=/root/repo/gpurun_out/bench_round
rm -rf $OUT; mkdir -p $OUT
cd /root/repo
for w in autoets_m5_positive autoets_m5 ets_aaa_fixed_m5 ets_aaa_m5 autoets_stress autoarima_m5 autoarima_css_m5 autoets_m24 autoets_hourly168; do
  extra=""
  [ $w = ets_aaa_fixed_m5 ] && extra="--steps 20"
  timeout 600 python bench.py --workload $w $extra 2>$OUT/$w.err | tail -1 > $OUT/bench_$w.json
done
for w in autoets_m5_positive autoets_m5 autoarima_css_m5; do
  timeout 300 python bench.py --workload $w --scaling strong --simulate-world 8 --steps 5 --cpu-sample 0 --e2e-steps 0 2>>$OUT/strong.err | tail -1 > $OUT/bench_strong8_shard0_$w.json
done
timeout 300 python bench.py --workload autoets_stress --scaling strong --simulate-world 8 --n-series 1000000 --steps 3 --cpu-sample 0 --e2e-steps 0 2>>$OUT/strong.err | tail -1 > $OUT/bench_strong8_shard0_autoets_stress_1M.json
for w in 2 4; do
  timeout 300 python bench.py --workload autoets_m5_positive --scaling strong --simulate-world $w --steps 5 --cpu-sample 0 --e2e-steps 0 2>>$OUT/strong.err | tail -1 > $OUT/bench_strong${w}_shard0_autoets_m5_positive.json
  timeout 300 python bench.py --workload autoets_stress --scaling strong --simulate-world $w --n-series 1000000 --steps 3 --cpu-sample 0 --e2e-steps 0 2>>$OUT/strong.err | tail -1 > $OUT/bench_strong${w}_shard0_autoets_stress_1M.json
done
# BASELINE config 5 whole on ONE GPU (1M x 1,024: 8.2 GB of series, per-spec gather blocks of 772k columns = 158 GB)
timeout 900 python bench.py --workload autoets_stress --n-series 1000000 --steps 2 --cpu-sample 0 --e2e-steps 0 2>>$OUT/strong.err | tail -1 > $OUT/bench_autoets_stress_1M_one_gpu.json
# the default call shape (periods detected per series), the single-call latencies and the concurrent C workers
for v in positive intermittent; do for m in AutoETS AutoARIMA; do
  REPS=4 ANOFOX_HIP_TIMING=1 timeout 300 python tools/time_autodetect_full.py 30490 $m $v 2>&1 | grep -E "forecast_batch:|period detection|calls:|series/s" | sed "s/^/[$m $v] /" >> $OUT/autodetect_full.txt
done; done
timeout 300 python tools/time_single_call.py 2>&1 | grep -v amdgpu.ids > $OUT/single_call_latency.txt
find $OUT -size 0 -delete
ls -la $OUT
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('/root/repo/gpurun_out/bench_round/bench_*.json')):
    try:
        d = json.load(open(f))
        r = d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d['unit'], d['ms_per_step'], 'frac', r.get('frac'), 'traffic', r.get('traffic'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
