#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box.  Output: gpurun_out/prof_round/ (copy the summaries to profiles/ with
# the round prefix: tools/summarize_profiles.py prints what to keep).
#   1. kernel traces (--kernel-trace --stats) of the default bench line (AutoETS, 30-spec positive batch), the intermittent batch,
#      the fixed-parameter config (BASELINE config 2) and AutoARIMA
#   2. PMC passes for the default line, one counter group per run, kernel trace only (gpurun refuses --pmc with other traces):
#      FETCH_SIZE, WRITE_SIZE (HBM traffic, corrected as MI355X_MICROARCH.md prescribes) and the SQ issue counters
# Usage (from the repo root):  bash tools/profile_round.sh
OUT=/root/repo/gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/$name -o t -- python3 /root/repo/bench.py "$@" > $OUT/$name.log 2>&1
}
run trace_autoets_positive --steps 2 --warmup 1 --cpu-sample 0 --e2e-steps 0
run trace_autoets_m5 --workload autoets_m5 --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0
run trace_ets_aaa_fixed --workload ets_aaa_fixed_m5 --steps 20 --warmup 2 --cpu-sample 0
run trace_autoarima_m5 --workload autoarima_m5 --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0
run trace_autoarima_css_m5 --workload autoarima_css_m5 --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0
run trace_autoets_hourly168 --workload autoets_hourly168 --steps 2 --warmup 1 --cpu-sample 0 --e2e-steps 0
# the default call shape (params := MAP{}: periods detected per series) through the host-buffer batch entry, two calls
rocprofv3 --kernel-trace --stats -d $OUT/trace_autodetect_autoets -o t -- python3 /root/repo/tools/time_autodetect_full.py 30490 AutoETS > $OUT/trace_autodetect_autoets.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o p -- python3 /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 > $OUT/pmc_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmcfixed_$c -o p -- python3 /root/repo/bench.py --workload ets_aaa_fixed_m5 --steps 4 --warmup 0 --cpu-sample 0 > $OUT/pmcfixed_$c.log 2>&1
done
# AutoARIMA (CSS method): HBM traffic of the row-per-lane W reads and the issue counters, per kernel
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmcarima_$n -o p -- python3 /root/repo/bench.py --workload autoarima_css_m5 --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 > $OUT/pmcarima_$n.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $OUT/pmc_SQ_INSTS -o p -- python3 /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 > $OUT/pmc_SQ_INSTS.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_SQ_CYCLES -o p -- python3 /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 > $OUT/pmc_SQ_CYCLES.log 2>&1
python3 /root/repo/tools/summarize_profiles.py $OUT
# occupancy picture of the default line and the intermittent batch: per-spec start / end / busy time and the number of round
# kernels in flight per 25 ms window
for t in trace_autoets_positive trace_autoets_m5; do
  db=$(find $OUT/$t -name "*.db" | head -1)
  [ -n "$db" ] && python3 /root/repo/tools/timeline.py $db > $OUT/timeline_${t#trace_}.txt 2>&1
done
# only the summaries travel back (gpurun merges at most 64 MiB): drop the rocpd databases and per-process trace directories
find $OUT -name "*.db" -delete
find $OUT -mindepth 1 -type d -empty -delete
rm -rf $OUT/../r2_*_trace $OUT/../*_trace 2>/dev/null
du -sh $OUT
ls $OUT
