#!/bin/bash
# Copy the summaries of the last tools/bench_round.sh + tools/profile_round.sh run (gpurun_out/) into profiles/ under the round prefix.
# Usage: bash tools/collect_profiles.sh r04
R=${1:?round prefix}
cd /root/repo
for f in gpurun_out/bench_round/bench_*.json; do cp $f profiles/${R}_$(basename $f); done
for f in autodetect_full.txt single_call_latency.txt; do [ -f gpurun_out/bench_round/$f ] && cp gpurun_out/bench_round/$f profiles/${R}_$f; done
P=gpurun_out/prof_round
for f in rocprof_summary.txt pmc_traffic.json pmc_traffic_arima.json pmc_traffic_fixed.json pmc_FETCH_SIZE_by_kernel.csv pmc_WRITE_SIZE_by_kernel.csv timeline_autoets_m5.txt timeline_autoets_positive.txt; do
  [ -f $P/$f ] && cp $P/$f profiles/${R}_$f
done
for f in $P/trace_*_kernel_stats.csv; do cp $f profiles/${R}_$(basename $f); done
ls profiles | grep "^${R}_" | wc -l
