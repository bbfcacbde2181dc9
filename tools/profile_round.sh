#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box.  Output: gpurun_out/prof_round/ (copy the summaries to profiles/).
#   1. kernel trace of the default bench line (AutoETS, 30-spec positive batch), of the intermittent batch and of AutoARIMA
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, kernel trace only) for the default line
# Usage (from the repo root):  bash tools/profile_round.sh
OUT=/root/repo/gpurun_out/prof_round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace -d $OUT/$name -o t -- python3 /root/repo/bench.py "$@" > $OUT/$name.log 2>&1
}
run trace_autoets_positive --steps 2 --warmup 1 --cpu-sample 0
run trace_autoets_m5 --workload autoets_m5 --steps 3 --warmup 1 --cpu-sample 0
run trace_autoarima_m5 --workload autoarima_m5 --steps 1 --warmup 0 --cpu-sample 0
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o p -- python3 /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $OUT/pmc_$c.log 2>&1
done
python3 /root/repo/tools/summarize_profiles.py $OUT
ls $OUT
