#!/bin/bash
# The round's whole evidence cycle on ONE GPU box, in the order that matters if the call is cut short: bench lines, traces + PMC,
# wave residency, fuzz.  Usage: gpurun --timeout 3000 -- 'bash tools/evidence_round.sh'; then bash tools/collect_profiles.sh rNN
cd /root/repo
bash tools/bench_round.sh > gpurun_out/bench_round.log 2>&1
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
mkdir -p gpurun_out/wave
ANOFOX_HIP_TUNE="wave_trace=/root/repo/gpurun_out/wave/wt.bin" python bench.py --steps 1 --warmup 1 --cpu-sample 0 --e2e-steps 0 --also 0 > gpurun_out/wave/bench.log 2>&1
python tools/wave_trace.py gpurun_out/wave/wt.bin 10 > gpurun_out/wave/wave_residency.txt 2>&1
rm -f gpurun_out/wave/wt.bin
bash tools/fuzz_round.sh > gpurun_out/fuzz_round.log 2>&1
tail -5 gpurun_out/fuzz_round.txt
