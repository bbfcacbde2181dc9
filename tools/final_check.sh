#!/bin/bash
# Last call of a round: what the driver runs at round end (smoke, the plain bench line -- which now finds the PMC summaries stamped
# with its own build and quotes roofline.traffic), then fuzz seeds the round's campaign did not use -> gpurun_out/final_check/
cd /root/repo
OUT=gpurun_out/final_check; mkdir -p $OUT
SHA=$(sha256sum anofox-forecast_amd/libanofox_fcst_hip.so | cut -c1-16)
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
F=$OUT/fuzz_extra.txt
echo "# differential fuzz, seeds beyond the round's campaign, build $SHA" > $F
for cmd in "tools/fuzz_long.py 1024 12" "tools/fuzz_parity.py 200 53" "tools/fuzz_arima_ml.py 120 7"; do
  echo "## python $cmd" >> $F
  timeout 900 python $cmd 2>&1 | grep -v amdgpu.ids >> $F
done
for seed in 63 64; do
  echo "## FUZZ_COUNTS=1 python tools/fuzz_parity.py 200 $seed" >> $F
  FUZZ_COUNTS=1 timeout 900 python tools/fuzz_parity.py 200 $seed 2>&1 | grep -v amdgpu.ids >> $F
done
grep -c "0 mismatches" $F; grep "mismatches" $F | grep -v " 0 mismatches" | head
