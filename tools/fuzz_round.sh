#!/bin/bash
# The round's differential fuzz campaign on the GPU box (HIP path vs oracle, bit for bit) -> gpurun_out/fuzz_round.txt
cd /root/repo
SHA=$(sha256sum anofox-forecast_amd/libanofox_fcst_hip.so | cut -c1-16)
OUT=gpurun_out/fuzz_round.txt
echo "# differential fuzz on the MI355X box, build $SHA" > $OUT
for cmd in "tools/fuzz_long.py 1024 ${1:-11}" "tools/fuzz_parity.py 200 51" "tools/fuzz_parity.py 200 52" "tools/fuzz_arima_ml.py 120 6"; do
  echo "## python $cmd" >> $OUT
  timeout 900 python $cmd 2>&1 | grep -v amdgpu.ids >> $OUT
done
# round 6: count series through the compact-storage kernels (float / uint16 copy of the block), two seeds
for seed in 61 62; do
  echo "## FUZZ_COUNTS=1 python tools/fuzz_parity.py 200 $seed" >> $OUT
  FUZZ_COUNTS=1 timeout 900 python tools/fuzz_parity.py 200 $seed 2>&1 | grep -v amdgpu.ids >> $OUT
done
tail -40 $OUT
