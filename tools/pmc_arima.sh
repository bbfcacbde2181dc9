#!/bin/bash
# PMC counter passes over the AutoARIMA fit kernels (bounded batch); results land in gpurun_out/pmc_arima_*/
cd /tmp && export TMPDIR=/tmp
N=${1:-4096}
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d /root/repo/gpurun_out/pmc_arima_$n -o p -- python3 /root/repo/bench.py --workload autoarima_m5 --n-series $N --steps 1 --warmup 0 --cpu-sample 0 > /root/repo/gpurun_out/pmc_arima_$n.log 2>&1
done
ls /root/repo/gpurun_out | grep pmc_arima
