#!/bin/bash
# PMC counter passes over one AutoETS bench step; results land in gpurun_out/pmc_ets_*/
cd /tmp && export TMPDIR=/tmp
WL=${1:-autoets_m5_positive}
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d /root/repo/gpurun_out/pmc_ets_$n -o p -- python3 /root/repo/bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 0 > /root/repo/gpurun_out/pmc_ets_$n.log 2>&1
done
ls /root/repo/gpurun_out | grep pmc_ets
