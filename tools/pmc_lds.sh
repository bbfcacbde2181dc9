cd /tmp && export TMPDIR=/tmp
for v in N R4; do
  cp /root/repo/abtest/lib$v.so /root/repo/anofox-forecast_amd/libanofox_fcst_hip.so
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d /root/repo/gpurun_out/r04_pmc_lds_$v -o p -- python3 /root/repo/bench.py --workload ets_amdn_stress --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 > /root/repo/gpurun_out/r04_pmc_lds_$v.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d /root/repo/gpurun_out/r04_pmc_cyc_$v -o p -- python3 /root/repo/bench.py --workload ets_amdn_stress --steps 1 --warmup 0 --cpu-sample 0 --e2e-steps 0 > /root/repo/gpurun_out/r04_pmc_cyc_$v.log 2>&1
done
cp /root/repo/abtest/libN.so /root/repo/anofox-forecast_amd/libanofox_fcst_hip.so
cd /root/repo
python3 - <<'PY'
import sqlite3,glob
for v in ("N","R4"):
  for kind in ("lds","cyc"):
    for db in glob.glob(f"gpurun_out/r04_pmc_{kind}_{v}/**/*.db", recursive=True):
        con=sqlite3.connect(db)
        q="select counter_name, sum(value) from counters_collection where kernel_name like '%ets_round_kernel%' group by counter_name"
        print(v, kind, dict(con.execute(q).fetchall()))
PY
