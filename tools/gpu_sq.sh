#!/bin/bash
# SQ / GRBM counters of the hot kernels: separate rocprofv3 passes (8 SQ slots + 2 GRBM per pass; no trace domain beside
# --kernel-trace), the program directly after `--`.  Summarise with tools/sq_summary.py -> profiles/r3_sq_<tag>.json.
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-sq}
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_IFETCH_LEVEL SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
P3="SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_BUSY_CU_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/${T}_p$i -- python3 $R/tools/sq_workload.py > $R/gpurun_out/${T}_p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/${T}_p$i.log; exit $i; }
done
python3 $R/tools/sq_summary.py $R/gpurun_out/${T}_p1 $R/gpurun_out/${T}_p2 $R/gpurun_out/${T}_p3 $R/gpurun_out/${T}_summary.json
