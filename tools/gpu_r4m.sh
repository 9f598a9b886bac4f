#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python tools/batch_prove_contexts.py 2>&1 | tail -n 1
cd /tmp; export TMPDIR=/tmp
LOG_M=16 REPS=8 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4m_trace -o runc -- python3 $R/tools/prove_profile.py > $R/gpurun_out/r4m_trace.txt 2>&1
grep "ms per" $R/gpurun_out/r4m_trace.txt
python3 $R/tools/prove_chain.py $R/gpurun_out/r4m_trace/runc_kernel_trace.csv 4 | tail -n 60
