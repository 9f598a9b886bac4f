#!/bin/bash
# kernel trace of tools/prove_profile.py under the options in $OPTS -> gpurun_out/<tag>_trace; prints the chain summary
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-t}
cd /tmp; export TMPDIR=/tmp
REPS=8 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${T}_trace -o runc -- python3 $R/tools/prove_profile.py > $R/gpurun_out/${T}_trace.txt 2>&1 || exit 1
grep "ms per" $R/gpurun_out/${T}_trace.txt
python3 $R/tools/prove_chain.py $R/gpurun_out/${T}_trace/runc_kernel_trace.csv 4 | tail -n 22
