#!/bin/bash
# HBM-side traffic of the accumulation kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (no trace domains beside --pmc
# except --kernel-trace), one multi-exponentiation in flight.  Summarise with tools/pmc_summary.py.
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-pmc}
cd /tmp; export TMPDIR=/tmp
CMD="bench.py --steps 3 --warmup 1 --no-pipeline --no-extras --no-prove --no-cpu-baseline --no-config5 --no-diag-clock"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_fetch -- python3 $R/$CMD > $R/gpurun_out/${T}_fetch.json 2> $R/gpurun_out/${T}_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}_write -- python3 $R/$CMD > $R/gpurun_out/${T}_write.json 2> $R/gpurun_out/${T}_write.err || exit 2
ls $R/gpurun_out/${T}_fetch/* | head -3
