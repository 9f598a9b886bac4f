#!/bin/bash
# End-of-round measurement on the GPU box: full GPU suite, the default bench line, and the rocprofv3 kernel stats of (a) the default bench
# command and (b) the one-in-flight configuration the roofline is quoted on.  Results under gpurun_out/<tag>_*; copy into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-final}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1 || { tail -n 30 gpurun_out/${T}_tests.log; exit 1; }
tail -n 2 gpurun_out/${T}_tests.log
python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err || exit 2
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_default -o runc -- python3 $R/bench.py --no-cpu-baseline --no-config5 > $R/gpurun_out/${T}_bench_profiled.json 2> $R/gpurun_out/${T}_bench_profiled.err || exit 3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_nopipe -o runc -- python3 $R/bench.py --no-pipeline --no-extras --no-prove --no-cpu-baseline > $R/gpurun_out/${T}_bench_nopipe.json 2> $R/gpurun_out/${T}_bench_nopipe.err || exit 4
cd $R
python tools/show_bench.py gpurun_out/${T}_bench_default.json gpurun_out/${T}_bench_profiled.json gpurun_out/${T}_bench_nopipe.json
