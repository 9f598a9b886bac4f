#!/bin/bash
# One GPU-box job of the measure loop (scratch helper): tests, proof timing, MSM sizes, optional profiles, bench.
#   bash tools/gpu_job.sh <tag> [full|quick] [serial] [g2]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-x}; MODE=${2:-quick}
cd $R
if [ "$MODE" = full ]; then TESTS="tests"; else TESTS="tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_errors.py"; fi
python -m pytest $TESTS -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1 || { tail -n 30 gpurun_out/${T}_tests.log; exit 1; }
tail -n 2 gpurun_out/${T}_tests.log
python tools/prove_profile.py > gpurun_out/${T}_prove_time.txt 2>&1 || exit 2
PRE=0 python tools/prove_profile.py >> gpurun_out/${T}_prove_time.txt 2>&1 || exit 2
grep "ms per" gpurun_out/${T}_prove_time.txt
python tools/msm_sizes.py > gpurun_out/${T}_msm_sizes.txt 2>&1 || exit 6
cd /tmp; export TMPDIR=/tmp
if [[ " $* " == *" serial "* ]]; then
AMD_SERIALIZE_KERNEL=3 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prove_serial -o runc -- python3 $R/tools/prove_profile.py > $R/gpurun_out/${T}_prove_serial.txt 2>&1 || exit 3
fi
if [[ " $* " == *" g2 "* ]]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_g2 -o runc -- python3 $R/tools/g2_msm_profile.py > $R/gpurun_out/${T}_g2.txt 2>&1 || exit 4
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_nopipe -o runc -- python3 $R/bench.py --no-pipeline --no-extras --no-prove --no-cpu-baseline > $R/gpurun_out/${T}_nopipe.json 2> $R/gpurun_out/${T}_nopipe.err || exit 7
cd $R
python bench.py --no-cpu-baseline --no-config5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || exit 5
python tools/show_bench.py gpurun_out/${T}_bench.json 2>/dev/null | head -40
