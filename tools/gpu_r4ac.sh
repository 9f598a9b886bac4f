#!/bin/bash
# round 4: the bit-decomposed last reduction step (k_dimbits) for batches -- batch tests, kernel stats of the batch prover, proofs per second
set -e
R=$PWD
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_errors.py -m gpu -x -q -k "batch" > gpurun_out/r4ac_tests.log 2>&1 || { tail -30 gpurun_out/r4ac_tests.log; exit 1; }
tail -1 gpurun_out/r4ac_tests.log
timeout -k 10 120 python tools/fuzz_batch_msm.py 60 5150 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
K=32 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4ac_prof -o runc -- python3 $R/tools/batch_prove_profile.py > $R/gpurun_out/r4ac.log 2>&1
tail -2 $R/gpurun_out/r4ac.log
cd $R
timeout -k 10 400 python tools/batch_prove_contexts.py 2>&1 | tail -1

