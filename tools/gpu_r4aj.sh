#!/bin/bash
# round 4: batches over tables of window multiples -- tests, then proofs per second at 2^16 with a table key
set -e
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_errors.py -m gpu -x -q -k "batch or batches" > gpurun_out/r4aj_tests.log 2>&1 || { tail -40 gpurun_out/r4aj_tests.log; exit 1; }
tail -1 gpurun_out/r4aj_tests.log
PRE=1 KS=32 timeout -k 10 300 python tools/batch_prove_contexts.py 2>&1 | tail -1
KS=32 timeout -k 10 300 python tools/batch_prove_contexts.py 2>&1 | tail -1
