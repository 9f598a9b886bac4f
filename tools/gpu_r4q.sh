#!/bin/bash
# round 4: the batch prover in two halves -- its tests, then proofs per second from one host thread
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_prover.py tests/test_gpu_errors.py -m gpu -x -q > gpurun_out/r4q_tests.log 2>&1 || { tail -30 gpurun_out/r4q_tests.log; exit 1; }
tail -2 gpurun_out/r4q_tests.log
timeout -k 10 400 python tools/batch_prove_contexts.py > gpurun_out/r4q_contexts.log 2>&1
cat gpurun_out/r4q_contexts.log
