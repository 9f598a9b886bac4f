#!/bin/bash
# round 4: fixed-base tables for the multiples of delta, 4-bit windows for s*A and r*B1 -- prover tests, host phases of a batch, proofs per second
set -e
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_prover.py tests/test_saver.py tests/test_gpu_shims.py -m gpu -x -q > gpurun_out/r4ah_tests.log 2>&1 || { tail -30 gpurun_out/r4ah_tests.log; exit 1; }
tail -1 gpurun_out/r4ah_tests.log
timeout -k 10 200 python tools/fuzz_prove.py 90 424242 2>&1 | tail -1
K=32 REPS=10 timeout -k 10 200 python tools/batch_prove_profile.py 2>&1 | tail -2
VSP_OPTS_UNUSED=1 LOG_M=16,20 timeout -k 10 300 python tools/prove_phases.py 2>&1 | tail -4
KS=32 timeout -k 10 300 python tools/batch_prove_contexts.py 2>&1 | tail -1
