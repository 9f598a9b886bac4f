#!/bin/bash
# round 4: soak of the three fuzzers on the LAST build (29-bit product with two register maps, lane-pair product without the a-side selection)
set -e
mkdir -p gpurun_out
timeout -k 10 340 python tools/fuzz_prove.py 300 9001 > gpurun_out/r4z_fuzz_prove.log 2>&1 || { tail -5 gpurun_out/r4z_fuzz_prove.log; exit 1; }
tail -1 gpurun_out/r4z_fuzz_prove.log
MAX_NC=300000 timeout -k 10 280 python tools/fuzz_prove.py 240 9002 > gpurun_out/r4z_fuzz_prove_large.log 2>&1 || { tail -5 gpurun_out/r4z_fuzz_prove_large.log; exit 1; }
tail -1 gpurun_out/r4z_fuzz_prove_large.log
timeout -k 10 340 python tools/fuzz_msm.py 300 9003 > gpurun_out/r4z_fuzz_msm.log 2>&1 || { tail -5 gpurun_out/r4z_fuzz_msm.log; exit 1; }
tail -1 gpurun_out/r4z_fuzz_msm.log | cut -c1-60
timeout -k 10 220 python tools/fuzz_batch_msm.py 180 9004 > gpurun_out/r4z_fuzz_batch.log 2>&1 || { tail -5 gpurun_out/r4z_fuzz_batch.log; exit 1; }
tail -1 gpurun_out/r4z_fuzz_batch.log
