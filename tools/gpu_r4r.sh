#!/bin/bash
# round 4: soak of the three fuzzers on the final build (new seeds), progress lines every minute
set -e
mkdir -p gpurun_out
timeout -k 10 330 python tools/fuzz_prove.py 300 4242 > gpurun_out/r4r_fuzz_prove.log 2>&1 || { tail -5 gpurun_out/r4r_fuzz_prove.log; exit 1; }
tail -1 gpurun_out/r4r_fuzz_prove.log
timeout -k 10 330 python tools/fuzz_msm.py 300 4243 > gpurun_out/r4r_fuzz_msm.log 2>&1 || { tail -5 gpurun_out/r4r_fuzz_msm.log; exit 1; }
tail -1 gpurun_out/r4r_fuzz_msm.log
timeout -k 10 270 python tools/fuzz_batch_msm.py 240 4244 > gpurun_out/r4r_fuzz_batch.log 2>&1 || { tail -5 gpurun_out/r4r_fuzz_batch.log; exit 1; }
tail -1 gpurun_out/r4r_fuzz_batch.log
