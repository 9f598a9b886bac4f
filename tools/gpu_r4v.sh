#!/bin/bash
# round 4: soak at larger sizes (constraint systems up to 300 000 constraints; the MSM fuzzer's full size range), new seeds
set -e
mkdir -p gpurun_out
MAX_NC=300000 timeout -k 10 460 python tools/fuzz_prove.py 420 77 > gpurun_out/r4v_fuzz_prove.log 2>&1 || { tail -5 gpurun_out/r4v_fuzz_prove.log; exit 1; }
tail -1 gpurun_out/r4v_fuzz_prove.log
timeout -k 10 460 python tools/fuzz_msm.py 420 78 > gpurun_out/r4v_fuzz_msm.log 2>&1 || { tail -5 gpurun_out/r4v_fuzz_msm.log; exit 1; }
tail -1 gpurun_out/r4v_fuzz_msm.log | cut -c1-80
