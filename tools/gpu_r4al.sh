#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 260 python tools/fuzz_batch_msm.py 220 8101 > gpurun_out/r4al_fuzz_batch.log 2>&1 || { tail -5 gpurun_out/r4al_fuzz_batch.log; exit 1; }
tail -1 gpurun_out/r4al_fuzz_batch.log
bash tools/gpu_final.sh r4final5
