#!/bin/bash
# long randomised runs on the round's last build
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 700 python tools/fuzz_msm.py 420 1 > gpurun_out/r4o_fuzz_msm_seed1.txt 2>&1; tail -n 1 gpurun_out/r4o_fuzz_msm_seed1.txt | cut -c1-120
timeout -k 10 400 python tools/fuzz_batch_msm.py 240 1 > gpurun_out/r4o_fuzz_batch.txt 2>&1; tail -n 1 gpurun_out/r4o_fuzz_batch.txt
MAX_NC=20000 timeout -k 10 500 python tools/fuzz_prove.py 300 1 > gpurun_out/r4o_fuzz_prove.txt 2>&1; tail -n 1 gpurun_out/r4o_fuzz_prove.txt
