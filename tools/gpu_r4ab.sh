#!/bin/bash
# round 4: kernel stats of the batch prover (K = 32 statements of 2^16 constraints per call, 10 calls)
set -e
R=$PWD
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
K=32 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4ab_prof -o runc -- python3 $R/tools/batch_prove_profile.py > $R/gpurun_out/r4ab.log 2>&1
tail -2 $R/gpurun_out/r4ab.log
