#!/bin/bash
# round 4: kernel stats of single proofs at 2^20 (plain key) and 2^16
set -e
R=$PWD
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
LOG_M=20 PRE=0 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4ae_p20 -o runc -- python3 $R/tools/prove_profile.py > $R/gpurun_out/r4ae_p20.log 2>&1
grep "per proof" $R/gpurun_out/r4ae_p20.log
LOG_M=16 PRE=0 REPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4ae_p16 -o runc -- python3 $R/tools/prove_profile.py > $R/gpurun_out/r4ae_p16.log 2>&1
grep "per proof" $R/gpurun_out/r4ae_p16.log
