#!/bin/bash
# round 4: the 29-bit product with two register maps (vsp_mm29 / vsp_mm29q) -- transform tests, then NTT and witness_map timings of
# variant A (plain asm statements: the compiler orders the four products of a group) and variant B (volatile: source order)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_domain.py tests/test_gpu_field.py tests/test_gpu_prover.py -m gpu -x -q > gpurun_out/r4w_tests.log 2>&1 || { tail -30 gpurun_out/r4w_tests.log; exit 1; }
tail -1 gpurun_out/r4w_tests.log
for v in a b a b; do
  if [ $v = b ]; then export VSP_LIB_PATH=$PWD/vote_saver_protocol_amd/libvsp_hip_b.so; else unset VSP_LIB_PATH; fi
  echo "variant $v"
  timeout -k 10 300 python tools/witness_map_time.py 2>&1 | tail -4
  timeout -k 10 300 python tools/ntt_time.py 2>&1 | tail -3
done
