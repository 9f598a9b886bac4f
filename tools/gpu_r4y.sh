#!/bin/bash
# round 4: same-box A/B of the transforms: the tree's library against libvsp_hip_pair.so (256 threads, two groups per thread interleaved)
set -e
mkdir -p gpurun_out
VSP_LIB_PATH=$PWD/vote_saver_protocol_amd/libvsp_hip_pair.so timeout -k 10 600 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_domain.py -m gpu -x -q > gpurun_out/r4y_tests.log 2>&1 || { tail -30 gpurun_out/r4y_tests.log; exit 1; }
tail -1 gpurun_out/r4y_tests.log
for v in base pair base pair base pair; do
  if [ $v = pair ]; then export VSP_LIB_PATH=$PWD/vote_saver_protocol_amd/libvsp_hip_pair.so; else unset VSP_LIB_PATH; fi
  echo "variant $v: $(timeout -k 10 200 python tools/ntt_time.py 2>&1 | tail -1); $(LOG_N=20 timeout -k 10 200 python tools/ntt_time.py 2>&1 | tail -1); $(LOG_N=16,20 R=60 timeout -k 10 200 python tools/witness_map_time.py 2>&1 | tail -2 | tr '\n' ' ')"
done
