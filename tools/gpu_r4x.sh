#!/bin/bash
# round 4: same-box A/B of the transforms: the previous commit's library (base) against the two-register-map product (new)
set -e
mkdir -p gpurun_out
for v in base new base new base new; do
  if [ $v = base ]; then export VSP_LIB_PATH=$PWD/vote_saver_protocol_amd/libvsp_hip_base.so; else unset VSP_LIB_PATH; fi
  echo "variant $v: $(timeout -k 10 200 python tools/ntt_time.py 2>&1 | tail -1); $(LOG_N=20 timeout -k 10 200 python tools/ntt_time.py 2>&1 | tail -1); $(LOG_N=20 R=60 timeout -k 10 200 python tools/witness_map_time.py 2>&1 | tail -1)"
done
