#!/bin/bash
set -e
for v in base c new base c new; do
  if [ $v = base ]; then export VSP_LIB_PATH=$PWD/vote_saver_protocol_amd/libvsp_hip_base.so; elif [ $v = c ]; then export VSP_LIB_PATH=$PWD/vote_saver_protocol_amd/libvsp_hip_c.so; else unset VSP_LIB_PATH; fi
  echo "variant $v: $(timeout -k 10 200 python tools/ntt_time.py 2>&1 | tail -1); $(LOG_N=20 timeout -k 10 200 python tools/ntt_time.py 2>&1 | tail -1)"
done
