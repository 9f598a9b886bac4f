#!/bin/bash
# the portable-product build (make portable) through the parity tests, file by file
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VSP_LIB_PATH=$R/vote_saver_protocol_amd/libvsp_hip_portable.so
for t in tests/test_gpu_field.py tests/test_gpu_ntt.py tests/test_gpu_domain.py "tests/test_gpu_msm.py -k golden or degenerate or c_oracle_random or skewed" tests/test_gpu_prover.py; do
  echo "== $t"; timeout -k 10 600 python -m pytest $t -x -q 2>&1 | tail -n 4
done
