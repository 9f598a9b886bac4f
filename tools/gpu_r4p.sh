#!/bin/bash
# A/B: merges with one-wave workgroups (-DVSP_MERGE_NT=64) against the default four-wave ones
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for lib in libvsp_hip.so libvsp_hip_m64.so libvsp_hip.so libvsp_hip_m64.so; do
  export VSP_LIB_PATH=$R/vote_saver_protocol_amd/$lib
  echo "$lib: $(LOG_M=16 REPS=40 python tools/prove_profile.py | tail -1)  2^16 plain: $(LOG_M=16 REPS=40 PRE=0 python tools/prove_profile.py | tail -1)  2^20: $(LOG_M=20 REPS=15 python tools/prove_profile.py | tail -1)  batch: $(KMAX=16 REPS=4 python tools/batch_prove_time.py | tail -1 | sed 's/.*K=8/K=8/')"
done
VSP_LIB_PATH=$R/vote_saver_protocol_amd/libvsp_hip_m64.so python -m pytest tests/test_gpu_msm.py -x -q -k "skewed or equal_partial or duplicate or one_point or batch or variant" 2>&1 | tail -n 2
