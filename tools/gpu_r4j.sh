#!/bin/bash
# round 4: the portable-product build through the parity tests; NTT stagger experiment; batch validation tests; PMC + SQ counters
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python tools/ntt_stagger.py > gpurun_out/r4j_stagger.txt 2>&1; tail -n 9 gpurun_out/r4j_stagger.txt
python -m pytest tests/test_gpu_errors.py -x -q -k "batch" 2>&1 | tail -n 3
VSP_LIB_PATH=$R/vote_saver_protocol_amd/libvsp_hip_portable.so timeout -k 10 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_ntt.py tests/test_gpu_domain.py tests/test_gpu_prover.py "tests/test_gpu_msm.py" -x -q -k "not 2p2 and not large and not 2p20 and not shard" > gpurun_out/r4j_portable_tests.log 2>&1; tail -n 4 gpurun_out/r4j_portable_tests.log
bash tools/gpu_pmc.sh r4pmc && python3 tools/pmc_summary.py gpurun_out/r4pmc_fetch gpurun_out/r4pmc_write gpurun_out/r4_pmc_hbm_traffic.json "bench.py --steps 3 --warmup 1 --no-pipeline --no-extras --no-prove --no-cpu-baseline --no-config5 --no-diag-clock" k_accum_G1_2p20_plain 2>&1 | tail -n 2
bash tools/gpu_sq.sh r4sq; ls gpurun_out/r4sq_summary.json
