#!/bin/bash
# A/B of the prefetching k_dimsum_mixed loop (option msm_dimsum_prefetch) on one box; why the portable build aborts
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for pf in 0 1 0 1; do VSP_OPTS=msm_dimsum_prefetch=$pf python bench.py --no-pipeline --no-extras --no-prove --no-cpu-baseline --no-config5 --no-diag-clock --steps 30 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('prefetch $pf: one in flight %.3f ms, accum %.3f' % (j['ms_per_step'], j['roofline']['avg_launch_ms']))"; done
cd /tmp; export TMPDIR=/tmp
for pf in 0 1; do VSP_OPTS=msm_dimsum_prefetch=$pf rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4k_pf$pf -o runc -- python3 $R/bench.py --no-pipeline --no-extras --no-prove --no-cpu-baseline --no-config5 --no-diag-clock --steps 30 > /dev/null 2>&1; grep "k_dimsum_mixed<vsp::Fp28" $R/gpurun_out/r4k_pf$pf/runc_kernel_stats.csv | cut -c1-60,150-260; done
cd $R
VSP_LIB_PATH=$R/vote_saver_protocol_amd/libvsp_hip_portable.so timeout -k 10 300 python -c "
import vote_saver_protocol_amd as v, numpy as np
ctx = v.Context(0); print('portable context ok')
dom = v.EvaluationDomain(ctx, 1 << 10); a = np.random.default_rng(1).integers(0, 1 << 62, size=(1 << 10, 4), dtype=np.uint64); print('fft', dom.fft(a)[0])
" 2>&1 | tail -n 8
