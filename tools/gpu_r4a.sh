#!/bin/bash
# round 4 GPU job: full GPU suite, the prefetch probe (generic and forced 28-bit), a fuzz run, the bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
T=${1:-r4a}
{ python tools/dimsum_prefetch_probe.py 2>&1 | tail -n 1; OPTS=msm_fp28=0 python tools/dimsum_prefetch_probe.py 2>&1 | tail -n 1; python tools/gpu_r4b.py 2>&1 | tail -n 2; MODE=pairs python tools/gpu_r4c.py 2>&1 | tail -n 2; } > gpurun_out/${T}_probe.txt 2>&1
cat gpurun_out/${T}_probe.txt
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_tests.log 2>&1 || { tail -n 30 gpurun_out/${T}_tests.log; exit 1; }
tail -n 2 gpurun_out/${T}_tests.log
timeout -k 10 400 python tools/fuzz_msm.py ${FUZZ_S:-240} 41 > gpurun_out/${T}_fuzz.txt 2>&1; tail -n 2 gpurun_out/${T}_fuzz.txt
python bench.py --no-cpu-baseline --no-config5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || { tail -n 5 gpurun_out/${T}_bench.err; exit 5; }
python tools/show_bench.py gpurun_out/${T}_bench.json 2>/dev/null | head -70
