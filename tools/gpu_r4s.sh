#!/bin/bash
# round 4: waves of the bucket reduction's first step (option msm_dimsum_maxw): one MSM in flight and four, headline only
set -e
timeout -k 10 500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r4s_tests.log 2>&1 || { tail -30 gpurun_out/r4s_tests.log; exit 1; }
tail -1 gpurun_out/r4s_tests.log
mkdir -p gpurun_out
for mw in 1024 1536 2048 3072; do
  VSP_OPTS="msm_dimsum_maxw=$mw" timeout -k 10 200 python bench.py --no-extras --no-prove --no-config5 --no-cpu-baseline --no-diag-clock --steps 40 > gpurun_out/r4s_pipe_$mw.json 2> gpurun_out/r4s_pipe_$mw.err
  VSP_OPTS="msm_dimsum_maxw=$mw" timeout -k 10 200 python bench.py --no-extras --no-prove --no-config5 --no-cpu-baseline --no-diag-clock --no-pipeline --steps 40 > gpurun_out/r4s_nopipe_$mw.json 2> gpurun_out/r4s_nopipe_$mw.err
  python3 - $mw <<'PY'
import json, sys
mw = sys.argv[1]
a = json.load(open("gpurun_out/r4s_pipe_%s.json" % mw)); b = json.load(open("gpurun_out/r4s_nopipe_%s.json" % mw))
print("maxw %s: pipelined %.3f ms/step, one in flight %.3f ms/step, verified %s %s" % (mw, a["ms_per_step"], b["ms_per_step"], a.get("verified"), b.get("verified")), flush=True)
PY
done
