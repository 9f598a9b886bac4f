#!/bin/bash
# round 4: a candidate build of the library (VSP_LIB_PATH) through the self-check probe, the bisect cases, a quick bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
T=$1; export VSP_LIB_PATH=$R/vote_saver_protocol_amd/libvsp_hip_$T.so
python tools/gpu_r4b.py 2>&1 | tail -n 2
MODE=pairs python tools/gpu_r4c.py 2>&1 | tail -n 2
python bench.py --no-cpu-baseline --no-config5 --no-diag-clock > gpurun_out/r4e_${T}_bench.json 2> gpurun_out/r4e_${T}_bench.err || { tail -n 5 gpurun_out/r4e_${T}_bench.err; exit 5; }
python - <<PY
import json
j = json.load(open("gpurun_out/r4e_${T}_bench.json"))
e = j["extras"]
print("$T", "ms_per_step %.3f" % j["ms_per_step"], "latency %.3f" % j["latency_ms_one_in_flight"], "accum %.3f" % j["roofline"]["avg_launch_ms"], "ntt %.3f / %.3f" % (e["ntt_2p22_ms"], e["ntt_2p22_inverse_ms"]),
      "g2 2^18 %.3f" % e["g2_msm_2p18_ms"], "prove 2^20 %.3f packed %.3f dense %.2f" % (e["prove_2p20_ms"], e["prove_2p20_packed_witness_ms"], e["prove_2p20_dense_witness_ms"]),
      "prove 2^16 %.3f" % e.get("prove_2p16_ms", -1), "secondary %.1f" % j["secondary"]["value"], "verified", j["verified_bit_exact"])
PY
