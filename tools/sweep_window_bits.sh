#!/bin/bash
# Window-size sweep (strong mode, one rank): ms per multi-exponentiation with three in flight and with one, per forced window size.
# usage (on the GPU box): bash tools/sweep_window_bits.sh "g1 23 16 18 19 20" "g2 24 16 20" ...
cd ${GRAFT_REPO_ROOT:-.}
for spec in "$@"; do
  set -- $spec; grp=$1; ln=$2; shift 2
  for c in "$@"; do
    timeout -k 10 200 python bench.py --scaling strong --total-log-n $ln --group $grp --steps 3 --warmup 1 --no-cpu-baseline --window-bits $c > gpurun_out/sw.json 2>gpurun_out/sw.err || { echo "FAIL $grp $ln $c"; tail -3 gpurun_out/sw.err; continue; }
    python3 -c "
import json; j=json.load(open('gpurun_out/sw.json')); print('$grp 2^$ln c=$c', round(j['ms_per_step'],2), 'ms/step, one in flight', round(j['latency_ms_one_in_flight'],2))"
  done
done
