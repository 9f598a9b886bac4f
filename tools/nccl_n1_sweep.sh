#!/bin/bash
cd $GRAFT_REPO_ROOT
port=29520
for args in "--pipeline-depth 3" "--pipeline-depth 4" "--pipeline-depth 4 --hw-queues 0" "--pipeline-depth 2" "--no-pipeline"; do
  port=$((port+1))
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-prove --no-extras --no-diag-clock --no-config5 $args > gpurun_out/n1.json 2> gpurun_out/n1.err || { echo "FAIL $args"; tail -3 gpurun_out/n1.err; continue; }
  python3 -c "
import json; j=json.load(open('gpurun_out/n1.json')); print('nccl N=1 $args:', round(j['ms_per_step'],3), 'ms/step', j['exchange']['backend'])"
done
