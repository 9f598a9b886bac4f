"""Where the exchange's time goes at N = 1 through RCCL (diagnostic): run under torch.distributed.run with one rank;
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 tools/exchange_timing.py
Times, per step of ShardedMsm.run with four in flight: the host time inside begin() after the record is ready (copy + all_gather call),
inside end() (wait + read-back + fold), and the step time with the RCCL exchange against the local one."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, torch.distributed as dist
import vote_saver_protocol_amd as v
from vote_saver_protocol_amd.sharded import ShardedMsm, TorchExchange, LocalExchange
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
ctx = v.Context(0)
n = 1 << 20
rng = np.random.default_rng(1)
ks = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ss = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
d_k = ctx.to_device(ks); d_s = ctx.to_device(ss)
d_b = v.fixed_base_mul(ctx, d_k, n, 1)
B = ctx.bases_from_device(d_b, n, 1)
class Timed(TorchExchange):
    tb = te = 0.0
    def begin(self, bases, slot, buf):
        torch_, dist_, g = self.torch, self.dist, bases.group
        send, recv = self.send[g][buf & 1], self.recv[g][buf & 1]
        bases.msm_finish_jacobian_device(slot, send.data_ptr(), self.stream.cuda_stream)
        t0 = time.perf_counter()
        with torch_.cuda.stream(self.stream):
            work = dist_.all_gather_into_tensor(recv, send, group=self.pg, async_op=True)
        Timed.tb += time.perf_counter() - t0
        return work, recv, g
    def end(self, h):
        t0 = time.perf_counter(); r = super().end(h); Timed.te += time.perf_counter() - t0; return r
for name, ex in (("rccl", Timed(ctx, dev)), ("local", LocalExchange(ctx))):
    job = ShardedMsm(B, ex)
    job.run(d_s, 8, 4); torch.cuda.synchronize()
    Timed.tb = Timed.te = 0.0
    K = 30
    t0 = time.perf_counter(); job.run(d_s, K, 4); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"{name}: {dt * 1e3:.3f} ms per step; host time in all_gather call {Timed.tb / K * 1e3:.3f} ms, in end() {Timed.te / K * 1e3:.3f} ms per step")
dist.destroy_process_group()
