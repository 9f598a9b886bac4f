"""Sharded multi-scalar multiplication: one MSM split by contiguous point chunk over the GPUs of a node.

    sum_i s_i P_i  =  sum_g  sum_{i in chunk g} s_i P_i

The call this serves is the one prove under bin/cli/include/nil/vote_saver/common.hpp:1132-1135: its multi-exponentiations are
independent sums over the proving key's queries, so rank g keeps bases [g n / N, (g + 1) n / N) resident (`shard_bounds`), runs the
ordinary single-GPU pipeline over its chunk and contributes ONE fixed-size record -- the Jacobian partial sum, 144 bytes (G1) or
288 bytes (G2).  The only exchange step is an all-gather of those records (RCCL over xGMI when the process group's backend is
"nccl"), after which every rank folds the N records locally: elliptic-curve addition is not an RCCL reduction operator, and N <= 8
additions cost microseconds.  No other collective touches the data path; the NTT stays on one GPU (BASELINE.json north_star).

Where the record lives.  The last step of a multi-exponentiation is a chain of c * W dependent doublings that the host runs ~50x
faster than a GPU lane, so the record is completed on the host.  With a device-side collective
(`TorchExchange`, backend nccl) `vsp_msm_finish_jacobian_device` writes it to a pinned ring entry and queues an asynchronous copy
into the send buffer on the exchange stream; the all-gather follows on that stream, and `vsp_fold_jacobian_device` copies the N
records back behind it and folds.  The host thread never blocks on a copy, and the exchange of step k completes while the
multi-exponentiation of step k + 1 is awaited (`ShardedMsm.run`).  With a host-side collective (backend gloo: CPU tests, rehearsals
with several ranks on one GPU) the record never leaves the host.

Nothing here computes field or curve arithmetic: records come from the C ABI (`Bases.msm_launch` / `msm_finish_jacobian[_device]`)
and are folded by it (`vsp_fold_jacobian[_device]`).  torch is imported only by `TorchExchange`.
"""
import ctypes as C

import numpy as np

from . import _lib

RECORD_WORDS = {1: 18, 2: 36}     # uint64 words of a Jacobian record: X, Y, Z canonical (G1 144 bytes, G2 288 bytes)
AFFINE_WORDS = {1: 12, 2: 24}


def shard_bounds(total, world, rank):
    """contiguous point chunk of rank `rank` of `world`: [lo, hi).  Chunks cover [0, total) exactly once and differ by at most one point."""
    return rank * total // world, (rank + 1) * total // world


def _fold_host(ctx, records, group):
    records = np.ascontiguousarray(records, dtype=np.uint64).reshape(-1, RECORD_WORDS[group])
    out = np.zeros(AFFINE_WORDS[group], np.uint64)
    inf = C.c_int(0)
    lib = _lib.load()
    rc = lib.vsp_fold_jacobian(ctx.h if ctx is not None else None, group, records.ctypes.data_as(C.c_void_p), records.shape[0],
                               out.ctypes.data_as(C.c_void_p), C.byref(inf))
    if rc != 0:
        raise RuntimeError(f"vsp_fold_jacobian failed ({rc})")
    return out


class LocalExchange:
    """world size 1: the rank's own record is the whole sum (same code path as the collective ones, no collective)."""
    world, rank, device_records, backend = 1, 0, False, "none"

    def __init__(self, ctx=None):
        self.ctx = ctx

    def begin(self, bases, slot, buf):
        return bases.msm_finish_jacobian(slot), bases.group

    def end(self, handle):
        rec, group = handle
        return _fold_host(self.ctx, rec, group)

    def ranks_seen(self):
        return 1


class TorchExchange:
    """All-gather of the ranks' records over an initialised torch.distributed process group.

    backend "nccl" (= RCCL on ROCm): records travel device-to-device over xGMI; the send and receive buffers (two of each per
    group, so step k's exchange overlaps step k + 1) live in device memory and the collective runs on a stream of its own.
    backend "gloo": host buffers (CPU tests; rehearsals where several ranks share one GPU, which RCCL refuses)."""

    def __init__(self, ctx, device=None, process_group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.ctx, self.pg = torch, dist, ctx, process_group
        self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        self.backend = dist.get_backend(process_group)
        self.device_records = self.backend == "nccl"
        if self.device_records:
            if ctx is None or device is None:
                raise ValueError("TorchExchange: a device-side collective needs the library context and the rank's torch device")
            self.device = device
            self.stream = torch.cuda.Stream(device=device)
            self.send = {g: [torch.zeros(w, dtype=torch.int64, device=device) for _ in range(2)] for g, w in RECORD_WORDS.items()}
            self.recv = {g: [torch.zeros(w * self.world, dtype=torch.int64, device=device) for _ in range(2)] for g, w in RECORD_WORDS.items()}

    def begin(self, bases, slot, buf):
        """wait for the multi-exponentiation on `slot`, put its record where the collective reads it, start the all-gather"""
        torch, dist, g = self.torch, self.dist, bases.group
        if self.device_records:
            send, recv = self.send[g][buf & 1], self.recv[g][buf & 1]
            bases.msm_finish_jacobian_device(slot, send.data_ptr(), self.stream.cuda_stream)      # pinned ring entry -> async copy on self.stream
            with torch.cuda.stream(self.stream):
                work = dist.all_gather_into_tensor(recv, send, group=self.pg, async_op=True)
            return work, recv, g
        rec = np.ascontiguousarray(bases.msm_finish_jacobian(slot), dtype=np.uint64)
        mine = torch.from_numpy(rec.view(np.int64).copy())
        parts = [torch.zeros(RECORD_WORDS[g], dtype=torch.int64) for _ in range(self.world)]
        work = dist.all_gather(parts, mine, group=self.pg, async_op=True)
        return work, parts, g

    def end(self, handle):
        """wait for the all-gather, fold the N records locally -> the affine sum (on every rank)"""
        work, recv, g = handle
        if self.device_records:
            with self.torch.cuda.stream(self.stream):
                work.wait()                                     # self.stream now follows the collective
            out = np.zeros(AFFINE_WORDS[g], np.uint64)
            inf = C.c_int(0)
            self.ctx.check(self.ctx.lib.vsp_fold_jacobian_device(self.ctx.h, g, C.c_void_p(recv.data_ptr()), self.world,
                                                                 C.c_void_p(self.stream.cuda_stream), out.ctypes.data_as(C.c_void_p), C.byref(inf)))
            return out
        work.wait()
        recs = np.stack([p.numpy().view(np.uint64) for p in recv])
        return _fold_host(self.ctx, recs, g)

    def ranks_seen(self):
        """the number of ranks the collective library itself reaches: an all-reduce of ones over the process group"""
        t = self.torch.ones(1, dtype=self.torch.int64, device=self.device if self.device_records else "cpu")
        self.dist.all_reduce(t, group=self.pg)
        return int(t.item())


class ShardedMsm:
    """One multi-exponentiation over bases split by contiguous chunk over the ranks of `exchange`.

    bases: this rank's resident chunk (`api.Bases`, or anything with .group, msm_launch(slot, d_scalars),
    msm_finish_jacobian(slot) and -- for device-side exchanges -- msm_finish_jacobian_device(slot, dptr, stream)).
    d_scalars: this rank's slice of the scalar vector, in device memory."""

    SLOTS = (1, 2, 4, 5)       # work slots with streams of their own (slot 0 shares the context's stream, 3 is left to the caller)

    def __init__(self, bases, exchange):
        self.bases, self.exchange, self.group = bases, exchange, bases.group

    def msm(self, d_scalars, slot=0):
        """blocking: this rank's partial sum, the exchange, the fold -> the affine result of the WHOLE problem on every rank"""
        self.bases.msm_launch(slot, d_scalars)
        return self.exchange.end(self.exchange.begin(self.bases, slot, 0))

    def run(self, d_scalars, steps, depth=3):
        """`steps` multi-exponentiations back to back with `depth` in flight over the work slots; the exchange of step k completes
        while step k + 1's multi-exponentiation is awaited.  Returns the folded result of the last step."""
        sl = self.SLOTS[:min(len(self.SLOTS), depth)] if depth > 1 else (0,)       # one in flight: the context's own stream
        D = len(sl)
        x, bases = self.exchange, self.bases
        res = pending = None
        for k in range(min(D - 1, steps)):
            bases.msm_launch(sl[k % D], d_scalars)
        for k in range(steps):
            if D == 1:
                bases.msm_launch(sl[0], d_scalars)
            elif k + D - 1 < steps:
                bases.msm_launch(sl[(k + D - 1) % D], d_scalars)
            h = x.begin(bases, sl[k % D], k & 1)
            if pending is not None:
                res = x.end(pending)
            pending = h
        if pending is not None:
            res = x.end(pending)
        return res
