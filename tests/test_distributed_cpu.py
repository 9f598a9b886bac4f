"""CPU, gloo, world sizes 2, 4 and 8 (the node size the driver scales to): the sharded multi-exponentiation of the package (vote_saver_protocol_amd/sharded.py)
through its REAL exchange code -- `ShardedMsm.run` / `.msm`, `TorchExchange.begin` / `.end` over a torch.distributed process group,
the library's host fold (vsp_fold_jacobian; no GPU needed) -- with a stand-in for the per-rank multi-exponentiation: this container has
no GPU, so each rank's Jacobian partial sum (144 bytes G1, 288 bytes G2) comes from the oracle over the rank's contiguous chunk.  On
the GPU box the same records come from vsp_msm_launch / vsp_msm_finish_jacobian[_device] (tests/test_gpu_msm.py checks those against
the oracle, tests/test_gpu_sharded.py runs the module over the real thing)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _jacobian_record(aff12, scale):
    """affine canonical -> a Jacobian record with a non-trivial Z: (x z^2, y z^3, z)."""
    import bls12_381 as o
    x = sum(int(aff12[i]) << (64 * i) for i in range(6)); y = sum(int(aff12[6 + i]) << (64 * i) for i in range(6))
    if x == 0 and y == 0:
        return np.array(o.int_to_limbs(1, 6) + o.int_to_limbs(1, 6) + [0] * 6, dtype=np.uint64)
    z = scale % o.P
    return np.array(o.int_to_limbs(x * z * z % o.P, 6) + o.int_to_limbs(y * z * z * z % o.P, 6) + o.int_to_limbs(z, 6), dtype=np.uint64)


def _jacobian_record_g2(aff24, scale):
    """the 288-byte G2 record (X, Y, Z in Fp2 as c0, c1) with a non-trivial Z = (z0, z1)"""
    import bls12_381 as o
    F = o.Fp2Ops
    c = [sum(int(aff24[6 * k + i]) << (64 * i) for i in range(6)) for k in range(4)]
    x, y = (c[0], c[1]), (c[2], c[3])
    if x == (0, 0) and y == (0, 0):
        return np.array(o.int_to_limbs(1, 6) + [0] * 6 + o.int_to_limbs(1, 6) + [0] * 6 + [0] * 12, dtype=np.uint64)
    z = (scale % o.P, (scale * 7 + 3) % o.P)
    z2 = F.mul(z, z); z3 = F.mul(z2, z)
    X, Y = F.mul(x, z2), F.mul(y, z3)
    out = []
    for v2 in (X, Y, z):
        out += o.int_to_limbs(v2[0], 6) + o.int_to_limbs(v2[1], 6)
    return np.array(out, dtype=np.uint64)


class OracleShard:
    """stand-in for api.Bases over this rank's chunk: msm_launch / msm_finish_jacobian with the records made by the oracle"""

    def __init__(self, group, bases, rank):
        self.group, self.bases, self.rank, self.slots, self.launches = group, bases, rank, {}, 0

    def msm_launch(self, slot, scalars, n=None, first=0):
        assert slot not in self.slots, "slot launched twice without a finish"
        self.slots[slot] = scalars; self.launches += 1

    def msm_finish_jacobian(self, slot):
        import cref
        ss = self.slots.pop(slot)
        if self.group == 1:
            part = cref.msm_g1(self.bases, ss) if len(ss) else np.zeros(12, np.uint64)
            return _jacobian_record(part, 0x1234567 + self.rank + self.launches)
        part = cref.msm_g2(self.bases, ss) if len(ss) else np.zeros(24, np.uint64)
        return _jacobian_record_g2(part, 0x7654321 + self.rank + self.launches)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import cref
    import vote_saver_protocol_amd as v
    from conftest import rand_fr_array
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fails = []
    x = v.TorchExchange(None)                                   # host-side collective: no library context, no device
    if not (x.world == world and x.rank == rank and x.backend == "gloo" and not x.device_records and x.ranks_seen() == world):
        fails.append("exchange setup")
    n = 403                                                     # not a multiple of the world size: ragged chunks
    ks, ss = rand_fr_array(n, 1), rand_fr_array(n, 2)           # the same global problem on every rank
    bases = cref.g1_batch_mul_gen(ks)
    lo, hi = v.shard_bounds(n, world, rank)                     # contiguous point chunk of this rank
    job = v.ShardedMsm(OracleShard(1, bases[lo:hi], rank), x)
    full = cref.msm_g1(bases, ss)
    if not np.array_equal(job.msm(ss[lo:hi]), full):
        fails.append("G1 blocking")
    for depth, steps in ((1, 2), (3, 5), (4, 3)):               # the pipelined loop: every step's exchange folds to the same point
        if not np.array_equal(job.run(ss[lo:hi], steps, depth), full):
            fails.append(f"G1 run depth {depth}")
    # fewer points than ranks: some chunks are empty (their record is the point at infinity, Z = 0)
    lo3, hi3 = v.shard_bounds(3, world, rank)
    small = v.ShardedMsm(OracleShard(1, bases[lo3:hi3], rank), x)
    if not np.array_equal(small.msm(ss[lo3:hi3]), cref.msm_g1(bases[:3], ss[:3])):
        fails.append("G1 empty chunks")
    # partial sums that cancel: P - P over two chunks -> infinity (all-zero affine)
    if world == 2:
        pair = np.stack([bases[0], bases[0]])
        one, minus_one = np.array([[1, 0, 0, 0]], np.uint64), np.array([cref_r_minus_1()], np.uint64)
        canc = v.ShardedMsm(OracleShard(1, pair[rank:rank + 1], rank), x)
        if canc.msm(one if rank == 0 else minus_one).any():
            fails.append("G1 cancelling chunks")
    # the G2 half of BASELINE config 5: 288-byte records through the same exchange
    n2 = 61
    b2 = cref.g2_batch_mul_gen(ks[:n2])
    lo2, hi2 = v.shard_bounds(n2, world, rank)
    job2 = v.ShardedMsm(OracleShard(2, b2[lo2:hi2], rank), x)
    full2 = cref.msm_g2(b2, ss[:n2])
    if not np.array_equal(job2.msm(ss[lo2:hi2]), full2) or not np.array_equal(job2.run(ss[lo2:hi2], 3, 2), full2):
        fails.append("G2")
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if not fails else "FAIL: " + ", ".join(fails))
    dist.destroy_process_group()


def cref_r_minus_1():
    import bls12_381 as o
    return o.int_to_limbs(o.R - 1, 4)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_msm_exchange(tmp_path, world):
    port = 29500 + (os.getpid() % 1000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def test_local_exchange_is_the_same_code_path_without_a_collective():
    """world size 1 (bench.py --gpus 1 outside torch.distributed.run): LocalExchange through ShardedMsm.run, host fold"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cref
    import vote_saver_protocol_amd as v
    from conftest import rand_fr_array
    ks, ss = rand_fr_array(50, 3), rand_fr_array(50, 4)
    bases = cref.g1_batch_mul_gen(ks)
    job = v.ShardedMsm(OracleShard(1, bases, 0), v.LocalExchange(None))
    assert np.array_equal(job.run(ss, 4, 3), cref.msm_g1(bases, ss)) and np.array_equal(job.msm(ss), cref.msm_g1(bases, ss))


# ------------------------------------------------------------------------------------------------ bench.py's replica-proving leg (N > 1)
class _FakeProver:
    """stand-in for the package in bench.bench_prove_replicas: contexts, a constraint system, a key, launch / finish -- the "proof" is a
    digest of everything the rank received (constraint system, witness, toxic waste, r, s), so equal proof bytes on every rank prove the
    broadcast delivered the same instance, and the launch / finish bookkeeping checks the ring discipline (at most one proof in flight per
    context, every launch finished exactly once)"""

    def __init__(self):
        import hashlib
        self.hashlib = hashlib
        self.launched = self.finished = 0
        outer = self

        class Context:
            def __init__(self): self.in_flight = None; self.closed = False
            def close(self): assert self.in_flight is None; self.closed = True
        class R1CS:
            def __init__(self, ctx, nc, ni, nv, A, B, C):
                h = outer.hashlib.sha256(np.array([nc, ni, nv], np.uint64).tobytes())
                for trip in (A, B, C):
                    for a in trip: h.update(np.ascontiguousarray(a).tobytes())
                self.digest = h.digest()
            def free(self): pass
        class Keypair:
            def __init__(self, ctx, dcs, tox, precompute=True):
                self.pk = outer.hashlib.sha256(dcs.digest + np.ascontiguousarray(tox).tobytes()).digest()
            def device_bytes(self): return 12345
            def part(self, name): return np.zeros((1, 12), np.uint64)
            def free(self): pass
        class PackedWitness:
            def __init__(self, wit): self.raw = np.ascontiguousarray(wit).tobytes(); self.nbytes = len(self.raw) // 8
        self.Context, self.R1CS, self.Keypair, self.PackedWitness = Context, R1CS, Keypair, PackedWitness

    def _proof(self, pk, src, r, s):
        raw = src.raw if hasattr(src, "raw") else np.ascontiguousarray(src).tobytes()
        d = self.hashlib.sha256(pk + raw + np.ascontiguousarray(r).tobytes() + np.ascontiguousarray(s).tobytes()).digest()
        blob = (d * 6)[:192]
        w = np.frombuffer(blob, np.uint64)
        return w[:12].copy(), w[:24].copy(), w[12:24].copy(), blob

    def groth16_prove(self, ctx, dcs, pk, wit, r, s):
        assert ctx.in_flight is None
        return self._proof(pk, wit, r, s)

    def groth16_prove_launch(self, ctx, dcs, pk, src, r, s):
        assert ctx.in_flight is None, "two proofs in flight on one context"
        ctx.in_flight = self._proof(pk, src, r, s); self.launched += 1

    def groth16_prove_finish(self, ctx):
        assert ctx.in_flight is not None, "finish without launch"
        out, ctx.in_flight = ctx.in_flight, None; self.finished += 1
        return out


def _replica_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib.util, json
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    cpu = torch.device("cpu")

    def allmax(x):
        t = torch.tensor([x], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); return float(t.item())
    comm = {"rank": rank, "world": world, "barrier": dist.barrier, "allmax": allmax,
            "bcast": lambda arrays: bench.bcast_arrays(dist, torch, cpu, rank, arrays),
            "gather": lambda blob: bench.gather_bytes(dist, torch, cpu, world, blob)}

    def instance():                                              # rank 0 only: the other ranks must get it through the broadcast
        assert rank == 0
        rng = np.random.default_rng(99)
        nc, ni = 500, 30
        trip = tuple((np.arange(nc + 1, dtype=np.uint32), rng.integers(0, nc, nc, dtype=np.uint32), rng.integers(0, 1 << 63, (nc, 4), dtype=np.uint64)) for _ in range(3))
        return nc, ni, nc + ni, trip, rng.integers(0, 1 << 63, (nc + ni, 4), dtype=np.uint64), rng.integers(0, 1 << 63, (5, 4), dtype=np.uint64), \
            rng.integers(0, 1 << 63, 4, dtype=np.uint64), rng.integers(0, 1 << 63, 4, dtype=np.uint64)

    fake = _FakeProver()
    ctxs = []
    def make_ctx():
        c = fake.Context(); ctxs.append(c); return c
    res, proof, pub, parts = bench.bench_prove_replicas(make_ctx, fake, comm, instance, 9, total=11, contexts=3)
    ok = (res["n_gpus"] == world and res["proofs_per_gpu"] == 11 and res["every_rank_same_proof_bytes"] and res["proofs_per_s"] > 0
          and res["constraints"] == 500 and fake.launched == fake.finished == 3 + (2 * 3 + 2) + (11 + 2) and all(c.closed for c in ctxs) and len(ctxs) == 3
          and pub.shape == (30, 4) and (parts is not None) == (rank == 0))
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "FAIL: " + json.dumps({k: (x if isinstance(x, (int, float, bool, str)) else str(x)) for k, x in res.items()}) + f" launched {fake.launched} finished {fake.finished}")
    dist.destroy_process_group()


def test_replica_proving_leg_of_the_bench_at_world_2(tmp_path):
    """bench.py --gpus N, N > 1: every rank proves over its own key (bench_prove_replicas) -- here with a stand-in prover over gloo: the instance
    reaches every rank through the broadcast, the ring keeps one proof in flight per context, the line carries all ranks' proofs / the slowest
    rank's time, and the ranks' proof bytes are compared"""
    world = 2
    port = 29500 + (os.getpid() % 1000) + 17
    mp.spawn(_replica_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"
