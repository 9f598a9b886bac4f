"""CPU, world_size 2, gloo: the exchange step of the sharded MSM -- each rank's Jacobian partial
sum (144 bytes G1, 288 bytes G2) is all-gathered and folded locally by the library's host code (vsp_fold_jacobian; no GPU needed).  The per-shard
MSM itself is produced by the oracle here because this container has no GPU; on the GPU box the same records come
from vsp_msm_resident_jacobian (tests/test_gpu_msm.py checks those against the oracle)."""
import os
import sys

import numpy as np
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


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import cref
    import vote_saver_protocol_amd as v
    from conftest import rand_fr_array
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 400
    ks, ss = rand_fr_array(n, 1), rand_fr_array(n, 2)          # same global problem on every rank
    bases = cref.g1_batch_mul_gen(ks)
    lo, hi = rank * n // world, (rank + 1) * n // world          # contiguous point chunk of this rank
    part = cref.msm_g1(bases[lo:hi], ss[lo:hi])
    rec = _jacobian_record(part, 0x1234567 + rank)
    gathered = [torch.zeros(18, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(rec.view(np.int64)))
    recs = np.stack([g.numpy().view(np.uint64) for g in gathered])
    lib = v.load()
    out = np.zeros(12, np.uint64)
    import ctypes as C
    inf = C.c_int(0)
    rc = lib.vsp_fold_jacobian(None, 1, recs.ctypes.data_as(C.c_void_p), world, out.ctypes.data_as(C.c_void_p), C.byref(inf))
    full = cref.msm_g1(bases, ss)
    ok = rc == 0 and np.array_equal(out, full) and inf.value == 0
    # the G2 half of BASELINE config 5: 288-byte records through the same exchange
    n2 = 60
    b2 = cref.g2_batch_mul_gen(ks[:n2])
    lo2, hi2 = rank * n2 // world, (rank + 1) * n2 // world
    rec2 = _jacobian_record_g2(cref.msm_g2(b2[lo2:hi2], ss[lo2:hi2]), 0x7654321 + rank)
    g2 = [torch.zeros(36, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(g2, torch.from_numpy(rec2.view(np.int64)))
    recs2 = np.stack([g.numpy().view(np.uint64) for g in g2])
    out2 = np.zeros(24, np.uint64)
    rc2 = lib.vsp_fold_jacobian(None, 2, recs2.ctypes.data_as(C.c_void_p), world, out2.ctypes.data_as(C.c_void_p), C.byref(inf))
    ok = ok and rc2 == 0 and np.array_equal(out2, cref.msm_g2(b2, ss[:n2])) and inf.value == 0
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "FAIL")
    dist.destroy_process_group()


def test_sharded_msm_exchange_world2(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"
