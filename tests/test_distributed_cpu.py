"""CPU, world_size 2, gloo: the exchange step of the sharded MSM -- each rank's Jacobian partial
sum is all-gathered and folded locally by the library's host code (vsp_fold_jacobian; no GPU needed).  The per-shard
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
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if ok else "FAIL")
    dist.destroy_process_group()


def test_sharded_msm_exchange_world2(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"
