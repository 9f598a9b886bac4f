"""GPU: the package's sharded multi-exponentiation (vote_saver_protocol_amd/sharded.py) over the REAL per-rank pipeline.

One GPU is all a test box has, so: (a) the world-1 paths in process (LocalExchange; the device-side record and fold entry points);
(b) the RCCL code path -- TorchExchange with backend "nccl": device send / receive buffers, all_gather_into_tensor on the exchange
stream, vsp_fold_jacobian_device -- with a process group of ONE rank; (c) TWO ranks sharing GPU 0 over gloo (RCCL refuses two ranks on
one device), each with its own context and resident chunk: the N > 1 arithmetic end to end on hardware.  Results against the C oracle."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, rand_fr_array

import vote_saver_protocol_amd as v

pytestmark = pytest.mark.gpu


def test_sharded_msm_world1_local_exchange(ctx, cref):
    n = 3000
    ks, ss = rand_fr_array(n, 21), rand_fr_array(n, 22)
    for group, gen, ref in ((1, cref.g1_batch_mul_gen, cref.msm_g1), (2, cref.g2_batch_mul_gen, cref.msm_g2)):
        m = n if group == 1 else 700
        bases = gen(ks[:m])
        B = ctx.upload_bases(bases, group); d_s = ctx.to_device(ss[:m])
        try:
            job = v.ShardedMsm(B, v.LocalExchange(ctx))
            want = ref(bases, ss[:m])
            assert np.array_equal(job.msm(d_s), want)
            for depth in (1, 2, 4):
                assert np.array_equal(job.run(d_s, 5, depth), want)
        finally:
            ctx.dfree(d_s); B.free()


def test_record_left_in_device_memory_and_folded_from_there(ctx, cref):
    """vsp_msm_finish_jacobian_device / vsp_fold_jacobian_device: the record reaches the caller's device buffer by the asynchronous copy,
    equals the host-side record, and N of them fold from device memory to the oracle's sum; the pinned ring survives more finishes than
    it has entries"""
    n = 2500
    ks, ss = rand_fr_array(n, 31), rand_fr_array(n, 32)
    for group, gen, ref, words in ((1, cref.g1_batch_mul_gen, cref.msm_g1, 18), (2, cref.g2_batch_mul_gen, cref.msm_g2, 36)):
        m = n if group == 1 else 600
        bases = gen(ks[:m])
        cuts = [(0, m // 3), (m // 3, m // 3), (m // 3, m)]           # three "ranks", the middle one with an empty chunk
        B = ctx.upload_bases(bases, group); d_s = ctx.to_device(ss[:m])
        d_recs = ctx.dmalloc(len(cuts) * words * 8)
        try:
            for rounds in range(3):                                   # 9 finishes per group > the ring's 4 entries
                for i, (lo, hi) in enumerate(cuts):
                    B.msm_launch(1, d_s + lo * 32, hi - lo, lo)
                    B.msm_finish_jacobian_device(1, d_recs + i * words * 8)
                got = v.fold_jacobian_device(ctx, d_recs, len(cuts), group)
                assert np.array_equal(got, ref(bases, ss[:m]))
            host = np.zeros((len(cuts), words), np.uint64); ctx.d2h(host, d_recs)
            for i, (lo, hi) in enumerate(cuts):                       # a record is one of many projective forms of its point: compare the points
                B.msm_launch(2, d_s + lo * 32, hi - lo, lo)
                assert np.array_equal(v.fold_jacobian(ctx, B.msm_finish_jacobian(2)[None], group), v.fold_jacobian(ctx, host[i][None], group))
                assert np.array_equal(v.fold_jacobian(ctx, host[i][None], group), ref(bases[lo:hi], ss[lo:hi]) if hi > lo else np.zeros(12 * group, np.uint64))
            assert not v.fold_jacobian_device(ctx, d_recs, 0, group).any()          # empty fold = infinity
        finally:
            ctx.dfree(d_recs); ctx.dfree(d_s); B.free()
    lib = ctx.lib
    out = np.zeros(12, np.uint64); inf = C.c_int(0)
    assert lib.vsp_fold_jacobian_device(ctx.h, 3, None, 0, None, out.ctypes.data_as(C.c_void_p), C.byref(inf)) == -1
    assert lib.vsp_fold_jacobian_device(ctx.h, 1, None, 2, None, out.ctypes.data_as(C.c_void_p), C.byref(inf)) == -1
    assert lib.vsp_msm_finish_jacobian_device(ctx.h, 9, C.c_void_p(8), None) == -1
    assert lib.vsp_msm_finish_jacobian_device(ctx.h, 1, None, None) == -1


def _rank(rank, world, port, backend, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import cref
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    fails = []
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = v.Context(0)
    x = v.TorchExchange(ctx, dev)
    if x.ranks_seen() != world or x.device_records != (backend == "nccl"):
        fails.append("exchange setup")
    ks, ss = rand_fr_array(5000, 41), rand_fr_array(5000, 42)      # the same global problem on every rank
    for group, gen, ref, m in ((1, cref.g1_batch_mul_gen, cref.msm_g1, 5000), (2, cref.g2_batch_mul_gen, cref.msm_g2, 900)):
        bases = gen(ks[:m])
        lo, hi = v.shard_bounds(m, world, rank)
        B = ctx.upload_bases(bases[lo:hi], group)
        d_s = torch.from_numpy(ss[lo:hi].view(np.int64).copy()).to(dev)
        job = v.ShardedMsm(B, x)
        want = ref(bases, ss[:m])
        if not np.array_equal(job.msm(d_s), want):
            fails.append(f"group {group} blocking")
        for depth in (1, 3):
            if not np.array_equal(job.run(d_s, 6, depth), want):
                fails.append(f"group {group} depth {depth}")
        B.free()
    ctx.close()
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write("ok" if not fails else "FAIL: " + ", ".join(fails))
    dist.destroy_process_group()


def _spawn(tmp_path, world, backend):
    port = 29700 + (os.getpid() % 1000) + (7 if backend == "nccl" else 0)
    mp.spawn(_rank, args=(world, port, backend, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}.txt").read() == "ok"


def test_rccl_code_path_with_one_rank(tmp_path):
    """backend nccl (= RCCL): device buffers, all_gather_into_tensor on the exchange stream, the fold from device memory"""
    _spawn(tmp_path, 1, "nccl")


def test_two_ranks_sharing_the_gpu_over_gloo(tmp_path):
    """N = 2 on hardware: two processes, one context and one resident chunk each, host-side all-gather of the records"""
    _spawn(tmp_path, 2, "gloo")
