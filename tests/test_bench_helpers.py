"""CPU checks of bench.py's sharding and checker helpers (the GPU legs themselves run on the box): contiguous chunk bounds of the
sharded MSM and the device-side big sum the strong-scaling legs verify against."""
import importlib.util
import os

import numpy as np
import torch

import bls12_381 as o
from conftest import ROOT, fr_ints

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_shard_bounds_cover_the_problem_exactly_once():
    for total in (1, 7, 1 << 20, (1 << 26), 1000003):
        for world in (1, 2, 3, 4, 8):
            cuts = [bench.shard_bounds(total, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_device_dot_product_mod_r_matches_big_ints():
    g = torch.Generator(); g.manual_seed(5)
    n = 5000
    k64 = torch.randint(-(1 << 63), (1 << 63) - 1, (n,), dtype=torch.int64, generator=g)
    s4 = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, generator=g)
    s4[:, 3] &= 0x3FFFFFFFFFFFFFFF
    k64[0] = -1; k64[1] = -(1 << 63); s4[2] = -1; s4[2, 3] = 0x3FFFFFFFFFFFFFFF      # all-ones patterns
    ks = [int(x) & 0xFFFFFFFFFFFFFFFF for x in k64.tolist()]
    ss = fr_ints(s4.numpy().view(np.uint64))
    assert bench.dot_mod_r_device(torch, k64, s4) == sum(a * b for a, b in zip(ks, ss)) % o.R
    assert bench.R_MOD == o.R


def test_run_sharded_keeps_depth_in_flight_and_folds_every_step():
    """the timed loop (vote_saver_protocol_amd.sharded.ShardedMsm.run, driven by bench.run_sharded): at most `depth` multi-exponentiations
    in flight, every step's record goes through exchange begin / end exactly once and in order, the exchange of step k ends only after
    step k + 1's record exists, the last fold is what is returned"""
    for depth in (1, 2, 3, 4):
        for steps in (1, 2, 3, 7):
            log = []

            class Bases:
                group = 1
                def __init__(self): self.in_flight = 0; self.count = 0
                def msm_launch(self, slot, d_s): self.in_flight += 1; assert self.in_flight <= depth; log.append(("launch", slot))
                def msm_finish_jacobian(self, slot): self.in_flight -= 1; self.count += 1; log.append(("finish", slot)); return np.array([self.count])

            class Problem:
                pass
            prob = Problem(); prob.bases = Bases(); prob.d_s = None; prob.group = 1
            begun, ended = [], []

            class Exchange:
                def begin(self, bases, slot, buf):
                    rec = bases.msm_finish_jacobian(slot)
                    begun.append(int(rec[0])); assert buf == (len(begun) - 1) & 1
                    return int(rec[0])

                def end(self, h):
                    ended.append(h); assert len(begun) >= min(h + 1, steps)      # step h's exchange ends after step h + 1 has begun (or at the very end)
                    return ("folded", h)

            from vote_saver_protocol_amd.sharded import ShardedMsm
            res = ShardedMsm(prob.bases, Exchange()).run(prob.d_s, steps, depth)
            assert begun == list(range(1, steps + 1)) and ended == begun and res == ("folded", steps)
            assert prob.bases.in_flight == 0
            slots = {x[1] for x in log if x[0] == "launch"}
            assert len(slots) <= depth and (slots == {0} if depth == 1 else 0 not in slots)      # one in flight: the context's stream; else the slots' own

            # bench.run_sharded around it: every work slot is used once before the warm-up (set-up: first use allocates), then W + K steps
            class Quiet:
                def begin(self, bases, slot, buf): return bases.msm_finish_jacobian(slot)
                def end(self, h): return h
            prob.bases = Bases(); log.clear()
            el, _ = bench.run_sharded(prob, steps, 2, Quiet(), lambda: None, depth)
            assert el >= 0 and prob.bases.count == (depth if depth > 1 else 0) + 2 + steps
            if depth > 1:
                assert {x[1] for x in log[:2 * depth] if x[0] == "launch"} == set(ShardedMsm.SLOTS[:depth])


def test_launcher_refuses_cleanly_without_gpus_and_never_imports_torch_first():
    """`python bench.py --gpus N` as a plain process: the parent counts GPUs from the KFD topology (no HIP call) and, on a box with fewer
    GPUs than ranks, returns 3 without starting anything"""
    import subprocess, sys
    if bench.visible_gpu_count() not in (0, None):
        import pytest
        pytest.skip("GPUs present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and r.stdout == "" and "nothing was launched" in r.stderr


def test_prove_ring_keeps_one_proof_in_flight_per_context_and_finishes_every_launch():
    """bench.prove_ring (the one-host-thread mode behind `secondary`): with a ring of C contexts, C - 1 proofs are in flight while one is
    awaited, every context holds at most one proof, launches and finishes alternate in ring order, and exactly total + C - 1 proofs run"""
    for C in (1, 2, 3, 4):
        for total in (1, 5, 12):
            log = []

            class Ctx:
                def __init__(self, i): self.i = i; self.busy = False

            class V:
                @staticmethod
                def groth16_prove_launch(c, dcs, pk, src, r, s):
                    assert not c.busy, "two proofs in flight on one context"
                    c.busy = True; log.append(("L", c.i))

                @staticmethod
                def groth16_prove_finish(c):
                    assert c.busy, "finish without launch"
                    c.busy = False; log.append(("F", c.i)); return ("proof", c.i)

            ring = [Ctx(i) for i in range(C)]
            dt, last = bench.prove_ring(V, ring, None, None, None, None, None, total)
            assert dt >= 0 and last[0] == "proof"
            assert sum(1 for x in log if x[0] == "L") == total + C - 1 == sum(1 for x in log if x[0] == "F")
            assert not any(c.busy for c in ring)
            fin = [x[1] for x in log if x[0] == "F"]
            assert fin == [k % C for k in range(total + C - 1)]                       # finishes in ring order
            in_flight = 0; peak = 0
            for kind, _ in log:
                in_flight += 1 if kind == "L" else -1; peak = max(peak, in_flight)
            assert peak == C
