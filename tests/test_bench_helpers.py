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
    """the timed loop of bench.py: at most `depth` multi-exponentiations in flight, every step's record goes through exchange begin / end
    exactly once and in order, the exchange of step k ends only after step k + 1's record exists, the last fold is what is returned"""
    for depth in (1, 2, 3, 4):
        for steps in (1, 2, 3, 7):
            log = []

            class Bases:
                def __init__(self): self.in_flight = 0; self.count = 0
                def msm_launch(self, slot, d_s): self.in_flight += 1; assert self.in_flight <= depth - 1 + 1; log.append(("launch", slot))
                def msm_finish_jacobian(self, slot): self.in_flight -= 1; self.count += 1; log.append(("finish", slot)); return np.array([self.count])
                def msm_jacobian(self, d_s): self.count += 1; log.append(("blocking",)); return np.array([self.count])

            class Problem:
                pass
            prob = Problem(); prob.bases = Bases(); prob.d_s = None; prob.group = 1
            begun, ended = [], []

            def begin(rec, group, buf):
                begun.append(int(rec[0])); assert buf == (len(begun) - 1) & 1
                return int(rec[0])

            def end(h):
                ended.append(h); assert len(begun) >= min(h + 1, steps)      # step h's exchange ends after step h + 1 has begun (or at the very end)
                return ("folded", h)

            el, res = bench.run_sharded(prob, steps, 0, (begin, end), lambda: None, depth)
            assert begun == list(range(1, steps + 1)) and ended == begun and res == ("folded", steps) and el >= 0
            assert prob.bases.in_flight == 0
