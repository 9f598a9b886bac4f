import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vote_saver_protocol_amd as v
import cref, bls12_381 as o
from conftest import rand_fr_array, L
ctx = v.Context(0)
N = 600
ks = rand_fr_array(N, seed=52)
bases_all = cref.g2_batch_mul_gen(ks)
ss_all = rand_fr_array(N, seed=60)
for n in (1, 2, 3, 8, 40, 200, 600):
    res = []
    for c in (5, 8, 9, 10, 11, 12, 13, 14, 16):
        ctx.set_option("msm_window_bits", c)
        bases, ss = bases_all[:n], ss_all[:n]
        exp = cref.msm_g2(bases, ss, mixed=True)
        got = v.multiexp(ctx, bases, ss, 2)
        res.append((c, bool(np.array_equal(got, exp))))
    print(n, res)
