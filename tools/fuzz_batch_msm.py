"""Randomised parity run of vsp_msm_resident_batch against the C oracle, vector by vector (diagnostic):  python tools/fuzz_batch_msm.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vote_saver_protocol_amd as v
import cref, bls12_381 as o
from conftest import rand_fr_array, L
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = v.Context(0)
pool1 = cref.g1_batch_mul_gen(rand_fr_array(20000, seed=3000 + seed)); pool2 = cref.g2_batch_mul_gen(rand_fr_array(2500, seed=4000 + seed))
t0 = time.time(); it = 0; vecs_done = 0; tables = 0; last = time.time()
while time.time() - t0 < budget:
    it += 1
    if time.time() - last > 60:
        print("... %d batches, %.0f s" % (it, time.time() - t0), flush=True); last = time.time()
    group = 1 if rng.random() < 0.75 else 2
    pool = pool1 if group == 1 else pool2
    n = int(np.exp(rng.uniform(0, np.log(len(pool))))) or 1
    idx = rng.integers(0, len(pool), size=n)
    if rng.random() < 0.15: idx = idx[rng.integers(0, max(1, n // 50), size=n)]
    bases = pool[idx].copy()
    if rng.random() < 0.3 and n > 3:
        for k in rng.integers(0, n, size=max(1, n // 100)): bases[k] = 0
    K = int(rng.integers(1, 13)); stride = n + int(rng.integers(0, 9))
    vecs = np.zeros((K, stride, 4), np.uint64)
    for k in range(K):
        ss = rand_fr_array(n, seed=int(rng.integers(1, 1 << 30)))
        kind = rng.choice(["uniform", "boolean", "small", "equal", "edges", "zero"])
        if kind == "boolean":
            m = rng.random(n) < 0.9; ss[m] = 0; ss[m, 0] = rng.integers(0, 2, size=int(m.sum()), dtype=np.uint64)
        elif kind == "small": ss[:] = 0; ss[:, 0] = rng.integers(0, 70000, size=n, dtype=np.uint64)
        elif kind == "equal": ss[:] = ss[0]
        elif kind == "zero": ss[:] = 0
        elif kind == "edges":
            for i in range(0, n, 7): ss[i] = L(o.R - 1 - int(rng.integers(0, 3)), 4)
        vecs[k, :n] = ss
    ctx.set_option("msm_window_bits", int(rng.choice([0, 0, 0, 7, 9, 12, 16])))
    ctx.set_option("msm_glv", int(rng.choice([1, 1, 0, 2])))
    ctx.set_option("msm_split", int(rng.choice([0, 0, 16, 64])))
    B = ctx.upload_bases(bases, group); d_s = ctx.to_device(vecs.reshape(-1, 4))
    table = None
    if rng.random() < 0.35:                                    # over a table of window multiples: one bucket set per vector
        table = (int(rng.integers(8, 17)), bool(rng.random() < 0.5))
        B.precompute(table[0], split=table[1])
    first = int(rng.integers(0, n)) if rng.random() < 0.3 else 0
    got, inf = B.msm_batch(d_s + 32 * first, K, n=n - first, first=first, stride=stride)
    msm = cref.msm_g1 if group == 1 else cref.msm_g2
    for k in range(K):
        exp = msm(bases[first:], vecs[k, first:n], mixed=True)
        if not np.array_equal(got[k], exp):
            print("MISMATCH", dict(it=it, seed=seed, group=group, n=n, K=K, k=k, first=first, table=table)); sys.exit(1)
    vecs_done += K
    B.free(); ctx.dfree(d_s)
    tables = tables + 1 if table else tables
print("batch fuzz ok: %d batches (%d over tables of window multiples), %d vectors in %.0f s" % (it, tables, vecs_done, time.time() - t0))
