"""REPS batches of K proofs at 2^LOG_M constraints over a plain key, for rocprofv3 --kernel-trace --stats"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cref, bls12_381 as o
import vote_saver_protocol_amd as v
lg = int(os.environ.get("LOG_M", "16")); reps = int(os.environ.get("REPS", "10")); K = int(os.environ.get("K", "16"))
ni = 30; nc = (1 << lg) - ni - 2
ctx = v.Context(0)
gen = o.splitmix64(16)
cs, wit = cref.R1CS.synth(nc, ni, 40, ballot=(25, 3))
tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
kp = v.Keypair(ctx, dcs, tox, precompute=0)
R = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(K)], np.uint64); S = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(K)], np.uint64)
W = np.ascontiguousarray(np.broadcast_to(wit, (K,) + wit.shape))
if os.environ.get("PINNED"): W = ctx.host_register(W)      # page-locked witnesses: the K x num_vars x 32-byte copy is an asynchronous DMA
v.groth16_prove_batch(ctx, dcs, kp.pk, W, R, S)
ctx.stats_reset()                                          # (the first call also builds the key's fixed-base tables of delta)
t0 = time.perf_counter()
for _ in range(reps): v.groth16_prove_batch(ctx, dcs, kp.pk, W, R, S)
print("K=%d 2^%d: %.2f ms per batch" % (K, lg, (time.perf_counter() - t0) / reps * 1e3))
print("  host phases per batch (ms): " + ", ".join("%s %.2f" % (k, ctx.stat("prove_batch_" + k + "_ms") / reps) for k in ("launch", "delta", "witness_finishes", "sA_rB1", "h_finish", "assembly")))
