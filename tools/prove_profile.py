"""REPS single proofs at 2^LOG_M constraints (PRE = 1: key with window-multiple tables, 0: plain), one context, for rocprofv3 --kernel-trace --stats"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cref, bls12_381 as o
import vote_saver_protocol_amd as v
lg = int(os.environ.get("LOG_M", "20")); reps = int(os.environ.get("REPS", "10")); pre = int(os.environ.get("PRE", "0"))
ni = 30; nc = (1 << lg) - ni - 2
ctx = v.Context(0)
gen = o.splitmix64(16)
cs, wit = cref.R1CS.synth(nc, ni, 40, ballot=(25, 3))
tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
kp = v.Keypair(ctx, dcs, tox, precompute=pre)
r = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64); s = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64)
w = ctx.host_register(np.ascontiguousarray(wit))
v.groth16_prove(ctx, dcs, kp.pk, w, r, s)
t0 = time.perf_counter()
for _ in range(reps): v.groth16_prove(ctx, dcs, kp.pk, w, r, s)
print("2^%d precompute=%d: %.2f ms per proof" % (lg, pre, (time.perf_counter() - t0) / reps * 1e3))
