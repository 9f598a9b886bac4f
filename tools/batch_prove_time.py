"""vsp_groth16_prove_batch: proofs per second by batch size at 2^LOG_M constraints over one PLAIN resident key, against single proofs on the
same key (and each proof of the batch checked against the single one)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cref, bls12_381 as o
import vote_saver_protocol_amd as v
lg = int(os.environ.get("LOG_M", "16")); reps = int(os.environ.get("REPS", "6"))
ni = 30; nc = (1 << lg) - ni - 2
ctx = v.Context(0)
gen = o.splitmix64(16)
cs, wit = cref.R1CS.synth(nc, ni, 40, ballot=(25, 3))
tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
kp = v.Keypair(ctx, dcs, tox, precompute=0)
Kmax = int(os.environ.get("KMAX", "32"))
R = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(Kmax)], np.uint64); S = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(Kmax)], np.uint64)
single = [v.groth16_prove(ctx, dcs, kp.pk, wit, R[k], S[k])[3] for k in range(min(Kmax, 4))]
t0 = time.perf_counter()
for _ in range(20): v.groth16_prove(ctx, dcs, kp.pk, wit, R[0], S[0])
one_ms = (time.perf_counter() - t0) / 20 * 1e3
out = ["single %.2f ms = %.0f/s" % (one_ms, 1e3 / one_ms)]
for K in [k for k in (1, 2, 4, 8, 16, 32, 64) if k <= Kmax]:
    W = np.ascontiguousarray(np.broadcast_to(wit, (K,) + wit.shape))
    A, B, C, proofs = v.groth16_prove_batch(ctx, dcs, kp.pk, W, R[:K], S[:K])
    ok = all(proofs[k] == single[k] for k in range(min(K, len(single))))
    t0 = time.perf_counter()
    for _ in range(reps): v.groth16_prove_batch(ctx, dcs, kp.pk, W, R[:K], S[:K])
    dt = (time.perf_counter() - t0) / reps
    out.append("K=%d: %.2f ms per batch = %.0f proofs/s%s" % (K, dt * 1e3, K / dt, "" if ok else " MISMATCH"))
print("2^%d constraints, plain key: " % lg + "; ".join(out))
