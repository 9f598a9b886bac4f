"""batched proving from C contexts at once (a host thread each, one plain resident key): proofs per second at 2^LOG_M constraints, batch K"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cref, bls12_381 as o
import vote_saver_protocol_amd as v
lg = int(os.environ.get("LOG_M", "16")); reps = int(os.environ.get("REPS", "8"))
ni = 30; nc = (1 << lg) - ni - 2
ctx = v.Context(0)
if os.environ.get("PRE_WINDOW"): ctx.set_option("generate_precompute_window", int(os.environ["PRE_WINDOW"]))
gen = o.splitmix64(16)
cs, wit = cref.R1CS.synth(nc, ni, 40, ballot=(25, 3))
tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
kp = v.Keypair(ctx, dcs, tox, precompute=int(os.environ.get("PRE", "0")))
ctxs = [ctx] + [v.Context(0) for _ in range(3)]
out = []
for K in [int(x) for x in os.environ.get("KS", "8,16,32").split(",")]:
    R = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(K)], np.uint64); S = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(K)], np.uint64)
    W = np.ascontiguousarray(np.broadcast_to(wit, (K,) + wit.shape))
    for c in ctxs: v.groth16_prove_batch(c, dcs, kp.pk, W, R, S)
    for C in (1, 2, 3, 4):
        def worker(c):
            for _ in range(reps): v.groth16_prove_batch(c, dcs, kp.pk, W, R, S)
        th = [threading.Thread(target=worker, args=(c,)) for c in ctxs[:C]]
        t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        out.append("K=%d x %d contexts: %.0f/s" % (K, C, C * reps * K / (time.perf_counter() - t0)))
    # ONE host thread, a batch in flight on each of C contexts (vsp_groth16_prove_batch_launch / _finish)
    for C in (2, 3):
        ring = ctxs[:C]
        total = C * reps
        t0 = time.perf_counter()
        for i in range(total):
            c = ring[i % C]
            if i >= C: v.groth16_prove_batch_finish(c)
            v.groth16_prove_batch_launch(c, dcs, kp.pk, W, R, S)
        for i in range(total, total + C): v.groth16_prove_batch_finish(ring[i % C])
        out.append("K=%d one thread x %d in flight: %.0f/s" % (K, C, total * K / (time.perf_counter() - t0)))
print("2^%d: " % lg + "; ".join(out))
