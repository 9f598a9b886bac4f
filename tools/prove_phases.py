"""host-side phases of a single proof (context stats): launch / host work under the GPU / wait (incl. the Horner chains of the five finishes) / assembly"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cref, bls12_381 as o
import vote_saver_protocol_amd as v
for lg in [int(x) for x in os.environ.get("LOG_M", "16,20").split(",")]:
    ni = 30; nc = (1 << lg) - ni - 2
    ctx = v.Context(0)
    gen = o.splitmix64(16)
    cs, wit = cref.R1CS.synth(nc, ni, 40, ballot=(25, 3))
    tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
    for pre in (0, 1):
        kp = v.Keypair(ctx, dcs, tox, precompute=pre)
        r = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64); s = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64)
        w = ctx.host_register(np.ascontiguousarray(wit))
        for _ in range(3): v.groth16_prove(ctx, dcs, kp.pk, w, r, s)
        ctx.stats_reset(); reps = 20
        t0 = time.perf_counter()
        for _ in range(reps): v.groth16_prove(ctx, dcs, kp.pk, w, r, s)
        dt = (time.perf_counter() - t0) / reps * 1e3
        print("2^%d precompute=%d: %.2f ms per proof; " % (lg, pre, dt) + ", ".join("%s %.2f" % (k, ctx.stat("prove_" + k + "_ms") / reps) for k in ("launch", "host_overlap", "wait", "assembly")))
        ctx.host_unregister(w); kp.free()
    dcs.free(); cs.free(); ctx.close()
