"""Proof latency and phase split over constraint counts (diagnostic):  python tools/prove_sizes.py [log sizes...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cref, bls12_381 as o
import vote_saver_protocol_amd as v
sizes = [int(x) for x in sys.argv[1:]] or [12, 14, 16, 18]
ctx = v.Context(0)
gen = o.splitmix64(5)
for lg in sizes:
    ni = 30; nc = (1 << lg) - ni - 2
    cs, wit = cref.R1CS.synth(nc, ni, 4, ballot=(25, 7))
    tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
    r = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64); s = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64)
    wit = ctx.host_register(np.ascontiguousarray(wit))
    kp = v.Keypair(ctx, dcs, tox, precompute=1)
    for _ in range(3): v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
    ctx.stats_reset(); reps = 20
    t0 = time.perf_counter()
    for _ in range(reps): v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
    ms = (time.perf_counter() - t0) / reps * 1e3
    ph = {k: round(ctx.stat("prove_" + k + "_ms") / reps, 3) for k in ("launch", "host_overlap", "wait", "assembly")}
    print("2^%d constraints: %.3f ms per proof" % (lg, ms), ph, flush=True)
    kp.free(); dcs.free(); cs.free()
