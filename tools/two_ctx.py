"""Proof rate with one and with two contexts over one resident key (diagnostic):  python tools/two_ctx.py [reps]
   env OPTS="name=value,name=value" sets context options on every context."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cref, bls12_381 as o
import vote_saver_protocol_amd as v
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lg = int(os.environ.get("LOG_M", "20"))
opts = [kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv]
def mk():
    c = v.Context(0)
    for k, val in opts: c.set_option(k, int(val))
    return c
ni = 30; nc = (1 << lg) - ni - 2
ctx = mk()
gen = o.splitmix64(5)
cs, wit = cref.R1CS.synth(nc, ni, 4, ballot=(25, 7))
tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
r = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64); s = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64)
wit = ctx.host_register(np.ascontiguousarray(wit))
kp = v.Keypair(ctx, dcs, tox, precompute=int(os.environ.get("PRE", "1")))
v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
t0 = time.perf_counter()
for _ in range(reps): v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
one = (time.perf_counter() - t0) / reps * 1e3
for n_ctx in (2, 3):
    cs_ = [ctx] + [mk() for _ in range(n_ctx - 1)]
    for c in cs_[1:]:
        v.groth16_prove(c, dcs, kp.pk, wit, r, s); v.groth16_prove(c, dcs, kp.pk, wit, r, s)
    def worker(c):
        for _ in range(reps): v.groth16_prove(c, dcs, kp.pk, wit, r, s)
    th = [threading.Thread(target=worker, args=(c,)) for c in cs_]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    many = (time.perf_counter() - t0) / (n_ctx * reps) * 1e3
    print("OPTS=%s: one context %.2f ms per proof; %d contexts %.2f ms per proof (%.1f proofs/s)" % (os.environ.get("OPTS", ""), one, n_ctx, many, 1e3 / many), flush=True)
    for c in cs_[1:]: c.close()
