"""2^LOG_M-constraint proofs per second from N host threads with a context each over one resident key (the real circuit's likely size is
2^15..2^16 constraints): N = 1..CTX, under the GPU_MAX_HW_QUEUES of the environment."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cref, bls12_381 as o
import vote_saver_protocol_amd as v
lg = int(os.environ.get("LOG_M", "16")); nctx = int(os.environ.get("CTX", "8")); per = int(os.environ.get("PER", "60"))
ni = 30; nc = (1 << lg) - ni - 2
ctx = v.Context(0)
gen = o.splitmix64(16)
cs, wit = cref.R1CS.synth(nc, ni, 40, ballot=(25, 3))
tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
kp = v.Keypair(ctx, dcs, tox, precompute=int(os.environ.get("PRE", "1")))
r = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64); s = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64)
wit = ctx.host_register(np.ascontiguousarray(wit))
ctxs = [ctx] + [v.Context(0) for _ in range(nctx - 1)]
for c in ctxs: v.groth16_prove(c, dcs, kp.pk, wit, r, s)
out = []
for k in [x for x in (1, 2, 4, 6, 8, 12, 16) if x <= nctx]:
    def worker(c):
        for _ in range(per): v.groth16_prove(c, dcs, kp.pk, wit, r, s)
    th = [threading.Thread(target=worker, args=(c,)) for c in ctxs[:k]]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    out.append("%d: %.0f" % (k, k * per / (time.perf_counter() - t0)))
print("GPU_MAX_HW_QUEUES=%s 2^%d proofs/s by contexts  " % (os.environ.get("GPU_MAX_HW_QUEUES"), lg) + "  ".join(out))
