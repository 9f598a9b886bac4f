"""Do independent transforms overlap?  K contexts (one stream each), one 2^LOG_N vector each: R rounds of (inverse transform, coset
transform) queued back to back on every context, all waited for at the end -- against the same work on one context."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vote_saver_protocol_amd as v  # noqa: E402
lg = int(os.environ.get("LOG_N", "20")); K = int(os.environ.get("K", "3")); R = int(os.environ.get("R", "20"))
rng = np.random.default_rng(3)
a = rng.integers(0, 1 << 62, size=(1 << lg, 4), dtype=np.uint64)
g7 = np.array([7, 0, 0, 0], np.uint64)
ctxs = [v.Context(0) for _ in range(K)]
doms = [v.EvaluationDomain(c, 1 << lg) for c in ctxs]
ds = [c.to_device(a) for c in ctxs]
for c, dom, d in zip(ctxs, doms, ds):
    dom.fft_device(d, inverse=True); dom.fft_device(d, coset=g7); c.synchronize()
def run(n_ctx, rounds):
    t0 = time.perf_counter()
    for _ in range(rounds):
        for i in range(n_ctx):
            doms[i].fft_device(ds[i], inverse=True); doms[i].fft_device(ds[i], coset=g7)
    for i in range(n_ctx):
        ctxs[i].synchronize()
    return (time.perf_counter() - t0) / rounds * 1e3
one = run(1, R * K) 
many = run(K, R)
print("2^%d: inverse + coset transform pair: %.3f ms on one stream; %d pairs on %d streams: %.3f ms = %.3f ms per pair" % (lg, one, K, K, many, many / K))
