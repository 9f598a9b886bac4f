"""Times the forward NTT at 2^LOG_N (default 22) on device-resident data (diagnostic)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vote_saver_protocol_amd as v
lg = int(os.environ.get("LOG_N", "22"))
ctx = v.Context(0)
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 62, size=(1 << lg, 4), dtype=np.uint64)
d = ctx.to_device(a)
dom = v.EvaluationDomain(ctx, 1 << lg)
dom.fft_device(d); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    dom.fft_device(d)
ctx.synchronize()
print("NTT 2^%d: %.4f ms" % (lg, (time.perf_counter() - t0) / 20 * 1e3))
