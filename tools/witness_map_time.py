"""witness_map on vectors resident in device memory (vsp_witness_map_h_device): ms per call.
(Round 3 tried the three iFFT -> coset-FFT chains on three streams with an event fork / join per call: 1.04 against 0.97 ms at 2^20 and
0.54 against 0.38 ms at 2^16 -- the cross-stream waits cost more than the overlap returns; tools/ntt_concurrent.py shows the overlap that
DEEP independent queues do get: 0.23 against 0.29 ms per transform pair at 2^20.)"""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vote_saver_protocol_amd as v  # noqa: E402
lg = int(os.environ.get("LOG_N", "20")); R = int(os.environ.get("R", "30"))
ctx = v.Context(0)
rng = np.random.default_rng(5)
n = 1 << lg
vec = [rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64) for _ in range(3)]
d = [ctx.to_device(x) for x in vec]
dH = ctx.dmalloc(n * 32)
def call():
    ctx.check(ctx.lib.vsp_witness_map_h_device(ctx.h, C.c_void_p(d[0]), C.c_void_p(d[1]), C.c_void_p(d[2]), lg, C.c_void_p(dH)))
out = {}
for mode in (0, 0):
    call(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        call()
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / R * 1e3
    h = np.zeros((n, 4), np.uint64); ctx.d2h(h, dH)
    out.setdefault(mode, []).append((ms, h[:4].copy()))
    print("2^%d witness_map: %.3f ms" % (lg, ms))
