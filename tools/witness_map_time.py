"""witness_map on vectors resident in device memory (vsp_witness_map_h_device): ms per call, the batched form (round 4: three transforms per
launch, the pointwise step fused into the last transform's first pass) against the sequence of rounds 1-3 (option witness_map_batched = 0);
both must give the same H word for word.
(Round 3 tried the three iFFT -> coset-FFT chains on three streams with an event fork / join per call: 1.04 against 0.97 ms at 2^20 and
0.54 against 0.38 ms at 2^16 -- the cross-stream waits cost more than the overlap returns.)"""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vote_saver_protocol_amd as v  # noqa: E402
R = int(os.environ.get("R", "30"))
ctx = v.Context(0)
rng = np.random.default_rng(5)
for lg in [int(x) for x in os.environ.get("LOG_N", "12,16,20").split(",")]:
    n = 1 << lg
    vec = [rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64) for _ in range(3)]
    dH = ctx.dmalloc(n * 32)
    res = {}
    for batched in (0, 1, 0, 1):
        ctx.set_option("witness_map_batched", batched)
        d = [ctx.to_device(x) for x in vec]                     # witness_map overwrites its inputs
        ctx.check(ctx.lib.vsp_witness_map_h_device(ctx.h, C.c_void_p(d[0]), C.c_void_p(d[1]), C.c_void_p(d[2]), lg, C.c_void_p(dH)))
        h = np.zeros((n, 4), np.uint64); ctx.d2h(h, dH)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(R):
            ctx.lib.vsp_witness_map_h_device(ctx.h, C.c_void_p(d[0]), C.c_void_p(d[1]), C.c_void_p(d[2]), lg, C.c_void_p(dH))
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / R * 1e3
        res.setdefault(batched, []).append((ms, h))
        for x in d: ctx.dfree(x)
    same = all(np.array_equal(res[0][0][1], x[1]) for b in (0, 1) for x in res[b])
    print("2^%d witness_map: sequence %.3f / %.3f ms, batched %.3f / %.3f ms, same H: %s" % (lg, res[0][0][0], res[0][1][0], res[1][0][0], res[1][1][0], same))
    ctx.dfree(dH)
