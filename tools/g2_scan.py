"""G2 2^LOG_N MSM: endomorphism split on / off x part length (diagnostic)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vote_saver_protocol_amd as v
lg = int(os.environ.get("LOG_N", "18")); group = int(os.environ.get("GROUP", "2"))
n = 1 << lg
rng = np.random.default_rng(1)
ks = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ks[:, 3] >>= np.uint64(2)
ss = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ss[:, 3] >>= np.uint64(2)
ref = None
for glv in (0, 2):
    ctx = v.Context(0)
    ctx.set_option("msm_glv", glv)
    d_k = ctx.to_device(ks); d_pts = v.fixed_base_mul(ctx, d_k, n, group)
    bases = ctx.bases_from_device(d_pts, n, group); d_s = ctx.to_device(ss)
    for T in (0, 24, 32, 48, 64, 96):
        ctx.set_option("msm_split", T)
        out, _ = bases.msm(d_s)
        ref = out if ref is None else ref
        ctx.stats_reset()
        t0 = time.perf_counter()
        for _ in range(5):
            bases.msm(d_s)
        dt = (time.perf_counter() - t0) / 5 * 1e3
        print("G%d 2^%d glv=%d T=%3d: %.3f ms  (accum %.3f ms, c=%d W=%d T_used=%d) same=%s" % (group, lg, glv, T, dt, ctx.stat("msm_accum_ms") / 5, ctx.stat("msm_window_bits"),
              ctx.stat("msm_windows"), ctx.stat("msm_split"), np.array_equal(out, ref)), flush=True)
    ctx.close()
