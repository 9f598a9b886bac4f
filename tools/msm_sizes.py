"""MSM timing over sizes / groups with and without the endomorphism split:  python tools/msm_sizes.py   (diagnostic)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vote_saver_protocol_amd as v  # noqa: E402
rng = np.random.default_rng(1)
plan = ((1, (16, 18, 19, 20)), (2, (16, 18, 19)))
if os.environ.get("SIZES"):                                   # e.g. SIZES="1:21,22;2:19,20"
    plan = tuple((int(p.split(":")[0]), tuple(int(x) for x in p.split(":")[1].split(","))) for p in os.environ["SIZES"].split(";"))
for group, lgs in plan:
    for lg in lgs:
        n = 1 << lg
        ks = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ks[:, 3] >>= np.uint64(2)
        ss = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ss[:, 3] >>= np.uint64(2)
        res = {}
        for glv in (0, 2):
            ctx = v.Context(0)
            ctx.set_option("msm_glv", glv)
            d_k = ctx.to_device(ks)
            d_pts = v.fixed_base_mul(ctx, d_k, n, group)
            bases = ctx.bases_from_device(d_pts, n, group)
            d_s = ctx.to_device(ss)
            out, _ = bases.msm(d_s)
            t0 = time.perf_counter()
            for _ in range(5):
                bases.msm(d_s)
            blocking = (time.perf_counter() - t0) / 5 * 1e3
            sl = (1, 2, 4)
            for k in range(2): bases.msm_launch(sl[k], d_s)
            t0 = time.perf_counter(); K = 9
            for k in range(K):
                if k + 2 < K + 2: bases.msm_launch(sl[(k + 2) % 3], d_s)
                bases.msm_finish_jacobian(sl[k % 3])
            for k in range(2): bases.msm_finish_jacobian(sl[(K + k) % 3])
            piped = (time.perf_counter() - t0) / (K + 2) * 1e3
            res[glv] = (blocking, piped, out.copy(), ctx.stat("msm_windows"), ctx.stat("msm_endomorphism_split"))
            ctx.close()
        same = np.array_equal(res[0][2], res[2][2])
        print("G%d 2^%d  plain: %.3f ms blocking, %.3f ms pipelined (W=%d)   split: %.3f / %.3f (W=%d, on=%d)   same result: %s" % (
            group, lg, res[0][0], res[0][1], res[0][3], res[2][0], res[2][1], res[2][3], res[2][4], same), flush=True)
