"""Plan counts of a dense MSM (buckets, parts, buckets split in few / many parts) -- diagnostic:  python tools/msm_counts.py [group] [log_n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vote_saver_protocol_amd as v  # noqa: E402
group = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 18
n = 1 << lg
rng = np.random.default_rng(1)
ks = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ks[:, 3] >>= np.uint64(2)
ss = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ss[:, 3] >>= np.uint64(2)
ctx = v.Context(0)
ctx.set_option("msm_debug_counts", 1)
if os.environ.get("GLV"): ctx.set_option("msm_glv", int(os.environ["GLV"]))
d_k = ctx.to_device(ks)
d_pts = v.fixed_base_mul(ctx, d_k, n, group)
bases = ctx.bases_from_device(d_pts, n, group)
d_s = ctx.to_device(ss)
for lanes in (0, 8, 16, 32):
    ctx.set_option("msm_dimsum_lanes", lanes)
    bases.msm(d_s)
    t0 = time.perf_counter()
    for _ in range(5):
        bases.msm(d_s)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print("G%d 2^%d dimsum_lanes=%d: %.3f ms" % (group, lg, lanes, ms), {k: ctx.stat(k) for k in ("msm_window_bits", "msm_windows", "msm_split", "msm_buckets", "msm_parts",
          "msm_medium_buckets", "msm_heavy_buckets", "msm_endomorphism_split")}, flush=True)
