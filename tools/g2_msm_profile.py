"""Runs a few dense G2 multi-exponentiations (2^18 points) so that rocprofv3 --kernel-trace --stats shows where the time goes:
   cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d <dir> -- python3 $GRAFT_REPO_ROOT/tools/g2_msm_profile.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vote_saver_protocol_amd as v  # noqa: E402

log_n = int(os.environ.get("LOG_N", "18"))
n = 1 << log_n
ctx = v.Context(0)
if os.environ.get("WBITS"):
    ctx.set_option("msm_window_bits", int(os.environ["WBITS"]))
rng = np.random.default_rng(1)
ks = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ks[:, 3] >>= np.uint64(2)
ss = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ss[:, 3] >>= np.uint64(2)
d_k = ctx.to_device(ks)
d_pts = ctx.dmalloc(n * 192)
ctx.check(ctx.lib.vsp_fixed_base_mul_g2(ctx.h, d_k, n, d_pts))
bases = ctx.bases_from_device(d_pts, n, group=2)
d_s = ctx.to_device(ss)
bases.msm(d_s)
t0 = time.perf_counter()
for _ in range(5):
    r = bases.msm(d_s)
print("G2 2^%d window_bits=%s: %.3f ms per MSM" % (log_n, os.environ.get("WBITS", "auto"), (time.perf_counter() - t0) / 5 * 1e3))
