"""Runs forward NTTs of 2^LOG_N (default 22) so that rocprofv3 --kernel-trace --stats shows the pass times:
   cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d <dir> -- python3 $GRAFT_REPO_ROOT/tools/ntt_profile.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vote_saver_protocol_amd as v  # noqa: E402
lg = int(os.environ.get("LOG_N", "22"))
ctx = v.Context(0)
rng = np.random.default_rng(3)
a = rng.integers(0, 1 << 62, size=(1 << lg, 4), dtype=np.uint64)
d = ctx.to_device(a)
dom = v.EvaluationDomain(ctx, 1 << lg)
dom.fft_device(d); ctx.synchronize()
for mode in ("fwd", "inv", "coset"):
    t0 = time.perf_counter()
    for _ in range(10):
        if mode == "fwd": dom.fft_device(d)
        elif mode == "inv": dom.fft_device(d, inverse=True)
        else: dom.fft_device(d, coset=np.array([7, 0, 0, 0], np.uint64))
    ctx.synchronize()
    print("NTT 2^%d %s: %.3f ms" % (lg, mode, (time.perf_counter() - t0) / 10 * 1e3))
