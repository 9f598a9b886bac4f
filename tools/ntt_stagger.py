"""experiment: does staggering the start of every other workgroup of an NTT pass break the lockstep of the two workgroups of a CU?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vote_saver_protocol_amd as v
ctx = v.Context(0)
lg = int(os.environ.get("LOG_N", "22"))
a = ctx.to_device(np.random.default_rng(1).integers(0, 1 << 62, size=(1 << lg, 4), dtype=np.uint64))
dom = v.EvaluationDomain(ctx, 1 << lg)
def run(reps=20):
    dom.fft_device(a); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): dom.fft_device(a)
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("baseline %.3f ms" % run())
for shift in (0, 1, 2, 3, 8, 9):
    out = []
    for st in (1, 2, 4, 8):
        ctx.set_option("ntt_stagger", st); ctx.set_option("ntt_stagger_shift", shift)
        out.append("%d: %.3f" % (st, run()))
    print("shift %d  " % shift + "  ".join(out))
ctx.set_option("ntt_stagger", 0)
print("baseline again %.3f ms" % run())
