"""The kernels whose SQ counters profiles/r3_sq_*.json holds, one multi-exponentiation / transform at a time (nothing pipelined):
   G1 2^20 plain bases through the generated accumulation routine (k_accum28) and through the compiler-allocated one (k_accum28_cxx),
   G2 2^18, G1 2^22 through the staged sort (k_ms_*), forward NTT 2^22.  Run under rocprofv3 by tools/gpu_sq.sh (the program goes directly after `--`)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vote_saver_protocol_amd as v  # noqa: E402

REPS = int(os.environ.get("SQ_REPS", "3"))
ctx = v.Context(0)
rng = np.random.default_rng(3)


def rand_fr(n):
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return a


for group, lg in ((1, 20), (2, 18)):
    n = 1 << lg
    d_k = ctx.to_device(rand_fr(n)); d_s = ctx.to_device(rand_fr(n))
    d_b = v.fixed_base_mul(ctx, d_k, n, group)
    B = ctx.bases_from_device(d_b, n, group)
    ctx.dfree(d_b); ctx.dfree(d_k)
    for asm in ((1, 0) if group == 1 else (1,)):
        ctx.set_option("msm_accum28_asm", asm)
        for _ in range(REPS):
            B.msm(d_s)
    ctx.dfree(d_s); B.free()
# the staged sort (k_ms_*) and the 17-bit windows over folded scalars: G1 2^22, unsplit
n = 1 << 22
d_k = ctx.to_device(rand_fr(n)); d_s = ctx.to_device(rand_fr(n))
d_b = v.fixed_base_mul(ctx, d_k, n, 1)
B = ctx.bases_from_device(d_b, n, 1)
ctx.dfree(d_b); ctx.dfree(d_k)
for _ in range(2):
    B.msm(d_s)
ctx.dfree(d_s); B.free()
lg = 22
d = ctx.to_device(rand_fr(1 << lg))
dom = v.EvaluationDomain(ctx, 1 << lg)
for _ in range(REPS + 1):
    dom.fft_device(d)
ctx.synchronize()
print("sq_workload done")
