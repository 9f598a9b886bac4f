"""Probe for the open finding of DESIGN.md 3.7: builds of the library with -DVSP_DIMSUM_PREFETCH_EXPERIMENT=<k> (k_dimsum_mixed keeps the
next bucket record in flight across the current addition) return wrong sums when EQUAL points meet in the bucket reduction.
   make -C vote_saver_protocol_amd/csrc BUILD=build_pf1 OUT=../libvsp_hip_pf1.so EXTRA=-DVSP_DIMSUM_PREFETCH_EXPERIMENT=1
   VSP_LIB_PATH=vote_saver_protocol_amd/libvsp_hip_pf1.so [OPTS=msm_fp28=0] python tools/dimsum_prefetch_probe.py
Measured (round 3): k = 1 (the variant), 2 (s_waitcnt 0 before every addition) and 3 (unconditional next load) fail on "one point" and
"two points" inputs for every size tried but (34 points, 5-bit windows), on the 28-bit and on the 12 x 32-bit form; k = 5 (the same
prefetch, the addition called with its operands SWAPPED: tmp = cur; tmp += acc; acc = tmp) passes everything, as does the shipped loop;
k = 6 (no prefetch, the summand a named const object) passes; k = 7 (prefetch, explicit copies: sum = acc; sum += cur; acc = sum) fails.
So: wrong exactly when the record loaded one iteration EARLIER is the addition's second (read-only) operand and stays live through it."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vote_saver_protocol_amd as v, cref, bls12_381 as o
from conftest import rand_fr_array, g1_limbs
ctx = v.Context(0)
for kv in [x for x in os.environ.get("OPTS", "").split(",") if x]:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
one = cref.g1_batch_mul_gen(rand_fr_array(1, seed=91))
many = cref.g1_batch_mul_gen(rand_fr_array(300, seed=93))
res = []
for label, mk in (("one point", lambda n: np.repeat(one, n, axis=0)), ("distinct", lambda n: many[:n].copy()), ("two points", lambda n: np.concatenate([np.repeat(one, n // 2, axis=0), np.repeat(many[:1], n - n // 2, axis=0)]))):
    for n, wb in ((34, 8), (34, 5), (200, 8), (200, 11), (300, 13)):
        bases = mk(n); ss = rand_fr_array(n, seed=92 + n)
        want = cref.msm_g1(bases, ss, mixed=True)
        ctx.set_option("msm_window_bits", wb)
        B = ctx.upload_bases(bases, 1); d_s = ctx.to_device(ss)
        got, _ = B.msm(d_s)
        res.append((label, n, wb, bool(np.array_equal(got, want))))
        B.free(); ctx.dfree(d_s)
print(os.environ.get("VSP_LIB_PATH", "shipped"), os.environ.get("OPTS", ""), [r for r in res if not r[3]] or "all ok")
