"""Companion of tools/dimsum_prefetch_probe.py: runs configurations whose bucket sums are multiples of one or two points (generic
12 x 32-bit form) on a build with -DVSP_DEBUG_DUMP, reads the dumped bucket sums and digit sums, recomputes every digit sum from the bucket
sums with big integers and lists the sums the kernel got wrong (and what their summands were) -- a check of the bucket reduction's
INTERMEDIATE values, not only of the final point.
   VSP_LIB_PATH=vote_saver_protocol_amd/libvsp_hip_dd1.so python tools/dimsum_dump_check.py"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
d = tempfile.mkdtemp(); os.environ["VSP_DEBUG_DUMP_DIR"] = d
import vote_saver_protocol_amd as v, cref, bls12_381 as o
from conftest import rand_fr_array
ctx = v.Context(0)
ctx.set_option("msm_fp28", 0)
one = cref.g1_batch_mul_gen(rand_fr_array(1, seed=91))
two = cref.g1_batch_mul_gen(rand_fr_array(2, seed=93))
P = o.P; RINV = pow(1 << 384, -1, P)
def fp(words): return (sum(int(w) << (32 * i) for i, w in enumerate(words)) * RINV) % P
def point(r):                       # r: 48 uint32 words X | Y | ZZ | ZZZ  ->  affine or None
    X, Y, ZZ, ZZZ = fp(r[0:12]), fp(r[12:24]), fp(r[24:36]), fp(r[36:48])
    if ZZ == 0: return None
    return (X * pow(ZZ, -1, P) % P, Y * pow(ZZZ, -1, P) % P)
G1 = o.G1
total_bad = 0
for label, n, wb in (("one point", 34, 8), ("one point", 200, 8), ("two points", 200, 11), ("one point", 1500, 13), ("two points", 600, 10)):
    ctx.set_option("msm_window_bits", wb)
    bases = np.repeat(one, n, axis=0) if label == "one point" else np.concatenate([np.repeat(two[:1], n // 2, axis=0), np.repeat(two[1:], n - n // 2, axis=0)])
    ss = rand_fr_array(n, seed=92 + n)
    want = cref.msm_g1(bases, ss, mixed=True)
    B = ctx.upload_bases(bases, 1); d_s = ctx.to_device(ss)
    got, _ = B.msm(d_s)
    B.free(); ctx.dfree(d_s)
    c, W, Wr, Bk, q0, q1, q2, rec = [int(x) for x in open(d + "/geom.txt").read().split()]
    bk = np.fromfile(d + "/buckets.bin", dtype=np.uint32).reshape(-1, rec // 4)
    dm = np.fromfile(d + "/dims.bin", dtype=np.uint32).reshape(-1, rec // 4)
    n0, n1, n2 = 1 << q0, 1 << q1, (1 << q2) if q2 else 0
    per_w = n0 + n1 + n2
    bad = 0
    for w in range(Wr):
        pts = [point(bk[w * Bk + b]) for b in range(Bk)]
        if all(p is None for p in pts): continue
        for dgt, nd, base in ((0, n0, 0), (1, n1, n0), (2, n2, n0 + n1)):
            for val in range(nd):
                if dgt == 0: idxs = [(i << q0) | val for i in range(Bk >> q0)]
                elif dgt == 1: idxs = [((i >> q0) << (q1 + q0)) | (val << q0) | (i & (n0 - 1)) for i in range(Bk >> q1)]
                else: idxs = [(val << (q1 + q0)) | i for i in range(Bk >> q2)]
                acc = None
                for b in idxs: acc = G1.add(acc, pts[b])
                gotp = point(dm[w * per_w + base + val])
                if gotp != acc:
                    bad += 1
                    if bad <= 3:
                        live = [b for b in idxs if pts[b] is not None]
                        print(f"  window {w} digit {dgt} value {val}: WRONG; non-empty summands at buckets {live}, {len({pts[b] for b in live})} distinct points; kernel gave {'infinity' if gotp is None else 'a point'}, expected {'infinity' if acc is None else 'a point'}")
    total_bad += bad
    print(f"{label}, n = {n}, c = {c} (digits {q0}+{q1}+{q2}): result correct {bool(np.array_equal(got, want))}, digit sums wrong {bad} of {Wr * per_w}")
print("TOTAL wrong digit sums:", total_bad)
