"""round 4 diagnostic: the known-answer check's input (pairs of equal / opposite points under equal scalars) through option combinations,
28-bit kernels forced without the check (msm_fp28 = 2), against the C oracle"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vote_saver_protocol_amd as v, cref, bls12_381 as o
from conftest import rand_fr_array, g1_limbs
n = int(os.environ.get("N", "4096"))
ks = rand_fr_array(n, seed=7); ss = rand_fr_array(n, seed=8)
R = o.R
def ival(row): return sum(int(row[j]) << (64 * j) for j in range(4))
def limbs(x): return [(x >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
MODE = os.environ.get("MODE", "pairs")          # pairs: equal and opposite points under equal scalars; equal: only equal; opposite: only opposite; distinct
for i in range(1, n):
    if MODE == "distinct" or (MODE == "equal" and i % 16 == 9) or (MODE == "opposite" and i % 16 == 5): continue
    if i % 16 == 5: ks[i] = ks[i - 1]; ss[i] = ss[i - 1]
    elif i % 16 == 9: ks[i] = limbs(R - ival(ks[i - 1])); ss[i] = ss[i - 1]
for i in range(n):
    if i % 7 == 3 and MODE == "pairs": ss[i] = 0; ss[i, 0] = i & 1
bases = cref.g1_batch_mul_gen(ks)
want = cref.msm_g1(bases, ss, mixed=True)
for fp28 in (2,):
    for split in (0,):
        for dimbits in (0, 1):
            for glv in (0,):
                ctx = v.Context(0)
                ctx.set_option("msm_fp28", fp28); ctx.set_option("msm_glv", glv)
                if split: ctx.set_option("msm_split", split)
                if dimbits >= 0: ctx.set_option("msm_dimbits", dimbits)
                B = ctx.upload_bases(bases, 1); d_s = ctx.to_device(ss)
                got, _ = B.msm(d_s)
                print("fp28", fp28, "split", split, "dimbits", dimbits, "glv", glv, "c", int(ctx.stat("msm_window_bits")), "OK" if np.array_equal(got, want) else "WRONG", flush=True)
                if ctx.stat("dbg_eqx_total") >= 0 and os.environ.get("DBG"):
                    print("   equal-x taken", ctx.stat("dbg_eqx_total"), "doubling", ctx.stat("dbg_eqx_doubling"), "cancel", ctx.stat("dbg_eqx_cancel"))
                    ops = [int(ctx.stat("dbg_eqx_op%d" % i)) for i in range(120)]
                    P = o.P; RP = 1 << 392; RINV = pow(RP, -1, P)
                    def val(l): return sum(x << (28 * i) for i, x in enumerate(l))
                    names = ["acc.X", "acc.Y", "acc.ZZ", "acc.ZZZ", "q.X", "q.Y", "q.ZZ", "q.ZZZ"]
                    vals = [val(ops[14 * k:14 * k + 14]) for k in range(8)]
                    for nm, x in zip(names, vals): print("   ", nm, "limbs ok" if all(l < (1 << 28) for l in ops[14 * names.index(nm):14 * names.index(nm) + 13]) else "LOOSE", "value/p %.3f" % (x / P), "residue", hex(x * RINV % P)[:20])
                    print("    block", ops[112], "thread", ops[113], "grid", ops[114], "blockDim", ops[115])
                    def aff(X, Y, ZZ, ZZZ):
                        X, Y, ZZ, ZZZ = (t * RINV % P for t in (X, Y, ZZ, ZZZ))
                        return None if ZZ == 0 else (X * pow(ZZ, -1, P) % P, Y * pow(ZZZ, -1, P) % P)
                    a, b = aff(*vals[:4]), aff(*vals[4:])
                    print("    acc on curve", a is None or (a[1] * a[1] - a[0] ** 3 - 4) % P == 0, "q on curve", b is None or (b[1] * b[1] - b[0] ** 3 - 4) % P == 0, "same point", a == b, "opposite", a is not None and b is not None and a[0] == b[0] and (a[1] + b[1]) % P == 0)
                B.free(); ctx.dfree(d_s); ctx.close()
