"""round 4 diagnostic: window results (k_dimweight's output) of the same multi-exponentiation from two builds, record by record"""
import sys, os, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import vote_saver_protocol_amd as v, cref
    from conftest import rand_fr_array
    n = 4096
    bases = cref.g1_batch_mul_gen(rand_fr_array(n, seed=7)); ss = rand_fr_array(n, seed=8)
    ctx = v.Context(0)
    ctx.set_option("msm_fp28", 2); ctx.set_option("msm_glv", 0); ctx.set_option("msm_dimbits", 0)
    B = ctx.upload_bases(bases, 1); d_s = ctx.to_device(ss)
    got, _ = B.msm(d_s)
    print("result", "OK" if np.array_equal(got, cref.msm_g1(bases, ss, mixed=True)) else "WRONG")
    sys.exit(0)
outs = {}
for tag in ("x0d", "x0d", "x3d"):
    d = os.path.join(ROOT, "gpurun_out", "dump_" + tag + ("_b" if tag in outs else "")); os.makedirs(d, exist_ok=True)
    env = dict(os.environ, VSP_LIB_PATH=os.path.join(ROOT, "vote_saver_protocol_amd", "libvsp_hip_%s.so" % tag), VSP_DEBUG_DUMP_DIR=d)
    print(tag, subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip())
    outs[tag + ("_b" if tag in outs else "")] = np.fromfile(os.path.join(d, "winres.bin"), dtype=np.uint32).reshape(-1, 4, 4, 12)      # [window][record D0 D1 D2 Tot][X Y ZZ ZZZ][12 limbs]
a, a2, b = outs["x0d"], outs["x0d_b"], outs["x3d"]
print("windows", a.shape[0], "x0d run twice identical:", bool(np.array_equal(a, a2)))
import bls12_381 as o
P = o.P; RI = pow(1 << 384, -1, P)
def aff(rec):
    X, Y, ZZ, ZZZ = (sum(int(x) << (32 * i) for i, x in enumerate(c)) * RI % P for c in rec)
    return None if ZZ == 0 else (X * pow(ZZ, -1, P) % P, Y * pow(ZZZ, -1, P) % P)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
for w in range(a.shape[0]):
    for r in range(4):
        pa, pb = aff(a[w, r]), aff(b[w, r])
        if pa != pb:
            on = pa is None or (pa[1] * pa[1] - pa[0] ** 3 - 4) % P == 0
            print("window", w, "record", ("D0", "D1", "D2", "Tot")[r], "differs; the wrong build's value is", "a curve point" if on else "NOT on the curve", "infinity" if pa is None else "")
