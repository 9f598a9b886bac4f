"""TEST INFRASTRUCTURE -- regenerates tests/golden/*.json from the pure-Python big-int oracle.

The reference tree holds no vectors for this path, so these fixtures are the
build's own: produced by oracle/bls12_381.py (independent of both the C restatement and the HIP code),
committed, and checked by the CPU tests (C oracle vs fixture) and the GPU tests (HIP path vs fixture).
Run:  python oracle/gen_golden.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bls12_381 as o  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def hx(v):
    return "%x" % v


def pt1(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def pt2(p):
    return None if p is None else [[hx(p[0][0]), hx(p[0][1])], [hx(p[1][0]), hx(p[1][1])]]


def main():
    os.makedirs(OUT, exist_ok=True)
    gen = o.splitmix64(20260101)

    def rfr():
        return o.rand_fr(gen)

    def rfp():
        return (rfr() * rfr() + rfr()) % o.P

    # ---- field vectors
    fp_edge = [0, 1, 2, o.P - 1, o.P - 2, (o.P - 1) // 2, (1 << 380)]
    fr_edge = [0, 1, 2, o.R - 1, o.R - 2, (o.R - 1) // 2, (1 << 254)]
    fp_cases = [(a, b) for a in fp_edge for b in fp_edge[:4]] + [(rfp(), rfp()) for _ in range(24)]
    fr_cases = [(a, b) for a in fr_edge for b in fr_edge[:4]] + [(rfr(), rfr()) for _ in range(24)]
    field = {
        "fp": [{"a": hx(a), "b": hx(b), "mul": hx(a * b % o.P), "add": hx((a + b) % o.P), "sub": hx((a - b) % o.P),
                "inv_a": hx(pow(a, o.P - 2, o.P))} for a, b in fp_cases],
        "fr": [{"a": hx(a), "b": hx(b), "mul": hx(a * b % o.R), "add": hx((a + b) % o.R), "sub": hx((a - b) % o.R),
                "inv_a": hx(pow(a, o.R - 2, o.R))} for a, b in fr_cases],
    }
    fp2_cases = [((rfp(), rfp()), (rfp(), rfp())) for _ in range(12)] + [((0, 1), (0, 1)), ((o.P - 1, 0), (5, o.P - 1))]
    field["fp2"] = [{"a": [hx(a[0]), hx(a[1])], "b": [hx(b[0]), hx(b[1])],
                     "mul": [hx(x) for x in o.Fp2Ops.mul(a, b)], "sqr_a": [hx(x) for x in o.Fp2Ops.sqr(a)],
                     "inv_a": [hx(x) for x in o.Fp2Ops.inv(a)]} for a, b in fp2_cases]
    json.dump(field, open(os.path.join(OUT, "field.json"), "w"), indent=0)

    # ---- curve vectors
    curve = {}
    for name, cur, enc in (("g1", o.G1, pt1), ("g2", o.G2, pt2)):
        P1 = cur.mul(cur.gen, rfr()); P2 = cur.mul(cur.gen, rfr())
        k = rfr()
        curve[name] = {
            "gen": enc(cur.gen), "P1": enc(P1), "P2": enc(P2),
            "P1_plus_P2": enc(cur.add(P1, P2)), "dbl_P1": enc(cur.add(P1, P1)),
            "P1_minus_P1": enc(cur.add(P1, cur.neg(P1))), "k": hx(k), "k_P1": enc(cur.mul(P1, k)),
            "r_minus_1_gen": enc(cur.mul(cur.gen, o.R - 1)),
        }
    json.dump(curve, open(os.path.join(OUT, "curve.json"), "w"), indent=0)

    # ---- MSM vectors: bases k_i*G, scalars with edge cases (0, 1, r-1, duplicates, inverse points, infinity)
    msm = []
    for name, cur, enc in (("g1", o.G1, pt1), ("g2", o.G2, pt2)):
        for n in ((1, 2, 3, 17, 64) if name == "g1" else (1, 3, 17)):
            ks = [rfr() for _ in range(n)]
            ss = [rfr() for _ in range(n)]
            pts = [cur.mul(cur.gen, k) for k in ks]
            if n >= 17:
                ss[0] = 0; ss[1] = 1; ss[2] = o.R - 1; ss[3] = 1
                pts[5] = pts[4]                       # duplicate base
                pts[7] = cur.neg(pts[6]); ss[7] = ss[6]   # inverse pair with equal scalars: cancels
                pts[8] = None                         # infinity base
                ss[9] = ss[10]                        # equal scalars
            res = cur.msm_naive(pts, ss)
            msm.append({"group": name, "n": n, "bases": [enc(p) for p in pts], "scalars": [hx(s) for s in ss], "result": enc(res)})
    json.dump(msm, open(os.path.join(OUT, "msm.json"), "w"), indent=0)

    # ---- NTT vectors (n in 2, 4, 8 against the O(n^2) DFT as well; 1024 by the radix-2 routine)
    nttv = []
    for n in (1, 2, 4, 8, 1024):
        a = [rfr() for _ in range(n)]
        if n >= 8:
            a[0] = 0; a[1] = o.R - 1
        lg = n.bit_length() - 1
        fwd = o.ntt(a)
        if n <= 8:
            assert fwd == o.dft_naive(a, o.fr_root_of_unity(lg))
        nttv.append({"n": n, "input": [hx(x) for x in a], "fft": [hx(x) for x in fwd],
                     "inverse_fft": [hx(x) for x in o.ntt(a, inverse=True)],
                     "coset_fft_g7": [hx(x) for x in o.ntt(a, coset=7)],
                     "inverse_coset_fft_g7": [hx(x) for x in o.ntt(a, inverse=True, coset=7)]})
    json.dump(nttv, open(os.path.join(OUT, "ntt.json"), "w"), indent=0)
    print("golden fixtures written to", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
