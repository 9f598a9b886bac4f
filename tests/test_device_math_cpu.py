"""CPU build of the kernel arithmetic: vote_saver_protocol_amd/csrc/{field,curve}.h compiled by g++ with the
very 32-bit-limb types the gfx950 kernels use (and the 64-bit-limb types of the library's host-side finishing
code), checked against the golden fixtures and the Python oracle.  No GPU, no HIP."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import bls12_381 as o
from conftest import I, L, ROOT, dec1, dec2, g1_limbs, g2_limbs, load_golden


@pytest.fixture(scope="module")
def mc(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("cpu_build") / "libmathchk.so")
    flags = os.environ.get("VSP_MATHCHK_FLAGS", "-O2").split()      # tests/test_sanitizers_cpu.py re-runs this file with sanitizer flags
    subprocess.check_call(["g++"] + flags + ["-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "cpu_build", "math_check.cpp")])
    return C.CDLL(so)


def call(lib, fn, *args, nout):
    out = np.zeros(nout, np.uint64)
    cargs = [a.ctypes.data_as(C.c_void_p) if isinstance(a, np.ndarray) else a for a in args]
    getattr(lib, fn)(*cargs, out.ctypes.data_as(C.c_void_p))
    return out


def test_field_ops_vs_golden(mc):
    g = load_golden("field.json")
    for c in g["fp"]:
        a, b = int(c["a"], 16), int(c["b"], 16)
        for fn in ("chk_fp_mul", "chk_hfp_mul"):
            assert I(call(mc, fn, L(a, 6), L(b, 6), nout=6)) == int(c["mul"], 16)
        assert I(call(mc, "chk_fp_add", L(a, 6), L(b, 6), nout=6)) == int(c["add"], 16)
        assert I(call(mc, "chk_fp_sub", L(a, 6), L(b, 6), nout=6)) == int(c["sub"], 16)
    for c in g["fp"][-6:]:
        a = int(c["a"], 16)
        assert I(call(mc, "chk_fp_inv", L(a, 6), nout=6)) == int(c["inv_a"], 16)
        assert I(call(mc, "chk_hfp_inv", L(a, 6), nout=6)) == int(c["inv_a"], 16)
    for c in g["fr"]:
        a, b = int(c["a"], 16), int(c["b"], 16)
        assert I(call(mc, "chk_fr_mul", L(a, 4), L(b, 4), nout=4)) == int(c["mul"], 16)
        assert I(call(mc, "chk_fr_mul_mixed", L(a, 4), L(b, 4), nout=4)) == int(c["mul"], 16)   # the NTT butterfly product
        assert I(call(mc, "chk_fr_add", L(a, 4), L(b, 4), nout=4)) == int(c["add"], 16)
        assert I(call(mc, "chk_fr_sub", L(a, 4), L(b, 4), nout=4)) == int(c["sub"], 16)
    for c in g["fr"][-6:]:
        a = int(c["a"], 16)
        assert I(call(mc, "chk_fr_inv", L(a, 4), nout=4)) == int(c["inv_a"], 16)
        assert I(call(mc, "chk_hfr_inv", L(a, 4), nout=4)) == int(c["inv_a"], 16)
    for c in g["fp2"]:
        a = np.concatenate([L(int(c["a"][0], 16), 6), L(int(c["a"][1], 16), 6)])
        b = np.concatenate([L(int(c["b"][0], 16), 6), L(int(c["b"][1], 16), 6)])
        r = call(mc, "chk_fp2_mul", a, b, nout=12)
        assert [I(r[:6]), I(r[6:])] == [int(x, 16) for x in c["mul"]]
        r = call(mc, "chk_fp2_sqr", a, nout=12)
        assert [I(r[:6]), I(r[6:])] == [int(x, 16) for x in c["sqr_a"]]
        r = call(mc, "chk_fp2_inv", a, nout=12)
        assert [I(r[:6]), I(r[6:])] == [int(x, 16) for x in c["inv_a"]]


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_xyzz_group_law_all_cases(mc, group):
    """mixed add / full add / double / negated add, incl. P+P, P+(-P), infinity operands, and the
    XYZZ<->Jacobian record conversion, for device (32-bit) and host (64-bit) limb types."""
    g = load_golden("curve.json")[group]
    if group == "g1":
        cur, lim, frm, dec, nl, fns = o.G1, g1_limbs, o.g1_from_limbs, dec1, 12, ("chk_g1_op", "chk_hg1_op")
    else:
        cur, lim, frm, dec, nl, fns = o.G2, g2_limbs, o.g2_from_limbs, dec2, 24, ("chk_g2_op", "chk_hg2_op")
    P1, P2 = dec(g["P1"]), dec(g["P2"])
    cases = [(P1, P2), (P1, P1), (P1, cur.neg(P1)), (P1, None), (None, P2), (None, None)]
    for fn in fns:
        assert frm(call(mc, fn, 0, lim(P1), lim(P2), nout=nl)) == dec(g["P1_plus_P2"])
        assert frm(call(mc, fn, 2, lim(P1), lim(P2), nout=nl)) == dec(g["dbl_P1"])
        for A, B in cases:
            assert frm(call(mc, fn, 0, lim(A), lim(B), nout=nl)) == cur.add(A, B)
            assert frm(call(mc, fn, 1, lim(A), lim(B), nout=nl)) == cur.add(A, B)
            assert frm(call(mc, fn, 2, lim(A), lim(B), nout=nl)) == cur.add(A, A)
            assert frm(call(mc, fn, 3, lim(A), lim(B), nout=nl)) == cur.add(A, cur.neg(B))


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_host_scalar_multiplications_agree_with_the_oracle(mc, group):
    """the prover's host-side multiplications (round 4): 4-bit windows for s * A and r * B1, a fixed-base table of 32 x 255 multiples for the
    multiples of delta -- against the bit-by-bit loop and the oracle's k * P, for scalars with zero bytes, all-ones bytes, 0, 1 and r - 1"""
    g = load_golden("curve.json")[group]
    if group == "g1":
        cur, lim, frm, dec, nl, fn = o.G1, g1_limbs, o.g1_from_limbs, dec1, 12, "chk_hg1_mul"
    else:
        cur, lim, frm, dec, nl, fn = o.G2, g2_limbs, o.g2_from_limbs, dec2, 24, "chk_hg2_mul"
    P = dec(g["P1"])
    gen = o.splitmix64(1234)
    ks = [0, 1, 2, 255, 256, o.R - 1, (1 << 255) - 19 - o.R, 0xFF00FF00FF00FF00FF00FF00FF00FF00, 0x0100000000000000000000000000000000000000000000000000000000000001]
    ks += [o.rand_fr(gen) for _ in range(3)]
    for k in ks:
        want = cur.mul(P, k % o.R) if k % o.R else None
        for mode in (0, 1, 2):
            assert frm(call(mc, fn, mode, lim(P), L(k, 4), nout=nl)) == want, (hex(k), mode)
