"""GPU: the kernels' Montgomery arithmetic (hand-scheduled v_mad_u64_u32 product, add, sub, inverse) against the
golden field vectors (edge values 0, 1, p-1, ...) and against python big-int arithmetic on random values."""
import ctypes as C

import numpy as np
import pytest

import bls12_381 as o
from conftest import fr_ints_fast, load_golden

pytestmark = pytest.mark.gpu


def limbs_arr(vals, nl):
    return np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nl)] for v in vals], dtype=np.uint64)


def ints(arr, nl):
    a = np.asarray(arr).reshape(-1, nl)
    v = a[:, nl - 1].astype(object)
    for k in range(nl - 2, -1, -1):
        v = (v << 64) | a[:, k].astype(object)
    return v.tolist()


def run(ctx, field, op, a, b):
    out = np.zeros_like(a)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    ctx.check(ctx.lib.vsp_selftest_field(ctx.h, field, op, p(a), p(b), p(out), a.shape[0]))
    return out


@pytest.mark.parametrize("field,mod,nl,key", [(0, o.P, 6, "fp"), (1, o.R, 4, "fr")])
def test_field_golden_and_random(ctx, field, mod, nl, key):
    g = load_golden("field.json")[key]
    A = [int(c["a"], 16) for c in g]; B = [int(c["b"], 16) for c in g]
    rng = np.random.default_rng(5 + field)
    for _ in range(20000):
        A.append(int.from_bytes(rng.bytes(48), "little") % mod); B.append(int.from_bytes(rng.bytes(48), "little") % mod)
    # structured values that stress carries: all-ones limbs, single high bits
    for k in range(0, 32 * 2 * nl, 7):
        A.append(((1 << k) - 1) % mod); B.append((mod - 1 - (1 << (k % 200))) % mod)
    a, b = limbs_arr(A, nl), limbs_arr(B, nl)
    assert ints(run(ctx, field, 0, a, b), nl) == [x * y % mod for x, y in zip(A, B)]
    assert ints(run(ctx, field, 1, a, b), nl) == [(x + y) % mod for x, y in zip(A, B)]
    assert ints(run(ctx, field, 2, a, b), nl) == [(x - y) % mod for x, y in zip(A, B)]
    assert ints(run(ctx, field, 3, a, b), nl) == [x * x % mod for x in A]
    assert ints(run(ctx, field, 4, a, b), nl) == [x * y % mod for x, y in zip(A, B)]
    n_inv = 300
    assert ints(run(ctx, field, 5, a[:n_inv], b[:n_inv]), nl) == [pow(x, mod - 2, mod) for x in A[:n_inv]]


def test_fp28_lazy_field_against_big_integers(ctx):
    """The 14 x 28-bit field of the G1 accumulation (carry-free product, lazy subtractions against redundant multiples of p):
    round trip, product, and each subtraction form used by the mixed addition, on edge values and random ones."""
    mod, nl = o.P, 6
    g = load_golden("field.json")["fp"]
    A = [int(c["a"], 16) for c in g]; B = [int(c["b"], 16) for c in g]
    edge = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (1 << 380) - 1, (1 << 364) - 1, (1 << 28) - 1, 1 << 28, (1 << 56) - 1, mod >> 1, 3]
    for x in edge:
        for y in edge:
            A.append(x % mod); B.append(y % mod)
    rng = np.random.default_rng(28)
    for _ in range(20000):
        A.append(int.from_bytes(rng.bytes(48), "little") % mod); B.append(int.from_bytes(rng.bytes(48), "little") % mod)
    for k in range(0, 381, 3):                         # limbs of all ones / single bits around every 28-bit boundary
        A.append(((1 << k) - 1) % mod); B.append((mod - (1 << (k % 377))) % mod)
    a, b = limbs_arr(A, nl), limbs_arr(B, nl)
    assert ints(run(ctx, 0, 6, a, b), nl) == A
    assert ints(run(ctx, 0, 7, a, b), nl) == [x * y % mod for x, y in zip(A, B)]
    assert ints(run(ctx, 0, 8, a, b), nl) == [(x - y) * (x - y) % mod for x, y in zip(A, B)]
    assert ints(run(ctx, 0, 9, a, b), nl) == [(x - 3 * y) % mod for x, y in zip(A, B)]
    assert ints(run(ctx, 0, 10, a, b), nl) == [x * (y - x) % mod for x, y in zip(A, B)]
    assert ints(run(ctx, 0, 11, a, b), nl) == [(-y) * x % mod for x, y in zip(A, B)]
    assert ints(run(ctx, 0, 12, a, b), nl) == [(x * (y - x) - y * x) % mod for x, y in zip(A, B)]


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("form", [0, 1])
def test_full_addition_of_bucket_sums_lane_by_lane(ctx, group, form):
    """xyzz_add on its own (vsp_selftest_xyzz_add), 12 x 32-bit and 14 x 28-bit lazy form, G1 and the lane-pair G2: ordinary sums,
    the SAME point in two different representations (the doubling the round-4 kernels finish in the 28-bit form), a point and its
    negative, infinity on either side -- interleaved so that the lanes of one wave take different paths -- against the group law
    in plain integers."""
    rng = np.random.default_rng(40 + group)
    G, F = (o.G1, None) if group == 1 else (o.G2, o.Fp2Ops)
    nl = 6 * group
    n = 256
    pts = [G.mul(G.gen, int(rng.integers(1, 1 << 62))) for _ in range(24)]

    def fmul(a, b):
        return a * b % o.P if group == 1 else F.mul(a, b)

    def rep(pt):                                            # a random representation (x z^2, y z^3, z^2, z^3)
        if pt is None:
            return None
        z = int.from_bytes(rng.bytes(47), "little") + 1
        z = z if group == 1 else (z, int.from_bytes(rng.bytes(47), "little"))
        zz = fmul(z, z); zzz = fmul(zz, z)
        return (fmul(pt[0], zz), fmul(pt[1], zzz), zz, zzz)

    def words(r):
        if r is None:
            return [0] * (4 * nl)
        out = []
        for c in r:
            for comp in ((c,) if group == 1 else c):
                out += [(comp >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]
        return out

    A, B, want = [], [], []
    for i in range(n):
        p, q = pts[i % 24], pts[(7 * i + 3) % 24]
        kind = i % 8
        if kind in (1, 5):
            q = p                                           # doubling: equal points, different representations
        elif kind == 2:
            q = G.neg(p)                                    # cancellation
        elif kind == 3:
            p = None
        elif kind == 4:
            q = None
        elif kind == 6 and i % 16 == 6:
            p = q = None
        ra, rb = rep(p), rep(q)
        if kind == 5:
            rb = ra                                         # doubling of the very same record
        A.append(words(ra)); B.append(words(rb)); want.append(G.add(p, q))
    a, b = np.array(A, dtype=np.uint64), np.array(B, dtype=np.uint64)
    out = np.zeros_like(a)
    p_ = lambda x: x.ctypes.data_as(C.c_void_p)
    ctx.check(ctx.lib.vsp_selftest_xyzz_add(ctx.h, group, form, p_(a), p_(b), p_(out), n))
    bad = []
    for i in range(n):
        vals = ints(out[i], 6)
        comps = vals if group == 1 else [(vals[2 * k], vals[2 * k + 1]) for k in range(4)]
        X, Y, ZZ, ZZZ = comps
        zero = (ZZ == 0) if group == 1 else (ZZ == (0, 0))
        if zero:
            got = None
        elif group == 1:
            got = (X * pow(ZZ, -1, o.P) % o.P, Y * pow(ZZZ, -1, o.P) % o.P)
        else:
            got = (F.mul(X, F.inv(ZZ)), F.mul(Y, F.inv(ZZZ)))
        if got != want[i]:
            bad.append((i, i % 8))
    assert not bad, bad[:16]
