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
