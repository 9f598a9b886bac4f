"""CPU: mechanical check of the bound discipline of the 9 x 29-bit lazy Fr arithmetic the NTT butterflies run on
(vote_saver_protocol_amd/csrc/fr29.h, k_ntt29_pass in ntt.hip), with the constants parsed from the generated header:

 1. an exact limb-level model of vsp_mm29 (the column schedule of tools/gen_mont_asm.py body29) asserting that no column reaches
    2^64, and of the radix-2 / radix-4 butterflies, driven through whole transforms of random and of adversarial data -- every
    intermediate is checked (no 32-bit limb wraps in either direction, every subtrahend is dominated by the 2r constant, every
    product operand stays below 2^261) and the result is the plain big-integer DFT;
 2. a worst-case propagation (value bound and per-limb bound) through the butterfly, showing that 11 radix-4 steps from a canonical
    input (the deepest transform the pass plan allows is 2^28: 14 steps -- checked too) keep every constraint.
No GPU needed; tests/test_gpu_ntt.py and tests/test_gpu_domain.py compare the kernels with the oracle's transforms bit for bit."""
import os
import random
import re

import pytest

import bls12_381 as o
from conftest import ROOT

R = o.R
W, N = 29, 9
MASK = (1 << W) - 1
RP = 1 << (W * N)                      # R' = 2^261


def _consts():
    text = open(os.path.join(ROOT, "vote_saver_protocol_amd", "csrc", "mont_asm_gfx950.h")).read()
    return {name: [int(x.strip().rstrip("u"), 16) for x in body.split(",")]
            for name, body in re.findall(r"static constexpr uint32_t (FR29_\w+)\[9\] = \{([^}]*)\};", text)}


K = _consts()
RL, K2 = K["FR29_R"], K["FR29_K2_L1"]


def val(l):
    return sum(x << (W * i) for i, x in enumerate(l))


def tight(v):
    assert 0 <= v < 1 << (W * (N - 1) + 32)
    return [(v >> (W * i)) & MASK for i in range(N - 1)] + [v >> (W * (N - 1))]


class Bound(AssertionError):
    pass


def need(c, msg):
    if not c:
        raise Bound(msg)


def test_constants():
    assert val(RL) == R and RL == tight(R) and RL[0] == 1                 # r = 1 mod 2^29: m_k is a negation
    assert (-pow(R, -1, 1 << W)) % (1 << W) == MASK
    assert val(K["FR29_ONE"]) == RP % R and val(K["FR29_R2"]) == RP * RP % R
    assert val(K2) == 2 * R and all(x >= MASK for x in K2[:-1]) and all(x < 1 << 31 for x in K2)
    assert val(K["FR29_K4_L1"]) == 4 * R
    # the top limb of 2r (after lending one unit) dominates the top limb of anything below 1.9 r
    assert K2[-1] >= (19 * R // 10) >> (W * (N - 1))


def mm29(a, b):
    for x in (a, b):
        need(all(0 <= t < 1 << 32 for t in x), "operand limb outside 32 bits")
    need(val(a) < RP and val(b) < RP, "operand not below 2^261")
    m, r, acc = [0] * N, [0] * N, 0
    for k in range(2 * N - 1):
        for i in range(max(0, k - N + 1), min(k, N - 1) + 1):
            acc += a[i] * b[k - i]
        for i in (range(0, k) if k < N else range(k - N + 1, N)):
            acc += m[i] * RL[k - i]
        if k < N:
            m[k] = ((-acc) & 0xFFFFFFFF) & MASK                             # v_sub_u32 tmp, 0, lo ; v_and_b32
            acc += m[k] * RL[0]
            need(acc & MASK == 0, "Montgomery column not cleared")
        else:
            r[k - N] = acc & MASK
        need(acc < 1 << 64, "column %d overflows the 64-bit accumulator" % k)
        acc >>= W
    need(acc < 1 << 32, "top limb overflows")
    r[N - 1] = acc
    need(val(r) * RP == val(a) * val(b) + val(m) * R, "not (a b + m r) / R'")
    need(val(r) < val(a) * val(b) // RP + R + 1 and all(x <= MASK for x in r[:-1]), "output not tight / above a b / R' + r")
    return r


def add29(a, b):
    r = [x + y for x, y in zip(a, b)]
    need(all(t < 1 << 32 for t in r), "sum wraps")
    return r


def sub29(a, b):
    r = []
    for i in range(N):
        need(a[i] + K2[i] < 1 << 32, "a + 2r wraps")
        need(a[i] + K2[i] - b[i] >= 0, "a + 2r - b borrows in limb %d" % i)
        r.append(a[i] + K2[i] - b[i])
    return r


def norm29(a):
    r, c = [], 0
    for i in range(N - 1):
        t = a[i] + c
        need(t < 1 << 32, "carry pass wraps")
        r.append(t & MASK); c = t >> W
    need(a[N - 1] + c < 1 << 32, "carry pass wraps the top limb")
    return r + [a[N - 1] + c]


def csub29(v):
    need(val(v) < 2 * R, "conditional subtraction of a value >= 2r")
    return tight(val(v) - R) if val(v) >= R else list(v)


def tw(x):
    return tight(x * RP % R)


def radix4(x, w1, w2, w3, s_zero=False):
    x0, x1, x2, x3 = x
    if not s_zero:
        x1, x3 = mm29(x1, w1), mm29(x3, w1)
    a0, a1, a2, a3 = add29(x0, x1), sub29(x0, x1), add29(x2, x3), sub29(x2, x3)
    a2, a3 = mm29(a2, w2), mm29(a3, w3)
    return [norm29(add29(a0, a2)), norm29(add29(a1, a3)), norm29(sub29(a0, a2)), norm29(sub29(a1, a3))]


def model_ntt(vals, log_n):
    """radix-2 decimation in time exactly as k_ntt29_pass steps it (an odd stage count starts with one radix-2 stage, then radix-4
    steps), one 'pass' over the whole array, values lazy throughout, canonical at the end through the product with the Montgomery one"""
    n = 1 << log_n
    omega = o.fr_root_of_unity(log_n)
    a = [tight(vals[int(format(i, "0%db" % log_n)[::-1], 2)] if log_n else vals[i]) for i in range(n)]
    t = 0
    if log_n & 1:
        for q in range(n // 2):
            u, v = a[2 * q], a[2 * q + 1]
            a[2 * q], a[2 * q + 1] = norm29(add29(u, v)), norm29(sub29(u, v))
        t = 1
    while t < log_n:
        h = 1 << t
        for q in range(n // 4):
            mid_lo = q & (h - 1)
            mid0 = ((q >> t) << (t + 2)) | mid_lo
            e = [mid0, mid0 + h, mid0 + 2 * h, mid0 + 3 * h]
            w1 = tw(pow(omega, mid_lo << (log_n - 1 - t), R))
            w2 = tw(pow(omega, mid_lo << (log_n - 2 - t), R))
            w3 = tw(pow(omega, (mid_lo + h) << (log_n - 2 - t), R))
            out = radix4([a[i] for i in e], w1, w2, w3, s_zero=(t == 0))
            for i, v in zip(e, out):
                a[i] = v
        t += 2
    worst = max(val(v) for v in a)
    return [val(csub29(mm29(v, K["FR29_ONE"]))) for v in a], worst


@pytest.mark.parametrize("log_n", [1, 2, 3, 6, 7])
def test_exact_model_transform_matches_the_dft(log_n):
    rng = random.Random(log_n)
    n = 1 << log_n
    for kind in ("random", "max"):
        vals = [rng.randrange(R) for _ in range(n)] if kind == "random" else [R - 1] * n
        got, worst = model_ntt(vals, log_n)
        assert got == o.dft_naive(vals, o.fr_root_of_unity(log_n))
        assert worst < (1 + 4 * ((log_n + 1) // 2) + 1) * R                 # V + 4r per step (one more for the radix-2 stage)


def test_product_column_bound_at_the_loosest_operands():
    """the largest limbs a data operand reaches inside a butterfly: a3 = x2 + 2r - x3, below 2^29 + 2^30 per limb, against a tight twiddle"""
    a = [(1 << 29) - 1 + K2[i] for i in range(N - 1)] + [(40 * R) >> (W * (N - 1))]
    assert max(a[:-1]) < 1 << 31 and val(a) < RP
    mm29(a, tight(R - 1))
    mm29(a, [MASK] * (N - 1) + [(R - 1) >> (W * (N - 1))])                # every twiddle limb at its maximum
    with pytest.raises(Bound):
        mm29([0xFFFFFFFF] * N, [MASK] * N)


# ------------------------------------------------------------------------------------------------ worst-case propagation
class B:
    def __init__(self, v, l):
        self.v, self.l = int(v), list(l)          # value < v, limb i < l[i]

    @staticmethod
    def tight(v):
        v = int(v)
        return B(v, [1 << W] * (N - 1) + [(v >> (W * (N - 1))) + 1])


def b_mul(a, b):
    worst, carry = 0, 0
    for k in range(2 * N - 1):
        col = carry
        for i in range(max(0, k - N + 1), min(k, N - 1) + 1):
            col += (a.l[i] - 1) * (b.l[k - i] - 1)
        for i in (range(0, k + 1) if k < N else range(k - N + 1, N)):
            col += MASK * RL[k - i]
        worst = max(worst, col); carry = col >> W
    need(worst < 1 << 64, "worst-case column sum reaches 2^64")
    need(a.v <= RP and b.v <= RP, "product operand may reach 2^261")
    return B.tight(a.v * b.v // RP + R + 1)


def b_add(a, b):
    l = [x + y - 1 for x, y in zip(a.l, b.l)]
    need(all(x <= 1 << 32 for x in l), "sum may wrap")
    return B(a.v + b.v, l)


def b_sub(a, b):
    need(all(K2[i] >= b.l[i] - 1 for i in range(N)), "2r does not dominate the subtrahend's limbs")
    need(all(a.l[i] - 1 + K2[i] < 1 << 32 for i in range(N)), "a + 2r may wrap")
    return B(a.v + 2 * R, [a.l[i] + K2[i] for i in range(N)])


def b_norm(a):
    need(all(x <= (1 << 32) - 8 for x in a.l), "carry pass may wrap")
    return B.tight(a.v)


@pytest.mark.parametrize("steps", [11, 14])
def test_bounds_are_inductive_over_a_whole_transform(steps):
    """from a canonical input (or a coset-shifted one: a product output below 1.02 r) through `steps` radix-4 steps -- 11 for 2^22,
    14 for the largest supported transform 2^28 -- every constraint holds and the value stays below 2^261 = 70.4 r"""
    w = B.tight(R)                                                        # twiddles are canonical
    V = B.tight(R + R // 50)
    for step in range(steps):
        x1 = b_mul(V, w); assert x1.v < 1.9 * R
        a0, a1 = b_add(V, x1), b_sub(V, x1)
        p2, p3 = b_mul(a0, w), b_mul(a1, w); assert p3.v < 1.9 * R
        outs = [b_norm(b_add(a0, p2)), b_norm(b_add(a1, p3)), b_norm(b_sub(a0, p2)), b_norm(b_sub(a1, p3))]
        V = B.tight(max(x.v for x in outs))
        assert V.v <= (103 + 400 * (step + 1)) * R // 100                 # + 4r per step
    assert V.v < RP
    # leaving the lazy domain: the product with the scale (or the Montgomery one) is below 2r, one conditional subtraction follows
    assert b_mul(V, w).v < 2 * R


def test_the_model_notices_a_broken_bound():
    with pytest.raises(Bound):
        b_sub(B.tight(R), B.tight(3 * R))                                 # a subtrahend that is not a product output
    with pytest.raises(Bound):
        sub29(tight(5), tight(2 * R + 5))
    with pytest.raises(Bound):
        b_mul(B(RP, [1 << 32] * N), B.tight(R))


def test_fused_pointwise_load_of_witness_map_stays_inside_the_lazy_domain():
    """round 4: the first pass of witness_map's last transform computes its own input, (a b - c) / 2^261, from canonical a, b, c
    (k_ntt29_pass, NttPass29Args.fuse_b / fuse_c): two products of tight operands, the lazy subtraction (its subtrahend IS a product
    output), one carry pass.  Exact model on edge and random values against plain integers; then the worst-case propagation of a whole
    transform starting from that bound (3.03 r instead of a canonical input's 1.02 r) -- 11 steps (2^22) and 14 (2^28) keep every constraint."""
    rng = random.Random(29)
    one = [1] + [0] * (N - 1)
    rinv = pow(RP, -1, R)
    edge = [0, 1, 2, R - 1, R - 2, (R - 1) // 2, (1 << 254) - 1, (1 << 29) - 1, 1 << 29]
    cases = [(a, b, c) for a in edge for b in edge for c in edge] + [(rng.randrange(R), rng.randrange(R), rng.randrange(R)) for _ in range(300)]
    worst = 0
    for a, b, c in cases:
        v = norm29(sub29(mm29(tight(a), tight(b)), mm29(tight(c), one)))
        assert val(v) % R == (a * b - c) * rinv % R
        assert all(x <= MASK for x in v[:-1])
        worst = max(worst, val(v))
    assert worst < 3.03 * R
    w = B.tight(R)
    for steps in (11, 14):
        V = B.tight(int(3.03 * R))
        for step in range(steps):
            x1 = b_mul(V, w); assert x1.v < 1.9 * R
            a0, a1 = b_add(V, x1), b_sub(V, x1)
            p2, p3 = b_mul(a0, w), b_mul(a1, w); assert p3.v < 1.9 * R
            outs = [b_norm(b_add(a0, p2)), b_norm(b_add(a1, p3)), b_norm(b_sub(a0, p2)), b_norm(b_sub(a1, p3))]
            V = B.tight(max(x.v for x in outs))
        assert V.v < RP and b_mul(V, w).v < 2 * R
