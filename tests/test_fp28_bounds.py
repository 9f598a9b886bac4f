"""CPU: mechanical check of the bound discipline of the 14 x 28-bit lazy field (vote_saver_protocol_amd/csrc/fp28.h).

The accumulation kernels k_accum28 (G1 and G2) never reduce fully: products leave values below a*b/2^392 + p, subtractions are
a + K - b limb by limb with no borrow, and three carry passes per mixed addition bring limbs back to 28 bits.  The header argues
the bounds in comments; a wrong bound would be a silent soundness bug in a prover.  This file checks them two ways, with the
constants parsed from the generated header the kernels are built from:

 1. an EXACT limb-level model of vsp_mm28 / vsp_mm28x2 (the column schedule of tools/gen_mont_asm.py body28) and of madd28 /
    madd28_g2 / the full addition over XYZZ<Fp28> that the G1 merges and bucket reduction run (fp28.h), which asserts at every step that no column accumulator reaches 2^64, no 32-bit limb wraps in either
    direction, every product input is below 2^388 -- driven over random accumulations AND over adversarial accumulator states
    placed at the stated worst case (X just below 9.5 p with loose-as-allowed limbs, Y limbs just below 2^30, ...), and which
    compares every result with the group law computed in plain big integers;
 2. a worst-case propagation (value bound, limb bound per position) through the same formulas, showing the invariants of the
    header are inductive: the bounds after a mixed addition are inside the bounds assumed before it.
No GPU needed; the GPU side of the same routines is covered by tests/test_gpu_field.py and the MSM parity tests."""
import os
import random
import re

import pytest

import bls12_381 as o
from conftest import ROOT

P = o.P
W, N = 28, 14
MASK = (1 << W) - 1
RP = 1 << (W * N)                      # R' = 2^392
INV28 = (-pow(P, -1, 1 << W)) % (1 << W)


def _consts():
    text = open(os.path.join(ROOT, "vote_saver_protocol_amd", "csrc", "mont_asm_gfx950.h")).read()
    out = {}
    for name, body in re.findall(r"static constexpr uint32_t (FP28_\w+)\[14\] = \{([^}]*)\};", text):
        out[name] = [int(x.strip().rstrip("u"), 16) for x in body.split(",")]
    return out


K = _consts()
PL = K["FP28_P"]


def val(l):
    return sum(x << (W * i) for i, x in enumerate(l))


def tight(v):
    assert 0 <= v < 1 << (W * 13 + 32)
    return [(v >> (W * i)) & MASK for i in range(N - 1)] + [v >> (W * (N - 1))]


def test_constants_are_what_the_header_says():
    assert val(PL) == P and PL == tight(P)
    assert val(K["FP28_ONE"]) == RP % P and val(K["FP28_R2"]) == RP * RP % P
    assert val(K["FP28_2P"]) == 2 * P and val(K["FP28_3P"]) == 3 * P
    for c in (8, 32, 64):
        for lend in (1, 4):
            k = K["FP28_K%d_L%d" % (c, lend)]
            assert val(k) == c * P and all(0 <= x < 1 << 32 for x in k)
            # every limb but the top dominates lend * 2^28 - 1 (a limb of the value being subtracted)
            assert all(x >= lend * (1 << W) - 1 for x in k[:-1])


# ------------------------------------------------------------------------------------------------ exact model of the asm routines
class Bound(AssertionError):
    pass


def need(c, msg):
    if not c:
        raise Bound(msg)


def mm28(a, b, c=None, d=None):
    """vsp_mm28 / vsp_mm28x2 exactly as tools/gen_mont_asm.py body28 schedules them (one 64-bit accumulator, 28-bit shifts)."""
    for x in (a, b) + ((c, d) if c is not None else ()):
        need(all(0 <= t < 1 << 32 for t in x), "operand limb outside 32 bits")
    m = [0] * N
    r = [0] * N
    acc = 0
    for k in range(2 * N - 1):
        for i in range(max(0, k - N + 1), min(k, N - 1) + 1):
            acc += a[i] * b[k - i]
            if c is not None:
                acc += c[i] * d[k - i]
        for i in (range(0, k) if k < N else range(k - N + 1, N)):
            acc += m[i] * PL[k - i]
        if k < N:
            m[k] = (((acc & 0xFFFFFFFF) * INV28) & 0xFFFFFFFF) & MASK
            acc += m[k] * PL[0]
        else:
            r[k - N] = acc & MASK
        need(acc < 1 << 64, "column %d overflows the 64-bit accumulator" % k)
        if k < N:
            need(acc & MASK == 0, "Montgomery column not cleared")
        acc >>= W
    need(acc < 1 << 32, "top limb of the product overflows")
    r[N - 1] = acc
    va, vb = val(a), val(b)
    tot = va * vb + (val(c) * val(d) if c is not None else 0)
    need(val(r) * RP == tot + val(m) * P, "product is not (a b + m p) / R'")
    need(val(r) < tot // RP + P + 1, "product output above a b / R' + p")
    need(all(x <= MASK for x in r[:-1]), "product output not tight")
    return r


def sq28(a):
    """vsp_sq28 (tools/gen_mont_asm.py body28(sqr=True)): off-diagonal products once against 2a, diagonal ones a_i * a_i.  Column by
    column the totals equal those of mm28(a, a), so the same m_k and the same output -- asserted here, with the tight-operand
    precondition the kernel relies on."""
    need(all(0 <= t <= MASK for t in a[:-1]) and a[-1] < 1 << 28, "squaring operand not tight")
    a2 = [2 * t for t in a]
    m = [0] * N
    r = [0] * N
    acc = 0
    for k in range(2 * N - 1):
        for i in range(max(0, k - N + 1), min(k, N - 1) + 1):
            if i < k - i:
                acc += a[i] * a2[k - i]
            elif i == k - i:
                acc += a[i] * a[i]
        for i in (range(0, k) if k < N else range(k - N + 1, N)):
            acc += m[i] * PL[k - i]
        if k < N:
            m[k] = (((acc & 0xFFFFFFFF) * INV28) & 0xFFFFFFFF) & MASK
            acc += m[k] * PL[0]
        else:
            r[k - N] = acc & MASK
        need(acc < 1 << 64, "column %d of the square overflows" % k)
        acc >>= W
    need(acc < 1 << 32, "top limb of the square overflows")
    r[N - 1] = acc
    need(r == mm28(a, a), "dedicated square differs from the product a * a")
    return r


def sub28(a, k, b):
    r = []
    for i in range(N):
        t = a[i] + k[i] - b[i]
        need(0 <= a[i] + k[i] < 1 << 32, "a + K wraps a 32-bit limb")
        need(t >= 0, "a + K - b borrows in limb %d" % i)
        r.append(t)
    return r


def neg28(k, b):
    return sub28([0] * N, k, b)


def norm28(a):
    r, c = [], 0
    for i in range(N - 1):
        t = a[i] + c
        need(t < 1 << 32, "carry pass wraps")
        r.append(t & MASK); c = t >> W
    need(a[N - 1] + c < 1 << 32, "carry pass wraps the top limb")
    r.append(a[N - 1] + c)
    return r


def lin(a, ka, b, kb):
    r = [ka * x + kb * y for x, y in zip(a, b)]
    need(all(t < 1 << 32 for t in r), "limb-wise sum wraps")
    return r


ZERO = [0] * N


def is_zero_product(a, multiples):
    return any(a == tight(k * P) for k in range(multiples))


def madd28(acc, q, negate):
    """fp28.h madd28: returns (new accumulator, True) or (acc, False) in the equal-x case.  acc = (X, Y, ZZ, ZZZ) or None."""
    qx, qy = q
    if qx == ZERO and qy == ZERO:
        return acc, True
    qyn = neg28(K["FP28_K8_L1"], qy) if negate else qy
    if acc is None:
        one = K["FP28_ONE"]
        return (qx, norm28(qyn) if negate else qyn, one, one), True
    X, Y, ZZ, ZZZ = acc
    U2 = mm28(qx, ZZ)
    S2 = mm28(qyn, ZZZ)
    Pd = norm28(sub28(U2, K["FP28_K32_L1"], X))
    PP = sq28(Pd)
    if is_zero_product(PP, 2):
        return acc, False
    R = norm28(sub28(S2, K["FP28_K32_L1"], Y))
    PPP = mm28(Pd, PP)
    Q = mm28(X, PP)
    s = lin(PPP, 1, Q, 2)
    X3 = norm28(sub28(sq28(R), K["FP28_K8_L4"], s))
    Y3 = mm28(R, sub28(Q, K["FP28_K32_L1"], X3), neg28(K["FP28_K32_L1"], Y), PPP)
    return (X3, Y3, mm28(ZZ, PP), mm28(ZZZ, PPP)), True


def dbl28(a):
    """fp28.h xyzz_dbl28: the doubling the full addition takes at equal x, equal y (dbl-2008-s-1, a = 0)"""
    X, Y, ZZ, ZZZ = a
    U = lin(Y, 2, ZERO, 0)
    V = mm28(U, U)
    Wd = mm28(U, V)
    S = mm28(X, V)
    M = norm28(lin(sq28(X), 3, ZERO, 0))
    X3 = norm28(sub28(sq28(M), K["FP28_K8_L4"], lin(S, 2, ZERO, 0)))
    Y3 = mm28(M, sub28(S, K["FP28_K32_L1"], X3), neg28(K["FP28_K32_L1"], Y), Wd)
    return (X3, Y3, mm28(V, ZZ), mm28(Wd, ZZZ))


def add28(a, b):
    """fp28.h xyzz_add(XYZZ<Fp28>&, const XYZZ<Fp28>&): (sum, True); at equal x (2 b, False) or (None, False) -- finished in place."""
    if b is None:
        return a, True
    if a is None:
        return b, True
    X1, Y1, ZZ1, ZZZ1 = a
    X2, Y2, ZZ2, ZZZ2 = b
    U1, U2 = mm28(X1, ZZ2), mm28(X2, ZZ1)
    S1, S2 = mm28(Y1, ZZZ2), mm28(Y2, ZZZ1)
    Pd = norm28(sub28(U2, K["FP28_K8_L1"], U1))
    PP = sq28(Pd)
    R = norm28(sub28(S2, K["FP28_K8_L1"], S1))
    if is_zero_product(PP, 2):
        return (dbl28(b) if is_zero_product(sq28(R), 2) else None), False
    PPP, Q = mm28(Pd, PP), mm28(U1, PP)
    s = lin(PPP, 1, Q, 2)
    X3 = norm28(sub28(sq28(R), K["FP28_K8_L4"], s))
    Y3 = mm28(R, sub28(Q, K["FP28_K32_L1"], X3), neg28(K["FP28_K8_L1"], S1), PPP)
    return (X3, Y3, mm28(mm28(ZZ1, ZZ2), PP), mm28(mm28(ZZZ1, ZZZ2), PPP)), True


# G2: a value is a pair of component limb vectors (c0, c1) = the even and odd lane of a pair
def mulF2(a, b, KB):
    a0, a1 = a; b0, b1 = b
    even = mm28(a0, b0, a1, neg28(KB, b1))          # a0 b0 + a1 (K - b1)
    # fp28.h mulF2, odd lane: own a = a1 times the partner's b = b0, the partner's a = a0 times own b = b1  ->  a1 b0 + a0 b1
    odd = mm28(a1, b0, a0, b1)
    return even, odd


def sqrF2(a, KA):
    a0, a1 = a
    even = mm28(lin(a0, 1, a1, 1), sub28(a0, KA, a1))     # (a0 + a1)(a0 + K - a1)
    odd = mm28(a0, lin(a1, 2, ZERO, 0))                   # a0 * 2 a1
    return even, odd


def madd28_g2(acc, q, negate):
    (qx, qy) = q
    if all(c == ZERO for c in qx + qy):
        return acc, True
    qyn = tuple(neg28(K["FP28_K8_L1"], c) for c in qy) if negate else qy
    if acc is None:
        return (qx, qyn, (K["FP28_ONE"], ZERO), (K["FP28_ONE"], ZERO)), True
    X, Y, ZZ, ZZZ = acc
    K8, K32, K32L4, K64, K64L4, K8L4 = (K[n] for n in ("FP28_K8_L1", "FP28_K32_L1", "FP28_K32_L4", "FP28_K64_L1", "FP28_K64_L4", "FP28_K8_L4"))
    U2 = mulF2(qx, ZZ, K8)
    S2 = mulF2(qyn, ZZZ, K8)
    Pd = tuple(norm28(sub28(u, K32, x)) for u, x in zip(U2, X))
    PP = sqrF2(Pd, K64)
    if all(is_zero_product(c, 4) for c in PP):
        return acc, False
    R = tuple(norm28(sub28(s2, K32L4, y)) for s2, y in zip(S2, Y))
    PPP = mulF2(Pd, PP, K8)
    Q = mulF2(X, PP, K8)
    s = tuple(lin(a, 1, b, 2) for a, b in zip(PPP, Q))
    RR = sqrF2(R, K64)
    X3 = tuple(norm28(sub28(rr, K8L4, ss)) for rr, ss in zip(RR, s))
    D = tuple(sub28(qq, K32, x3) for qq, x3 in zip(Q, X3))
    t1 = mulF2(R, D, K64L4)
    t2 = mulF2(Y, PPP, K8)
    Y3 = tuple(sub28(a, K8, b) for a, b in zip(t1, t2))
    return (X3, Y3, mulF2(ZZ, PP, K8), mulF2(ZZZ, PPP, K8)), True


def dbl28_g2(a):
    """fp28.h xyzz_dbl28_g2: the doubling over Fp2 on lane pairs"""
    X, Y, ZZ, ZZZ = a
    K8, K32, K64L4, K8L4 = (K[n] for n in ("FP28_K8_L1", "FP28_K32_L1", "FP28_K64_L4", "FP28_K8_L4"))
    Yn = tuple(norm28(c) for c in Y)
    U = tuple(norm28(lin(c, 2, ZERO, 0)) for c in Yn)
    V = sqrF2(U, K32)
    Wd = mulF2(U, V, K8)
    S = mulF2(X, V, K8)
    M = tuple(norm28(lin(c, 3, ZERO, 0)) for c in sqrF2(X, K32))
    X3 = tuple(norm28(sub28(mm, K8L4, lin(ss, 2, ZERO, 0))) for mm, ss in zip(sqrF2(M, K8), S))
    D = tuple(sub28(ss, K32, x3) for ss, x3 in zip(S, X3))
    t1, t2 = mulF2(M, D, K64L4), mulF2(Yn, Wd, K8)
    Y3 = tuple(sub28(x, K8, y) for x, y in zip(t1, t2))
    return (X3, Y3, mulF2(V, ZZ, K8), mulF2(Wd, ZZZ, K8))


def add28_g2(a, b):
    """fp28.h xyzz_add(XYZZ<Fp28L>&, const XYZZ<Fp28L>&): the full addition over Fp2 on lane pairs; (sum, True); at equal x (2 b, False) or (None, False)"""
    if b is None:
        return a, True
    if a is None:
        return b, True
    X1, Y1, ZZ1, ZZZ1 = a
    X2, Y2, ZZ2, ZZZ2 = b
    K8, K32, K64L4, K8L4 = (K[n] for n in ("FP28_K8_L1", "FP28_K32_L1", "FP28_K64_L4", "FP28_K8_L4"))
    U1, U2 = mulF2(X1, ZZ2, K8), mulF2(X2, ZZ1, K8)
    S1, S2 = mulF2(Y1, ZZZ2, K8), mulF2(Y2, ZZZ1, K8)
    Pd = tuple(norm28(sub28(u2, K8, u1)) for u2, u1 in zip(U2, U1))
    PP = sqrF2(Pd, K32)
    R = tuple(norm28(sub28(s2, K8, s1)) for s2, s1 in zip(S2, S1))
    if all(is_zero_product(c, 2) for c in PP):
        return (dbl28_g2(b) if all(is_zero_product(c, 2) for c in sqrF2(R, K32)) else None), False
    PPP, Q = mulF2(Pd, PP, K8), mulF2(U1, PP, K8)
    s = tuple(lin(x, 1, y, 2) for x, y in zip(PPP, Q))
    X3 = tuple(norm28(sub28(rr, K8L4, ss)) for rr, ss in zip(sqrF2(R, K32), s))
    D = tuple(sub28(qq, K32, x3) for qq, x3 in zip(Q, X3))
    t1, t2 = mulF2(R, D, K64L4), mulF2(S1, PPP, K8)
    Y3 = tuple(sub28(x, K8, y) for x, y in zip(t1, t2))
    return (X3, Y3, mulF2(mulF2(ZZ1, ZZ2, K8), PP, K8), mulF2(mulF2(ZZZ1, ZZZ2, K8), PPP, K8)), True


# ------------------------------------------------------------------------------------------------ reference values in plain integers
RINV = pow(RP, -1, P)


def from28(l):
    return val(l) * RINV % P


def to28(x):
    return tight(x * RP % P)


def acc_affine(acc):
    X, Y, ZZ, ZZZ = (from28(c) for c in acc)
    return (X * pow(ZZ, -1, P) % P, Y * pow(ZZZ, -1, P) % P)


def acc_affine_g2(acc):
    F = o.Fp2Ops
    X, Y, ZZ, ZZZ = ((from28(c[0]), from28(c[1])) for c in acc)
    return (F.mul(X, F.inv(ZZ)), F.mul(Y, F.inv(ZZZ)))


def rand_points_g1(rng, n):
    return [o.G1.mul(o.G1.gen, rng.randrange(1, o.R)) for _ in range(n)]


def loosen(v, limb_cap, rng):
    """the value v written with limbs as LARGE as the cap allows (borrowing 2^28 units from the limb above wherever possible)"""
    l = tight(v)
    for i in range(N - 1):
        t = min((limb_cap - 1 - l[i]) >> W, l[i + 1])
        if t > 0 and rng.random() < 0.9:
            l[i] += t << W; l[i + 1] -= t
    assert val(l) == v and all(0 <= x < limb_cap for x in l[:-1])
    return l


def test_exact_model_random_accumulation_g1():
    rng = random.Random(1)
    pts = rand_points_g1(rng, 40)
    acc, ref = None, None
    for i, pt in enumerate(pts):
        negate = bool(i & 1)
        acc, ok = madd28(acc, (to28(pt[0]), to28(pt[1])), negate)
        assert ok
        ref = o.G1.add(ref, o.G1.neg(pt) if negate else pt)
        assert acc_affine(acc) == ref
        X, Y, ZZ, ZZZ = acc
        assert val(X) < 9.5 * P and val(Y) < 8 * P and val(ZZ) < 1.1 * P and val(ZZZ) < 1.1 * P      # the header's invariants
        assert all(x <= MASK for c in acc for x in c[:-1])
    # equal x: doubling and cancellation are both handed back untouched
    last = pts[-1] if not (len(pts) - 1) & 1 else o.G1.neg(pts[-1])
    single, _ = madd28(None, (to28(pts[0][0]), to28(pts[0][1])), False)
    assert madd28(single, (to28(pts[0][0]), to28(pts[0][1])), False) == (single, False)
    assert madd28(single, (to28(pts[0][0]), to28(pts[0][1])), True) == (single, False)
    assert last is not None


def test_exact_model_adversarial_states_g1():
    """accumulator states at the header's worst case: X tight just below 9.5 p, Y just below 8 p, ZZ / ZZZ just below 1.1 p, each
    congruent to a genuine accumulator, so the group law can still be checked; plus extreme table entries (coordinates p - 1, 1, 0)"""
    rng = random.Random(2)
    pts = rand_points_g1(rng, 12)
    base = o.G1.mul(o.G1.gen, 0xC0FFEE)

    def lift(x, cap_mult):          # the largest representative of x mod p below cap_mult * p
        kmax = int(cap_mult * 1000) * P // 1000
        v = x + ((kmax - x) // P) * P
        assert v % P == x % P and v < kmax
        return v

    for pt in pts:
        z = rng.randrange(1, P)
        zz, zzz = z * z % P, z * z * z % P
        X, Y = base[0] * zz % P, base[1] * zzz % P
        acc = (tight(lift(X * RP % P, 9.5)), tight(lift(Y * RP % P, 8.0)), tight(lift(zz * RP % P, 1.1)), tight(lift(zzz * RP % P, 1.1)))
        for negate in (False, True):
            new, ok = madd28(acc, (to28(pt[0]), to28(pt[1])), negate)
            assert ok and acc_affine(new) == o.G1.add(base, o.G1.neg(pt) if negate else pt)
            assert val(new[0]) < 9.5 * P and val(new[1]) < 1.5 * P
    # table entries at the edges of the canonical range
    for x in (0, 1, P - 1):
        acc = (tight(lift(5, 9.5)), tight(lift(7, 8.0)), tight(lift(11, 1.1)), tight(lift(13, 1.1)))
        for y in (1, P - 1):
            madd28(acc, (tight(x), tight(y)), True)          # not a curve point: only the bound assertions matter here


def test_exact_model_full_addition_g1():
    """the bucket sums of the G1 path are merged and reduced in the 28-bit form (XYZZ<Fp28>): random partial sums, states at the
    invariants' worst case on BOTH sides, and the equal-x cases (doubling, cancellation) that the kernel hands to the generic formulas"""
    rng = random.Random(5)
    pts = rand_points_g1(rng, 24)

    def accumulate(group):
        acc, ref = None, None
        for pt in group:
            acc, ok = madd28(acc, (to28(pt[0]), to28(pt[1])), False); assert ok
            ref = o.G1.add(ref, pt)
        return acc, ref

    sums = [accumulate(pts[i:i + 3]) for i in range(0, 24, 3)]
    while len(sums) > 1:                                   # a merge tree, as k_merge* / k_dimsum / k_dimweight fold
        nxt = []
        for (a, ra), (b, rb) in zip(sums[::2], sums[1::2]):
            c, ok = add28(a, b); assert ok
            assert acc_affine(c) == o.G1.add(ra, rb)
            assert val(c[0]) < 9.5 * P and val(c[1]) < 8 * P and val(c[2]) < 1.1 * P and val(c[3]) < 1.1 * P
            assert all(x <= MASK for comp in c for x in comp[:-1])
            nxt.append((c, o.G1.add(ra, rb)))
        sums = nxt
    total, ref = sums[0]
    assert acc_affine(total) == ref
    assert add28(None, total) == (total, True) and add28(total, None) == (total, True)
    dbl, ok = add28(total, total)                                                          # doubling: finished in the 28-bit form
    assert not ok and acc_affine(dbl) == o.G1.add(ref, ref)
    assert val(dbl[0]) < 9.5 * P and val(dbl[1]) < 8 * P and val(dbl[2]) < 1.1 * P and val(dbl[3]) < 1.1 * P
    assert all(x <= MASK for comp in dbl for x in comp[:-1])
    neg = (total[0], norm28(neg28(K["FP28_K32_L1"], total[1])), total[2], total[3])
    assert add28(total, neg) == (None, False)                                              # cancellation: infinity
    # the same point in two different representations (what coinciding partial sums look like), and a chain of doublings
    z = rng.randrange(2, P); zz, zzz = z * z % P, z * z * z % P
    other = tuple(to28(from28(c) * f % P) for c, f in zip(total, (zz, zzz, zz, zzz)))
    d2, ok = add28(total, other)
    assert not ok and acc_affine(d2) == o.G1.add(ref, ref)
    assert add28(other, neg) == (None, False)
    cur, rc = total, ref
    for _ in range(6):
        cur, ok = add28(cur, cur); rc = o.G1.add(rc, rc)
        assert not ok and acc_affine(cur) == rc

    def lift(x, cap_mult):
        kmax = int(cap_mult * 1000) * P // 1000
        return x + ((kmax - x) // P) * P

    def worst_state(pt):
        z = rng.randrange(1, P)
        zz, zzz = z * z % P, z * z * z % P
        return (tight(lift(pt[0] * zz % P * RP % P, 9.5)), tight(lift(pt[1] * zzz % P * RP % P, 8.0)),
                tight(lift(zz * RP % P, 1.1)), tight(lift(zzz * RP % P, 1.1)))

    for a, b in zip(pts[:8], pts[8:16]):
        c, ok = add28(worst_state(a), worst_state(b))
        assert ok and acc_affine(c) == o.G1.add(a, b)
        assert val(c[0]) < 9.5 * P and val(c[1]) < 1.5 * P
        d, ok = add28(worst_state(a), worst_state(a))                                      # doubling at the invariants' worst case
        assert not ok and acc_affine(d) == o.G1.add(a, a)
        assert val(d[0]) < 9.5 * P and val(d[1]) < 1.5 * P and val(d[2]) < 1.1 * P and val(d[3]) < 1.1 * P


def test_exact_model_random_and_adversarial_g2():
    rng = random.Random(3)
    F = o.Fp2Ops
    pts = [o.G2.mul(o.G2.gen, rng.randrange(1, o.R)) for _ in range(14)]
    to2 = lambda v: (to28(v[0]), to28(v[1]))
    acc, ref = None, None
    for i, pt in enumerate(pts):
        negate = bool(i % 3 == 1)
        acc, ok = madd28_g2(acc, (to2(pt[0]), to2(pt[1])), negate)
        assert ok
        ref = o.G2.add(ref, o.G2.neg(pt) if negate else pt)
        assert acc_affine_g2(acc) == ref
        X, Y, ZZ, ZZZ = acc
        assert all(val(c) < 11.7 * P for c in X) and all(val(c) < 10.4 * P for c in Y)
        assert all(x < 1 << 30 for c in Y for x in c[:-1]) and all(x <= MASK for c in X for x in c[:-1])
        assert all(val(c) < 4 * P for c in ZZ + ZZZ)
    single, _ = madd28_g2(None, (to2(pts[0][0]), to2(pts[0][1])), False)
    assert madd28_g2(single, (to2(pts[0][0]), to2(pts[0][1])), False)[1] is False
    assert madd28_g2(single, (to2(pts[0][0]), to2(pts[0][1])), True)[1] is False
    # adversarial: components lifted to the stated maxima, Y in the loosest limb form the header allows (limbs below 2^30)
    base = o.G2.mul(o.G2.gen, 0xBADC0DE)

    def lift(x, cap_mult):
        kmax = int(cap_mult * 1000) * P // 1000
        return x + ((kmax - x) // P) * P

    for pt in pts[:8]:
        z = (rng.randrange(1, P), rng.randrange(1, P))
        zz = F.mul(z, z); zzz = F.mul(zz, z)
        X, Y = F.mul(base[0], zz), F.mul(base[1], zzz)
        m = lambda v, cap: tuple(tight(lift(c * RP % P, cap)) for c in v)
        Yl = tuple(loosen(lift(c * RP % P, 10.4), 1 << 30, rng) for c in Y)
        acc = (m(X, 11.7), Yl, m(zz, 3.9), m(zzz, 3.9))
        for negate in (False, True):
            new, ok = madd28_g2(acc, (to2(pt[0]), to2(pt[1])), negate)
            assert ok and acc_affine_g2(new) == o.G2.add(base, o.G2.neg(pt) if negate else pt)


def test_exact_model_full_addition_g2():
    """the G2 bucket sums are merged and reduced in the lane-pair 28-bit form too: a merge tree over random partial sums, the equal-x
    cases, and operands lifted to the invariants' worst case (X < 11.7 p, Y < 10.4 p with limbs up to 2^30, ZZ, ZZZ < 3.9 p)"""
    rng = random.Random(6)
    F = o.Fp2Ops
    pts = [o.G2.mul(o.G2.gen, rng.randrange(1, o.R)) for _ in range(16)]
    to2 = lambda pt: ((to28(pt[0][0]), to28(pt[0][1])), (to28(pt[1][0]), to28(pt[1][1])))

    def accumulate(group):
        acc, ref = None, None
        for pt in group:
            acc, ok = madd28_g2(acc, to2(pt), False); assert ok
            ref = o.G2.add(ref, pt)
        return acc, ref

    sums = [accumulate(pts[i:i + 2]) for i in range(0, 16, 2)]
    while len(sums) > 1:
        nxt = []
        for (a, ra), (b, rb) in zip(sums[::2], sums[1::2]):
            c, ok = add28_g2(a, b); assert ok
            assert acc_affine_g2(c) == o.G2.add(ra, rb)
            for comp in range(2):
                assert val(c[0][comp]) < 11.7 * P and val(c[1][comp]) < 10.4 * P and val(c[2][comp]) < 3.9 * P and val(c[3][comp]) < 3.9 * P
                assert all(x <= MASK for k in (0, 2, 3) for x in c[k][comp][:-1]) and all(x < 1 << 30 for x in c[1][comp][:-1])
            nxt.append((c, o.G2.add(ra, rb)))
        sums = nxt
    total, ref = sums[0]

    def inside(c):
        for comp in range(2):
            assert val(c[0][comp]) < 11.7 * P and val(c[1][comp]) < 10.4 * P and val(c[2][comp]) < 3.9 * P and val(c[3][comp]) < 3.9 * P
            assert all(x <= MASK for k in (0, 2, 3) for x in c[k][comp][:-1]) and all(x < 1 << 30 for x in c[1][comp][:-1])

    dbl, ok = add28_g2(total, total)
    assert not ok and acc_affine_g2(dbl) == o.G2.add(ref, ref)
    inside(dbl)
    neg = (total[0], tuple(norm28(neg28(K["FP28_K32_L4"], c)) for c in total[1]), total[2], total[3])
    assert add28_g2(total, neg) == (None, False)
    cur, rc = total, ref
    for _ in range(4):
        cur, ok = add28_g2(cur, cur); rc = o.G2.add(rc, rc)
        assert not ok and acc_affine_g2(cur) == rc
        inside(cur)

    def lift(x, cap_mult):
        kmax = int(cap_mult * 1000) * P // 1000
        return x + ((kmax - x) // P) * P

    def worst_state(pt):
        z = (rng.randrange(1, P), rng.randrange(1, P))
        zz = F.mul(z, z); zzz = F.mul(zz, z)
        X, Y = F.mul(pt[0], zz), F.mul(pt[1], zzz)
        mont = lambda v, cap: tuple(tight(lift(c * RP % P, cap)) for c in v)
        Yl = tuple(loosen(lift(c * RP % P, 10.4), 1 << 30, rng) for c in Y)
        return (mont(X, 11.7), Yl, mont(zz, 3.9), mont(zzz, 3.9))

    for a, b in zip(pts[:6], pts[6:12]):
        c, ok = add28_g2(worst_state(a), worst_state(b))
        assert ok and acc_affine_g2(c) == o.G2.add(a, b)
        d, ok = add28_g2(worst_state(a), worst_state(a))                                   # doubling at the invariants' worst case
        assert not ok and acc_affine_g2(d) == o.G2.add(a, a)
        inside(d)


# ------------------------------------------------------------------------------------------------ worst-case propagation
class B:
    """upper bounds of a quantity: value < v, limb i < l[i] (exclusive)"""

    def __init__(self, v, l):
        self.v, self.l = v, list(l)

    @staticmethod
    def tight(v):
        v = int(v)
        return B(v, [1 << W] * (N - 1) + [(v >> (W * (N - 1))) + 1])


def b_mul(a, b, c=None, d=None):
    # every column: up to 14 products of each group + 14 reduction products + the carry from the column below (< 2^36 + ...)
    worst = 0
    carry = 0
    for k in range(2 * N - 1):
        col = carry
        for i in range(max(0, k - N + 1), min(k, N - 1) + 1):
            col += (a.l[i] - 1) * (b.l[k - i] - 1)
            if c is not None:
                col += (c.l[i] - 1) * (d.l[k - i] - 1)
        for i in (range(0, k + 1) if k < N else range(k - N + 1, N)):
            col += MASK * PL[k - i]
        worst = max(worst, col)
        carry = col >> W
    need(worst < 1 << 64, "worst-case column sum reaches 2^64")
    tot = a.v * b.v + (c.v * d.v if c is not None else 0)
    need(a.v < 1 << 388 and b.v < 1 << 388, "product input may exceed 2^388")
    out = tot // RP + P + 1
    need(out < 1 << (W * (N - 1) + 28), "product output may not fit")
    return B.tight(out)


def b_sub(a, kname, b):
    k = K[kname]
    need(all(k[i] >= b.l[i] - 1 for i in range(N)), "%s does not dominate the limbs being subtracted" % kname)
    need(all(a.l[i] - 1 + k[i] < 1 << 32 for i in range(N)), "a + %s wraps" % kname)
    return B(a.v + val(k), [a.l[i] + k[i] for i in range(N)])


def b_norm(a):
    need(all(x <= 1 << 32 for x in a.l), "limb above 32 bits before the carry pass")
    return B.tight(a.v)


def b_lin(a, ka, b, kb):
    l = [ka * (x - 1) + kb * (y - 1) + 1 for x, y in zip(a.l, b.l)]
    need(all(x <= 1 << 32 for x in l), "limb-wise sum wraps")
    return B(ka * a.v + kb * b.v, l)


def test_invariants_are_inductive_g1():
    """fp28.h:14-18: X tight < 9.5 p, Y tight < 8 p, ZZ, ZZZ tight < 1.1 p  ==>  the same after madd28, for any table entry < p"""
    X, Y, ZZ, ZZZ = B.tight(9.5 * P), B.tight(8 * P + 1), B.tight(1.1 * P), B.tight(1.1 * P)      # Y <= 8 p: the copy 8p - y of a table entry
    qx = B.tight(P)
    qy_neg = b_sub(B(1, [1] * N), "FP28_K8_L1", B.tight(P))            # 8p - y: limbs < 2^29
    for qy in (B.tight(P), qy_neg):
        U2, S2 = b_mul(qx, ZZ), b_mul(qy, ZZZ)
        Pd = b_norm(b_sub(U2, "FP28_K32_L1", X)); assert Pd.v < 33.2 * P
        PP = b_mul(Pd, Pd); assert PP.v < 1.45 * P                        # so the zero test against {0, p} is exhaustive (< 2p)
        R = b_norm(b_sub(S2, "FP28_K32_L1", Y)); assert R.v < 33.6 * P
        PPP, Q = b_mul(Pd, PP), b_mul(X, PP)
        s = b_lin(PPP, 1, Q, 2); assert max(s.l[:-1]) <= 3 * (1 << W)
        X3 = b_norm(b_sub(b_mul(R, R), "FP28_K8_L4", s)); assert X3.v < 9.5 * P
        D = b_sub(Q, "FP28_K32_L1", X3); assert max(D.l[:-1]) <= 1 << 30
        nY = b_sub(B(1, [1] * N), "FP28_K32_L1", Y); assert max(nY.l[:-1]) <= 1 << 29
        Y3 = b_mul(R, D, nY, PPP); assert Y3.v < 1.5 * P
        ZZ3, ZZZ3 = b_mul(ZZ, PP), b_mul(ZZZ, PPP)
        assert ZZ3.v < 1.1 * P and ZZZ3.v < 1.1 * P
        assert X3.v <= X.v and Y3.v <= Y.v                                # inductive
    # the first addition copies the table entry: Y = 8p - y normalised, X = x, ZZ = ZZZ = one: all inside the invariants
    assert b_norm(qy_neg).v <= Y.v and qx.v <= X.v


def test_invariants_are_inductive_full_addition_g1():
    """the same invariants on both operands  ==>  the same after xyzz_add over XYZZ<Fp28>"""
    X, Y, ZZ, ZZZ = B.tight(9.5 * P), B.tight(8 * P + 1), B.tight(1.1 * P), B.tight(1.1 * P)
    U1, U2, S1, S2 = b_mul(X, ZZ), b_mul(X, ZZ), b_mul(Y, ZZZ), b_mul(Y, ZZZ)
    assert U1.v < 1.01 * P and S1.v < 1.01 * P
    Pd = b_norm(b_sub(U2, "FP28_K8_L1", U1)); assert Pd.v < 9.02 * P
    PP = b_mul(Pd, Pd); assert PP.v < 1.04 * P                             # zero test against {0, p} exhaustive
    R = b_norm(b_sub(S2, "FP28_K8_L1", S1)); assert R.v < 9.02 * P
    PPP, Q = b_mul(Pd, PP), b_mul(U1, PP)
    s = b_lin(PPP, 1, Q, 2); assert max(s.l[:-1]) <= 3 * (1 << W)
    X3 = b_norm(b_sub(b_mul(R, R), "FP28_K8_L4", s)); assert X3.v < 9.1 * P
    D = b_sub(Q, "FP28_K32_L1", X3); assert max(D.l[:-1]) <= 1 << 30
    nS1 = b_sub(B(1, [1] * N), "FP28_K8_L1", S1); assert max(nS1.l[:-1]) <= 1 << 29
    Y3 = b_mul(R, D, nS1, PPP); assert Y3.v < 1.13 * P
    ZZ3, ZZZ3 = b_mul(b_mul(ZZ, ZZ), PP), b_mul(b_mul(ZZZ, ZZZ), PPP)
    assert ZZ3.v < 1.01 * P and ZZZ3.v < 1.01 * P
    assert X3.v <= X.v and Y3.v <= Y.v and ZZ3.v <= ZZ.v and ZZZ3.v <= ZZZ.v


def test_invariants_are_inductive_doubling_g1():
    """the invariants on the operand  ==>  the same after xyzz_dbl28 (the equal-x branch of the full addition), and the branch's own
    zero test (R^2 against {0, p}) is exhaustive"""
    X, Y, ZZ, ZZZ = B.tight(9.5 * P), B.tight(8 * P + 1), B.tight(1.1 * P), B.tight(1.1 * P)
    S1 = b_mul(Y, ZZZ)
    R = b_norm(b_sub(S1, "FP28_K8_L1", S1)); assert b_mul(R, R).v < 2 * P
    U = b_lin(Y, 2, B(1, [1] * N), 0); assert max(U.l[:-1]) <= 1 << 29
    V = b_mul(U, U); assert V.v < 1.11 * P
    Wd, S = b_mul(U, V), b_mul(X, V); assert Wd.v < 1.01 * P and S.v < 1.005 * P
    M = b_norm(b_lin(b_mul(X, X), 3, B(1, [1] * N), 0)); assert M.v < 3.2 * P
    X3 = b_norm(b_sub(b_mul(M, M), "FP28_K8_L4", b_lin(S, 2, B(1, [1] * N), 0))); assert X3.v < 9.01 * P
    D = b_sub(S, "FP28_K32_L1", X3); assert max(D.l[:-1]) <= 1 << 30
    nY = b_sub(B(1, [1] * N), "FP28_K32_L1", Y); assert max(nY.l[:-1]) <= 1 << 29
    Y3 = b_mul(M, D, nY, Wd); assert Y3.v < 1.06 * P
    ZZ3, ZZZ3 = b_mul(V, ZZ), b_mul(Wd, ZZZ)
    assert X3.v <= X.v and Y3.v <= Y.v and ZZ3.v <= ZZ.v and ZZZ3.v <= ZZZ.v


def test_invariants_are_inductive_doubling_g2():
    """the lane-pair invariants on the operand  ==>  the same after xyzz_dbl28_g2"""
    def pair_mul(a, b, kb):
        nb = b_sub(B(1, [1] * N), kb, b)
        e, od = b_mul(a, b, a, nb), b_mul(a, b, a, b)
        return e if e.v > od.v else od

    def pair_sqr(a, ka):
        e, od = b_mul(b_lin(a, 1, a, 1), b_sub(a, ka, a)), b_mul(a, b_lin(a, 2, B(1, [1] * N), 0))
        return e if e.v > od.v else od

    one = B(1, [1] * N)
    X, ZZ, ZZZ = B.tight(11.7 * P), B.tight(3.9 * P), B.tight(3.9 * P)
    Y = B(int(10.4 * P), [1 << 30] * (N - 1) + [(int(10.4 * P) >> (W * (N - 1))) + 1])
    S1 = pair_mul(Y, ZZZ, "FP28_K8_L1")
    R = b_norm(b_sub(S1, "FP28_K8_L1", S1)); assert pair_sqr(R, "FP28_K32_L1").v < 2 * P      # zero test against {0, p} exhaustive
    Yn = b_norm(Y)
    U = b_norm(b_lin(Yn, 2, one, 0)); assert U.v < 20.9 * P
    V = pair_sqr(U, "FP28_K32_L1"); assert V.v < 1.9 * P
    Wd, S = pair_mul(U, V, "FP28_K8_L1"), pair_mul(X, V, "FP28_K8_L1"); assert Wd.v < 1.1 * P and S.v < 1.06 * P
    M = b_norm(b_lin(pair_sqr(X, "FP28_K32_L1"), 3, one, 0)); assert M.v < 4.3 * P
    X3 = b_norm(b_sub(pair_sqr(M, "FP28_K8_L1"), "FP28_K8_L4", b_lin(S, 2, one, 0))); assert X3.v < 9.1 * P
    D = b_sub(S, "FP28_K32_L1", X3); assert max(D.l[:-1]) <= 1 << 30
    t1, t2 = pair_mul(M, D, "FP28_K64_L4"), pair_mul(Yn, Wd, "FP28_K8_L1")
    Y3 = b_sub(t1, "FP28_K8_L1", t2); assert Y3.v < 9.2 * P and max(Y3.l[:-1]) <= 1 << 30
    ZZ3, ZZZ3 = pair_mul(V, ZZ, "FP28_K8_L1"), pair_mul(Wd, ZZZ, "FP28_K8_L1"); assert ZZ3.v < 1.02 * P and ZZZ3.v < 1.02 * P
    assert X3.v <= X.v and Y3.v <= Y.v and ZZ3.v <= ZZ.v and ZZZ3.v <= ZZZ.v


def test_invariants_are_inductive_g2():
    """fp28.h (madd28_g2) for the lane-pair form: X tight < 11.7 p, Y limbs < 2^30 and value < 10.4 p, ZZ / ZZZ tight < 3.9 p.
    (Round 1's header said Y < 9.4 p; this propagation showed the true worst case of t1 + 8p - t2 is 10.3 p -- every constraint
    still holds with it, and the header now states 10.4 p.)"""
    def pair_mul(a, b, kb):            # worst lane: max over  a0 b0 + a1 (K - b1)  and  a0 b1 + a1 b0
        nb = b_sub(B(1, [1] * N), kb, b)
        e, od = b_mul(a, b, a, nb), b_mul(a, b, a, b)
        return e if e.v > od.v else od

    def pair_sqr(a, ka):
        s = b_lin(a, 1, a, 1)
        e, od = b_mul(s, b_sub(a, ka, a)), b_mul(a, b_lin(a, 2, B(1, [1] * N), 0))
        return e if e.v > od.v else od

    X, ZZ, ZZZ = B.tight(11.7 * P), B.tight(3.9 * P), B.tight(3.9 * P)
    Y = B(int(10.4 * P), [1 << 30] * (N - 1) + [(int(10.4 * P) >> (W * (N - 1))) + 1])
    qx = B.tight(P)
    for qy in (B.tight(P), b_sub(B(1, [1] * N), "FP28_K8_L1", B.tight(P))):
        U2, S2 = pair_mul(qx, ZZ, "FP28_K8_L1"), pair_mul(qy, ZZZ, "FP28_K8_L1")
        Pd = b_norm(b_sub(U2, "FP28_K32_L1", X)); assert Pd.v < 33.6 * P
        PP = pair_sqr(Pd, "FP28_K64_L1"); assert PP.v < 3.7 * P          # zero test against {0, p, 2p, 3p} is exhaustive (< 4p)
        R = b_norm(b_sub(S2, "FP28_K32_L4", Y)); assert R.v < 33.6 * P
        PPP, Q = pair_mul(Pd, PP, "FP28_K8_L1"), pair_mul(X, PP, "FP28_K8_L1")
        s = b_lin(PPP, 1, Q, 2)
        X3 = b_norm(b_sub(pair_sqr(R, "FP28_K64_L1"), "FP28_K8_L4", s)); assert X3.v < 11.7 * P
        D = b_sub(Q, "FP28_K32_L1", X3); assert max(D.l[:-1]) <= 1 << 30
        t1 = pair_mul(R, D, "FP28_K64_L4")
        t2 = pair_mul(Y, PPP, "FP28_K8_L1")
        Y3 = b_sub(t1, "FP28_K8_L1", t2); assert Y3.v < 10.4 * P and max(Y3.l[:-1]) <= 1 << 30
        ZZ3, ZZZ3 = pair_mul(ZZ, PP, "FP28_K8_L1"), pair_mul(ZZZ, PPP, "FP28_K8_L1")
        assert ZZ3.v < 3.9 * P and ZZZ3.v < 3.9 * P
        assert X3.v <= X.v and Y3.v <= Y.v


def test_invariants_are_inductive_full_addition_g2():
    """the lane-pair invariants on both operands  ==>  the same after xyzz_add over XYZZ<Fp28L>"""
    def pair_mul(a, b, kb):
        nb = b_sub(B(1, [1] * N), kb, b)
        e, od = b_mul(a, b, a, nb), b_mul(a, b, a, b)
        return e if e.v > od.v else od

    def pair_sqr(a, ka):
        e, od = b_mul(b_lin(a, 1, a, 1), b_sub(a, ka, a)), b_mul(a, b_lin(a, 2, B(1, [1] * N), 0))
        return e if e.v > od.v else od

    X, ZZ, ZZZ = B.tight(11.7 * P), B.tight(3.9 * P), B.tight(3.9 * P)
    Y = B(int(10.4 * P), [1 << 30] * (N - 1) + [(int(10.4 * P) >> (W * (N - 1))) + 1])
    U1, S1 = pair_mul(X, ZZ, "FP28_K8_L1"), pair_mul(Y, ZZZ, "FP28_K8_L1")
    assert U1.v < 1.06 * P and S1.v < 1.06 * P
    Pd = b_norm(b_sub(U1, "FP28_K8_L1", U1)); assert Pd.v < 9.1 * P
    PP = pair_sqr(Pd, "FP28_K32_L1"); assert PP.v < 1.31 * P                # zero test against {0, p} exhaustive (< 2p)
    R = b_norm(b_sub(S1, "FP28_K8_L1", S1)); assert R.v < 9.1 * P
    PPP, Q = pair_mul(Pd, PP, "FP28_K8_L1"), pair_mul(U1, PP, "FP28_K8_L1")
    s = b_lin(PPP, 1, Q, 2)
    X3 = b_norm(b_sub(pair_sqr(R, "FP28_K32_L1"), "FP28_K8_L4", s)); assert X3.v < 9.4 * P
    D = b_sub(Q, "FP28_K32_L1", X3); assert max(D.l[:-1]) <= 1 << 30
    t1, t2 = pair_mul(R, D, "FP28_K64_L4"), pair_mul(S1, PPP, "FP28_K8_L1")
    Y3 = b_sub(t1, "FP28_K8_L1", t2); assert Y3.v < 9.5 * P and max(Y3.l[:-1]) <= 1 << 30
    ZZ3 = pair_mul(pair_mul(ZZ, ZZ, "FP28_K8_L1"), PP, "FP28_K8_L1")
    ZZZ3 = pair_mul(pair_mul(ZZZ, ZZZ, "FP28_K8_L1"), PPP, "FP28_K8_L1")
    assert ZZ3.v < 1.02 * P and ZZZ3.v < 1.02 * P
    assert X3.v <= X.v and Y3.v <= Y.v and ZZ3.v <= ZZ.v and ZZZ3.v <= ZZZ.v


def test_the_model_notices_a_broken_bound():
    """the checks are live: a subtraction constant that does not dominate, or limbs too loose for a product, are caught"""
    with pytest.raises(Bound):
        b_sub(B.tight(P), "FP28_K8_L1", B(int(9 * P), [1 << 30] * N))          # L1 constant against a 2^30-limbed value
    big = B(1 << 387, [1 << 32] * N)
    with pytest.raises(Bound):
        b_mul(big, big)
    with pytest.raises(Bound):
        mm28([0xFFFFFFFF] * N, [0xFFFFFFFF] * N)
