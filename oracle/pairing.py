"""TEST INFRASTRUCTURE ONLY -- slow pure-Python BLS12-381 ate pairing, used solely to check that
proofs satisfy the Groth16 verification equation:

    e(A, B) = e(alpha_g1, beta_g2) * e(sum_i x_i * gamma_ABC_g1[i], gamma_g2) * e(C, delta_g2)

which is what the reference's verifier side computes (zk::verify / tvm.vergrth16,
share/tvm/voting_voter.sol:94).  Fp12 is represented as polynomials in w modulo
w^12 - 2 w^6 + 2 (w^6 = 1 + u), so no tower bookkeeping is needed.
"""
from bls12_381 import P, R, G1, G2

DEG = 12
ATE_LOOP = 15132376222941642752      # |x|
LOG_ATE_LOOP = 62


def f12(coeffs):
    return [c % P for c in coeffs] + [0] * (DEG - len(coeffs))


ONE = f12([1])
ZERO = f12([0])


def f12_add(a, b): return [(x + y) % P for x, y in zip(a, b)]
def f12_sub(a, b): return [(x - y) % P for x, y in zip(a, b)]
def f12_neg(a): return [(-x) % P for x in a]
def f12_scal(a, k): return [x * k % P for x in a]


def f12_mul(a, b):
    t = [0] * (2 * DEG - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    # reduce: w^12 = 2 w^6 - 2
    for k in range(2 * DEG - 2, DEG - 1, -1):
        c = t[k]
        if c:
            t[k - 6] += 2 * c
            t[k - 12] -= 2 * c
    return [x % P for x in t[:DEG]]


def _deg(p):
    d = len(p) - 1
    while d >= 0 and p[d] == 0:
        d -= 1
    return d


def f12_inv(a):
    """Extended Euclid over Fp[w] against the modulus polynomial."""
    mod = [2, 0, 0, 0, 0, 0, P - 2, 0, 0, 0, 0, 0, 1]
    lm, hm = [1] + [0] * DEG, [0] * (DEG + 1)
    low, high = list(a) + [0], mod
    while _deg(low) > 0:
        # r = high // low
        r = [0] * (DEG + 1)
        temp = list(high)
        dl = _deg(low)
        inv_lead = pow(low[dl], P - 2, P)
        for i in range(_deg(temp) - dl, -1, -1):
            q = temp[dl + i] * inv_lead % P
            r[i] = q
            if q:
                for c in range(dl + 1):
                    temp[c + i] = (temp[c + i] - low[c] * q) % P
        nm, new = list(hm), list(high)
        for i in range(DEG + 1):
            if lm[i] or low[i]:
                for j in range(DEG + 1 - i):
                    if r[j]:
                        nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                        new[i + j] = (new[i + j] - low[i] * r[j]) % P
        lm, low, hm, high = nm, new, lm, low
    c = pow(low[0], P - 2, P)
    return [x * c % P for x in lm[:DEG]]


def f12_pow(a, e):
    out, base = ONE, a
    while e:
        if e & 1:
            out = f12_mul(out, base)
        base = f12_mul(base, base)
        e >>= 1
    return out


W = f12([0, 1])
W2_INV = f12_inv(f12_mul(W, W))
W3_INV = f12_inv(f12_mul(f12_mul(W, W), W))


def twist(pt):
    """G2 affine over Fp2 (u^2 = -1) -> curve over Fp12 (untwist by w^2, w^3)."""
    (x0, x1), (y0, y1) = pt
    nx = f12([(x0 - x1) % P] + [0] * 5 + [x1])
    ny = f12([(y0 - y1) % P] + [0] * 5 + [y1])
    return (f12_mul(nx, W2_INV), f12_mul(ny, W3_INV))


def cast_g1(pt):
    return (f12([pt[0]]), f12([pt[1]]))


def _double(pt):
    x, y = pt
    m = f12_mul(f12_scal(f12_mul(x, x), 3), f12_inv(f12_scal(y, 2)))
    nx = f12_sub(f12_mul(m, m), f12_scal(x, 2))
    ny = f12_sub(f12_mul(m, f12_sub(x, nx)), y)
    return (nx, ny)


def _add(p1, p2):
    if p1 is None: return p2
    if p2 is None: return p1
    x1, y1 = p1; x2, y2 = p2
    if x1 == x2:
        return _double(p1) if y1 == y2 else None
    m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    nx = f12_sub(f12_sub(f12_mul(m, m), x1), x2)
    ny = f12_sub(f12_mul(m, f12_sub(x1, nx)), y1)
    return (nx, ny)


def _linefunc(p1, p2, t):
    x1, y1 = p1; x2, y2 = p2; xt, yt = t
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    if y1 == y2:
        m = f12_mul(f12_scal(f12_mul(x1, x1), 3), f12_inv(f12_scal(y1, 2)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    return f12_sub(xt, x1)


def miller_loop(q_g2, p_g1):
    """Miller loop f_{|x|,Q}(P) without the final exponentiation; ONE if either point is infinity."""
    if q_g2 is None or p_g1 is None:
        return ONE
    Q = twist(q_g2); Pp = cast_g1(p_g1)
    Rr = Q
    f = ONE
    for i in range(LOG_ATE_LOOP, -1, -1):
        f = f12_mul(f12_mul(f, f), _linefunc(Rr, Rr, Pp))
        Rr = _double(Rr)
        if ATE_LOOP & (1 << i):
            f = f12_mul(f, _linefunc(Rr, Q, Pp))
            Rr = _add(Rr, Q)
    return f


def final_exp(f):
    return f12_pow(f, (P ** 12 - 1) // R)


def pairing_product_is_one(pairs):
    """pairs: [(G1 affine, G2 affine)];  prod e(P_i, Q_i) == 1 ?"""
    f = ONE
    for p1, q2 in pairs:
        f = f12_mul(f, miller_loop(q2, p1))
    return final_exp(f) == ONE


def groth16_verify(vk, public_inputs, proof):
    """vk: dict(alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_ABC_g1=[...]) affine points;
    public_inputs: list of ints (without the leading 1); proof: (A, B, C) affine points."""
    A, B, Cc = proof
    if not (G1.is_on_curve(A) and G2.is_on_curve(B) and G1.is_on_curve(Cc)):
        return False
    acc = vk["gamma_ABC_g1"][0]
    for x, pt in zip(public_inputs, vk["gamma_ABC_g1"][1:]):
        acc = G1.add(acc, G1.mul(pt, x % R))
    return pairing_product_is_one([(G1.neg(A), B), (vk["alpha_g1"], vk["beta_g2"]),
                                   (acc, vk["gamma_g2"]), (Cc, vk["delta_g2"])])
