"""TEST INFRASTRUCTURE ONLY -- pure-Python big-integer oracle for the Groth16 hot path.

This file is a CPU restatement, in Python big-ints, of the *published algorithms* that
the reference (NilFoundation/vote-saver-protocol) reaches through its un-vendored
crypto3 submodules (``.gitmodules:5-12,47-48``: crypto3-multiprecision / -algebra /
-zk / -math, pinned versions unknown -- SURVEY.md section 0, F1/F2).  The reference
tree holds no source, tests or golden vectors for this path, so this oracle is
**"parity unpinned"** except for:

* public BLS12-381 constants (checked in ``self_check``),
* ``bin/cli/src/data.bin[0:192]`` of the reference -- a ZCash-compressed Groth16 proof
  A||B||C (fixture copy under ``tests/golden/data_bin_proof.hex``) -- which pins point
  (de)compression, on-curve and subgroup membership,
* mathematical uniqueness of the outputs (an MSM result as an affine point, an NTT
  result given omega) -- SURVEY.md section 8(c).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``vote_saver_protocol_amd``) never does.

Reference call sites this follows (relative to /root/reference):
  bin/cli/include/nil/vote_saver/common.hpp:148   curve = bls12<381>
  bin/cli/include/nil/vote_saver/common.hpp:107-129  G1/G2 coordinates (affine / jacobian a4=0)
  bin/cli/include/nil/vote_saver/common.hpp:1132-1135 the prove call (multiexp + evaluation_domain below it)
  bin/cli/src/protocol_exec.ipynb cell 0  wire sizes fr=32, g1=48, g2=96
"""

# --------------------------------------------------------------------------- constants
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G2_X = (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E)
G2_Y = (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)
B1 = 4            # G1: y^2 = x^3 + 4
B2 = (4, 4)       # G2: y^2 = x^3 + 4(1+u)
FR_TWO_ADICITY = 32
FR_GENERATOR = 7  # multiplicative generator of Fr*, also the coset shift of the evaluation domain
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R - 1) >> FR_TWO_ADICITY, R)   # order exactly 2^32
BLS_X = 0xD201000000010000   # |x|; the curve parameter is -x
BLS_X_IS_NEG = True


# --------------------------------------------------------------------------- field ops
class FpOps:
    """Prime field Fp, elements are ints in [0, P)."""
    zero = 0
    one = 1

    @staticmethod
    def add(a, b): return (a + b) % P
    @staticmethod
    def sub(a, b): return (a - b) % P
    @staticmethod
    def neg(a): return (-a) % P
    @staticmethod
    def mul(a, b): return (a * b) % P
    @staticmethod
    def sqr(a): return (a * a) % P
    @staticmethod
    def inv(a): return pow(a, P - 2, P)
    @staticmethod
    def is_zero(a): return a == 0
    @staticmethod
    def muli(a, k): return (a * k) % P
    @staticmethod
    def eq(a, b): return a == b


class Fp2Ops:
    """Fp2 = Fp[u]/(u^2+1), elements are (c0, c1)."""
    zero = (0, 0)
    one = (1, 0)

    @staticmethod
    def add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
    @staticmethod
    def sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
    @staticmethod
    def neg(a): return ((-a[0]) % P, (-a[1]) % P)
    @staticmethod
    def mul(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    @staticmethod
    def sqr(a):
        return ((a[0] + a[1]) * (a[0] - a[1]) % P, (2 * a[0] * a[1]) % P)
    @staticmethod
    def inv(a):
        t = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
        return (a[0] * t % P, (-a[1]) * t % P)
    @staticmethod
    def is_zero(a): return a[0] == 0 and a[1] == 0
    @staticmethod
    def muli(a, k): return (a[0] * k % P, a[1] * k % P)
    @staticmethod
    def eq(a, b): return a[0] == b[0] and a[1] == b[1]


def fp_sqrt(a):
    """sqrt in Fp (P = 3 mod 4); returns None if a is a non-residue."""
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a % P else None


def fp2_sqrt(a):
    """sqrt in Fp2 via the norm method; returns None if a is a non-residue."""
    a0, a1 = a
    if a1 == 0:
        s = fp_sqrt(a0)
        if s is not None:
            return (s, 0)
        s = fp_sqrt((-a0) % P)          # sqrt(a0) = s*u because u^2 = -1
        return (0, s) if s is not None else None
    n = fp_sqrt((a0 * a0 + a1 * a1) % P)
    if n is None:
        return None
    inv2 = pow(2, P - 2, P)
    for nn in (n, (-n) % P):
        t = (a0 + nn) * inv2 % P
        x0 = fp_sqrt(t)
        if x0 is None or x0 == 0:
            continue
        x1 = a1 * pow(2 * x0, P - 2, P) % P
        if Fp2Ops.sqr((x0, x1)) == (a0 % P, a1 % P):
            return (x0, x1)
    return None


# --------------------------------------------------------------------------- curves
class Curve:
    """Short Weierstrass y^2 = x^3 + b, a = 0.  Affine points are (x, y) or None (infinity).
    Jacobian points are (X, Y, Z) with Z == zero meaning infinity
    (the reference's jacobian_with_a4_0 form, common.hpp:117-121)."""

    def __init__(self, F, b, gen, name):
        self.F, self.b, self.gen, self.name = F, b, gen, name

    # ---- affine
    def is_on_curve(self, pt):
        if pt is None:
            return True
        F = self.F
        x, y = pt
        return F.eq(F.sqr(y), F.add(F.mul(F.sqr(x), x), self.b))

    def neg(self, pt):
        return None if pt is None else (pt[0], self.F.neg(pt[1]))

    def add(self, p1, p2):
        F = self.F
        if p1 is None: return p2
        if p2 is None: return p1
        x1, y1 = p1
        x2, y2 = p2
        if F.eq(x1, x2):
            if F.eq(y1, y2):
                if F.is_zero(y1):
                    return None
                lam = F.mul(F.muli(F.sqr(x1), 3), F.inv(F.muli(y1, 2)))
            else:
                return None
        else:
            lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        x3 = F.sub(F.sub(F.sqr(lam), x1), x2)
        y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
        return (x3, y3)

    # ---- jacobian (a = 0)
    def to_jac(self, pt):
        F = self.F
        return (F.one, F.one, F.zero) if pt is None else (pt[0], pt[1], F.one)

    def from_jac(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def jdbl(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z) or F.is_zero(Y):
            return (F.one, F.one, F.zero)
        A = F.sqr(X); B = F.sqr(Y); C = F.sqr(B)
        D = F.muli(F.sub(F.sub(F.sqr(F.add(X, B)), A), C), 2)
        E = F.muli(A, 3)
        Fq = F.sqr(E)
        X3 = F.sub(Fq, F.muli(D, 2))
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), F.muli(C, 8))
        Z3 = F.muli(F.mul(Y, Z), 2)
        return (X3, Y3, Z3)

    def jadd(self, J1, J2):
        F = self.F
        X1, Y1, Z1 = J1
        X2, Y2, Z2 = J2
        if F.is_zero(Z1): return J2
        if F.is_zero(Z2): return J1
        Z1Z1 = F.sqr(Z1); Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2); U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(Y1, F.mul(Z2, Z2Z2)); S2 = F.mul(Y2, F.mul(Z1, Z1Z1))
        if F.eq(U1, U2):
            if F.eq(S1, S2):
                return self.jdbl(J1)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, U1)
        Rr = F.sub(S2, S1)
        HH = F.sqr(H); HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.sqr(Rr), HHH), F.muli(V, 2))
        Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def mul(self, pt, k):
        """Scalar multiplication of an affine point, double-and-add, returns affine."""
        if pt is None:
            return None
        if k < 0:
            return self.mul(self.neg(pt), -k)
        acc = self.to_jac(None)
        base = self.to_jac(pt)
        while k:
            if k & 1:
                acc = self.jadd(acc, base)
            base = self.jdbl(base)
            k >>= 1
        return self.from_jac(acc)

    def in_subgroup(self, pt):
        return self.mul(pt, R) is None

    def msm_naive(self, points, scalars):
        """sum_i scalars[i] * points[i] -- the value algebra::multiexp must return
        (reached from common.hpp:1132-1135); plain double-and-add, no bucket method."""
        acc = self.to_jac(None)
        for pt, k in zip(points, scalars):
            if pt is None or k % R == 0:
                continue
            acc = self.jadd(acc, self.to_jac(self.mul(pt, k % R)))
        return self.from_jac(acc)


G1 = Curve(FpOps, B1, (G1_X, G1_Y), "G1")
G2 = Curve(Fp2Ops, B2, (G2_X, G2_Y), "G2")


# --------------------------------------------------------------------------- ZCash codec
def _lex_larger_fp(y):
    return y > (P - 1) // 2


def _lex_larger_fp2(y):
    # compare c1 first, then c0 (ZCash / IETF pairing-friendly-curves serialization)
    if y[1] != 0:
        return y[1] > (P - 1) // 2
    return y[0] > (P - 1) // 2


def g1_compress(pt):
    """48-byte ZCash compressed G1 (wire size g1_size=48, protocol_exec.ipynb cell 0)."""
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80
    if _lex_larger_fp(y):
        b[0] |= 0x20
    return bytes(b)


def g1_decompress(data):
    assert len(data) == 48
    flags = data[0]
    assert flags & 0x80, "uncompressed form not supported"
    if flags & 0x40:
        assert (flags & 0x3F) == 0 and not any(data[1:])
        return None
    x = int.from_bytes(bytes([flags & 0x1F]) + data[1:], "big")
    assert x < P
    y = fp_sqrt((x * x * x + B1) % P)
    assert y is not None, "x not on curve"
    if _lex_larger_fp(y) != bool(flags & 0x20):
        y = P - y
    return (x, y)


def g2_compress(pt):
    """96-byte ZCash compressed G2: x.c1 || x.c0 big-endian, flags in the first byte."""
    if pt is None:
        return bytes([0xC0]) + bytes(95)
    x, y = pt
    b = bytearray(x[1].to_bytes(48, "big") + x[0].to_bytes(48, "big"))
    b[0] |= 0x80
    if _lex_larger_fp2(y):
        b[0] |= 0x20
    return bytes(b)


def g2_decompress(data):
    assert len(data) == 96
    flags = data[0]
    assert flags & 0x80
    if flags & 0x40:
        assert (flags & 0x3F) == 0 and not any(data[1:])
        return None
    x1 = int.from_bytes(bytes([flags & 0x1F]) + data[1:48], "big")
    x0 = int.from_bytes(data[48:96], "big")
    assert x0 < P and x1 < P
    x = (x0, x1)
    y = fp2_sqrt(Fp2Ops.add(Fp2Ops.mul(Fp2Ops.sqr(x), x), B2))
    assert y is not None, "x not on curve"
    if _lex_larger_fp2(y) != bool(flags & 0x20):
        y = Fp2Ops.neg(y)
    return (x, y)


# --------------------------------------------------------------------------- Fr / NTT
def fr_root_of_unity(log_m):
    """omega for the radix-2 domain of size 2^log_m: root_of_unity^(2^(32-log_m))
    (libfqfft-lineage basic_radix2_domain)."""
    assert 0 <= log_m <= FR_TWO_ADICITY
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_TWO_ADICITY - log_m), R)


def dft_naive(a, omega):
    """O(n^2) DFT: out[k] = sum_j a[j] * omega^(jk) -- the definition evaluation_domain::fft meets."""
    n = len(a)
    out = []
    for k in range(n):
        wk = pow(omega, k, R)
        acc, w = 0, 1
        for j in range(n):
            acc = (acc + a[j] * w) % R
            w = w * wk % R
        out.append(acc)
    return out


def _bitrev(i, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def ntt(a, inverse=False, coset=None):
    """Serial radix-2 DIT NTT with the semantics of the evaluation_domain family:
       fft:           A[k] = sum_j a[j] w^(jk)
       inverse_fft:   fft with w^-1, then scale by m^-1
       coset fft:     a[j] *= g^j first;   inverse coset: inverse_fft then a[j] *= g^-j."""
    n = len(a)
    log_m = n.bit_length() - 1
    assert 1 << log_m == n
    a = [x % R for x in a]
    omega = fr_root_of_unity(log_m)
    if inverse:
        omega = pow(omega, R - 2, R)
    if coset is not None and not inverse:
        g, w = coset % R, 1
        for i in range(n):
            a[i] = a[i] * w % R
            w = w * g % R
    # bit-reverse then butterflies
    for i in range(n):
        j = _bitrev(i, log_m)
        if i < j:
            a[i], a[j] = a[j], a[i]
    m = 1
    while m < n:
        wm = pow(omega, n // (2 * m), R)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = w * a[k + j + m] % R
                u = a[k + j]
                a[k + j] = (u + t) % R
                a[k + j + m] = (u - t) % R
                w = w * wm % R
        m *= 2
    if inverse:
        ninv = pow(n, R - 2, R)
        a = [x * ninv % R for x in a]
        if coset is not None:
            gi, w = pow(coset % R, R - 2, R), 1
            for i in range(n):
                a[i] = a[i] * w % R
                w = w * gi % R
    return a


# --------------------------------------------------------------------------- deterministic inputs
def splitmix64(seed):
    """Generator of the synthetic inputs named in SURVEY.md section 8(d)."""
    state = seed & 0xFFFFFFFFFFFFFFFF
    while True:
        state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        yield z ^ (z >> 31)


def rand_fr(gen):
    """256 bits from splitmix64 (little-endian limbs), reduced mod r."""
    v = 0
    for i in range(4):
        v |= next(gen) << (64 * i)
    return v % R


# --------------------------------------------------------------------------- limb helpers (C-ABI layout)
def int_to_limbs(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def limbs_to_int(l):
    v = 0
    for i, x in enumerate(l):
        v |= int(x) << (64 * i)
    return v


def g1_to_limbs(pt):
    """affine G1 -> 12 u64 (x then y, canonical little-endian); infinity = all zero."""
    if pt is None:
        return [0] * 12
    return int_to_limbs(pt[0], 6) + int_to_limbs(pt[1], 6)


def g1_from_limbs(l):
    x, y = limbs_to_int(l[0:6]), limbs_to_int(l[6:12])
    return None if (x == 0 and y == 0) else (x, y)


def g2_to_limbs(pt):
    """affine G2 -> 24 u64 (x.c0, x.c1, y.c0, y.c1); infinity = all zero."""
    if pt is None:
        return [0] * 24
    (x0, x1), (y0, y1) = pt
    return int_to_limbs(x0, 6) + int_to_limbs(x1, 6) + int_to_limbs(y0, 6) + int_to_limbs(y1, 6)


def g2_from_limbs(l):
    v = [limbs_to_int(l[6 * i:6 * i + 6]) for i in range(4)]
    if not any(v):
        return None
    return ((v[0], v[1]), (v[2], v[3]))


# --------------------------------------------------------------------------- self-check
def self_check():
    assert G1.is_on_curve(G1.gen) and G2.is_on_curve(G2.gen)
    assert G1.in_subgroup(G1.gen) and G2.in_subgroup(G2.gen)
    assert pow(FR_ROOT_OF_UNITY, 1 << 32, R) == 1 and pow(FR_ROOT_OF_UNITY, 1 << 31, R) == R - 1
    assert FR_ROOT_OF_UNITY == 10238227357739495823651030575849232062558860180284477541189508159991286009131
    a = [3, 1, 4, 1, 5, 9, 2, 6]
    assert ntt(a) == dft_naive(a, fr_root_of_unity(3))
    assert ntt(ntt(a), inverse=True) == a
    assert ntt(ntt(a, coset=7), inverse=True, coset=7) == a
    assert g1_decompress(g1_compress(G1.gen)) == G1.gen
    assert g2_decompress(g2_compress(G2.gen)) == G2.gen
    return True


if __name__ == "__main__":
    self_check()
    print("oracle self-check ok")
