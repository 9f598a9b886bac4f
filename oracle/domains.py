"""TEST INFRASTRUCTURE ONLY -- evaluation domains over BLS12-381 Fr, Python big-int restatement.  Parity unpinned
(the reference's crypto3-math submodule is absent, /root/reference/.gitmodules:47-48; no golden vector exists there).

Follows the libfqfft lineage crypto3-math descends from [UPSTREAM-KNOWLEDGE]:
  evaluation_domain/domains/basic_radix2_domain.tcc, step_radix2_domain.tcc, get_evaluation_domain.tcc
  (crypto3: math/domains/{basic_radix2,step_radix2}_domain.hpp, math/algorithms/make_evaluation_domain.hpp).
Every method is additionally checked in tests/ against its mathematical definition (naive evaluation at the
domain's elements, Lagrange interpolation identity, Z vanishing on the domain), so the restatement is pinned by
construction even though the reference pins nothing.

A step domain of size m = big_m + small_m (big_m = 2^(ceil_log2(m)-1), small_m a power of two) is the union of
the big_m-th roots of unity and the coset omega * <small_m-th roots>, omega a primitive (2*big_m)-th root.
"""
try:
    from .bls12_381 import R, FR_GENERATOR, FR_TWO_ADICITY, fr_root_of_unity, ntt
except ImportError:      # tests put oracle/ itself on sys.path
    from bls12_381 import R, FR_GENERATOR, FR_TWO_ADICITY, fr_root_of_unity, ntt


def _inv(x):
    return pow(x % R, R - 2, R)


def ceil_log2(n):
    l = 0
    while (1 << l) < n:
        l += 1
    return l


class DomainError(ValueError):
    pass


class BasicRadix2Domain:
    kind = "basic_radix2"

    def __init__(self, m):
        # basic_radix2_domain.tcc ctor: m > 1 required, m a power of two, log m <= two-adicity
        if m <= 1:
            raise DomainError("basic_radix2(): expected m > 1")
        self.log_m = ceil_log2(m)
        if m != 1 << self.log_m:
            raise DomainError("basic_radix2(): expected m == 1<<log2(m)")
        if self.log_m > FR_TWO_ADICITY:
            raise DomainError("basic_radix2(): expected logm <= s")
        self.m = m
        self.omega = fr_root_of_unity(self.log_m)

    def fft(self, a):
        assert len(a) == self.m
        return ntt(a)

    def inverse_fft(self, a):
        assert len(a) == self.m
        return ntt(a, inverse=True)

    def coset_fft(self, a, g):
        return ntt(a, coset=g)

    def inverse_coset_fft(self, a, g):
        return ntt(a, inverse=True, coset=g)

    def get_domain_element(self, idx):
        return pow(self.omega, idx, R)

    def compute_vanishing_polynomial(self, t):
        return (pow(t, self.m, R) - 1) % R

    def evaluate_all_lagrange_polynomials(self, t):
        return _radix2_lagrange(self.m, t)

    def add_poly_z(self, coeff, H):
        assert len(H) == self.m + 1
        H = list(H)
        H[self.m] = (H[self.m] + coeff) % R
        H[0] = (H[0] - coeff) % R
        return H

    def divide_by_z_on_coset(self, P, g=FR_GENERATOR):
        zi = _inv(self.compute_vanishing_polynomial(g))
        return [x * zi % R for x in P]


def _radix2_lagrange(m, t):
    """basic_radix2_domain_aux.tcc _basic_radix2_evaluate_all_lagrange_polynomials."""
    if m == 1:
        return [1]
    omega = fr_root_of_unity(ceil_log2(m))
    t %= R
    # t in the domain: the indicator vector
    if pow(t, m, R) == 1:
        out, w = [], 1
        for i in range(m):
            out.append(1 if w == t else 0)
            w = w * omega % R
        return out
    Z = (pow(t, m, R) - 1) % R
    l = Z * _inv(m) % R
    out, r = [], 1
    for i in range(m):
        out.append(l * _inv(t - r) % R)
        l = l * omega % R
        r = r * omega % R
    return out


class StepRadix2Domain:
    kind = "step_radix2"

    def __init__(self, m):
        if m <= 1:
            raise DomainError("step_radix2(): expected m > 1")
        self.m = m
        self.big_m = 1 << (ceil_log2(m) - 1)
        self.small_m = m - self.big_m
        if self.small_m != 1 << ceil_log2(self.small_m):
            raise DomainError("step_radix2(): expected small_m == 1<<log2(small_m)")
        if ceil_log2(m) > FR_TWO_ADICITY:
            raise DomainError("step_radix2(): expected logm <= s")
        self.omega = fr_root_of_unity(ceil_log2(m))          # order 2 * big_m
        self.big_omega = self.omega * self.omega % R
        self.small_omega = fr_root_of_unity(ceil_log2(self.small_m))

    # step_radix2_domain.tcc FFT
    def fft(self, a):
        assert len(a) == self.m
        big_m, small_m = self.big_m, self.small_m
        a = [x % R for x in a]
        c, d = [0] * big_m, [0] * big_m
        w = 1
        for i in range(big_m):
            if i < small_m:
                c[i] = (a[i] + a[i + big_m]) % R
                d[i] = w * (a[i] - a[i + big_m]) % R
            else:
                c[i] = a[i]
                d[i] = w * a[i] % R
            w = w * self.omega % R
        e = [0] * small_m
        compr = big_m // small_m
        for i in range(small_m):
            for j in range(compr):
                e[i] = (e[i] + d[i + j * small_m]) % R
        c = ntt(c) if big_m > 1 else c
        e = ntt(e) if small_m > 1 else e
        return c + e

    # step_radix2_domain.tcc iFFT
    def inverse_fft(self, a):
        assert len(a) == self.m
        big_m, small_m = self.big_m, self.small_m
        U0 = [x % R for x in a[:big_m]]
        U1 = [x % R for x in a[big_m:]]
        U0 = ntt(U0, inverse=True) if big_m > 1 else U0         # includes the 1/big_m scaling
        U1 = ntt(U1, inverse=True) if small_m > 1 else U1
        tmp, w = [], 1
        for i in range(big_m):
            tmp.append(U0[i] * w % R)
            w = w * self.omega % R
        out = [0] * self.m
        for i in range(small_m, big_m):
            out[i] = U0[i]
        compr = big_m // small_m
        for i in range(small_m):
            for j in range(1, compr):           # j = 0 is the term a[i] - a[i + big_m] being solved for
                U1[i] = (U1[i] - tmp[i + j * small_m]) % R
        omega_inv, w = _inv(self.omega), 1
        for i in range(small_m):
            U1[i] = U1[i] * w % R
            w = w * omega_inv % R
        over_two = _inv(2)
        for i in range(small_m):
            out[i] = (U0[i] + U1[i]) * over_two % R
            out[big_m + i] = (U0[i] - U1[i]) * over_two % R
        return out

    def coset_fft(self, a, g):
        w, b = 1, []
        for x in a:
            b.append(x * w % R)
            w = w * g % R
        return self.fft(b)

    def inverse_coset_fft(self, a, g):
        b = self.inverse_fft(a)
        gi, w, out = _inv(g), 1, []
        for x in b:
            out.append(x * w % R)
            w = w * gi % R
        return out

    def get_domain_element(self, idx):
        if idx < self.big_m:
            return pow(self.big_omega, idx, R)
        return self.omega * pow(self.small_omega, idx - self.big_m, R) % R

    def compute_vanishing_polynomial(self, t):
        return (pow(t, self.big_m, R) - 1) * (pow(t, self.small_m, R) - pow(self.omega, self.small_m, R)) % R

    def evaluate_all_lagrange_polynomials(self, t):
        big_m, small_m = self.big_m, self.small_m
        inner_big = _radix2_lagrange(big_m, t)
        inner_small = _radix2_lagrange(small_m, t * _inv(self.omega) % R)
        out = [0] * self.m
        L0 = (pow(t, small_m, R) - pow(self.omega, small_m, R)) % R
        omega_to_small_m = pow(self.omega, small_m, R)
        big_omega_to_small_m = pow(self.big_omega, small_m, R)
        elt = 1
        for i in range(big_m):
            out[i] = inner_big[i] * L0 % R * _inv(elt - omega_to_small_m) % R
            elt = elt * big_omega_to_small_m % R
        L1 = (pow(t, big_m, R) - 1) * _inv(pow(self.omega, big_m, R) - 1) % R
        for i in range(small_m):
            out[big_m + i] = L1 * inner_small[i] % R
        return out

    def add_poly_z(self, coeff, H):
        assert len(H) == self.m + 1
        H = list(H)
        w = pow(self.omega, self.small_m, R)
        H[self.m] = (H[self.m] + coeff) % R
        H[self.big_m] = (H[self.big_m] - coeff * w) % R
        H[self.small_m] = (H[self.small_m] - coeff) % R
        H[0] = (H[0] + coeff * w) % R
        return H

    def divide_by_z_on_coset(self, P, g=FR_GENERATOR):
        big_m, small_m = self.big_m, self.small_m
        Z0 = (pow(g, big_m, R) - 1) % R
        cZ0 = pow(g, small_m, R) * Z0 % R
        w1 = pow(self.omega, small_m, R)
        w2 = pow(self.omega, 2 * small_m, R)
        out, elt = list(P), 1
        for i in range(big_m):
            out[i] = out[i] * _inv(cZ0 * elt - w1 * Z0) % R
            elt = elt * w2 % R
        go = g * self.omega % R
        Z1i = _inv((pow(go, big_m, R) - 1) * (pow(go, small_m, R) - w1))
        for i in range(small_m):
            out[big_m + i] = out[big_m + i] * Z1i % R
        return out


def make_evaluation_domain(min_size):
    """get_evaluation_domain.tcc: the first of basic_radix2(min_size), extended_radix2(min_size), step_radix2(min_size),
    basic_radix2(big + rounded_small), extended_radix2(..), step_radix2(big + rounded_small), geometric, arithmetic that
    constructs.  For BLS12-381 Fr (two-adicity 32) and min_size <= 2^32 one of the radix-2 family always does; the extended
    domain needs m = 2^33 and the sequence domains need generators Fr does not define -- unreachable here."""
    if min_size <= 1:
        raise DomainError("make_evaluation_domain: min_size must exceed 1")
    log = ceil_log2(min_size)
    big = 1 << (log - 1)
    small = min_size - big
    rounded_small = 1 << ceil_log2(small)
    for ctor, m in ((BasicRadix2Domain, min_size), (StepRadix2Domain, min_size),
                    (BasicRadix2Domain, big + rounded_small), (StepRadix2Domain, big + rounded_small)):
        try:
            return ctor(m)
        except DomainError:
            continue
    raise DomainError("make_evaluation_domain: no radix-2 family domain of this size")


# ---- definitions the restatement is checked against (tests/test_oracle.py)
def evaluate_naive(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def lagrange_naive(points, t):
    out = []
    for i, xi in enumerate(points):
        num, den = 1, 1
        for j, xj in enumerate(points):
            if j != i:
                num = num * (t - xj) % R
                den = den * (xi - xj) % R
        out.append(num * _inv(den) % R)
    return out
