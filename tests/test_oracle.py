"""CPU tests: the oracle chain.  Python big-int oracle -> golden fixtures -> C restatement -> Groth16
pairing equation; and the one reference-supplied byte string (bin/cli/src/data.bin[0:192])."""
import os

import numpy as np
import pytest

import bls12_381 as o
from conftest import GOLDEN, I, L, dec1, dec2, fr_array, fr_ints, g1_limbs, g2_limbs, load_golden


def test_python_oracle_self_check():
    assert o.self_check()


def test_reference_data_bin_proof_kat():
    """/root/reference/bin/cli/src/data.bin[0:192] = ZCash-compressed Groth16 proof A||B||C."""
    d = bytes.fromhex(open(os.path.join(GOLDEN, "data_bin_proof.hex")).read().strip())
    assert len(d) == 192
    A = o.g1_decompress(d[0:48]); B = o.g2_decompress(d[48:144]); Cc = o.g1_decompress(d[144:192])
    assert o.G1.is_on_curve(A) and o.G2.is_on_curve(B) and o.G1.is_on_curve(Cc)
    assert o.G1.in_subgroup(A) and o.G2.in_subgroup(B) and o.G1.in_subgroup(Cc)
    assert o.g1_compress(A) + o.g2_compress(B) + o.g1_compress(Cc) == d


def test_c_oracle_field_vs_golden(cref):
    g = load_golden("field.json")
    for c in g["fp"]:
        a, b = int(c["a"], 16), int(c["b"], 16)
        assert I(cref.fp_mul(L(a, 6), L(b, 6))) == int(c["mul"], 16)
        assert I(cref.fp_inv(L(a, 6))) == int(c["inv_a"], 16)
    for c in g["fr"]:
        a, b = int(c["a"], 16), int(c["b"], 16)
        assert I(cref.fr_mul(L(a, 4), L(b, 4))) == int(c["mul"], 16)
        assert I(cref.fr_inv(L(a, 4))) == int(c["inv_a"], 16)


def test_c_oracle_curve_vs_golden(cref):
    g = load_golden("curve.json")
    c = g["g1"]
    P1, P2 = dec1(c["P1"]), dec1(c["P2"])
    assert o.g1_from_limbs(cref.g1_add(g1_limbs(P1), g1_limbs(P2))) == dec1(c["P1_plus_P2"])
    assert o.g1_from_limbs(cref.g1_add(g1_limbs(P1), g1_limbs(P1))) == dec1(c["dbl_P1"])
    assert o.g1_from_limbs(cref.g1_add(g1_limbs(P1), g1_limbs(o.G1.neg(P1)))) is None
    assert o.g1_from_limbs(cref.g1_mul(g1_limbs(P1), L(int(c["k"], 16), 4))) == dec1(c["k_P1"])
    assert o.g1_from_limbs(cref.g1_mul(g1_limbs(o.G1.gen), L(o.R - 1, 4))) == dec1(c["r_minus_1_gen"])
    assert o.g1_from_limbs(cref.g1_mul(g1_limbs(o.G1.gen), L(o.R, 4))) is None
    c = g["g2"]
    P1, P2 = dec2(c["P1"]), dec2(c["P2"])
    assert o.g2_from_limbs(cref.g2_add(g2_limbs(P1), g2_limbs(P2))) == dec2(c["P1_plus_P2"])
    assert o.g2_from_limbs(cref.g2_add(g2_limbs(P1), g2_limbs(P1))) == dec2(c["dbl_P1"])
    assert o.g2_from_limbs(cref.g2_mul(g2_limbs(P1), L(int(c["k"], 16), 4))) == dec2(c["k_P1"])


def test_c_oracle_msm_vs_golden(cref):
    for case in load_golden("msm.json"):
        scal = fr_array([int(s, 16) for s in case["scalars"]])
        if case["group"] == "g1":
            bases = np.stack([g1_limbs(dec1(p)) for p in case["bases"]])
            for mixed in (False, True):
                assert o.g1_from_limbs(cref.msm_g1(bases, scal, mixed)) == dec1(case["result"]), case["n"]
        else:
            bases = np.stack([g2_limbs(dec2(p)) for p in case["bases"]])
            for mixed in (False, True):
                assert o.g2_from_limbs(cref.msm_g2(bases, scal, mixed)) == dec2(case["result"]), case["n"]


def test_c_oracle_ntt_vs_golden(cref):
    for case in load_golden("ntt.json"):
        a = fr_array([int(x, 16) for x in case["input"]])
        g7 = L(7, 4)
        assert fr_ints(cref.ntt_fr(a)) == [int(x, 16) for x in case["fft"]]
        assert fr_ints(cref.ntt_fr(a, inverse=True)) == [int(x, 16) for x in case["inverse_fft"]]
        assert fr_ints(cref.ntt_fr(a, coset=g7)) == [int(x, 16) for x in case["coset_fft_g7"]]
        assert fr_ints(cref.ntt_fr(a, inverse=True, coset=g7)) == [int(x, 16) for x in case["inverse_coset_fft_g7"]]


def test_c_oracle_batch_mul_and_msm_identity(cref):
    """bases k_i*G: multiexp(bases, s) must equal (sum k_i s_i)*G -- the size-independent property the
    full-size GPU test uses."""
    gen = o.splitmix64(99)
    n = 50
    ks = [o.rand_fr(gen) for _ in range(n)]; ss = [o.rand_fr(gen) for _ in range(n)]
    B1 = cref.g1_batch_mul_gen(fr_array(ks)); B2 = cref.g2_batch_mul_gen(fr_array(ks))
    assert o.g1_from_limbs(B1[3]) == o.G1.mul(o.G1.gen, ks[3])
    assert o.g2_from_limbs(B2[3]) == o.G2.mul(o.G2.gen, ks[3])
    e = sum(a * b for a, b in zip(ks, ss)) % o.R
    assert o.g1_from_limbs(cref.msm_g1(B1, fr_array(ss))) == o.G1.mul(o.G1.gen, e)
    assert o.g2_from_limbs(cref.msm_g2(B2, fr_array(ss))) == o.G2.mul(o.G2.gen, e)


DOMAIN_SIZES = [2, 3, 4, 5, 6, 7, 9, 10, 12, 17, 20, 24, 33, 40, 48, 65, 100]


@pytest.mark.parametrize("min_size", DOMAIN_SIZES)
def test_python_domains_meet_their_definitions(min_size):
    """make_evaluation_domain + basic/step radix-2 domains of the Python oracle against the definitions: fft = evaluation at the
    domain's elements, Lagrange basis by the product formula, Z = prod (t - x_i), add_poly_z, divide_by_z_on_coset."""
    import random
    import domains as dm
    rnd = random.Random(min_size)
    d = dm.make_evaluation_domain(min_size)
    m = d.m
    assert m >= min_size
    pts = [d.get_domain_element(i) for i in range(m)]
    assert len(set(pts)) == m
    a = [rnd.randrange(o.R) for _ in range(m)]
    ev = d.fft(a)
    assert ev == [dm.evaluate_naive(a, x) for x in pts]
    assert d.inverse_fft(ev) == a
    cev = d.coset_fft(a, 7)
    assert cev == [dm.evaluate_naive(a, 7 * x % o.R) for x in pts]
    assert d.inverse_coset_fft(cev, 7) == a
    t = rnd.randrange(o.R)
    assert d.evaluate_all_lagrange_polynomials(t) == dm.lagrange_naive(pts, t)
    z = 1
    for x in pts:
        z = z * (t - x) % o.R
    assert d.compute_vanishing_polynomial(t) == z
    assert dm.evaluate_naive(d.add_poly_z(5, [0] * (m + 1)), t) == 5 * z % o.R
    P = [rnd.randrange(o.R) for _ in range(m)]
    q = d.divide_by_z_on_coset(P)
    assert all(q[i] * d.compute_vanishing_polynomial(7 * pts[i] % o.R) % o.R == P[i] for i in range(m))


def test_make_evaluation_domain_selection():
    """get_evaluation_domain order: exact power of two, exact step size, else big + rounded_small."""
    import domains as dm
    exp = {2: (2, "basic_radix2"), 3: (3, "step_radix2"), 7: (8, "basic_radix2"), 11: (12, "step_radix2"), 65: (65, "step_radix2"),
           100: (128, "basic_radix2"), 600: (640, "step_radix2"), (1 << 20) + 5: ((1 << 20) + 8, "step_radix2"),
           (1 << 20) - 5: (1 << 20, "basic_radix2"), (1 << 20) + (1 << 19) + 1: (1 << 21, "basic_radix2")}
    for k, (m, kind) in exp.items():
        d = dm.make_evaluation_domain(k)
        assert (d.m, d.kind) == (m, kind), k
    with pytest.raises(dm.DomainError):
        dm.make_evaluation_domain(1)


@pytest.mark.parametrize("min_size", DOMAIN_SIZES + [320, 1025, 1500])
def test_c_oracle_domains_vs_python(cref, min_size):
    import domains as dm
    d = dm.make_evaluation_domain(min_size)
    c = cref.Domain(min_size)
    assert (c.m, c.is_step) == (d.m, d.kind == "step_radix2")
    gen = o.splitmix64(min_size)
    a = [o.rand_fr(gen) for _ in range(d.m)]
    A = fr_array(a); g7 = L(7, 4)
    assert fr_ints(c.fft(A)) == d.fft(a)
    assert fr_ints(c.inverse_fft(A)) == d.inverse_fft(a)
    assert fr_ints(c.coset_fft(A, g7)) == d.coset_fft(a, 7)
    assert fr_ints(c.inverse_coset_fft(A, g7)) == d.inverse_coset_fft(a, 7)
    assert fr_ints(c.divide_by_z_on_coset(A, g7)) == d.divide_by_z_on_coset(a, 7)
    t = o.rand_fr(gen)
    assert fr_ints(c.evaluate_all_lagrange_polynomials(L(t, 4))) == d.evaluate_all_lagrange_polynomials(t)
    assert I(c.compute_vanishing_polynomial(L(t, 4))) == d.compute_vanishing_polynomial(t)
    for idx in (0, 1, d.m // 2, d.m - 1):
        assert I(c.get_domain_element(idx)) == d.get_domain_element(idx)


@pytest.mark.parametrize("nc,ni", [(120, 4), (70, 3), (129, 2)])
def test_c_oracle_groth16_proof_verifies(cref, nc, ni):
    """generator + witness_map + prover of the C restatement produce a proof that satisfies the
    Groth16 pairing equation under an independent pure-Python pairing -- on a basic radix-2 domain (125 -> 128) and on step
    domains (74 -> 80 = 64 + 16; 132 -> 132 = 128 + 4)."""
    import pairing as pg
    gen = o.splitmix64(5)
    cs, wit = cref.R1CS.synth(nc, ni, 4)
    assert cs.m == {(120, 4): 128, (70, 3): 80, (129, 2): 132}[(nc, ni)]
    assert cs.is_satisfied(wit)
    bad = wit.copy(); bad[60] = L(2, 4)          # a boolean or product wire set to 2 breaks its constraint
    assert not cs.is_satisfied(bad)
    tox = fr_array([o.rand_fr(gen) for _ in range(5)])
    kp = cref.Keypair(cs, tox)
    r, s = L(o.rand_fr(gen), 4), L(o.rand_fr(gen), 4)
    A, B, Cc = kp.prove(wit, r, s)
    vk = dict(alpha_g1=o.g1_from_limbs(kp.part("alpha_g1")[0]), beta_g2=o.g2_from_limbs(kp.part("beta_g2")[0]),
              gamma_g2=o.g2_from_limbs(kp.part("gamma_g2")[0]), delta_g2=o.g2_from_limbs(kp.part("delta_g2")[0]),
              gamma_ABC_g1=[o.g1_from_limbs(x) for x in kp.part("gamma_ABC_g1")])
    pub = [I(wit[i]) for i in range(ni)]
    proof = (o.g1_from_limbs(A), o.g2_from_limbs(B), o.g1_from_limbs(Cc))
    assert pg.groth16_verify(vk, pub, proof)
    pub[1] = (pub[1] + 1) % o.R
    assert not pg.groth16_verify(vk, pub, proof)
    kp.free(); cs.free()
