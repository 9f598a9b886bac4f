"""GPU parity: evaluation_domain<Fr> handles -- make_evaluation_domain's choice, the step radix-2 domain's transforms,
evaluate_all_lagrange_polynomials, divide_by_z_on_coset, get_domain_element, compute_vanishing_polynomial, add_poly_z --
against the C oracle (same algorithms, serial) and the Python oracle (checked against the definitions in test_oracle.py)."""
import numpy as np
import pytest

import bls12_381 as o
import domains as dm
from conftest import I, L, fr_ints, rand_fr_array

import vote_saver_protocol_amd as v

pytestmark = pytest.mark.gpu

G7 = L(7, 4)
SIZES = [2, 3, 5, 6, 9, 12, 17, 24, 40, 65, 100, 320, 1025, 1500, 2049, 5031, (1 << 16) + 1, (1 << 16) + (1 << 15), (1 << 17) + 5]


@pytest.mark.parametrize("min_size", SIZES)
def test_domain_all_operations_vs_c_oracle(ctx, cref, min_size):
    ref = cref.Domain(min_size)
    dom = v.make_evaluation_domain(ctx, min_size)
    assert (dom.m, dom.kind == "step_radix2") == (ref.m, ref.is_step)
    a = rand_fr_array(dom.m, seed=min_size)
    assert np.array_equal(dom.fft(a), ref.fft(a))
    assert np.array_equal(dom.inverse_fft(a), ref.inverse_fft(a))
    assert np.array_equal(dom.coset_fft(a, G7), ref.coset_fft(a, G7))
    assert np.array_equal(dom.inverse_coset_fft(a, G7), ref.inverse_coset_fft(a, G7))
    assert np.array_equal(dom.inverse_fft(dom.fft(a)), a)
    assert np.array_equal(dom.divide_by_z_on_coset(a), ref.divide_by_z_on_coset(a, G7))
    t = rand_fr_array(1, seed=7 * min_size)[0]
    assert np.array_equal(dom.evaluate_all_lagrange_polynomials(t), ref.evaluate_all_lagrange_polynomials(t))
    assert np.array_equal(dom.compute_vanishing_polynomial(t), ref.compute_vanishing_polynomial(t))
    for idx in (0, 1, dom.m // 2, dom.m - 1):
        assert np.array_equal(dom.get_domain_element(idx), ref.get_domain_element(idx))
    dom.free()


@pytest.mark.parametrize("min_size", [3, 12, 40, 65, 128])
def test_domain_vs_python_oracle_definitions(ctx, min_size):
    """Independent of the C oracle: fft = evaluation at the domain elements; Lagrange at a domain element is an indicator;
    add_poly_z adds coeff * Z."""
    pd = dm.make_evaluation_domain(min_size)
    dom = v.make_evaluation_domain(ctx, min_size)
    a = rand_fr_array(dom.m, seed=min_size + 1)
    ai = fr_ints(a)
    pts = [pd.get_domain_element(i) for i in range(pd.m)]
    assert fr_ints(dom.fft(a)) == [dm.evaluate_naive(ai, x) for x in pts]
    for k in (0, pd.m // 2, pd.m - 1):
        ind = fr_ints(dom.evaluate_all_lagrange_polynomials(L(pts[k], 4)))
        assert ind == [1 if i == k else 0 for i in range(pd.m)]
    t = 123456789
    H = rand_fr_array(dom.m + 1, seed=5)
    got = fr_ints(dom.add_poly_z(L(5, 4), H))
    assert got == pd.add_poly_z(5, fr_ints(H))
    assert (dm.evaluate_naive(got, t) - dm.evaluate_naive(fr_ints(H), t)) % o.R == 5 * pd.compute_vanishing_polynomial(t) % o.R
    dom.free()


def test_step_domain_large_vs_c_oracle(ctx, cref):
    """2^20 + 2^17 elements (the shape of a real circuit's domain): all four transforms bit for bit, and witness_map over it."""
    min_size = (1 << 20) + (1 << 17) - 11
    ref = cref.Domain(min_size)
    dom = v.make_evaluation_domain(ctx, min_size)
    assert dom.m == (1 << 20) + (1 << 17) and dom.kind == "step_radix2"
    a = rand_fr_array(dom.m, seed=21)
    A = dom.fft(a)
    assert np.array_equal(A, ref.fft(a))
    assert np.array_equal(dom.inverse_fft(A), a)
    assert np.array_equal(dom.inverse_coset_fft(a, G7), ref.inverse_coset_fft(a, G7))
    assert np.array_equal(dom.inverse_coset_fft(dom.coset_fft(a, G7), G7), a)
    dom.free()


def test_step_domain_tiny_small_part_large(ctx, cref):
    """m = 2^18 + 1: the strided sums run over 2^18 terms for a single output (the tree of partial sums)."""
    ref = cref.Domain((1 << 18) + 1)
    dom = v.make_evaluation_domain(ctx, (1 << 18) + 1)
    a = rand_fr_array(dom.m, seed=22)
    assert np.array_equal(dom.fft(a), ref.fft(a))
    assert np.array_equal(dom.inverse_fft(a), ref.inverse_fft(a))
    assert np.array_equal(dom.divide_by_z_on_coset(a), ref.divide_by_z_on_coset(a, G7))
    t = rand_fr_array(1, seed=23)[0]
    assert np.array_equal(dom.evaluate_all_lagrange_polynomials(t), ref.evaluate_all_lagrange_polynomials(t))
    dom.free()
