"""GPU parity: vsp_ntt_fr / vsp_witness_map_h (HIP) vs the oracle -- bit-exact canonical Fr values.
Mirrors how crypto3-math's own evaluation_domain tests are shaped (fft vs naive evaluation, fft/inverse
round trip, coset round trip); those tests are absent from the reference snapshot."""
import numpy as np
import pytest

import bls12_381 as o
from conftest import L, fr_array, fr_ints, fr_ints_fast, load_golden, rand_fr_array

import vote_saver_protocol_amd as v

pytestmark = pytest.mark.gpu
G7 = L(7, 4)


def test_ntt_golden_vectors(ctx):
    for case in load_golden("ntt.json"):
        a = fr_array([int(x, 16) for x in case["input"]])
        dom = v.EvaluationDomain(ctx, case["n"])
        assert fr_ints(dom.fft(a)) == [int(x, 16) for x in case["fft"]]
        assert fr_ints(dom.inverse_fft(a)) == [int(x, 16) for x in case["inverse_fft"]]
        assert fr_ints(dom.coset_fft(a, G7)) == [int(x, 16) for x in case["coset_fft_g7"]]
        assert fr_ints(dom.inverse_coset_fft(a, G7)) == [int(x, 16) for x in case["inverse_coset_fft_g7"]]


@pytest.mark.parametrize("log_m", [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 15, 16, 17])
def test_ntt_vs_c_oracle_all_modes(ctx, cref, log_m):
    """one, two and three pass plans; forward / inverse / coset / inverse coset"""
    a = rand_fr_array(1 << log_m, seed=100 + log_m)
    if log_m >= 2:
        a[0] = 0; a[1] = L(o.R - 1, 4)
    dom = v.EvaluationDomain(ctx, 1 << log_m)
    g5 = L(5, 4)
    for inverse, coset in ((False, None), (True, None), (False, G7), (True, G7), (False, g5), (True, g5)):
        got = dom._run(a, inverse, coset)
        exp = cref.ntt_fr(a, inverse=inverse, coset=coset)
        assert np.array_equal(got, exp), (log_m, inverse, coset is not None)


def test_ntt_small_vs_naive_dft(ctx):
    for n in (2, 4, 8, 16):
        a = rand_fr_array(n, seed=n)
        got = fr_ints(v.EvaluationDomain(ctx, n).fft(a))
        assert got == o.dft_naive(fr_ints(a), o.fr_root_of_unity(n.bit_length() - 1))


def test_ntt_rejects_bad_sizes(ctx):
    with pytest.raises(ValueError):
        v.EvaluationDomain(ctx, 11)                      # 8 + 3: neither a power of two nor a step size
    with pytest.raises(ValueError):
        v.api.BasicRadix2Domain(ctx, 12)                 # basic_radix2_domain throws unless m is a power of two
    with pytest.raises(ValueError):
        v.api.StepRadix2Domain(ctx, 16)
    dom = v.EvaluationDomain(ctx, 8)
    with pytest.raises(ValueError):
        dom.fft(rand_fr_array(4, 1))                     # wrong length
    with pytest.raises(ValueError):
        v.make_evaluation_domain(ctx, 1)
    d9, d11, d16 = (v.make_evaluation_domain(ctx, k) for k in (9, 11, 13))
    assert (d9.m, d9.kind, d11.m, d11.kind, d16.m, d16.kind) == (9, "step_radix2", 12, "step_radix2", 16, "basic_radix2")


def test_ntt_full_size_properties(ctx):
    """BASELINE config 3 size (2^22): round trips, linearity and the DFT definition at two outputs --
    size-independent properties; test_ntt_2p20_vs_c_oracle checks a 2^20 transform bit for bit."""
    log_m = 22
    n = 1 << log_m
    a = rand_fr_array(n, seed=3)
    b = rand_fr_array(n, seed=4)
    a[:, 3] >>= np.uint64(1); b[:, 3] >>= np.uint64(1)          # a, b < 2^253 so a + b < r needs no reduction
    dom = v.EvaluationDomain(ctx, n)
    A = dom.fft(a)
    assert np.array_equal(dom.inverse_fft(A), a)
    assert np.array_equal(dom.inverse_coset_fft(dom.coset_fft(a, G7), G7), a)
    # DFT definition: A[0] = sum a_j ; A[1] = sum a_j w^j (Horner over all 2^22 inputs)
    ai = fr_ints_fast(a)
    assert sum(ai) % o.R == fr_ints(A[0:1])[0]
    w = o.fr_root_of_unity(log_m)
    acc = 0
    for x in reversed(ai):
        acc = (acc * w + x) % o.R
    assert acc == fr_ints(A[1:2])[0]
    # linearity: fft(a + b) == fft(a) + fft(b); a + b by a vectorised multi-limb add
    c = np.zeros_like(a)
    carry = np.zeros(n, np.uint64)
    for k in range(4):
        t = a[:, k] + b[:, k]
        c1 = (t < a[:, k]).astype(np.uint64)
        t2 = t + carry
        c2 = (t2 < t).astype(np.uint64)
        c[:, k] = t2
        carry = c1 + c2
    B = dom.fft(b)
    Cf = dom.fft(c)
    sample = list(range(0, n, n // 64)) + [n - 1]
    Ai, Bi, Ci = fr_ints(A[sample]), fr_ints(B[sample]), fr_ints(Cf[sample])
    assert all((x + y) % o.R == z for x, y, z in zip(Ai, Bi, Ci))


def test_ntt_2p20_vs_c_oracle(ctx, cref):
    a = rand_fr_array(1 << 20, seed=20)
    dom = v.EvaluationDomain(ctx, 1 << 20)
    assert np.array_equal(dom.fft(a), cref.ntt_fr(a))
    assert np.array_equal(dom.inverse_coset_fft(a, G7), cref.ntt_fr(a, inverse=True, coset=G7))


@pytest.mark.parametrize("nc,ni", [(5, 1), (60, 3), (1000, 10), (5000, 30)])
def test_witness_map_vs_c_oracle(ctx, cref, nc, ni):
    cs, wit = cref.R1CS.synth(nc, ni, seed=nc)
    H, Az, Bz, Cz = cs.witness_map(wit, want_abc=True)
    got = v.witness_map_h(ctx, Az, Bz, Cz)
    assert np.array_equal(got, H)
    cs.free()
