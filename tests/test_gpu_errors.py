"""GPU: error behaviour of the C ABI -- negative status codes and a readable message, never an abort
(the reference's convention is BOOST_ASSERT -> std::exit(1), bin/cli/src/main.cpp:24-33)."""
import ctypes as C

import numpy as np
import pytest

import vote_saver_protocol_amd as v
from conftest import rand_fr_array

pytestmark = pytest.mark.gpu
NULL = None


def test_status_codes_and_messages(ctx, cref):
    lib = ctx.lib
    out = np.zeros(12, np.uint64); inf = C.c_int(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    # null pointers
    assert lib.vsp_msm_g1(ctx.h, NULL, NULL, 5, p(out), C.byref(inf)) == -1
    assert "null" in ctx.last_error()
    assert lib.vsp_ntt_fr(ctx.h, NULL, 3, 0, NULL) == -1
    a = rand_fr_array(8, 1)
    assert lib.vsp_ntt_fr(ctx.h, p(a), 29, 0, NULL) == -4                      # log_m > 28: unsupported
    zero_g = np.zeros(4, np.uint64)
    assert lib.vsp_ntt_fr(ctx.h, p(a), 3, 0, p(zero_g)) == -1                  # coset generator 0
    assert lib.vsp_msm_resident(ctx.h, NULL, 0, 1, NULL, p(out), C.byref(inf)) == -1
    assert lib.vsp_fold_jacobian(ctx.h, 3, p(np.zeros(18, np.uint64)), 1, p(out), C.byref(inf)) == -1   # group must be 1 or 2
    assert lib.vsp_msm_finish_jacobian(ctx.h, 7, p(np.zeros(18, np.uint64))) == -1
    assert lib.vsp_groth16_prove(ctx.h, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == -1
    assert not lib.vsp_groth16_generate(ctx.h, NULL, NULL, 0)
    assert not lib.vsp_r1cs_upload(ctx.h, 1, 2, 1, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL)
    # a null context never crashes
    assert lib.vsp_synchronize(NULL) == -1
    assert lib.vsp_msm_g1(NULL, NULL, NULL, 0, NULL, NULL) == -1
    assert lib.vsp_last_error(NULL) == b"null context"
    assert not lib.vsp_create(9999)                                             # no such device
    # column index out of range in a constraint system
    rp = np.array([0, 1], np.uint32); ci = np.array([7], np.uint32); co = np.array([[1, 0, 0, 0]], np.uint64)
    assert not lib.vsp_r1cs_upload(ctx.h, 1, 0, 2, p(rp), p(ci), p(co), p(rp), p(ci), p(co), p(rp), p(ci), p(co))
    assert "column" in ctx.last_error()
    # evaluation-domain handles
    assert not lib.vsp_domain_create(ctx.h, 1) and "min_size" in ctx.last_error()          # a domain has at least two elements
    assert not lib.vsp_domain_create(ctx.h, (1 << 28) + 1)                                   # larger than 2^28
    assert lib.vsp_domain_size(NULL) == 0 and lib.vsp_domain_kind(NULL) == -1
    dom = lib.vsp_domain_create(ctx.h, 12)
    assert dom and lib.vsp_domain_size(dom) == 12 and lib.vsp_domain_kind(dom) == 1
    assert lib.vsp_domain_fft(ctx.h, dom, NULL, 0, NULL) == -1
    assert lib.vsp_domain_fft(ctx.h, NULL, p(a), 0, NULL) == -1
    a12 = rand_fr_array(12, 4)
    assert lib.vsp_domain_fft(ctx.h, dom, p(a12), 0, p(zero_g)) == -1                        # coset generator 0
    o4 = np.zeros(4, np.uint64)
    assert lib.vsp_domain_element(ctx.h, dom, 12, p(o4)) == -1                               # index out of range
    assert lib.vsp_domain_lagrange(ctx.h, dom, NULL, p(a12)) == -1
    assert lib.vsp_domain_witness_map_h(ctx.h, dom, p(a12), p(a12), NULL, p(a12)) == -1
    lib.vsp_domain_free(ctx.h, dom); lib.vsp_domain_free(ctx.h, NULL)
    # wire format: malformed encodings are refused with a status, not decoded
    enc = (C.c_uint8 * 48)(*([0x1f] + [0] * 47))                                              # compression flag missing
    assert lib.vsp_g1_decompress(enc, 1, p(out), C.byref(inf)) == -1
    # the context still works afterwards
    b = cref.g1_batch_mul_gen(rand_fr_array(10, 2)); s = rand_fr_array(10, 3)
    assert np.array_equal(v.multiexp(ctx, b, s, 1), cref.msm_g1(b, s))


def test_boundary_validation_of_scalars_and_bases(ctx, cref):
    """include/vsp.h: VSP_ERR_ARG for a scalar >= r, a base coordinate >= p, a base off the curve (the reference's field types
    cannot hold such values; a raw-limb boundary must refuse them instead of returning another point)."""
    import bls12_381 as o
    from conftest import L, g1_limbs, g2_limbs
    n = 300
    bases = cref.g1_batch_mul_gen(rand_fr_array(n, 5)); ss = rand_fr_array(n, 6)
    good = cref.msm_g1(bases, ss)
    # --- scalars: r itself, r + 5, 2^256 - 1, at the first, a middle and the last position
    for bad_val in (o.R, o.R + 5, (1 << 256) - 1):
        for pos in (0, 137, n - 1):
            s2 = ss.copy(); s2[pos] = L(bad_val, 4)
            with pytest.raises(v.VspError, match="canonical"):
                v.multiexp(ctx, bases, s2, 1)
    s2 = ss.copy(); s2[3] = L(o.R - 1, 4)                                      # the largest canonical value is fine
    assert np.array_equal(v.multiexp(ctx, bases, s2, 1), cref.msm_g1(bases, s2))
    big = 5000                                                                 # above the census threshold, resident + pipelined form
    bb = cref.g1_batch_mul_gen(rand_fr_array(big, 7)); sb = rand_fr_array(big, 8); sb[4321] = L(o.R, 4)
    B = ctx.upload_bases(bb, 1); d_s = ctx.to_device(sb)
    with pytest.raises(v.VspError, match="canonical"):
        B.msm(d_s)
    B.msm_launch(2, d_s)
    with pytest.raises(v.VspError, match="canonical"):
        B.msm_finish_jacobian(2)
    sb[4321] = 0; ctx.h2d(d_s, sb)
    got, _ = B.msm(d_s)                                                        # the slot recovers
    assert np.array_equal(got, cref.msm_g1(bb, sb, mixed=True))
    ctx.dfree(d_s); B.free()
    # --- bases: coordinate >= p
    b2 = bases.copy(); b2[7, :6] = L(o.P, 6)
    with pytest.raises(v.VspError, match="canonical"):
        ctx.upload_bases(b2, 1)
    b2 = bases.copy(); b2[n - 1, 6:] = L(o.P + 1, 6)
    with pytest.raises(v.VspError, match="canonical"):
        v.multiexp(ctx, b2, ss, 1)
    # --- bases: not on the curve (x of one point, y of another)
    b3 = bases.copy(); b3[11, 6:] = bases[12, 6:]
    with pytest.raises(v.VspError, match="curve"):
        ctx.upload_bases(b3, 1)
    ctx.set_option("bases_check_curve", 0)                                     # the check is an option; the coordinate check is not
    try:
        ctx.upload_bases(b3, 1).free()
        with pytest.raises(v.VspError, match="canonical"):
            ctx.upload_bases(b2, 1)
    finally:
        ctx.set_option("bases_check_curve", 1)
    g2b = cref.g2_batch_mul_gen(rand_fr_array(40, 9))
    g2bad = g2b.copy(); g2bad[5, 12:18] = g2b[6, 12:18]
    with pytest.raises(v.VspError, match="curve"):
        ctx.upload_bases(g2bad, 2)
    g2bad = g2b.copy(); g2bad[0, 6:12] = L(o.P, 6)
    with pytest.raises(v.VspError, match="canonical"):
        ctx.upload_bases(g2bad, 2)
    inf = bases.copy(); inf[3] = 0                                             # infinity stays legal
    assert np.array_equal(v.multiexp(ctx, inf, ss, 1), cref.msm_g1(inf, ss))
    assert np.array_equal(v.multiexp(ctx, bases, ss, 1), good)


def test_prover_refuses_non_canonical_inputs_and_recovers(ctx, cref):
    import bls12_381 as o
    from conftest import L, fr_array
    gen = o.splitmix64(9)
    cs, wit = cref.R1CS.synth(600, 4, 9)
    tox = fr_array([o.rand_fr(gen) for _ in range(5)])
    dcs = v.R1CS(ctx, 600, 4, cs.num_vars, *cs.export())
    kp = v.Keypair(ctx, dcs, tox)
    r, s = L(o.rand_fr(gen), 4), L(o.rand_fr(gen), 4)
    ok = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
    with pytest.raises(v.VspError, match="canonical"):
        v.groth16_prove(ctx, dcs, kp.pk, wit, L(o.R, 4), s)
    w2 = wit.copy(); w2[77] = L(o.R + 1, 4)
    with pytest.raises(v.VspError, match="canonical"):                         # reported when the multi-exponentiations finish; all slots drained
        v.groth16_prove(ctx, dcs, kp.pk, w2, r, s)
    again = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
    assert all(np.array_equal(a, b) for a, b in zip(ok[:3], again[:3])) and ok[3] == again[3]
    A, B, Cm = cs.export()
    co = A[2].copy(); co[0] = L(o.R, 4)
    with pytest.raises(v.VspError, match="canonical"):
        v.R1CS(ctx, 600, 4, cs.num_vars, (A[0], A[1], co), B, Cm)
    kp.free(); dcs.free(); cs.free()


def test_fp28_kernel_known_answer_check_and_fallback(cref):
    """The 28-bit-limb accumulation kernels are checked THROUGH k_accum28 against the generic kernel when a context builds its first
    table; a failed check switches the context to the generic kernel, results unchanged."""
    c = v.Context(0)
    try:
        n = 3000
        bases = cref.g1_batch_mul_gen(rand_fr_array(n, 21)); ss = rand_fr_array(n, 22)
        b2 = cref.g2_batch_mul_gen(rand_fr_array(1100, 23))
        exp = cref.msm_g1(bases, ss)
        assert np.array_equal(v.multiexp(c, bases, ss, 1), exp)
        assert c.stat("msm_fp28_selfcheck_g1") == 1.0
        B2 = c.upload_bases(b2, 2)
        assert c.stat("msm_fp28_selfcheck_g2") == 1.0
        B2.free()
    finally:
        c.close()
    c = v.Context(0)
    try:
        c.set_option("msm_fp28_selfcheck_fault", 1)                            # test hook: the comparison reports a mismatch
        assert np.array_equal(v.multiexp(c, bases, ss, 1), exp)                # generic 12 x 32-bit kernel took over
        assert c.stat("msm_fp28_selfcheck_g1") == -1.0 and "known-answer" in c.last_error()
        B = c.upload_bases(bases, 1).precompute(0)
        d_s = c.to_device(ss)
        got, _ = B.msm(d_s)
        assert np.array_equal(got, exp)
        c.dfree(d_s); B.free()
    finally:
        c.close()


def test_ntt29_kernel_known_answer_check_and_fallback(cref):
    """The 9 x 29-bit butterfly kernel is checked THROUGH k_ntt29_pass against the 8 x 32-bit kernel before a context's first transform
    (VERDICT round 2, item 9); a failed check switches the context to the 8 x 32-bit kernel, results unchanged -- also through the
    step-domain glue and a whole proof."""
    g7 = np.array([7, 0, 0, 0], np.uint64)
    a = rand_fr_array(1 << 12, 91)
    with v.Context(0) as c:
        dom = v.EvaluationDomain(c, 1 << 12)
        assert np.array_equal(dom.coset_fft(a, g7), cref.ntt_fr(a, coset=g7))
        assert c.stat("ntt_fr29_selfcheck") == 1.0 and c.stat("ntt_fr29") == 1
        dom.free()
    with v.Context(0) as c:
        c.set_option("ntt_fr29_selfcheck_fault", 1)                            # test hook: the comparison reports a mismatch
        sd = v.make_evaluation_domain(c, 4096 + 300)                           # step domain first: the check runs ahead of its table set-up
        sref = cref.Domain(4096 + 300)
        a3 = rand_fr_array(sd.m, 92)
        g5 = np.array([5, 0, 0, 0], np.uint64)
        assert np.array_equal(sd.coset_fft(a3, g5), sref.coset_fft(a3, g5))
        assert c.stat("ntt_fr29_selfcheck") == -1.0 and c.stat("ntt_fr29") == 0 and "known-answer" in c.last_error()
        assert np.array_equal(sd.inverse_coset_fft(a3, g5), sref.inverse_coset_fft(a3, g5))
        dom = v.EvaluationDomain(c, 1 << 12)
        assert np.array_equal(dom.inverse_coset_fft(a, g7), cref.ntt_fr(a, inverse=True, coset=g7)) and c.stat("ntt_fr29") == 0
        cs, wit = cref.R1CS.synth(500, 3, 93)
        tox, r, s = rand_fr_array(5, 94), rand_fr_array(1, 95)[0], rand_fr_array(1, 96)[0]
        dcs = v.R1CS(c, 500, 3, cs.num_vars, *cs.export())
        kp, ref = v.Keypair(c, dcs, tox), cref.Keypair(cs, tox)
        pa, pb, pc, _ = v.groth16_prove(c, dcs, kp.pk, wit, r, s)
        ea, eb, ec = ref.prove(wit, r, s)
        assert np.array_equal(pa, ea) and np.array_equal(pb, eb) and np.array_equal(pc, ec)
        kp.free(); dcs.free(); ref.free(); cs.free(); dom.free(); sd.free()


def test_r1cs_upload_refuses_null_arrays_and_bad_row_pointers(ctx):
    """vsp_r1cs_upload with nnz > 0 and a NULL column / coefficient array returned a crash, not VSP_ERR_ARG (small)"""
    lib = ctx.lib
    rp = np.array([0, 1, 2], np.uint32); ci = np.array([0, 1], np.uint32); co = np.zeros((2, 4), np.uint64); co[:, 0] = 1
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    good = [p(rp), p(ci), p(co)]
    h = lib.vsp_r1cs_upload(ctx.h, 2, 1, 2, *(good * 3))
    assert h
    lib.vsp_r1cs_free(ctx.h, h)
    for hole in (1, 2, 4, 8):                                                  # a NULL col_* / coef_* with nnz = 2
        args = good * 3; args[hole] = None
        assert not lib.vsp_r1cs_upload(ctx.h, 2, 1, 2, *args) and "null column" in ctx.last_error()
    rp_bad = np.array([0, 2, 1], np.uint32)                                    # decreasing row pointers
    assert not lib.vsp_r1cs_upload(ctx.h, 2, 1, 2, p(rp_bad), p(ci), p(co), *(good * 2)) and "row pointers" in ctx.last_error()
    rp_bad = np.array([1, 1, 2], np.uint32)                                    # not starting at 0
    assert not lib.vsp_r1cs_upload(ctx.h, 2, 1, 2, p(rp_bad), p(ci), p(co), *(good * 2)) and "row pointers" in ctx.last_error()
    rp0 = np.zeros(3, np.uint32)                                               # nnz = 0: NULL arrays are fine
    h = lib.vsp_r1cs_upload(ctx.h, 2, 1, 2, p(rp0), None, None, *(good * 2))
    assert h
    lib.vsp_r1cs_free(ctx.h, h)


def test_batch_entry_points_validate_every_vector_and_recover(ctx, cref):
    """vsp_msm_resident_batch / vsp_groth16_prove_batch (round 4): a scalar >= r in ANY vector of a batch is refused (the census counts the
    vectors together), batch sizes outside 1..64 and strides shorter than a vector are refused, a batch of one equals the single call,
    the largest batch works, and after every refusal the context proves again"""
    import bls12_381 as o
    from conftest import L, fr_array
    n, K = 5000, 6
    bases = cref.g1_batch_mul_gen(rand_fr_array(n, 15))
    vecs = np.stack([rand_fr_array(n, 20 + k) for k in range(K)])
    B = ctx.upload_bases(bases, 1)
    d_s = ctx.to_device(vecs.reshape(-1, 4))
    got, _ = B.msm_batch(d_s, K)
    for k in range(K):
        assert np.array_equal(got[k], cref.msm_g1(bases, vecs[k]))
    one, _ = B.msm_batch(d_s, 1)
    assert np.array_equal(one[0], B.msm(d_s)[0])
    for k_bad in (0, 3, K - 1):                                               # a non-canonical scalar in the first, a middle, the last vector
        bad = vecs.copy(); bad[k_bad, 4321] = L(o.R, 4)
        ctx.h2d(d_s, bad.reshape(-1, 4))
        with pytest.raises(v.VspError, match="canonical"):
            B.msm_batch(d_s, K)
    ctx.h2d(d_s, vecs.reshape(-1, 4))
    assert np.array_equal(B.msm_batch(d_s, K)[0], got)                        # the slot recovers
    for bad_K, stride in ((0, n), (65, n), (2, n - 1)):
        with pytest.raises(v.VspError):
            B.msm_batch(d_s, bad_K, stride=stride)
    small = cref.g1_batch_mul_gen(rand_fr_array(40, 16)); sv = np.stack([rand_fr_array(40, 50 + k) for k in range(64)])
    Bs = ctx.upload_bases(small, 1); d_sv = ctx.to_device(sv.reshape(-1, 4))
    g64, _ = Bs.msm_batch(d_sv, 64)
    assert all(np.array_equal(g64[k], cref.msm_g1(small, sv[k])) for k in range(64))
    Bs.free(); ctx.dfree(d_sv); B.free(); ctx.dfree(d_s)
    # the prover
    gen = o.splitmix64(19)
    cs, wit = cref.R1CS.synth(700, 4, 19)
    tox = fr_array([o.rand_fr(gen) for _ in range(5)])
    dcs = v.R1CS(ctx, 700, 4, cs.num_vars, *cs.export())
    kp = v.Keypair(ctx, dcs, tox)
    r, s = L(o.rand_fr(gen), 4), L(o.rand_fr(gen), 4)
    ok = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
    W = np.stack([wit, wit, wit]); R = np.stack([r, r, r]); S = np.stack([s, s, s])
    assert v.groth16_prove_batch(ctx, dcs, kp.pk, W, R, S)[3] == [ok[3]] * 3
    R2 = R.copy(); R2[2] = L(o.R, 4)
    with pytest.raises(v.VspError, match="canonical"):
        v.groth16_prove_batch(ctx, dcs, kp.pk, W, R2, S)
    W2 = W.copy(); W2[1, 77] = L(o.R + 1, 4)
    with pytest.raises(v.VspError, match="canonical"):                         # found by the census of the batch, every slot drained
        v.groth16_prove_batch(ctx, dcs, kp.pk, W2, R, S)
    assert v.groth16_prove_batch(ctx, dcs, kp.pk, W, R, S)[3] == [ok[3]] * 3
    assert v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)[3] == ok[3]
    kp.free(); dcs.free(); cs.free()
