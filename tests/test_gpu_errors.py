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
