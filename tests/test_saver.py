"""SAVER wrapper: elgamal_verifiable<bls12_381> around the prover, as the reference's vote phase calls it
(bin/cli/include/nil/vote_saver/common.hpp:921-931 generate_keypair, :1131-1135 encrypt, :1138-1145 rerandomize, :1164-1169
verify_encryption, :1220-1223 decrypt, :1282-1283 verify_decryption).

CPU part (no GPU): the two restatements -- oracle/saver.py (big integers) and oracle/vsp_ref.c ref_saver_* (C) -- agree bit for bit
with each other and with the library's host-only entry points (vsp_saver_keygen, vsp_saver_rerandomize); the scheme's equations hold
under the independent pairing: a ballot verifies, a rerandomized ballot verifies, tampered ones do not, the tally of added
ciphertexts decrypts to the sum of the votes and verify_decryption accepts exactly that.
GPU part: vsp_saver_encrypt (ciphertext on the host while the GPU proves) against the oracles, at the reference's msg_size = 25."""
import ctypes as C

import numpy as np
import pytest

import bls12_381 as o
import saver as sv
from conftest import I, L, fr_array

import vote_saver_protocol_amd as v


def _setup(cref, n, nc, ni, seed, ballot=None):
    gen = o.splitmix64(seed)
    cs, wit = cref.R1CS.synth(nc, ni, seed, ballot=ballot)
    tox = fr_array([o.rand_fr(gen) for _ in range(5)])
    kp = cref.Keypair(cs, tox)
    rnd = fr_array([o.rand_fr(gen) for _ in range(3 * n + 2)])
    parts = {k: kp.part(k) for k in ("gamma_ABC_g1", "delta_g1", "gamma_g1", "alpha_g1", "beta_g2", "gamma_g2", "delta_g2")}
    return gen, cs, wit, kp, rnd, parts


def _gg_vk(parts):
    return dict(alpha_g1=o.g1_from_limbs(parts["alpha_g1"][0]), beta_g2=o.g2_from_limbs(parts["beta_g2"][0]),
                gamma_g2=o.g2_from_limbs(parts["gamma_g2"][0]), delta_g2=o.g2_from_limbs(parts["delta_g2"][0]),
                gamma_ABC_g1=[o.g1_from_limbs(x) for x in parts["gamma_ABC_g1"]])


def _ct_points(ct):
    return [o.g1_from_limbs(row) for row in np.asarray(ct).reshape(-1, 12)]


def test_restatements_and_host_library_agree(cref):
    n = 3
    gen, cs, wit, kp, rnd, parts = _setup(cref, n, 40, 6, seed=31)
    gabc = parts["gamma_ABC_g1"]
    pk_c, sk_c, vk_c = cref.saver_keygen(n, parts["delta_g1"][0], parts["gamma_g1"][0], gabc[:n + 1], rnd)
    pk_p, sk_p, vk_p = sv.keygen(n, o.g1_from_limbs(parts["delta_g1"][0]), o.g1_from_limbs(parts["gamma_g1"][0]),
                                 [o.g1_from_limbs(x) for x in gabc], [I(x) for x in rnd])
    assert np.array_equal(pk_c, sv.pk_to_words(pk_p)) and np.array_equal(vk_c, sv.vk_to_words(vk_p)) and I(sk_c) == sk_p
    assert sv.pk_from_words(pk_c, n) == pk_p and sv.vk_from_words(vk_c, n) == vk_p
    # the library's host-only key generation (no context needed)
    lib = v.load()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    pk_l = np.zeros(lib.vsp_saver_pk_words(n), np.uint64); sk_l = np.zeros(4, np.uint64); vk_l = np.zeros(lib.vsp_saver_vk_words(n), np.uint64)
    gabc_n = np.ascontiguousarray(gabc[:n + 1])
    assert lib.vsp_saver_keygen(None, n, p(parts["delta_g1"][0].copy()), p(parts["gamma_g1"][0].copy()), p(gabc_n), p(rnd), p(pk_l), p(sk_l), p(vk_l)) == 0
    assert np.array_equal(pk_l, pk_c) and np.array_equal(vk_l, vk_c) and np.array_equal(sk_l, sk_c)
    assert pk_l.shape[0] == cref.saver_pk_words(n) and vk_l.shape[0] == cref.saver_vk_words(n)
    # ciphertext: C vs big integers, message blocks 0, 1 and a larger value
    msg = fr_array([1, 0, 5]); r_enc = L(o.rand_fr(gen), 4)
    ct_c = cref.saver_encrypt_ct(n, pk_c, gabc[:n + 1], msg, r_enc)
    ct_p = sv.encrypt_ct(pk_p, [o.g1_from_limbs(x) for x in gabc], [1, 0, 5], I(r_enc))
    assert _ct_points(ct_c) == ct_p
    # rerandomize: C vs big integers vs the library's host entry point
    A = cref.g1_mul(np.array(o.g1_to_limbs(o.G1.gen), np.uint64), L(o.rand_fr(gen), 4))
    B = cref.g2_mul(np.array(o.g2_to_limbs(o.G2.gen), np.uint64), L(o.rand_fr(gen), 4))
    Cc = cref.g1_mul(np.array(o.g1_to_limbs(o.G1.gen), np.uint64), L(o.rand_fr(gen), 4))
    rnd3 = fr_array([o.rand_fr(gen) for _ in range(3)])
    ct2_c, A2, B2, C2 = cref.saver_rerandomize(n, pk_c, parts["delta_g2"][0], rnd3, ct_c, A, B, Cc)
    ct2_p, (pA, pB, pC) = sv.rerandomize(pk_p, o.g2_from_limbs(parts["delta_g2"][0]), [I(x) for x in rnd3], ct_p,
                                         (o.g1_from_limbs(A), o.g2_from_limbs(B), o.g1_from_limbs(Cc)))
    assert _ct_points(ct2_c) == ct2_p and o.g1_from_limbs(A2) == pA and o.g2_from_limbs(B2) == pB and o.g1_from_limbs(C2) == pC
    spk = lib.vsp_saver_pk_load(None, n, p(pk_c), p(gabc_n))
    assert spk and lib.vsp_saver_pk_msg_size(spk) == n
    ct_l, A_l, B_l, C_l = ct_c.copy(), A.copy(), B.copy(), Cc.copy(); proof = np.zeros(192, np.uint8)
    assert lib.vsp_saver_rerandomize(None, spk, p(parts["delta_g2"][0].copy()), p(rnd3), p(ct_l), p(A_l), p(B_l), p(C_l), p(proof)) == 0
    assert np.array_equal(ct_l, ct2_c) and np.array_equal(A_l, A2) and np.array_equal(B_l, B2) and np.array_equal(C_l, C2)
    assert proof.tobytes() == o.g1_compress(pA) + o.g2_compress(pB) + o.g1_compress(pC)
    # refusals: z1 = 0, a non-canonical value, a ciphertext element off the curve
    d2 = parts["delta_g2"][0].copy()
    bad = rnd3.copy(); bad[1] = 0
    assert lib.vsp_saver_rerandomize(None, spk, p(d2), p(bad), p(ct_l), p(A_l), p(B_l), p(C_l), None) == -1
    bad = rnd3.copy(); bad[0] = L(o.R, 4)
    assert lib.vsp_saver_rerandomize(None, spk, p(d2), p(bad), p(ct_l), p(A_l), p(B_l), p(C_l), None) == -1
    off = ct_l.copy(); off[1, 6:] = ct_l[2, 6:]
    assert lib.vsp_saver_rerandomize(None, spk, p(d2), p(rnd3), p(off), p(A_l), p(B_l), p(C_l), None) == -1
    lib.vsp_saver_pk_free(None, spk)
    badpk = pk_c.copy(); badpk[12 + 6:12 + 12] = pk_c[24 + 6:24 + 12]
    assert not lib.vsp_saver_pk_load(None, n, p(badpk), p(gabc_n))
    kp.free(); cs.free()


def test_ballot_verifies_rerandomizes_tallies_and_decrypts(cref):
    """the protocol round trip on the CPU restatement with the pairing as judge: two ballots, msg_size 3"""
    n, ni = 3, 6
    gen, cs, wit, kp, rnd, parts = _setup(cref, n, 40, ni, seed=77)
    gabc_l = parts["gamma_ABC_g1"]; gabc = [o.g1_from_limbs(x) for x in gabc_l]
    gg_vk = _gg_vk(parts)
    pk, rho, vk = sv.keygen(n, o.g1_from_limbs(parts["delta_g1"][0]), o.g1_from_limbs(parts["gamma_g1"][0]), gabc, [I(x) for x in rnd])
    pk_words = sv.pk_to_words(pk)
    P2 = np.array(o.g1_to_limbs(pk["gamma_inverse_sum_s_g1"]), np.uint64)
    # the synthetic witness fixes the public inputs: its first n entries are "the message" of this ballot
    msg = [I(wit[i]) for i in range(n)]
    rest = [I(wit[i]) for i in range(n, ni)]
    r_enc, r, s = (L(o.rand_fr(gen), 4) for _ in range(3))
    A, B, Cc = kp.prove(wit, r, s, P1=P2, r_enc=r_enc)
    ct = sv.encrypt_ct(pk, gabc, msg, I(r_enc))
    proof = (o.g1_from_limbs(A), o.g2_from_limbs(B), o.g1_from_limbs(Cc))
    assert sv.verify_encryption(pk, gg_vk, ct, proof, rest)
    # a plain Groth16 proof (no SAVER addend) does not verify against the ciphertext, nor does a ciphertext under other randomness
    A0, B0, C0 = kp.prove(wit, r, s)
    assert not sv.verify_encryption(pk, gg_vk, ct, (o.g1_from_limbs(A0), o.g2_from_limbs(B0), o.g1_from_limbs(C0)), rest)
    assert not sv.verify_encryption(pk, gg_vk, sv.encrypt_ct(pk, gabc, msg, I(r_enc) + 1), proof, rest)
    swapped = list(ct); swapped[1], swapped[2] = swapped[2], swapped[1]
    assert not sv.verify_encryption(pk, gg_vk, swapped, proof, rest)
    # rerandomize (C restatement, checked against the big-integer one above): still verifies, every element changed
    rnd3 = fr_array([o.rand_fr(gen) for _ in range(3)])
    ct2_l, A2, B2, C2 = cref.saver_rerandomize(n, pk_words, parts["delta_g2"][0], rnd3, np.array([o.g1_to_limbs(c) for c in ct], np.uint64), A, B, Cc)
    ct2 = _ct_points(ct2_l); proof2 = (o.g1_from_limbs(A2), o.g2_from_limbs(B2), o.g1_from_limbs(C2))
    assert sv.verify_encryption(pk, gg_vk, ct2, proof2, rest)
    assert all(a != b for a, b in zip(ct, ct2)) and proof2[0] != proof[0] and proof2[1] != proof[1] and proof2[2] != proof[2]
    # tally: the sum of two small ballots decrypts to the sum of the votes; the decryption proof verifies for that result only
    # (the synthetic witness' public inputs are full-size field elements, far outside any searchable range, so the tally is made of
    # two ballots encrypted here; their well-formedness is the first equation of verify_encryption, checked on its own)
    msg_a, msg_b = [1, 0, 0], [2, 0, 1]
    ct_a, ct_b = sv.encrypt_ct(pk, gabc, msg_a, o.rand_fr(gen)), sv.encrypt_ct(pk, gabc, msg_b, o.rand_fr(gen))
    import pairing as pg
    for c in (ct_a, ct_b):
        assert pg.pairing_product_is_one([(c[j], pk["t_g2"][j]) for j in range(n + 1)] + [(o.G1.neg(c[n + 1]), o.G2.gen)])
    agg = sv.add_ciphertexts([ct_a, ct_b])
    got, nu = sv.decrypt(rho, vk, gabc, agg, max_value=16)
    assert got == [3, 0, 1]
    assert sv.verify_decryption(vk, gabc, agg, got, nu)
    assert not sv.verify_decryption(vk, gabc, agg, [3, 1, 1], nu)
    assert not sv.verify_decryption(vk, gabc, agg, got, o.G1.add(nu, o.G1.gen))
    kp.free(); cs.free()


@pytest.mark.gpu
def test_gpu_encrypt_and_rerandomize_msg_size_25(ctx, cref):
    """vsp_saver_encrypt at the reference's msg_size = 25 (common.hpp:163): ciphertext and proof bit for bit against the C
    restatement (ciphertext) and the oracle's prover with the SAVER addend (proof); rerandomized by the library; both verify under
    the pairing.  The message is what the synthetic witness carries in its first 25 public inputs (full-size field elements: the
    generic m_i * G_i path; a one-hot ballot takes the 0 / 1 shortcuts, covered by the CPU test above)."""
    n, ni, nc = 25, 30, 900
    gen, cs, wit, kp, rnd, parts = _setup(cref, n, nc, ni, seed=2025)
    msg_full = wit[:n].copy()
    gabc_l = np.ascontiguousarray(parts["gamma_ABC_g1"])
    pk_w, sk, vk_w = v.saver_generate_keypair(ctx, rnd, gabc_l, parts["delta_g1"][0], parts["gamma_g1"][0], n)
    pk_c, sk_c, vk_c = cref.saver_keygen(n, parts["delta_g1"][0], parts["gamma_g1"][0], gabc_l[:n + 1], rnd)
    assert np.array_equal(pk_w, pk_c) and np.array_equal(vk_w, vk_c) and np.array_equal(sk, sk_c)
    spk = v.SaverPublicKey(ctx, pk_w, gabc_l[:n + 1], n)
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
    q = [ctx.upload_bases(kp.part(nm), g) for nm, g in (("A_query", 1), ("B_query_g1", 1), ("B_query_g2", 2), ("H_query", 1), ("L_query", 1))]
    pk = v.ProvingKey(ctx, kp.part("alpha_g1")[0], kp.part("beta_g1")[0], kp.part("beta_g2")[0], kp.part("delta_g1")[0], kp.part("delta_g2")[0], *q)
    r_enc, r, s = (L(o.rand_fr(gen), 4) for _ in range(3))
    ct, (A, B, Cc), proof = v.saver_encrypt(ctx, spk, dcs, pk, msg_full, wit, r_enc, r, s)
    P2 = pk_c[-12:]
    eA, eB, eC = kp.prove(wit, r, s, P1=P2, r_enc=r_enc)
    assert np.array_equal(A, eA) and np.array_equal(B, eB) and np.array_equal(Cc, eC)
    assert np.array_equal(ct, cref.saver_encrypt_ct(n, pk_c, gabc_l[:n + 1], msg_full, r_enc))
    assert proof == o.g1_compress(o.g1_from_limbs(eA)) + o.g2_compress(o.g2_from_limbs(eB)) + o.g1_compress(o.g1_from_limbs(eC))
    gg_vk = _gg_vk(parts); pkd = sv.pk_from_words(pk_w, n)
    rest = [I(wit[i]) for i in range(n, ni)]
    assert sv.verify_encryption(pkd, gg_vk, _ct_points(ct), (o.g1_from_limbs(A), o.g2_from_limbs(B), o.g1_from_limbs(Cc)), rest)
    rnd3 = fr_array([o.rand_fr(gen) for _ in range(3)])
    ct2, (A2, B2, C2), proof2 = v.saver_rerandomize(ctx, spk, parts["delta_g2"][0], rnd3, ct, (A, B, Cc))
    rc = cref.saver_rerandomize(n, pk_c, parts["delta_g2"][0], rnd3, ct, A, B, Cc)
    assert np.array_equal(ct2, rc[0]) and np.array_equal(A2, rc[1]) and np.array_equal(B2, rc[2]) and np.array_equal(C2, rc[3])
    assert sv.verify_encryption(pkd, gg_vk, _ct_points(ct2), (o.g1_from_limbs(A2), o.g2_from_limbs(B2), o.g1_from_limbs(C2)), rest)
    assert np.array_equal(v.g1_decompress(proof2[:48]), A2) and np.array_equal(v.g2_decompress(proof2[48:144]), B2)
    # refusals: a message that is not the witness' public input, non-canonical randomness
    bad = msg_full.copy(); bad[3] = L(I(bad[3]) ^ 1, 4)
    with pytest.raises(v.VspError, match="message"):
        v.saver_encrypt(ctx, spk, dcs, pk, bad, wit, r_enc, r, s)
    with pytest.raises(v.VspError, match="canonical"):
        v.saver_encrypt(ctx, spk, dcs, pk, msg_full, wit, L(o.R, 4), r, s)
    # the prover is unaffected by the hook: a plain proof right after is the oracle's plain proof
    pa, pb, pc, _ = v.groth16_prove(ctx, dcs, pk, wit, r, s)
    ea, eb, ec = kp.prove(wit, r, s)
    assert np.array_equal(pa, ea) and np.array_equal(pb, eb) and np.array_equal(pc, ec)
    spk.free(); pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()


@pytest.mark.gpu
def test_gpu_one_hot_ballot_tally_of_three_voters(ctx, cref):
    """the protocol end to end at msg_size = 25 with real ballots: three voters vote (one-hot messages in the first 25 public inputs,
    common.hpp:1029-1040), each ballot = vsp_saver_encrypt + vsp_saver_rerandomize and verifies; the tally of the three ciphertexts
    (common.hpp:1208-1216) decrypts to the vote counts and the decryption proof verifies."""
    n, ni, nc = 25, 30, 600
    votes = (7, 3, 7)
    gen = o.splitmix64(99)
    tox = fr_array([o.rand_fr(gen) for _ in range(5)])
    rnd = fr_array([o.rand_fr(gen) for _ in range(3 * n + 2)])
    cts = []
    spk = pkd = vkd = rho = gabc = None
    for voter, vote in enumerate(votes):
        cs, wit = cref.R1CS.synth(nc, ni, 5, ballot=(n, vote))                 # the same circuit, another ballot
        assert [I(x) for x in wit[:n]] == [1 if i == vote else 0 for i in range(n)]
        kp = cref.Keypair(cs, tox)
        parts = {k: kp.part(k) for k in ("gamma_ABC_g1", "delta_g1", "gamma_g1", "alpha_g1", "beta_g2", "gamma_g2", "delta_g2")}
        gabc_l = np.ascontiguousarray(parts["gamma_ABC_g1"])
        if spk is None:
            pk_w, sk, vk_w = v.saver_generate_keypair(ctx, rnd, gabc_l, parts["delta_g1"][0], parts["gamma_g1"][0], n)
            spk = v.SaverPublicKey(ctx, pk_w, gabc_l[:n + 1], n)
            pkd, vkd, rho = sv.pk_from_words(pk_w, n), sv.vk_from_words(vk_w, n), I(sk)
            gabc = [o.g1_from_limbs(x) for x in gabc_l]
        dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
        q = [ctx.upload_bases(kp.part(nm), g) for nm, g in (("A_query", 1), ("B_query_g1", 1), ("B_query_g2", 2), ("H_query", 1), ("L_query", 1))]
        pk = v.ProvingKey(ctx, kp.part("alpha_g1")[0], kp.part("beta_g1")[0], kp.part("beta_g2")[0], kp.part("delta_g1")[0], kp.part("delta_g2")[0], *q)
        r_enc, r, s = (L(o.rand_fr(gen), 4) for _ in range(3))
        ct, abc, _ = v.saver_encrypt(ctx, spk, dcs, pk, wit[:n], wit, r_enc, r, s)
        ct, abc, _ = v.saver_rerandomize(ctx, spk, parts["delta_g2"][0], fr_array([o.rand_fr(gen) for _ in range(3)]), ct, abc)
        if voter == 0:                                                          # the pairing check is slow in pure Python: one ballot
            assert sv.verify_encryption(pkd, _gg_vk(parts), _ct_points(ct), (o.g1_from_limbs(abc[0]), o.g2_from_limbs(abc[1]), o.g1_from_limbs(abc[2])),
                                        [I(wit[i]) for i in range(n, ni)])
        # the ciphertext travels as a blob (common.hpp:471-474) and comes back intact
        assert np.array_equal(v.g1_vector_from_blob(v.g1_vector_to_blob(ct)), ct)
        cts.append(_ct_points(ct))
        pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()
    agg = sv.add_ciphertexts(cts)
    # decrypt three positions (each costs pairings in pure Python): the two that received votes and an empty one
    sub = [0, 3, 7]
    pick = lambda lst: [lst[i] for i in sub]
    vk3 = dict(rho_g2=vkd["rho_g2"], rho_sv_g2=pick(vkd["rho_sv_g2"]), rho_rhov_g2=pick(vkd["rho_rhov_g2"]))
    agg3 = [agg[0]] + [agg[i + 1] for i in sub] + [agg[n + 1]]
    gabc3 = [gabc[0]] + [gabc[i + 1] for i in sub]
    got, nu = sv.decrypt(rho, vk3, gabc3, agg3, max_value=8)
    assert got == [0, 1, 2]
    assert sv.verify_decryption(vk3, gabc3, agg3, got, nu) and not sv.verify_decryption(vk3, gabc3, agg3, [0, 2, 1], nu)
    spk.free()
