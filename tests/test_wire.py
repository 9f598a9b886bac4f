"""Wire formats: the big-endian blobs of the reference's marshaling_policy (common.hpp:168-203, 462-485, 749-799).
CPU part: the library's host-only encoders / decoders against an independent Python restatement (oracle/wire.py) and against the bytes
the reference itself holds (bin/cli/src/data.bin: the proof, and the head of an extended verification key whose GT element also pins
the test-side pairing's Fp12 arithmetic).  GPU part: the proving-key blob, parsed on the GPU, proves like the key it came from."""
import os

import numpy as np
import pytest

import bls12_381 as o
import pairing as pg
import wire as w
from conftest import GOLDEN, I, L, fr_array, g1_limbs, g2_limbs

import vote_saver_protocol_amd as v


def test_data_bin_gt_element_pins_the_pairing_field(cref):
    """data.bin[196:772): 12 little-endian Fp values in tower order.  Read into the pairing's Fp12 it is a genuine GT element:
    g != 1, g^r = 1 -- a reference-produced value under the oracle's own Fp12 multiplication; and the three points after it decode,
    lie on their curves and in the order-r subgroups."""
    head = bytes.fromhex(open(os.path.join(GOLDEN, "data_bin_vk_head.hex")).read().strip())
    assert len(head) == 820
    h4, g, gamma_g2, delta_g2, delta_g1 = w.parse_vk_head(head)
    assert h4 == 0
    assert g != pg.ONE and pg.f12_pow(g, o.R) == pg.ONE
    assert pg.f12_mul(g, pg.f12_inv(g)) == pg.ONE
    # a GT element is unitary: g^(p^6) = g^-1, i.e. conjugation (w -> -w) inverts it
    conj = [x if k % 2 == 0 else (-x) % o.P for k, x in enumerate(g)]
    assert pg.f12_mul(g, conj) == pg.ONE
    assert w.gt_to_tower_le(g) == head[4:580]
    for q in (gamma_g2, delta_g2):
        assert o.G2.is_on_curve(q) and o.G2.in_subgroup(q)
    assert o.G1.is_on_curve(delta_g1) and o.G1.in_subgroup(delta_g1)
    # the library decodes the same three points from the same bytes
    assert o.g2_from_limbs(v.g2_decompress(head[580:676])) == gamma_g2 and o.g2_from_limbs(v.g2_decompress(head[676:772])) == delta_g2
    assert o.g1_from_limbs(v.g1_decompress(head[772:820])) == delta_g1
    # ... and a verification-key blob built from them has data.bin's bytes as its head
    gabc = [o.G1.mul(o.G1.gen, k) for k in (3, 5)]; gamma_g1 = o.G1.mul(o.G1.gen, 7)
    blob = v.vk_to_blob(head[4:580], g2_limbs(gamma_g2), g2_limbs(delta_g2), g1_limbs(delta_g1), np.stack([g1_limbs(p) for p in gabc]), g1_limbs(gamma_g1), head=0)
    assert blob[:820] == head and blob == w.vk_blob(0, g, gamma_g2, delta_g2, delta_g1, gabc, gamma_g1)
    back = v.vk_from_blob(blob)
    assert back["head"] == 0 and back["alpha_g1_beta_g2"] == head[4:580] and o.g2_from_limbs(back["gamma_g2"]) == gamma_g2
    assert [o.g1_from_limbs(x) for x in back["gamma_ABC_g1"]] == gabc and o.g1_from_limbs(back["gamma_g1"]) == gamma_g1
    with pytest.raises(ValueError):
        v.vk_from_blob(blob[:-1])
    with pytest.raises(ValueError):
        v.vk_from_blob(blob + b"\x00")


def test_alpha_beta_pairing_round_trips_through_the_tower_encoding():
    """e(alpha, beta) of a key made here, written in the tower / little-endian layout and read back"""
    a, b = o.G1.mul(o.G1.gen, 1234567), o.G2.mul(o.G2.gen, 7654321)
    g = pg.final_exp(pg.miller_loop(b, a))
    enc = w.gt_to_tower_le(g)
    assert len(enc) == 576 and w.gt_from_tower_le(enc) == g


def test_scalar_vector_ciphertext_and_proof_blobs():
    gen = o.splitmix64(8)
    vals = [0, 1, o.R - 1] + [o.rand_fr(gen) for _ in range(30)]
    blob = v.fr_vector_to_blob(fr_array(vals))
    assert blob == w.fr_vector(vals) and len(blob) == 8 + 32 * len(vals)            # 8-byte count + 32-byte elements (notebook cell 0)
    assert [I(x) for x in v.fr_vector_from_blob(blob)] == vals == w.fr_vector_parse(blob)
    assert v.fr_vector_to_blob(np.zeros((0, 4), np.uint64)) == bytes(8) and v.fr_vector_from_blob(bytes(8)).shape == (0, 4)
    with pytest.raises(ValueError):
        v.fr_vector_to_blob(fr_array([o.R]))                                         # not canonical
    with pytest.raises(ValueError):
        v.fr_vector_from_blob(w.be(1, 8) + w.be(o.R, 32))
    with pytest.raises(ValueError):
        v.fr_vector_from_blob(blob[:-3])
    with pytest.raises(ValueError):
        v.fr_vector_from_blob(w.be(1 << 60, 8) + bytes(32))                         # a count larger than the blob
    pts = [o.G1.mul(o.G1.gen, o.rand_fr(gen)) for _ in range(5)] + [None]
    ctb = v.g1_vector_to_blob(np.stack([g1_limbs(p) for p in pts]))
    assert ctb == w.g1_vector(pts)
    assert [o.g1_from_limbs(x) for x in v.g1_vector_from_blob(ctb)] == pts
    bad = bytearray(ctb); bad[8] &= 0x7F
    with pytest.raises(ValueError):
        v.g1_vector_from_blob(bytes(bad))
    d = bytes.fromhex(open(os.path.join(GOLDEN, "data_bin_proof.hex")).read().strip())
    A, B, Cc = v.proof_from_blob(d)                                                 # the reference's own proof bytes
    assert o.g1_from_limbs(A) == o.g1_decompress(d[:48]) and o.g2_from_limbs(B) == o.g2_decompress(d[48:144]) and o.g1_from_limbs(Cc) == o.g1_decompress(d[144:])
    out = np.zeros(192, np.uint8)
    assert v.load().vsp_proof_to_blob(A.ctypes.data, B.ctypes.data, Cc.ctypes.data, out.ctypes.data) == 0 and out.tobytes() == d
    with pytest.raises(ValueError):
        v.proof_from_blob(d[:191])


@pytest.mark.gpu
def test_proving_key_blob_round_trip_on_the_gpu(ctx, cref):
    """deserialize_pk_crs (common.hpp:749-754, inside the reference's timed vote phase): a generated key is written as the "fast"
    proving-key blob -- byte for byte the independent Python encoding of the same key -- and read back ON THE GPU into a resident
    key (plain and with window multiples); proofs made with the loaded keys are the proofs of the original.  Malformed blobs are
    refused: truncated, a compressed-form record, a point off the curve, an inflated count."""
    nc, ni = 3000, 5
    gen = o.splitmix64(404)
    cs, wit = cref.R1CS.synth(nc, ni, 44)
    tox = fr_array([o.rand_fr(gen) for _ in range(5)])
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
    kp = v.Keypair(ctx, dcs, tox)
    r, s = L(o.rand_fr(gen), 4), L(o.rand_fr(gen), 4)
    ref = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
    blob = kp.to_blob()
    P1 = lambda name: [o.g1_from_limbs(x) for x in kp.part(name)]
    P2 = lambda name: [o.g2_from_limbs(x) for x in kp.part(name)]
    assert blob == w.pk_blob(P1("alpha_g1")[0], P1("beta_g1")[0], P2("beta_g2")[0], P1("delta_g1")[0], P2("delta_g2")[0],
                             P1("A_query"), P1("B_query_g1"), P2("B_query_g2"), P1("H_query"), P1("L_query"))
    assert any(p is None for p in P1("B_query_g1"))                                 # the key has points at infinity: the flag byte path
    for pre in (False, True):
        k2 = v.Keypair.from_blob(ctx, blob, precompute=pre)
        for name in ("A_query", "B_query_g1", "B_query_g2", "H_query", "L_query", "alpha_g1", "beta_g2", "delta_g1"):
            if not pre or name in ("alpha_g1", "beta_g2", "delta_g1"):
                assert np.array_equal(k2.part(name), kp.part(name)), name
        got = v.groth16_prove(ctx, dcs, k2.pk, wit, r, s)
        assert all(np.array_equal(a, b) for a, b in zip(got[:3], ref[:3])) and got[3] == ref[3]
        assert k2.part("gamma_ABC_g1").shape[0] == 0                                # not part of the proving key
        k2.free()
    for bad in (blob[:-5], blob + b"\x00"):
        with pytest.raises(v.VspError):
            v.Keypair.from_blob(ctx, bad)
    off_a = 672 + 8                                                                 # first A_query record
    b2 = bytearray(blob); b2[off_a] |= 0x80                                         # claims compressed form
    with pytest.raises(v.VspError, match="uncompressed"):
        v.Keypair.from_blob(ctx, bytes(b2))
    b2 = bytearray(blob); b2[off_a] |= 0x20                                         # the sign flag exists only in the compressed form
    with pytest.raises(v.VspError, match="uncompressed"):
        v.Keypair.from_blob(ctx, bytes(b2))
    inf_at = next(i for i, p in enumerate(P1("B_query_g1")) if p is None)           # an infinity record (G1 half of a B_query pair) with a payload
    off_b = 672 + 8 + 96 * len(P1("A_query")) + 8 + 288 * inf_at + 192
    assert blob[off_b] == 0x40
    b2 = bytearray(blob); b2[off_b + 50] = 1
    with pytest.raises(v.VspError, match="uncompressed"):
        v.Keypair.from_blob(ctx, bytes(b2))
    b3 = bytearray(blob); b3[off_a + 96 + 95] ^= 1                                  # y of the second point off by one
    with pytest.raises(v.VspError, match="curve"):
        v.Keypair.from_blob(ctx, bytes(b3))
    b4 = bytearray(blob); b4[672:680] = w.be(1 << 40, 8)
    with pytest.raises(v.VspError, match="count"):
        v.Keypair.from_blob(ctx, bytes(b4))
    b5 = bytearray(blob); b5[95] ^= 1                                               # alpha_g1 off the curve
    with pytest.raises(v.VspError):
        v.Keypair.from_blob(ctx, bytes(b5))
    kp.free(); dcs.free(); cs.free()
