"""TEST INFRASTRUCTURE ONLY -- independent restatement of the blob layouts of vote_saver_protocol_amd/csrc/wire.hip, in
plain Python over bytes and big integers, so the library's encoders / decoders are checked against a second implementation, and the
pieces of the reference's bin/cli/src/data.bin that pin the formats are decoded here (fixture tests/golden/data_bin_vk_head.hex).

What the reference pins (marshalling sources are absent submodules): the 192 proof bytes; the head of the extended verification key
-- 4 bytes, a GT element as 12 little-endian Fp in tower order, two compressed G2 and one compressed G1; 8-byte counts and 32-byte
scalars (protocol_exec.ipynb, read as text).  Everything else is the guess documented in wire.hip."""
from bls12_381 import P, g1_compress, g1_decompress, g2_compress, g2_decompress


def be(v, n):
    return int(v).to_bytes(n, "big")


def fr_vector(vals):
    return be(len(vals), 8) + b"".join(be(v, 32) for v in vals)


def fr_vector_parse(blob):
    n = int.from_bytes(blob[:8], "big")
    assert len(blob) == 8 + 32 * n
    return [int.from_bytes(blob[8 + 32 * i:40 + 32 * i], "big") for i in range(n)]


def g1_vector(pts):
    return be(len(pts), 8) + b"".join(g1_compress(p) for p in pts)


def g1_uncompressed(pt):
    if pt is None:
        return bytes([0x40]) + bytes(95)
    return be(pt[0], 48) + be(pt[1], 48)


def g2_uncompressed(pt):
    if pt is None:
        return bytes([0x40]) + bytes(191)
    (x0, x1), (y0, y1) = pt
    return be(x1, 48) + be(x0, 48) + be(y1, 48) + be(y0, 48)


def pk_blob(alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, A, B1, B2, H, Lq):
    out = g1_uncompressed(alpha_g1) + g1_uncompressed(beta_g1) + g2_uncompressed(beta_g2) + g1_uncompressed(delta_g1) + g2_uncompressed(delta_g2)
    out += be(len(A), 8) + b"".join(g1_uncompressed(p) for p in A)
    out += be(len(B2), 8) + b"".join(g2_uncompressed(q) + g1_uncompressed(p) for q, p in zip(B2, B1))
    out += be(len(H), 8) + b"".join(g1_uncompressed(p) for p in H)
    out += be(len(Lq), 8) + b"".join(g1_uncompressed(p) for p in Lq)
    return out


def gt_from_tower_le(b576):
    """576 bytes = 12 little-endian Fp, tower order index (k * 3 + j) * 2 + i for the coefficient of u^i v^j w^k (Fp2 = Fp[u]/(u^2+1),
    Fp6 = Fp2[v]/(v^3 - (1+u)), Fp12 = Fp6[w]/(w^2 - v))  ->  pairing.py's representation: polynomial in w modulo w^12 - 2 w^6 + 2
    (u = w^6 - 1, v = w^2)."""
    c = [int.from_bytes(b576[48 * t:48 * t + 48], "little") for t in range(12)]
    assert all(x < P for x in c)
    poly = [0] * 12
    for k in range(2):
        for j in range(3):
            a0, a1 = c[(k * 3 + j) * 2], c[(k * 3 + j) * 2 + 1]
            e = 2 * j + k                                   # v^j w^k = w^(2j + k), exponent 0..5
            poly[e] = (poly[e] + a0 - a1) % P               # a0 + a1 u = (a0 - a1) + a1 w^6
            poly[e + 6] = (poly[e + 6] + a1) % P
    return poly


def gt_to_tower_le(poly):
    out = b""
    for k in range(2):
        for j in range(3):
            e = 2 * j + k
            a1 = poly[e + 6] % P
            a0 = (poly[e] + a1) % P
            out += a0.to_bytes(48, "little") + a1.to_bytes(48, "little")
    return out


def vk_blob(head, gt_poly, gamma_g2, delta_g2, delta_g1, gamma_abc, gamma_g1):
    return (be(head, 4) + gt_to_tower_le(gt_poly) + g2_compress(gamma_g2) + g2_compress(delta_g2) + g1_compress(delta_g1) +
            be(len(gamma_abc), 8) + b"".join(g1_compress(p) for p in gamma_abc) + g1_compress(gamma_g1))


def parse_vk_head(b):
    """the part data.bin holds: head (4) | GT (576) | G2 | G2 | G1  ->  (head, gt polynomial, gamma_g2, delta_g2, delta_g1)"""
    return (int.from_bytes(b[:4], "big"), gt_from_tower_le(b[4:580]), g2_decompress(b[580:676]), g2_decompress(b[676:772]), g1_decompress(b[772:820]))
