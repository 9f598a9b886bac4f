"""TEST INFRASTRUCTURE ONLY -- big-integer restatement of the SAVER wrapper the reference puts around the prover
(SURVEY.md 8(f).3): elgamal_verifiable<bls12_381> of crypto3-pubkey (absent submodule, /root/reference/.gitmodules:50), i.e. the
scheme of Lee, Choi, Kim, Oh, "SAVER: SNARK-friendly, Additively-homomorphic, and Verifiable Encryption and decryption with
Rerandomization" (fig. 3), as the reference calls it:

    generate_keypair<elgamal_verifiable>(rnd[3*msg_size + 2], {gg_keypair, msg_size})        bin/cli/.../common.hpp:921-931
    encrypt<...>(m_field, {d(), pk_eid, gg_keypair, primary_input, auxiliary_input})         common.hpp:1131-1135   (this proves)
    rerandomize<...>(rnd[3], cipher_text.first, {pk_eid, gg_keypair, cipher_text.second})    common.hpp:1138-1145
    verify_encryption<...>(ct, {pk_eid, vk, proof, primary_input without the message})       common.hpp:1164-1169
    decrypt<...>(ct_agg, {sk_eid, vk_eid, gg_keypair}) / verify_decryption<...>(...)         common.hpp:1220-1223, 1282-1283

[UPSTREAM-KNOWLEDGE] member names (delta_s_g1, t_g1, t_g2, delta_sum_s_g1, gamma_inverse_sum_s_g1, rho_g2, rho_sv_g2, rho_rhov_g2)
and the order the random values are consumed in follow upstream as remembered and cannot be checked against its source here.
What IS checked: the scheme's own equations, with the independent pairing of oracle/pairing.py -- a ciphertext + proof made here
(or by the GPU library) passes verify_encryption, a rerandomized one passes too, tampered ones fail, decrypt returns the plaintext
and the tally of added ciphertexts, verify_decryption accepts exactly the true result.

Points are affine tuples of python ints (None = infinity) as in bls12_381.py; n = msg_size; G_i = gamma_ABC_g1[i], i = 1..n.
"""
from bls12_381 import G1, G2, R
import pairing as pg


def keygen(n, delta_g1, gamma_g1, gamma_abc, rnd):
    """rnd: 3n + 2 ints = s_1..s_n | v_1..v_n | t_0..t_n | rho.  -> (pk dict, sk = rho, vk dict)"""
    assert len(rnd) == 3 * n + 2 and len(gamma_abc) >= n + 1
    s, v, t, rho = rnd[:n], rnd[n:2 * n], rnd[2 * n:3 * n + 1], rnd[3 * n + 1]
    pk = dict(delta_g1=delta_g1,
              delta_s_g1=[G1.mul(delta_g1, s[i]) for i in range(n)],
              t_g1=[G1.mul(gamma_abc[i + 1], t[i + 1]) for i in range(n)],
              t_g2=[G2.mul(G2.gen, t[j]) for j in range(n + 1)],
              delta_sum_s_g1=G1.mul(delta_g1, (t[0] + sum(t[j + 1] * s[j] for j in range(n))) % R),
              gamma_inverse_sum_s_g1=G1.mul(gamma_g1, (-(1 + sum(s))) % R))
    vk = dict(rho_g2=G2.mul(G2.gen, rho),
              rho_sv_g2=[G2.mul(G2.gen, s[i] * v[i] % R) for i in range(n)],
              rho_rhov_g2=[G2.mul(G2.gen, rho * v[i] % R) for i in range(n)])
    return pk, rho, vk


def encrypt_ct(pk, gamma_abc, msg, r):
    """ciphertext c_0 | c_1..c_n | psi.  The proof is the Groth16 proof with C += r * gamma_inverse_sum_s_g1."""
    n = len(msg)
    ct = [G1.mul(pk["delta_g1"], r)]
    psi = G1.mul(pk["delta_sum_s_g1"], r)
    for i in range(n):
        ct.append(G1.add(G1.mul(pk["delta_s_g1"][i], r), G1.mul(gamma_abc[i + 1], msg[i] % R)))
        psi = G1.add(psi, G1.mul(pk["t_g1"][i], msg[i] % R))
    return ct + [psi]


def rerandomize(pk, delta_g2, rnd3, ct, proof):
    rp, z1, z2 = rnd3
    n = len(ct) - 2
    bases = [pk["delta_g1"]] + pk["delta_s_g1"] + [pk["delta_sum_s_g1"]]
    ct2 = [G1.add(c, G1.mul(b, rp)) for c, b in zip(ct, bases)]
    A, B, Cc = proof
    A2 = G1.mul(A, z1)
    B2 = G2.add(G2.mul(B, pow(z1, -1, R)), G2.mul(delta_g2, z2))
    C2 = G1.add(G1.add(Cc, G1.mul(A, z1 * z2 % R)), G1.mul(pk["gamma_inverse_sum_s_g1"], rp))
    assert len(ct2) == n + 2
    return ct2, (A2, B2, C2)


def verify_encryption(pk, gg_vk, ct, proof, pinput_rest):
    """gg_vk: dict(alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_ABC_g1).  pinput_rest: the primary input after the n message slots.
    (1) the ciphertext is well formed:  e(c_0, t_g2[0]) * prod_j e(c_j, t_g2[j]) = e(psi, H)
    (2) Groth16 with the ciphertext standing in for the message inputs:
        e(A, B) = e(alpha, beta) * e(gamma_ABC[0] + c_0 + .. + c_n + sum_k x_k gamma_ABC[n+1+k], gamma_g2) * e(C, delta_g2)"""
    n = len(ct) - 2
    A, B, Cc = proof
    if not (G1.is_on_curve(A) and G2.is_on_curve(B) and G1.is_on_curve(Cc) and all(G1.is_on_curve(c) for c in ct)):
        return False
    pairs = [(ct[j], pk["t_g2"][j]) for j in range(n + 1)] + [(G1.neg(ct[n + 1]), G2.gen)]
    if not pg.pairing_product_is_one(pairs):
        return False
    acc = gg_vk["gamma_ABC_g1"][0]
    for c in ct[:n + 1]:
        acc = G1.add(acc, c)
    for x, pt in zip(pinput_rest, gg_vk["gamma_ABC_g1"][n + 1:]):
        acc = G1.add(acc, G1.mul(pt, x % R))
    return pg.pairing_product_is_one([(G1.neg(A), B), (gg_vk["alpha_g1"], gg_vk["beta_g2"]), (acc, gg_vk["gamma_g2"]), (Cc, gg_vk["delta_g2"])])


def decrypt(rho, vk, gamma_abc, ct, max_value=64):
    """-> (messages, decryption proof nu = rho * c_0).  m_i is the discrete log of e(c_i, V_{n+i}) / e(nu, V_i) to the base
    e(G_i, V_{n+i}), searched up to max_value (a tally of added ballots is at most the number of voters)."""
    n = len(ct) - 2
    nu = G1.mul(ct[0], rho)
    out = []
    for i in range(n):
        lhs = pg.final_exp(pg.f12_mul(pg.miller_loop(vk["rho_rhov_g2"][i], ct[i + 1]), pg.miller_loop(vk["rho_sv_g2"][i], G1.neg(nu))))
        base = pg.final_exp(pg.miller_loop(vk["rho_rhov_g2"][i], gamma_abc[i + 1]))
        acc, m = pg.ONE, None
        for k in range(max_value + 1):
            if acc == lhs:
                m = k
                break
            acc = pg.f12_mul(acc, base)
        if m is None:
            raise ValueError("decrypt: message %d outside [0, %d]" % (i, max_value))
        out.append(m)
    return out, nu


def verify_decryption(vk, gamma_abc, ct, msgs, nu):
    """e(nu, H) = e(c_0, rho_g2)  and, for every i,  e(c_i, V_{n+i}) = e(nu, V_i) * e(m_i G_i, V_{n+i})"""
    n = len(ct) - 2
    if not pg.pairing_product_is_one([(nu, G2.gen), (G1.neg(ct[0]), vk["rho_g2"])]):
        return False
    for i in range(n):
        mg = G1.mul(gamma_abc[i + 1], msgs[i] % R)
        if not pg.pairing_product_is_one([(ct[i + 1], vk["rho_rhov_g2"][i]), (G1.neg(nu), vk["rho_sv_g2"][i]), (G1.neg(mg), vk["rho_rhov_g2"][i])]):
            return False
    return True


def add_ciphertexts(cts):
    """the tally: component-wise sum (common.hpp:1208-1216)"""
    agg = list(cts[0])
    for ct in cts[1:]:
        agg = [G1.add(a, b) for a, b in zip(agg, ct)]
    return agg


# ---- flat limb layouts shared with the C restatement (oracle/vsp_ref.c ref_saver_*) and the library (include/vsp.h vsp_saver_*)
def pk_to_words(pk):
    import numpy as np
    from bls12_381 import g1_to_limbs, g2_to_limbs
    w = list(g1_to_limbs(pk["delta_g1"]))
    for p in pk["delta_s_g1"] + pk["t_g1"]:
        w += g1_to_limbs(p)
    for q in pk["t_g2"]:
        w += g2_to_limbs(q)
    w += g1_to_limbs(pk["delta_sum_s_g1"]) + g1_to_limbs(pk["gamma_inverse_sum_s_g1"])
    return np.array(w, dtype=np.uint64)


def pk_from_words(w, n):
    from bls12_381 import g1_from_limbs, g2_from_limbs
    w = [int(x) for x in w]
    g1 = lambda off: g1_from_limbs(w[off:off + 12])
    g2 = lambda off: g2_from_limbs(w[off:off + 24])
    o_ds, o_t1, o_t2 = 12, 12 + 12 * n, 12 + 24 * n
    o_sum = o_t2 + 24 * (n + 1)
    return dict(delta_g1=g1(0), delta_s_g1=[g1(o_ds + 12 * i) for i in range(n)], t_g1=[g1(o_t1 + 12 * i) for i in range(n)],
                t_g2=[g2(o_t2 + 24 * j) for j in range(n + 1)], delta_sum_s_g1=g1(o_sum), gamma_inverse_sum_s_g1=g1(o_sum + 12))


def vk_to_words(vk):
    import numpy as np
    from bls12_381 import g2_to_limbs
    w = list(g2_to_limbs(vk["rho_g2"]))
    for q in vk["rho_sv_g2"] + vk["rho_rhov_g2"]:
        w += g2_to_limbs(q)
    return np.array(w, dtype=np.uint64)


def vk_from_words(w, n):
    from bls12_381 import g2_from_limbs
    w = [int(x) for x in w]
    g2 = lambda off: g2_from_limbs(w[off:off + 24])
    return dict(rho_g2=g2(0), rho_sv_g2=[g2(24 + 24 * i) for i in range(n)], rho_rhov_g2=[g2(24 + 24 * n + 24 * i) for i in range(n)])
