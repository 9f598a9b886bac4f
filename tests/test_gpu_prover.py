"""GPU parity: vsp_groth16_prove (HIP) vs the C oracle's r1cs_gg_ppzksnark prover on the same R1CS, proving
key, witness and (r, s) -- identical A, B, C and identical 192 proof bytes -- and the pairing equation."""
import numpy as np
import pytest

import bls12_381 as o
from conftest import I, L, fr_array

import vote_saver_protocol_amd as v

pytestmark = pytest.mark.gpu


def build(ctx, cref, nc, ni, seed, precompute=False):
    gen = o.splitmix64(seed)
    cs, wit = cref.R1CS.synth(nc, ni, seed)
    tox = fr_array([o.rand_fr(gen) for _ in range(5)])
    kp = cref.Keypair(cs, tox)
    A, B, Cm = cs.export()
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, A, B, Cm)
    q = [ctx.upload_bases(kp.part(n), g) for n, g in (("A_query", 1), ("B_query_g1", 1), ("B_query_g2", 2), ("H_query", 1), ("L_query", 1))]
    if precompute:
        for i, x in enumerate(q):
            if precompute == "all" or i in (0, 2, 3):       # "mixed": only some queries precomputed (no shared plan then)
                x.precompute(16 if precompute == "all" else 12)
    pk = v.ProvingKey(ctx, kp.part("alpha_g1")[0], kp.part("beta_g1")[0], kp.part("beta_g2")[0], kp.part("delta_g1")[0], kp.part("delta_g2")[0], *q)
    r, s = L(o.rand_fr(gen), 4), L(o.rand_fr(gen), 4)
    return cs, wit, kp, dcs, pk, q, r, s


@pytest.mark.parametrize("nc,ni", [(10, 2), (300, 5), (2000, 30)])
def test_prove_bit_exact_vs_oracle(ctx, cref, nc, ni):
    cs, wit, kp, dcs, pk, q, r, s = build(ctx, cref, nc, ni, seed=nc)
    A, B, Cc, proof = v.groth16_prove(ctx, dcs, pk, wit, r, s)
    eA, eB, eC = kp.prove(wit, r, s)
    assert np.array_equal(A, eA) and np.array_equal(B, eB) and np.array_equal(Cc, eC)
    exp_bytes = o.g1_compress(o.g1_from_limbs(eA)) + o.g2_compress(o.g2_from_limbs(eB)) + o.g1_compress(o.g1_from_limbs(eC))
    assert proof == exp_bytes and len(proof) == 192
    # SAVER term r_enc * P1 on C
    P1 = kp.part("A_query")[3]; renc = L(987654321, 4)
    A2, B2, C2, _ = v.groth16_prove(ctx, dcs, pk, wit, r, s, saver_P1=P1, saver_r_enc=renc)
    sA, sB, sC = kp.prove(wit, r, s, P1=P1, r_enc=renc)
    assert np.array_equal(A2, sA) and np.array_equal(B2, sB) and np.array_equal(C2, sC)
    pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()


@pytest.mark.parametrize("option,value", [("prove_witness_streams", 0), ("msm_dimbits", 0), ("msm_dimbits", 1), ("prove_early_assembly", 0),
                                          ("prove_plan_first", 1), ("prove_h_first", 0), ("msm_slot_normal_priority", 1), ("msm_glv", 0)])
def test_prove_is_the_same_proof_under_every_scheduling_option(cref, option, value):
    """the documented tuning options (include/vsp.h) change how the work is queued or which kernel variant folds the buckets, never the
    result: the same proof bytes as the oracle's, on a context of its own so that stream-creation options take effect"""
    c = v.Context(0)
    try:
        c.set_option(option, value)
        for precompute in (None, "all"):
            cs, wit, kp, dcs, pk, q, r, s = build(c, cref, 1500, 7, seed=31, precompute=precompute)
            for _ in range(2):                                  # twice: the second call reuses warm workspaces and cached censuses
                A, B, Cc, proof = v.groth16_prove(c, dcs, pk, wit, r, s)
                eA, eB, eC = kp.prove(wit, r, s)
                assert np.array_equal(A, eA) and np.array_equal(B, eB) and np.array_equal(Cc, eC), (option, value, precompute)
            pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()
    finally:
        c.close()


@pytest.mark.parametrize("mode", ["all", "mixed"])
def test_prove_with_precomputed_key_bit_exact(ctx, cref, mode):
    cs, wit, kp, dcs, pk, q, r, s = build(ctx, cref, 700, 4, seed=11, precompute=mode)
    A, B, Cc, proof = v.groth16_prove(ctx, dcs, pk, wit, r, s)
    eA, eB, eC = kp.prove(wit, r, s)
    assert np.array_equal(A, eA) and np.array_equal(B, eB) and np.array_equal(Cc, eC)
    pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()


def test_gpu_proof_satisfies_pairing_equation(ctx, cref):
    import pairing as pg
    cs, wit, kp, dcs, pk, q, r, s = build(ctx, cref, 500, 6, seed=77)
    A, B, Cc, _ = v.groth16_prove(ctx, dcs, pk, wit, r, s)
    vk = dict(alpha_g1=o.g1_from_limbs(kp.part("alpha_g1")[0]), beta_g2=o.g2_from_limbs(kp.part("beta_g2")[0]),
              gamma_g2=o.g2_from_limbs(kp.part("gamma_g2")[0]), delta_g2=o.g2_from_limbs(kp.part("delta_g2")[0]),
              gamma_ABC_g1=[o.g1_from_limbs(x) for x in kp.part("gamma_ABC_g1")])
    pub = [I(wit[i]) for i in range(6)]
    assert pg.groth16_verify(vk, pub, (o.g1_from_limbs(A), o.g2_from_limbs(B), o.g1_from_limbs(Cc)))
    pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()


def test_generate_and_prove_2p16_step_domain_pairing(ctx, cref):
    """End to end at a size that takes the production paths (LDS counting sort, precomputed key with a shared bucket plan, the
    all-ones bucket split over hundreds of parts, a step radix-2 domain of 2^16 + 2^13 elements): key generated on the GPU, proof
    made on the GPU, Groth16 pairing equation checked by the oracle's pairing."""
    import pairing as pg
    ni = 6
    nc = (1 << 16) + (1 << 13) - ni - 40
    gen = o.splitmix64(123)
    cs, wit = cref.R1CS.synth(nc, ni, 21)
    tox = fr_array([o.rand_fr(gen) for _ in range(5)])
    A, B, Cm = cs.export()
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, A, B, Cm)
    assert dcs.m == (1 << 16) + (1 << 13) and dcs.domain_kind == "step_radix2"
    kp = v.Keypair(ctx, dcs, tox, precompute=True)
    r, s = L(o.rand_fr(gen), 4), L(o.rand_fr(gen), 4)
    pa, pb, pc, proof = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
    vk = dict(alpha_g1=o.g1_from_limbs(kp.part("alpha_g1")[0]), beta_g2=o.g2_from_limbs(kp.part("beta_g2")[0]),
              gamma_g2=o.g2_from_limbs(kp.part("gamma_g2")[0]), delta_g2=o.g2_from_limbs(kp.part("delta_g2")[0]),
              gamma_ABC_g1=[o.g1_from_limbs(x) for x in kp.part("gamma_ABC_g1")])
    pub = [I(wit[i]) for i in range(ni)]
    assert pg.groth16_verify(vk, pub, (o.g1_from_limbs(pa), o.g2_from_limbs(pb), o.g1_from_limbs(pc)))
    # the 192 proof bytes decode back to the same points
    assert np.array_equal(v.g1_decompress(proof[0:48]), pa) and np.array_equal(v.g2_decompress(proof[48:144]), pb)
    assert np.array_equal(v.g1_decompress(proof[144:192]), pc)
    # a second proof with other randomness verifies too and differs
    r2 = L(o.rand_fr(gen), 4)
    qa, qb, qc, _ = v.groth16_prove(ctx, dcs, kp.pk, wit, r2, s)
    assert not np.array_equal(qa, pa)
    assert pg.groth16_verify(vk, pub, (o.g1_from_limbs(qa), o.g2_from_limbs(qb), o.g1_from_limbs(qc)))
    kp.free(); dcs.free(); cs.free()


def test_prove_rejects_mismatched_key(ctx, cref):
    cs, wit, kp, dcs, pk, q, r, s = build(ctx, cref, 40, 2, seed=3)
    cs2, wit2 = cref.R1CS.synth(41, 2, 3)
    A, B, Cm = cs2.export()
    dcs2 = v.R1CS(ctx, 41, 2, cs2.num_vars, A, B, Cm)
    with pytest.raises(v.VspError):
        v.groth16_prove(ctx, dcs2, pk, wit2, r, s)
    with pytest.raises(ValueError):
        v.groth16_prove(ctx, dcs, pk, wit[:-1], r, s)
    dcs2.free(); cs2.free(); pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()


@pytest.mark.parametrize("nc,ni,precompute", [(7, 1, False), (300, 5, False), (3000, 30, True)])
def test_generator_bit_exact_vs_oracle(ctx, cref, nc, ni, precompute):
    """vsp_groth16_generate (zk::generate, common.hpp:916-917) builds the same key as the oracle's generator from the same
    toxic waste: every query, the verification-key elements, and a proof made with the generated key."""
    gen = o.splitmix64(100 + nc)
    cs, wit = cref.R1CS.synth(nc, ni, nc)
    tox = fr_array([o.rand_fr(gen) for _ in range(5)])
    ref = cref.Keypair(cs, tox)
    A, B, Cm = cs.export()
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, A, B, Cm)
    kp = v.Keypair(ctx, dcs, tox, precompute=precompute)
    for name in v.api.KEY_PARTS:
        assert np.array_equal(kp.part(name), ref.part(name)), name
    r, s = L(o.rand_fr(gen), 4), L(o.rand_fr(gen), 4)
    pa, pb, pc, _ = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
    ea, eb, ec = ref.prove(wit, r, s)
    assert np.array_equal(pa, ea) and np.array_equal(pb, eb) and np.array_equal(pc, ec)
    kp.free(); dcs.free(); ref.free(); cs.free()


def test_generate_and_prove_2p20_config4_pairing(ctx, cref):
    """BASELINE config 4 at full size: a synthetic SAVER-shaped R1CS filling the 2^20 domain (2^20 - 32 constraints, 30 public
    inputs, 90 % boolean wires), key generated on the GPU, proof made on the GPU -- once over the plain key and once
    over the key with window multiples -- identical proofs, Groth16 pairing equation checked by the oracle's pairing, the SAVER
    addend r_enc * P1 moves C by exactly that point."""
    import pairing as pg
    ni = 30
    nc = (1 << 20) - ni - 2
    gen = o.splitmix64(2020)
    cs, wit = cref.R1CS.synth(nc, ni, 4)
    tox = fr_array([o.rand_fr(gen) for _ in range(5)])
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
    assert dcs.m == 1 << 20 and dcs.domain_kind == "basic_radix2"
    r, s = L(o.rand_fr(gen), 4), L(o.rand_fr(gen), 4)
    kp = v.Keypair(ctx, dcs, tox, precompute=False)
    pa, pb, pc, proof = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
    vk = dict(alpha_g1=o.g1_from_limbs(kp.part("alpha_g1")[0]), beta_g2=o.g2_from_limbs(kp.part("beta_g2")[0]),
              gamma_g2=o.g2_from_limbs(kp.part("gamma_g2")[0]), delta_g2=o.g2_from_limbs(kp.part("delta_g2")[0]),
              gamma_ABC_g1=[o.g1_from_limbs(x) for x in kp.part("gamma_ABC_g1")])
    pub = [I(wit[i]) for i in range(ni)]
    assert pg.groth16_verify(vk, pub, (o.g1_from_limbs(pa), o.g2_from_limbs(pb), o.g1_from_limbs(pc)))
    assert np.array_equal(v.g1_decompress(proof[0:48]), pa) and np.array_equal(v.g2_decompress(proof[48:144]), pb)
    P1 = kp.part("gamma_ABC_g1")[1]; renc = L(o.rand_fr(gen), 4)
    _, _, pc2, _ = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s, saver_P1=P1, saver_r_enc=renc)
    assert o.g1_from_limbs(pc2) == o.G1.add(o.g1_from_limbs(pc), o.G1.mul(o.g1_from_limbs(P1), I(renc)))
    kp.free()
    kp2 = v.Keypair(ctx, dcs, tox, precompute=True)                      # resident key with window multiples: the same proof
    qa, qb, qc, proof2 = v.groth16_prove(ctx, dcs, kp2.pk, wit, r, s)
    assert np.array_equal(qa, pa) and np.array_equal(qb, pb) and np.array_equal(qc, pc) and proof2 == proof
    kp2.free(); dcs.free(); cs.free()


def test_prove_in_two_halves_and_from_a_packed_witness(cref):
    """vsp_groth16_prove_launch / _finish (one host thread, several proofs in flight, one per context) and the packed witness
    (two class bits per wire + the dense values, expanded on the GPU) give the proof of vsp_groth16_prove, bit for bit; malformed
    packed maps and out-of-order calls are refused."""
    import ctypes as C
    with v.Context(0) as c1, v.Context(0) as c2, v.Context(0) as c3:
        cs, wit, kp, dcs, pk, q, r, s = build(c1, cref, 3000, 7, seed=808, precompute="all")
        ref = v.groth16_prove(c1, dcs, pk, wit, r, s)
        eA, eB, eC = kp.prove(wit, r, s)
        assert np.array_equal(ref[0], eA) and np.array_equal(ref[1], eB) and np.array_equal(ref[2], eC)
        pw = v.PackedWitness(wit)
        assert 0 < pw.n_dense < wit.shape[0] and pw.nbytes < wit.nbytes // 2          # the synthetic witness is 90 % boolean
        # one thread, three contexts over ONE key and constraint system: launch all, finish in order, twice around; plain and packed alternate
        ctxs = (c1, c2, c3)
        for c in ctxs[1:]:
            v.groth16_prove(c, dcs, pk, wit, r, s)                                     # first use: workspaces
        for rounds in range(2):
            for k, c in enumerate(ctxs):
                v.groth16_prove_launch(c, dcs, pk, pw if (k + rounds) & 1 else wit, r, s)
            for c in ctxs:
                got = v.groth16_prove_finish(c)
                assert all(np.array_equal(a, b) for a, b in zip(got[:3], ref[:3])) and got[3] == ref[3]
        # the SAVER term travels through the launch
        P1 = kp.part("A_query")[3]; renc = L(987654321, 4)
        want = v.groth16_prove(c1, dcs, pk, wit, r, s, saver_P1=P1, saver_r_enc=renc)
        v.groth16_prove_launch(c2, dcs, pk, pw, r, s, saver_P1=P1, saver_r_enc=renc)
        assert v.groth16_prove_finish(c2)[3] == want[3] != ref[3]
        # out-of-order calls
        with pytest.raises(v.VspError, match="no proof in flight"):
            v.groth16_prove_finish(c1)
        v.groth16_prove_launch(c1, dcs, pk, wit, r, s)
        with pytest.raises(v.VspError, match="already in flight"):
            v.groth16_prove_launch(c1, dcs, pk, wit, r, s)
        assert v.groth16_prove_finish(c1)[3] == ref[3]
        # malformed packed witnesses: an offset off by one, a dense count that does not match, the reserved class, bits beyond num_vars
        for damage in ("offset", "count", "class3", "tail"):
            bad = v.PackedWitness(wit)
            if damage == "offset":
                bad.word_offsets[len(bad.word_offsets) // 2] += 1
            elif damage == "count":
                bad.n_dense -= 1
            elif damage == "class3":
                bad.class_words[1] |= np.uint64(3)
            else:
                assert wit.shape[0] % 32                                               # 3007 wires: the last word is partly used
                bad.class_words[-1] |= np.uint64(1) << np.uint64(62)
            with pytest.raises(v.VspError, match="packed witness"):
                v.groth16_prove_launch(c1, dcs, pk, bad, r, s)
        assert v.groth16_prove(c1, dcs, pk, wit, r, s)[3] == ref[3]                    # the context is usable after the refusals
        # a scalar >= r among the dense values is caught where the plain witness's would be (the census of the multi-exponentiations)
        bad = v.PackedWitness(wit); bad.dense[0] = L(o.R, 4)
        v.groth16_prove_launch(c1, dcs, pk, bad, r, s)
        with pytest.raises(v.VspError, match="canonical"):
            v.groth16_prove_finish(c1)
        assert v.groth16_prove(c1, dcs, pk, wit, r, s)[3] == ref[3]
        pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()


@pytest.mark.parametrize("nc,ni,K", [(10, 2, 3), (300, 5, 4), (2000, 30, 8), (5000, 3, 2)])
def test_batch_of_proofs_is_byte_identical_to_single_proofs_and_the_oracle(ctx, cref, nc, ni, K):
    """vsp_groth16_prove_batch (round 4): K witnesses of one constraint system proved in one pass over a plain key -- every proof equals the
    oracle's r1cs_gg_ppzksnark proof and vsp_groth16_prove's 192 bytes for the same (witness, r, s), under K different (r, s); over a key with
    tables of window multiples too."""
    cs, wit0, kp, dcs, pk, q, r0, s0 = build(ctx, cref, nc, ni, seed=nc + K)
    gen = o.splitmix64(7 * nc + K)
    wits, rs, ss = [wit0], [r0], [s0]
    for k in range(1, K):
        w = wit0.copy()                                         # the same witness under other randomness (different witnesses: the next test)
        wits.append(w); rs.append(L(o.rand_fr(gen), 4)); ss.append(L(o.rand_fr(gen), 4))
    W = np.stack(wits); R = np.stack(rs); S = np.stack(ss)
    A, B, Cc, proofs = v.groth16_prove_batch(ctx, dcs, pk, W, R, S)
    for k in range(K):
        eA, eB, eC = kp.prove(wits[k], rs[k], ss[k])
        assert np.array_equal(A[k], eA) and np.array_equal(B[k], eB) and np.array_equal(Cc[k], eC), k
        sA, sB, sC, sproof = v.groth16_prove(ctx, dcs, pk, wits[k], rs[k], ss[k])
        assert proofs[k] == sproof and np.array_equal(A[k], sA), k
    # a key with tables of window multiples (end of round 4: one bucket set per witness and query): the same proofs; refused with msm_batch_tables = 0
    for x in q: x.precompute(12)
    assert v.groth16_prove_batch(ctx, dcs, pk, W, R, S)[3] == proofs
    ctx.set_option("msm_batch_tables", 0)
    try:
        with pytest.raises(v.VspError):
            v.groth16_prove_batch(ctx, dcs, pk, W, R, S)
    finally:
        ctx.set_option("msm_batch_tables", 1)
    assert v.groth16_prove(ctx, dcs, pk, wits[0], rs[0], ss[0])[3] == proofs[0]
    pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()


def test_batch_of_proofs_over_different_witnesses(ctx, cref):
    """the witnesses of a batch are independent vectors: K different satisfying assignments of one system (the synthetic system's boolean
    wires that no product wire reads may take either value), each proof against the oracle"""
    nc, ni, K = 1200, 6, 5
    cs, wit0, kp, dcs, pk, q, r0, s0 = build(ctx, cref, nc, ni, seed=77)
    A_, B_, C_ = cs.export()
    read_by_product = set()
    for j in range(nc):
        a, b, c = int(A_[1][j]), int(B_[1][j]), int(C_[1][j])
        if not (a == b == c):
            read_by_product.add(a); read_by_product.add(b)
    free_bool = [int(C_[1][j]) for j in range(nc) if int(A_[1][j]) == int(B_[1][j]) == int(C_[1][j]) and int(C_[1][j]) not in read_by_product]
    assert len(free_bool) > 100
    rng = np.random.default_rng(3)
    gen = o.splitmix64(99)
    wits, rs, ss = [], [], []
    for k in range(K):
        w = wit0.copy()
        for idx in free_bool:
            if rng.random() < 0.5:
                w[idx - 1] = 0; w[idx - 1, 0] = int(rng.integers(0, 2))
        assert cs.is_satisfied(w)
        wits.append(w); rs.append(L(o.rand_fr(gen), 4)); ss.append(L(o.rand_fr(gen), 4))
    A, B, Cc, proofs = v.groth16_prove_batch(ctx, dcs, pk, np.stack(wits), np.stack(rs), np.stack(ss))
    assert len({p for p in proofs}) == K
    for k in range(K):
        eA, eB, eC = kp.prove(wits[k], rs[k], ss[k])
        assert np.array_equal(A[k], eA) and np.array_equal(B[k], eB) and np.array_equal(Cc[k], eC), k
    pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()


def test_two_batches_in_flight_from_one_thread_over_two_contexts(ctx, cref):
    """vsp_groth16_prove_batch_launch / _finish (round 4): one host thread keeps a batch in flight on each of two contexts over one plain key;
    every proof of either batch is byte-identical to the blocking batch call; a second launch, a single-proof launch or a finish without a
    launch on a busy / idle context is refused and leaves the batch in flight intact."""
    nc, ni, K = 900, 4, 3
    cs, wit0, kp, dcs, pk, q, r0, s0 = build(ctx, cref, nc, ni, seed=41)
    gen = o.splitmix64(4141)
    def rand_batch():
        return (np.ascontiguousarray(np.broadcast_to(wit0, (K,) + wit0.shape)), np.stack([L(o.rand_fr(gen), 4) for _ in range(K)]),
                np.stack([L(o.rand_fr(gen), 4) for _ in range(K)]))
    batches = [rand_batch() for _ in range(4)]
    expect = [v.groth16_prove_batch(ctx, dcs, pk, *b)[3] for b in batches]
    with v.Context(0) as c1:
        with pytest.raises(v.VspError):
            ctx.check(ctx.lib.vsp_groth16_prove_batch_finish(ctx.h, None, None, None, None))        # nothing in flight
        with pytest.raises(RuntimeError):
            v.groth16_prove_batch_finish(ctx)
        ring = [ctx, c1]
        got = [None] * 4
        v.groth16_prove_batch_launch(ring[0], dcs, pk, *batches[0])
        v.groth16_prove_batch_launch(ring[1], dcs, pk, *batches[1])
        with pytest.raises(v.VspError):
            ctx.check(ctx.lib.vsp_groth16_prove_batch_launch(ctx.h, dcs.h, pk.h, v.api._ptr(batches[2][0]), K, v.api._ptr(batches[2][1]), v.api._ptr(batches[2][2])))
        with pytest.raises(v.VspError):
            v.groth16_prove_launch(ctx, dcs, pk, wit0, r0, s0)                                        # a single proof on a context with a batch in flight
        got[0] = v.groth16_prove_batch_finish(ring[0])[3]
        v.groth16_prove_batch_launch(ring[0], dcs, pk, *batches[2])
        got[1] = v.groth16_prove_batch_finish(ring[1])[3]
        v.groth16_prove_batch_launch(ring[1], dcs, pk, *batches[3])
        got[2] = v.groth16_prove_batch_finish(ring[0])[3]
        got[3] = v.groth16_prove_batch_finish(ring[1])[3]
        assert got == expect
        assert v.groth16_prove(c1, dcs, pk, wit0, batches[0][1][0], batches[0][2][0])[3] == expect[0][0]
    pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()


def test_contexts_that_proved_batches_return_their_device_memory(ctx, cref):
    """vsp_destroy frees the batch prover's buffers too (witnesses, the 3 K evaluation vectors, the K coefficient vectors), also after a
    batch that was launched and never finished: free device memory after the third create / prove / destroy cycle is what it was after the first"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    def free_bytes():
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value
    nc, ni, K = 16000, 4, 16                                     # ~50 MB of batch buffers per context
    cs, wit0, kp, dcs, pk, q, r0, s0 = build(ctx, cref, nc, ni, seed=5)
    W = np.ascontiguousarray(np.broadcast_to(wit0, (K,) + wit0.shape)); R = np.stack([r0] * K); S = np.stack([s0] * K)
    expect = v.groth16_prove_batch(ctx, dcs, pk, W, R, S)[3]
    seen = []
    for cycle in range(3):
        c1 = v.Context(0)
        assert v.groth16_prove_batch(c1, dcs, pk, W, R, S)[3] == expect
        v.groth16_prove_batch_launch(c1, dcs, pk, W, R, S)       # left in flight: destroy waits it out
        c1.close()
        seen.append(free_bytes())
    assert abs(seen[2] - seen[0]) < (8 << 20), seen
    assert v.groth16_prove_batch(ctx, dcs, pk, W, R, S)[3] == expect
    pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()


def test_batch_prover_with_and_without_the_shared_digit_sort(ctx, cref):
    """A, B1 and B2 of a batch multiply by the same K witness vectors: by default B1 and B2 take A's digit sort and bucket plan (option
    prove_batch_share_plan = 1); with 0 every multi-exponentiation sorts for itself.  Same proofs either way, equal to the single call's;
    the last reduction step of a batch (k_dimbits since the end of round 4, msm_dimbits = 0: k_dimweight) does not change them either."""
    nc, ni, K = 3000, 5, 6
    cs, wit0, kp, dcs, pk, q, r0, s0 = build(ctx, cref, nc, ni, seed=23)
    gen = o.splitmix64(2323)
    W = np.ascontiguousarray(np.broadcast_to(wit0, (K,) + wit0.shape))
    R = np.stack([L(o.rand_fr(gen), 4) for _ in range(K)]); S = np.stack([L(o.rand_fr(gen), 4) for _ in range(K)])
    singles = [v.groth16_prove(ctx, dcs, pk, wit0, R[k], S[k])[3] for k in range(K)]
    try:
        for share, dimbits in ((1, -1), (0, -1), (1, 0), (0, 0), (1, 1)):
            ctx.set_option("prove_batch_share_plan", share); ctx.set_option("msm_dimbits", dimbits)
            assert v.groth16_prove_batch(ctx, dcs, pk, W, R, S)[3] == singles, (share, dimbits)
    finally:
        ctx.set_option("prove_batch_share_plan", 1); ctx.set_option("msm_dimbits", -1)
    pk.free(); dcs.free(); [x.free() for x in q]; kp.free(); cs.free()


def test_batch_over_a_generated_key_with_tables_on_all_five_queries(ctx, cref):
    """the configuration DESIGN.md 3.3b recommends for batched proving at the real circuit's size: vsp_groth16_generate with tables of window
    multiples on all five queries (precompute = 17) and a chosen window -- every proof of a batch against the oracle's prover, and the two
    halves over the same key"""
    nc, ni, K = 2500, 4, 5
    gen = o.splitmix64(171)
    cs, wit = cref.R1CS.synth(nc, ni, 171)
    tox = np.array([L(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
    ref = cref.Keypair(cs, tox)
    W = np.ascontiguousarray(np.broadcast_to(wit, (K,) + wit.shape))
    R = np.stack([L(o.rand_fr(gen), 4) for _ in range(K)]); S = np.stack([L(o.rand_fr(gen), 4) for _ in range(K)])
    for window in (9, 14):
        kp = v.Keypair(ctx, dcs, tox, precompute=17, precompute_window=window)
        A, B, Cc, proofs = v.groth16_prove_batch(ctx, dcs, kp.pk, W, R, S)
        for k in range(K):
            eA, eB, eC = ref.prove(wit, R[k], S[k])
            assert np.array_equal(A[k], eA) and np.array_equal(B[k], eB) and np.array_equal(Cc[k], eC), (window, k)
        v.groth16_prove_batch_launch(ctx, dcs, kp.pk, W, R, S)
        assert v.groth16_prove_batch_finish(ctx)[3] == proofs
        assert v.groth16_prove(ctx, dcs, kp.pk, wit, R[0], S[0])[3] == proofs[0]
        kp.free()
    dcs.free(); ref.free(); cs.free()
