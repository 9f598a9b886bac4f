"""Randomised parity run of generator + prover (any constraint count: basic and step radix-2 domains) against the C oracle
(diagnostic):  python tools/fuzz_prove.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vote_saver_protocol_amd as v
import cref
from conftest import rand_fr_array

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = v.Context(0)
ctx2 = v.Context(0)
t0 = time.time(); it = 0; kinds = {"basic_radix2": 0, "step_radix2": 0}
last = time.time()
while time.time() - t0 < budget:
    it += 1
    if time.time() - last > 60:
        print("... %d cases, %.0f s" % (it, time.time() - t0), flush=True); last = time.time()
    nc = int(np.exp(rng.uniform(np.log(2), np.log(float(os.environ.get("MAX_NC", "5000"))))))
    ni = int(rng.integers(0, min(30, nc) + 1))
    s = int(rng.integers(1, 1 << 30))
    cs, wit = cref.R1CS.synth(nc, ni, s)
    tox = rand_fr_array(5, seed=s + 1)
    r, s_ = rand_fr_array(2, seed=s + 2)
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
    assert dcs.m == cs.m
    kinds[dcs.domain_kind] += 1
    pre = bool(rng.random() < 0.5)
    kp = v.Keypair(ctx, dcs, tox, precompute=pre)
    ref = cref.Keypair(cs, tox)
    ok = all(np.array_equal(kp.part(nm), ref.part(nm)) for nm in ("H_query", "L_query", "B_query_g2", "gamma_ABC_g1"))
    saver = rng.random() < 0.3
    P1 = ref.part("A_query")[0] if saver else None
    renc = rand_fr_array(1, seed=s + 3)[0] if saver else None
    pa, pb, pc, _ = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s_, saver_P1=P1, saver_r_enc=renc)
    ea, eb, ec = ref.prove(wit, r, s_, P1=P1, r_enc=renc)
    ok = ok and np.array_equal(pa, ea) and np.array_equal(pb, eb) and np.array_equal(pc, ec)
    if ok and rng.random() < 0.6:                             # the same statement(s) through the batch prover (plain and table keys; basic and step domains)
        K = int(rng.integers(1, 7))
        R = rand_fr_array(K, seed=s + 4); S = rand_fr_array(K, seed=s + 5)
        if rng.random() < 0.5:
            bA, bB, bC, _ = v.groth16_prove_batch(ctx, dcs, kp.pk, np.stack([wit] * K), R, S)
        else:                                                 # the two halves, a second batch in flight on a second context meanwhile
            v.groth16_prove_batch_launch(ctx, dcs, kp.pk, np.stack([wit] * K), R, S)
            v.groth16_prove_batch_launch(ctx2, dcs, kp.pk, np.stack([wit] * K), S, R)
            bA, bB, bC, _ = v.groth16_prove_batch_finish(ctx)
            cA, cB, cC, _ = v.groth16_prove_batch_finish(ctx2)
            xa, xb, xc = ref.prove(wit, S[K - 1], R[K - 1])
            ok = ok and np.array_equal(cA[K - 1], xa) and np.array_equal(cB[K - 1], xb) and np.array_equal(cC[K - 1], xc)
            kinds["two_halves"] = kinds.get("two_halves", 0) + 1
        for k in range(K):
            xa, xb, xc = ref.prove(wit, R[k], S[k])
            ok = ok and np.array_equal(bA[k], xa) and np.array_equal(bB[k], xb) and np.array_equal(bC[k], xc)
        kinds["batched"] = kinds.get("batched", 0) + 1
        if pre: kinds["batched_table_key"] = kinds.get("batched_table_key", 0) + 1
    kp.free(); dcs.free(); ref.free(); cs.free()
    if not ok:
        print("MISMATCH", dict(it=it, seed=seed, nc=nc, ni=ni, synth_seed=s, pre=pre, saver=saver)); sys.exit(1)
print("fuzz ok: %d constraint systems in %.0f s" % (it, time.time() - t0), kinds)
