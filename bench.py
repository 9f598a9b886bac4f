#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Groth16 prover hot path.

Metric (BASELINE.json): G1 MSM points/sec at 2^20 (config[1]: 2^20-point BLS12-381 G1 Pippenger MSM, random
scalars/points, bit-exact vs multiexp), whole job, inputs resident in HBM, PLAIN bases (no precomputed table).  One
"step" = one full MSM of the rank's resident shard (sort + bucket accumulation + bucket reduction + host Horner), then
-- when launched by torch.distributed.run -- the exchange step of the sharded MSM: an RCCL all-gather of the 144-byte
(G1) / 288-byte (G2) Jacobian partial sums and a local fold.

    python bench.py --gpus 1 --steps 20 --warmup 3                       # headline: 2^20 G1 points per GPU ("weak")
    python bench.py --gpus N ...                                         # N > 1 without RANK in the environment: this process starts
                                                                         # `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
                                                                         # before anything touches the GPU and returns its exit code
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...   # the driver's form
    ... bench.py --gpus N --scaling strong --total-log-n 26 --group g1    # BASELINE config 5: ONE 2^26-point problem, rank g holds chunk g
    ... bench.py --gpus N --scaling strong --total-log-n 24 --group g2

Without --scaling strong the per-GPU work is fixed (2^20 points per rank, "weak"); the default run additionally measures
BASELINE config 5 (2^26 G1 + 2^24 G2 points split over the N ranks) -- `scaling_strong` at the top level of the line (and
`extras.config5` in full) -- so the driver's N = 1, 2, 4, 8 lines carry the strong-scaling curve of that fixed problem next to the
headline.  The sharded multi-exponentiation itself is the package's (vote_saver_protocol_amd/sharded.py); this file only generates the
synthetic shards, times and verifies.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (bucket accumulation, k_accum28) with HIP events recorded on
the launch stream, from a leg with ONE multi-exponentiation in flight (so nothing else shares the GPU with the kernel; the
pipelined average is reported beside it); `cpu_baseline` times the oracle's serial BDLO12 (the reference algorithm, 1 thread like
the reference build) on the same inputs, `cpu_baseline_all_cores` the same split in chunks over every host core (the `chunks`
argument of multiexp) -- reported baselines only.  Nothing here reads /root/reference.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
BYTES_PER_PAIR = {1: 128, 2: 224}   # affine base (96 / 192 B) + 32 B scalar
MADS_PER_ADD = {1: 6 * 392 + 2 * 301 + 588, 2: 2 * (8 * 588 + 2 * 392)}   # v_mad_u64_u32 per mixed addition: 6 products, 2 squares, 1 dual product (G2: two lanes per point)
# Issue cost of a SIMD per wave-instruction, in SHADER CYCLES, with >= 3 waves on the SIMD (tools/ubench_issue.hip, profiles/r3_ubench_issue.txt:
# every wave stamps s_memtime, a SIMD's cost = (last wave's end - first wave's start) / instructions of all its waves): the dependent chain of
# v_mad_u64_u32 the product routines are, and the simple 32-bit VALU instructions around it.  (The figure of rounds 1-2 -- 6.6 cycles -- came from
# tools/ubench_valu.hip, whose separate asm statements let the compiler put an s_nop after every multiply-add, and was priced in wall time at the nominal clock.)
ISSUE_CYCLES_FALLBACK = {"mad": 4.59, "simple": 2.88}
N_SIMD = 1024
PMC_FILES = ("r4_pmc_hbm_traffic.json", "r3_pmc_hbm_traffic.json", "r2_pmc_hbm_traffic.json", "r1_h_pmc_hbm_traffic.json")   # newest first; see profiles/README.md


def rand_fr(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)     # < 2^254 < r: canonical
    return a


def to_ints(arr):
    a = np.asarray(arr).reshape(-1, 4)
    v = a[:, 3].astype(object)
    for k in (2, 1, 0):
        v = (v << 64) | a[:, k].astype(object)
    return v


def limbs(v, n):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def bench_prove_step_domain(ctx, v, cref, o, precompute=True):
    """A constraint count that is not a power of two, as the reference's real circuit has: 2^20 + 2^17 - 32 constraints.
    make_evaluation_domain selects the step radix-2 domain of 2^20 + 2^17 elements (a power-of-two-only prover would pay for
    2^21).  GPU generator, prove, pairing check."""
    import pairing as pg
    ni = 30
    nc = (1 << 20) + (1 << 17) - ni - 2
    gen = o.splitmix64(15)
    cs, wit = cref.R1CS.synth(nc, ni, 14)
    tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
    A, B, Cm = cs.export()
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, A, B, Cm)
    kp = v.Keypair(ctx, dcs, tox, precompute=precompute)
    r = limbs(o.rand_fr(gen), 4); s_ = limbs(o.rand_fr(gen), 4)
    pa, pb, pc, _ = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s_)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        pa, pb, pc, _ = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s_)
    dt = (time.perf_counter() - t0) / reps
    vk = dict(alpha_g1=o.g1_from_limbs(kp.part("alpha_g1")[0]), beta_g2=o.g2_from_limbs(kp.part("beta_g2")[0]),
              gamma_g2=o.g2_from_limbs(kp.part("gamma_g2")[0]), delta_g2=o.g2_from_limbs(kp.part("delta_g2")[0]),
              gamma_ABC_g1=[o.g1_from_limbs(x) for x in kp.part("gamma_ABC_g1")])
    pub = [int(x) for x in to_ints(wit[:ni]).tolist()]
    ok = pg.groth16_verify(vk, pub, (o.g1_from_limbs(pa), o.g2_from_limbs(pb), o.g1_from_limbs(pc)))
    out = {"prove_step_domain_constraints": nc, "prove_step_domain_m": int(dcs.m), "prove_step_domain_kind": dcs.domain_kind,
           "prove_step_domain_ms": dt * 1e3, "prove_step_domain_pairing_verified": bool(ok)}
    kp.free(); dcs.free(); cs.free()
    return out


def bench_prove_small(ctx, v, cref, o, log_m=16, precompute=True):
    """Proofs per second at the real circuit's likely size (depth 20 over a Pedersen / Jubjub Merkle path is 2^15..2^16
    constraints): latency of one proof, then 1..4 host threads with a context each over ONE resident key."""
    import threading
    ni = 30
    nc = (1 << log_m) - ni - 2
    gen = o.splitmix64(16)
    cs, wit = cref.R1CS.synth(nc, ni, 40, ballot=(25, 3))
    tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
    kp = v.Keypair(ctx, dcs, tox, precompute=precompute)
    r = limbs(o.rand_fr(gen), 4); s_ = limbs(o.rand_fr(gen), 4)
    wit = ctx.host_register(np.ascontiguousarray(wit))
    ref = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s_)
    each = []
    for _ in range(20):
        t0 = time.perf_counter(); v.groth16_prove(ctx, dcs, kp.pk, wit, r, s_); each.append(time.perf_counter() - t0)
    out = {f"prove_2p{log_m}_ms": float(np.median(each)) * 1e3, f"prove_2p{log_m}_constraints": nc, f"prove_2p{log_m}_key_bytes": int(kp.device_bytes())}
    ctxs = [ctx] + [v.Context(ctx.device) for _ in range(3)]
    for c in ctxs[1:]:
        v.groth16_prove(c, dcs, kp.pk, wit, r, s_)
    per_thread, same = 40, True
    for k in (1, 2, 3, 4):
        res = {}

        def worker(c, tag):
            for _ in range(per_thread):
                res[tag] = v.groth16_prove(c, dcs, kp.pk, wit, r, s_)
        th = [threading.Thread(target=worker, args=(c, i)) for i, c in enumerate(ctxs[:k])]
        t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        out[f"prove_2p{log_m}_{k}_contexts_proofs_per_s"] = k * per_thread / (time.perf_counter() - t0)
        same = same and all(res[i][3] == ref[3] for i in range(k))
    out[f"prove_2p{log_m}_contexts_same_proof"] = bool(same)
    for c in ctxs[1:]:
        c.close()
    # ---- batched proving (vsp_groth16_prove_batch, round 4): K witnesses in one pass over a PLAIN key, one context, one host thread calling
    try:
        kpp = v.Keypair(ctx, dcs, tox, precompute=False)
        single = v.groth16_prove(ctx, dcs, kpp.pk, wit, r, s_)
        t1 = []
        for _ in range(10):
            t0 = time.perf_counter(); v.groth16_prove(ctx, dcs, kpp.pk, wit, r, s_); t1.append(time.perf_counter() - t0)
        out[f"prove_2p{log_m}_plain_key_ms"] = float(np.median(t1)) * 1e3
        out[f"prove_2p{log_m}_plain_key_bytes"] = int(kpp.device_bytes())
        batch = {}
        for K in (4, 8, 16, 32):
            W = np.ascontiguousarray(np.broadcast_to(np.asarray(wit), (K,) + np.asarray(wit).shape))
            R = np.ascontiguousarray(np.broadcast_to(r, (K, 4))); S = np.ascontiguousarray(np.broadcast_to(s_, (K, 4)))
            got = v.groth16_prove_batch(ctx, dcs, kpp.pk, W, R, S)
            reps_b = 5
            t0 = time.perf_counter()
            for _ in range(reps_b):
                got = v.groth16_prove_batch(ctx, dcs, kpp.pk, W, R, S)
            dtb = (time.perf_counter() - t0) / reps_b
            batch[str(K)] = {"ms_per_batch": dtb * 1e3, "proofs_per_s": K / dtb, "every_proof_equals_the_single_call": bool(all(p == single[3] for p in got[3]))}
        out[f"prove_2p{log_m}_batched"] = batch
        out[f"prove_2p{log_m}_batched_best_proofs_per_s"] = max(b["proofs_per_s"] for b in batch.values())
        # a service's shape: a few host threads, each with a context of its own, each proving batches over the ONE resident key (the host
        # steps of one context's batch overlap the GPU's work on another's)
        multi = {}
        ctxb = [ctx] + [v.Context(ctx.device) for _ in range(3)]
        for C_, K in ((2, 16), (4, 32)):
            W = np.ascontiguousarray(np.broadcast_to(np.asarray(wit), (K,) + np.asarray(wit).shape))
            R = np.ascontiguousarray(np.broadcast_to(r, (K, 4))); S = np.ascontiguousarray(np.broadcast_to(s_, (K, 4)))
            for c in ctxb[:C_]:
                v.groth16_prove_batch(c, dcs, kpp.pk, W, R, S)
            reps_m = 4

            def bworker(c):
                for _ in range(reps_m):
                    v.groth16_prove_batch(c, dcs, kpp.pk, W, R, S)
            thb = [threading.Thread(target=bworker, args=(c,)) for c in ctxb[:C_]]
            t0 = time.perf_counter()
            for x in thb: x.start()
            for x in thb: x.join()
            multi[f"{C_}_contexts_x_batch_{K}"] = C_ * reps_m * K / (time.perf_counter() - t0)
        # ONE host thread, a batch in flight on each of two contexts (vsp_groth16_prove_batch_launch / _finish): it assembles one batch
        # while the card works on the other
        K = 32
        W = np.ascontiguousarray(np.broadcast_to(np.asarray(wit), (K,) + np.asarray(wit).shape))
        R = np.ascontiguousarray(np.broadcast_to(r, (K, 4))); S = np.ascontiguousarray(np.broadcast_to(s_, (K, 4)))
        ring, total, same_b = ctxb[:2], 8, True
        t0 = time.perf_counter()
        for i in range(total + 2):
            c = ring[i % 2]
            if i >= 2:
                same_b = same_b and all(p == single[3] for p in v.groth16_prove_batch_finish(c)[3])
            if i < total:
                v.groth16_prove_batch_launch(c, dcs, kpp.pk, W, R, S)
        multi["one_thread_two_batches_of_32_in_flight"] = total * K / (time.perf_counter() - t0)
        multi["one_thread_every_proof_equals_the_single_call"] = bool(same_b)
        # the same ring over a key whose FIVE queries carry 14-bit tables of window multiples (small at this size): one bucket set per witness
        # and query instead of one per window -- only where the tables are cheap (the real circuit's size, not 2^20)
        if log_m <= 17:
            kpt = v.Keypair(ctx, dcs, tox, precompute=17, precompute_window=14)
            ring3, same_t = ctxb[:3], True
            for c in ring3:
                v.groth16_prove_batch(c, dcs, kpt.pk, W, R, S)
            t0 = time.perf_counter()
            for i in range(total + 3):
                c = ring3[i % 3]
                if i >= 3:
                    same_t = same_t and all(p == single[3] for p in v.groth16_prove_batch_finish(c)[3])
                if i < total:
                    v.groth16_prove_batch_launch(c, dcs, kpt.pk, W, R, S)
            multi["one_thread_three_batches_of_32_in_flight_table_key"] = total * K / (time.perf_counter() - t0)
            multi["table_key_every_proof_equals_the_single_call"] = bool(same_t)
            multi["table_key_bytes"] = int(kpt.device_bytes())
            kpt.free()
        for c in ctxb[1:]:
            c.close()
        out[f"prove_2p{log_m}_batched_multi_context_proofs_per_s"] = multi
        kpp.free()
    except Exception as e:                         # secondary measurement: never take the bench line down
        out[f"prove_2p{log_m}_batched_error"] = repr(e)
    ctx.host_unregister(wit)
    kp.free(); dcs.free(); cs.free()
    return out


def bench_prove(ctx, v, cref, o, dev, torch, log_m, precompute=True):
    """BASELINE config 4: full r1cs_gg_ppzksnark prove on a synthetic satisfiable R1CS filling a 2^log_m domain
    (90 % boolean wires, 30 public inputs).  The proving key is built on the GPU (generator batch
    exponentiation), the proof is verified with the oracle's pairing, the CPU leg is the oracle's serial prover."""
    import pairing as pg
    ni = 30
    nc = (1 << log_m) - ni - 2
    gen = o.splitmix64(5)
    cs, wit = cref.R1CS.synth(nc, ni, 4, ballot=(25, 7))      # the first 25 public inputs are a one-hot ballot (msg_size = 25, common.hpp:163, 1029-1040)
    tox_i = [o.rand_fr(gen) for _ in range(5)]
    tox = np.array([o.int_to_limbs(x, 4) for x in tox_i], dtype=np.uint64)
    A, B, Cm = cs.export()
    dcs = v.R1CS(ctx, nc, ni, cs.num_vars, A, B, Cm)
    t0 = time.perf_counter()
    kp = v.Keypair(ctx, dcs, tox, precompute=precompute)         # zk::generate on the GPU (incl. window-multiple precompute)
    setup_s = time.perf_counter() - t0
    pk = kp.pk
    alpha_g1, beta_g2, gamma_g2, delta_g2 = (kp.part(nm)[0] for nm in ("alpha_g1", "beta_g2", "gamma_g2", "delta_g2"))
    gamma_abc = kp.part("gamma_ABC_g1")
    r = limbs(o.rand_fr(gen), 4); s_ = limbs(o.rand_fr(gen), 4)
    wit = ctx.host_register(np.ascontiguousarray(wit))                     # page-locked once (vsp_host_register): the witness copy is an asynchronous DMA
    pa, pb, pc, proof = v.groth16_prove(ctx, dcs, pk, wit, r, s_)          # warm-up (twiddles, workspaces)
    reps = 10
    ctx.stats_reset()
    each = []
    for _ in range(reps):
        t0 = time.perf_counter()
        pa, pb, pc, proof = v.groth16_prove(ctx, dcs, pk, wit, r, s_)
        each.append(time.perf_counter() - t0)
    dt = float(np.median(each))                              # the median: one proof in a few hundred takes twice as long (a 14 ms reading in a 5-proof mean)
    phases = {k: ctx.stat("prove_" + k + "_ms") / reps for k in ("launch", "host_overlap", "wait", "assembly")}
    # throughput mode: contexts are independent and keys / constraint systems are plain device resources, so two host threads
    # with a context each prove concurrently over ONE resident key (the tails of one proof overlap the bulk of the other)
    import threading
    n_ctx = int(os.environ.get("VSP_BENCH_PROVE_CONTEXTS", "2"))
    extra_ctx = [v.Context(ctx.device) for _ in range(n_ctx - 1)]
    per_thread = 20                                        # (7 proofs per thread gave figures that jumped 30 % between runs)
    outs = {}

    def worker(c, tag):
        v.groth16_prove(c, dcs, pk, wit, r, s_)
        for _ in range(per_thread):
            outs[tag] = v.groth16_prove(c, dcs, pk, wit, r, s_)

    for c in extra_ctx:
        v.groth16_prove(c, dcs, pk, wit, r, s_)            # warm the other contexts' workspaces
    th = [threading.Thread(target=worker, args=(c, i)) for i, c in enumerate([ctx] + extra_ctx)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt2 = (time.perf_counter() - t0) / (n_ctx * (per_thread + 1))
    same = all(np.array_equal(outs[i][0], pa) and np.array_equal(outs[i][1], pb) and np.array_equal(outs[i][2], pc) for i in range(n_ctx))
    # ---- ONE host thread, proofs in flight on three contexts (vsp_groth16_prove_launch / _finish), the witness in the packed form
    #      (two class bits per wire + the dense values: vsp_witness_pack; a witness generator would emit it directly)
    t0 = time.perf_counter(); pw = v.PackedWitness(wit); pack_ms = (time.perf_counter() - t0) * 1e3
    v.groth16_prove_launch(ctx, dcs, pk, pw, r, s_); got = v.groth16_prove_finish(ctx)
    packed_same = bool(np.array_equal(got[0], pa) and np.array_equal(got[1], pb) and np.array_equal(got[2], pc))
    tp = []
    for _ in range(reps):
        t0 = time.perf_counter(); v.groth16_prove_launch(ctx, dcs, pk, pw, r, s_); v.groth16_prove_finish(ctx); tp.append(time.perf_counter() - t0)
    packed_ms = float(np.median(tp)) * 1e3
    ring = [ctx] + extra_ctx + ([v.Context(ctx.device)] if n_ctx < 3 else [])
    for c in ring[n_ctx:]:
        v.groth16_prove(c, dcs, pk, wit, r, s_)
    one_thread = {}
    for src, tag in ((pw, "packed"), (wit, "plain")):
        total = 60
        dt3, last = prove_ring(v, ring, dcs, pk, src, r, s_, total)
        one_thread[tag] = {"proofs_per_s": total / dt3, "ms_per_proof": dt3 / total * 1e3, "contexts": len(ring),
                           "same_proof": bool(np.array_equal(last[0], pa) and np.array_equal(last[1], pb) and np.array_equal(last[2], pc))}
    for c in ring[n_ctx:]:
        c.close()
    for c in extra_ctx:
        c.close()
    # ---- key memory: the same proofs over a PLAIN key -- no table of window multiples; the 28-bit copy keeps the endomorphism
    #      image beside every point (1.3 GB at 2^20 against 19 GB) -- latency, the ring's throughput, bytes resident
    plain = {}
    try:
        t0 = time.perf_counter(); kp_plain = v.Keypair(ctx, dcs, tox, precompute=False); plain_setup = time.perf_counter() - t0
        ring_all = [ctx, v.Context(ctx.device), v.Context(ctx.device)]
        for c in ring_all:
            v.groth16_prove(c, dcs, kp_plain.pk, wit, r, s_)
        tpl = []
        for _ in range(reps):
            t0 = time.perf_counter(); got = v.groth16_prove(ctx, dcs, kp_plain.pk, wit, r, s_); tpl.append(time.perf_counter() - t0)
        dtp, lastp = prove_ring(v, ring_all, dcs, kp_plain.pk, pw, r, s_, 60)
        dense_p = ctx.host_register(rand_fr(cs.num_vars, 99))
        v.groth16_prove(ctx, dcs, kp_plain.pk, dense_p, r, s_)
        tdp = []
        for _ in range(5):
            t0 = time.perf_counter(); v.groth16_prove(ctx, dcs, kp_plain.pk, dense_p, r, s_); tdp.append(time.perf_counter() - t0)
        ctx.host_unregister(dense_p); del dense_p
        Kb = 4
        Wb = np.ascontiguousarray(np.broadcast_to(np.asarray(wit), (Kb,) + np.asarray(wit).shape))
        Rb = np.ascontiguousarray(np.broadcast_to(r, (Kb, 4))); Sb = np.ascontiguousarray(np.broadcast_to(s_, (Kb, 4)))
        gotb = v.groth16_prove_batch(ctx, dcs, kp_plain.pk, Wb, Rb, Sb)
        t0 = time.perf_counter()
        for _ in range(3):
            gotb = v.groth16_prove_batch(ctx, dcs, kp_plain.pk, Wb, Rb, Sb)
        dtb = (time.perf_counter() - t0) / 3
        batch4 = {"batch": Kb, "ms_per_batch": dtb * 1e3, "proofs_per_s": Kb / dtb, "every_proof_equals_the_single_call": bool(all(p == proof for p in gotb[3]))}
        del Wb
        plain = {f"prove_2p{log_m}_plain_key_ms": float(np.median(tpl)) * 1e3, f"prove_2p{log_m}_plain_key_bytes": int(kp_plain.device_bytes()),
                 f"prove_2p{log_m}_plain_key_proofs_per_s": 60 / dtp, f"prove_2p{log_m}_plain_key_build_s": plain_setup,
                 f"prove_2p{log_m}_plain_key_dense_witness_ms": float(np.median(tdp)) * 1e3, f"prove_2p{log_m}_plain_key_batched": batch4,
                 f"prove_2p{log_m}_plain_key_same_proof": bool(np.array_equal(got[0], pa) and np.array_equal(got[1], pb) and np.array_equal(got[2], pc)
                                                               and np.array_equal(lastp[0], pa) and np.array_equal(lastp[2], pc))}
        for c in ring_all[1:]:
            c.close()
        kp_plain.free()
    except Exception as e:                         # secondary measurement: never take the bench line down
        plain = {f"prove_2p{log_m}_plain_key_error": repr(e)}
    # ---- the same prover on a DENSE witness (every wire a uniform field element: no zeros, no ones -- the all-ones bucket and the small windows
    #      of a boolean witness are gone, every witness multi-exponentiation is a full-width one).  Not a satisfying assignment: timing only.
    dense = ctx.host_register(rand_fr(cs.num_vars, 99))
    v.groth16_prove(ctx, dcs, pk, dense, r, s_)
    td = []
    for _ in range(5):
        t0 = time.perf_counter(); v.groth16_prove(ctx, dcs, pk, dense, r, s_); td.append(time.perf_counter() - t0)
    ctx.host_unregister(dense)
    dense_ms = float(np.median(td)) * 1e3
    del dense
    key_bytes = kp.device_bytes()
    # ---- the reference parses the proving key INSIDE its timed vote phase (main.cpp:446-456 around common.hpp:1002-1006): the "fast" blob of this
    #      very key -> a resident key, converted on the GPU (byte order, curve check, subgroup policy, Montgomery form, 28-bit tables, window multiples)
    pk_load = {}
    try:
        blob = kp.to_blob()
        t0 = time.perf_counter(); k2 = v.Keypair.from_blob(ctx, blob, precompute=precompute); pk_load_s = time.perf_counter() - t0
        pa2, pb2, pc2, _ = v.groth16_prove(ctx, dcs, k2.pk, wit, r, s_)
        pk_load = {f"pk_from_blob_2p{log_m}_s": pk_load_s, f"pk_blob_2p{log_m}_bytes": len(blob),
                   f"pk_from_blob_2p{log_m}_same_proof": bool(np.array_equal(pa2, pa) and np.array_equal(pb2, pb) and np.array_equal(pc2, pc)),
                   f"pk_from_blob_2p{log_m}_subgroup_checks": ctx.stat("bases_subgroup_checks")}
        k2.free(); del blob
    except Exception as e:
        pk_load = {f"pk_from_blob_2p{log_m}_error": repr(e)}
    vk = dict(alpha_g1=o.g1_from_limbs(alpha_g1), beta_g2=o.g2_from_limbs(beta_g2), gamma_g2=o.g2_from_limbs(gamma_g2),
              delta_g2=o.g2_from_limbs(delta_g2), gamma_ABC_g1=[o.g1_from_limbs(x) for x in gamma_abc])
    pub = [int(x) for x in to_ints(wit[:ni]).tolist()]
    ok = pg.groth16_verify(vk, pub, (o.g1_from_limbs(pa), o.g2_from_limbs(pb), o.g1_from_limbs(pc)))
    out = {f"prove_2p{log_m}_dense_witness_ms": dense_ms, f"prove_2p{log_m}_key_bytes": int(key_bytes), **pk_load, **plain,
           f"prove_2p{log_m}_packed_witness_ms": packed_ms, f"prove_2p{log_m}_packed_witness_same_proof": packed_same,
           f"prove_2p{log_m}_witness_pack_host_ms": pack_ms, f"prove_2p{log_m}_packed_witness_bytes": int(pw.nbytes), f"prove_2p{log_m}_plain_witness_bytes": int(wit.nbytes),
           f"prove_2p{log_m}_one_thread_pipelined": one_thread,
           f"prove_2p{log_m}_ms": dt * 1e3, f"prove_2p{log_m}_proofs_per_s": 1.0 / dt, f"prove_2p{log_m}_pairing_verified": bool(ok),
           f"prove_2p{log_m}_constraints": nc, f"generate_2p{log_m}_gpu_s": setup_s, f"prove_2p{log_m}_key_precomputed": bool(precompute),
           f"prove_2p{log_m}_phase_ms": phases,
           f"prove_2p{log_m}_ms_mean_max": [float(np.mean(each)) * 1e3, float(np.max(each)) * 1e3], f"prove_2p{log_m}_two_contexts_ms_per_proof": dt2 * 1e3, f"prove_2p{log_m}_two_contexts_proofs_per_s": 1.0 / dt2,
           f"prove_2p{log_m}_two_contexts_same_proof": bool(same)}
    # the reference's vote phase around the same proof (common.hpp:1131-1145): encrypt<elgamal_verifiable> (ciphertext of the 25
    # message blocks on the host while the GPU proves, proof with the SAVER addend) + rerandomize; verified by the oracle's pairing
    try:
        import saver as sv
        nmsg = 25
        rnd = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(3 * nmsg + 2)], dtype=np.uint64)
        gabc_l = np.ascontiguousarray(gamma_abc)
        delta_g1, gamma_g1 = kp.part("delta_g1")[0], kp.part("gamma_g1")[0]
        pk_w, _, _ = v.saver_generate_keypair(ctx, rnd, gabc_l, delta_g1, gamma_g1, nmsg)
        t0 = time.perf_counter(); spk = v.SaverPublicKey(ctx, pk_w, gabc_l[:nmsg + 1], nmsg); load_s = time.perf_counter() - t0
        r_enc = limbs(o.rand_fr(gen), 4)
        rnd3 = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(3)], dtype=np.uint64)
        wit_pinned = wit
        v.saver_encrypt(ctx, spk, dcs, pk, wit[:nmsg], wit_pinned, r_enc, r, s_)
        t_enc = t_rer = 0.0
        for _ in range(reps):
            t0 = time.perf_counter()
            ct, abc, _ = v.saver_encrypt(ctx, spk, dcs, pk, wit[:nmsg], wit_pinned, r_enc, r, s_)
            t1 = time.perf_counter()
            ct2, abc2, _ = v.saver_rerandomize(ctx, spk, delta_g2, rnd3, ct, abc)
            t_enc += t1 - t0; t_rer += time.perf_counter() - t1
        pkd = sv.pk_from_words(pk_w, nmsg)
        okv = sv.verify_encryption(pkd, vk, [o.g1_from_limbs(x) for x in ct2], (o.g1_from_limbs(abc2[0]), o.g2_from_limbs(abc2[1]), o.g1_from_limbs(abc2[2])), pub[nmsg:])
        out.update({f"vote_phase_2p{log_m}_encrypt_ms": t_enc / reps * 1e3, f"vote_phase_2p{log_m}_rerandomize_ms": t_rer / reps * 1e3,
                    f"vote_phase_2p{log_m}_ms": (t_enc + t_rer) / reps * 1e3, f"vote_phase_2p{log_m}_msg_size": nmsg,
                    f"vote_phase_2p{log_m}_pk_load_once_s": load_s, f"vote_phase_2p{log_m}_verify_encryption": bool(okv)})
        if f"pk_from_blob_2p{log_m}_s" in out:
            # the reference's timed region (main.cpp:446-456): deserialise the keys, then encrypt + rerandomize -- here: proving-key blob -> resident key on
            # the GPU, the SAVER key's fixed-base tables, one vote.  (Circuit construction and witness generation, also inside the reference's region, are out of scope.)
            out[f"vote_phase_2p{log_m}_including_key_load_ms"] = (out[f"pk_from_blob_2p{log_m}_s"] + load_s) * 1e3 + (t_enc + t_rer) / reps * 1e3
        spk.free()
    except Exception as e:                         # secondary measurement: never take the bench line down
        out[f"vote_phase_2p{log_m}_error"] = repr(e)
    ctx.host_unregister(wit)
    kp.free(); dcs.free(); cs.free()
    # CPU legs on bounded samples: the oracle's serial generator + prover at 2^14 and at 2^16 (the real circuit's likely size, SURVEY.md
    # section 0), and the GPU on the same instances, bit for bit
    for lg_s in (14, 16):
        out.update(cpu_prove_leg(ctx, v, cref, o, lg_s, tox, r, s_, ni=ni, precompute=precompute))
    return out



def prove_ring(v, ring, dcs, pk, src, r, s_, total):
    """ONE host thread, len(ring) - 1 proofs in flight on a ring of contexts (vsp_groth16_prove_launch / _finish): while proof k is being
    awaited, proofs k + 1 .. are running.  Returns (seconds for `total` proofs, the last proof)."""
    depth = len(ring)
    for k in range(depth - 1):
        v.groth16_prove_launch(ring[k], dcs, pk, src, r, s_)
    t0 = time.perf_counter()
    last = None
    for k in range(total):
        nxt = k + depth - 1
        v.groth16_prove_launch(ring[nxt % depth], dcs, pk, src, r, s_)
        last = v.groth16_prove_finish(ring[k % depth])
    dt = time.perf_counter() - t0
    for k in range(total, total + depth - 1):
        last = v.groth16_prove_finish(ring[k % depth])
    return dt, last


def bcast_arrays(dist, torch, coll_dev, rank, arrays):
    """rank 0's list of numpy arrays on every rank (shapes and dtypes first, then the bytes): how the ONE synthetic constraint system of
    the replica-proving leg reaches the ranks that did not generate it"""
    meta = [[(a.shape, a.dtype.str) for a in arrays] if rank == 0 else None]
    dist.broadcast_object_list(meta, src=0)
    out = []
    for i, (shape, dt) in enumerate(meta[0]):
        if rank == 0:
            t = torch.from_numpy(np.ascontiguousarray(arrays[i]).view(np.uint8).reshape(-1)).to(coll_dev)
        else:
            t = torch.empty(int(np.prod(shape)) * np.dtype(dt).itemsize, dtype=torch.uint8, device=coll_dev)
        dist.broadcast(t, src=0)
        out.append(arrays[i] if rank == 0 else t.cpu().numpy().view(np.dtype(dt)).reshape(shape).copy())
    return out


def gather_bytes(dist, torch, coll_dev, world, blob):
    """every rank's `blob` (equal lengths) on every rank: the checker's comparison of the replicas' proofs"""
    t = torch.from_numpy(np.frombuffer(bytes(blob), np.uint8).copy()).to(coll_dev)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return [bytes(x.cpu().numpy().tobytes()) for x in parts]


def bench_prove_replicas(make_ctx, v, comm, instance, log_m, total=60, contexts=3, precompute=True):
    """The other half of BASELINE.json's metric at any N ("prove: replicas"): EVERY rank holds the proving key of the same
    2^log_m-constraint system resident on its own GPU and proves `total` times from one host thread with `contexts` - 1 proofs in flight,
    packed witness; value = all ranks' proofs / the slowest rank's time, between two barriers.  No collective on the data path: a proof
    needs the whole key, and a vote is one proof -- votes are what is spread over the GPUs.
    make_ctx(): a fresh library context on this rank's GPU.  comm: rank, world, barrier(), allmax(x), bcast(arrays), gather(blob).
    instance(): rank 0 only -- (num_constraints, num_inputs, num_vars, (A, B, C) CSR triples, witness, toxic waste, r, s)."""
    rank, world = comm["rank"], comm["world"]
    if rank == 0:
        nc, ni, nv, (A, B, Cm), wit, tox, r, s_ = instance()
        head = np.array([nc, ni, nv], np.uint64)
        arrays = [head, *A, *B, *Cm, wit, tox, r, s_]
    else:
        arrays = None
    arrays = comm["bcast"](arrays)
    nc, ni, nv = (int(x) for x in arrays[0])
    A, B, Cm = tuple(arrays[1:4]), tuple(arrays[4:7]), tuple(arrays[7:10])
    wit, tox, r, s_ = arrays[10:14]
    ring = [make_ctx() for _ in range(contexts)]
    ctx = ring[0]
    dcs = v.R1CS(ctx, nc, ni, nv, A, B, Cm)
    t0 = time.perf_counter()
    kp = v.Keypair(ctx, dcs, tox, precompute=precompute)
    setup_s = time.perf_counter() - t0
    pw = v.PackedWitness(wit)
    ref = v.groth16_prove(ctx, dcs, kp.pk, wit, r, s_)              # the blocking entry point, plain witness: what every pipelined proof must equal
    for c in ring:                                                  # warm every context's workspaces
        v.groth16_prove_launch(c, dcs, kp.pk, pw, r, s_); v.groth16_prove_finish(c)
    prove_ring(v, ring, dcs, kp.pk, pw, r, s_, 2 * contexts)
    comm["barrier"](); t0 = time.perf_counter()
    dt_local, last = prove_ring(v, ring, dcs, kp.pk, pw, r, s_, total)
    comm["barrier"]()
    elapsed = comm["allmax"](time.perf_counter() - t0)
    same_local = all(np.array_equal(a, b) for a, b in zip(last[:3], ref[:3]))
    blobs = comm["gather"](bytes(ref[3]) + bytes([1 if same_local else 0]))
    out = {"proofs_per_s": world * total / elapsed, "ms_per_proof_per_gpu": elapsed / total * 1e3, "n_gpus": world, "proofs_per_gpu": total,
           "contexts_per_gpu": contexts, "constraints": nc, "log_m": log_m, "key_build_s": setup_s, "key_precomputed": bool(precompute),
           "key_bytes_per_gpu": int(kp.device_bytes()),
           "every_rank_same_proof_bytes": bool(all(b == blobs[0] and b[-1] == 1 for b in blobs)),
           "slowest_rank_s": elapsed, "this_rank_s": dt_local,
           "mode": "replicas: every rank its own resident key, ONE host thread per rank, %d proofs in flight (vsp_groth16_prove_launch / _finish), packed witness" % (contexts - 1)}
    proof = ref
    pub = wit[:ni]
    parts = {nm: kp.part(nm) for nm in ("alpha_g1", "beta_g2", "gamma_g2", "delta_g2", "gamma_ABC_g1")} if rank == 0 else None
    for c in ring[1:]:
        c.close()
    kp.free(); dcs.free()
    ctx.close()
    return out, proof, pub, parts


def cpu_prove_leg(ctx, v, cref, o, lg_s, tox, r, s_, ni=30, precompute=True):
    """The reference's prover on the host beside the GPU's, same instance (north_star: the CPU path timed on the same box in the same run):
    the oracle's serial generator + r1cs_gg_ppzksnark prover at 2^lg_s constraints (a BOUNDED sample of config 4: the 2^20 instance would
    be ~16x the 2^16 one), then this library on the very same key and witness, bit for bit."""
    nc_s = (1 << lg_s) - ni - 2
    cs2, wit2 = cref.R1CS.synth(nc_s, ni, 4)
    t0 = time.perf_counter()
    kp = cref.Keypair(cs2, tox)
    keygen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    eA, eB, eC = kp.prove(wit2, r, s_)
    cpu_dt = time.perf_counter() - t0
    q2 = [ctx.upload_bases(kp.part(nm), g) for nm, g in (("A_query", 1), ("B_query_g1", 1), ("B_query_g2", 2), ("H_query", 1), ("L_query", 1))]
    if precompute:
        for q in q2:
            q.precompute(0)
    pk2 = v.ProvingKey(ctx, kp.part("alpha_g1")[0], kp.part("beta_g1")[0], kp.part("beta_g2")[0], kp.part("delta_g1")[0], kp.part("delta_g2")[0], *q2)
    A2, B2, C2 = cs2.export()
    dcs2 = v.R1CS(ctx, nc_s, ni, cs2.num_vars, A2, B2, C2)
    v.groth16_prove(ctx, dcs2, pk2, wit2, r, s_)
    each = []
    for _ in range(5):
        t0 = time.perf_counter()
        gA, gB, gC, _ = v.groth16_prove(ctx, dcs2, pk2, wit2, r, s_)
        each.append(time.perf_counter() - t0)
    gpu_dt = float(np.median(each))
    out = {f"prove_2p{lg_s}_cpu_oracle_s": cpu_dt, f"prove_2p{lg_s}_cpu_oracle_keygen_s": keygen_s, f"prove_2p{lg_s}_gpu_ms": gpu_dt * 1e3,
           f"prove_2p{lg_s}_gpu_over_cpu": cpu_dt / gpu_dt,
           f"prove_2p{lg_s}_bit_exact_vs_cpu": bool(np.array_equal(gA, eA) and np.array_equal(gB, eB) and np.array_equal(gC, eC))}
    pk2.free(); dcs2.free(); [q.free() for q in q2]; kp.free(); cs2.free()
    return out


def issue_costs():
    """cycles of a SIMD per wave-instruction from the committed microbenchmark output (4 waves per SIMD rows), else the fallback constants"""
    out, src = dict(ISSUE_CYCLES_FALLBACK), "bench.py constants"
    try:
        import re
        txt = open(os.path.join(ROOT, "profiles", "r3_ubench_issue.txt")).read()
        def grab(label):
            m = re.search(re.escape(label) + r"\s+4 wave/SIMD.*?with 4 waves: ([0-9.]+) cycles", txt)
            return float(m.group(1)) if m else None
        a, b = grab("v_mad_u64_u32 (VCC, ONE dependent chain, as the routines)"), grab("v_add_u32")
        if a and b:
            out, src = {"mad": a, "simple": b}, "profiles/r3_ubench_issue.txt (tools/ubench_issue.hip, 4 waves per SIMD)"
    except Exception:
        pass
    return out, src


def accum_loop_measured_cycles():
    """the generated accumulation loop on L1-resident / gathered rows, THREE waves per SIMD as the kernel runs (tools/ubench_madd28.hip, committed
    output): cycles of a SIMD per mixed addition of a wave -- a floor the kernel has been SEEN at, so the ceiling of roofline_valu is never above it"""
    import re
    best, src = None, None
    for f in ("r4_ubench_madd28.txt", "r3_ubench_madd28.txt"):
        try:
            for line in open(os.path.join(ROOT, "profiles", f)):
                m = re.match(r"accum28 asm loop.*?3 wave/SIMD:.*?->\s+(\d+) cycles of a SIMD per wave-op", line)
                if m and (best is None or int(m.group(1)) < best):
                    best, src = int(m.group(1)), "profiles/" + f
        except OSError:
            continue
        if best is not None:
            break
    return best, src


def accum_instr_mix():
    """the instruction mix of ONE iteration (one mixed addition) of the generated G1 accumulation loop, counted from the generator's own
    instruction list (tools/gen_accum28_asm.py): multiply-adds, other VALU, SALU, VMEM"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_accum28_asm", os.path.join(ROOT, "tools", "gen_accum28_asm.py"))
    g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
    ins = g.gen_body()
    a, b = ins.index(".Lvsp_acc28_loop:"), max(i for i, x in enumerate(ins) if x.startswith("s_cbranch_scc1 .Lvsp_acc28_loop"))
    body = [x for x in ins[a:b + 1] if not x.endswith(":")]
    init = ins.index(".Lvsp_acc28_noinit:")                     # the first-point branch is taken once per part, not per addition
    skip = set(range(max(i for i, x in enumerate(ins[:init]) if x.startswith("s_cbranch_scc0 .Lvsp_acc28_noinit")) + 1, init))
    body = [x for i, x in enumerate(ins[a:b + 1], a) if not x.endswith(":") and i not in skip]
    mix = {"v_mad_u64_u32": sum(x.startswith("v_mad_u64_u32") for x in body), "valu_other": sum(x.startswith("v_") and not x.startswith("v_mad_u64_u32") for x in body),
           "salu": sum(x.startswith("s_") for x in body), "vmem": sum(x.startswith("global_") for x in body)}
    return mix


def diag_clock_ghz(dev_index, d_bases_canon, n, d_scalars_ptr, seconds=2.0):
    """The clock the chip holds inside the G1 accumulation loop (MI355X_MICROARCH.md, DVFS item 6): the DIAGNOSTIC build of the same
    sources (libvsp_hip_diag.so: s_memtime / s_memrealtime stamps around the loop; the shipped library executes no stamp) runs the
    headline multi-exponentiation back to back for `seconds` on the same bases and scalars.  None when that build is absent."""
    import ctypes as C
    from vote_saver_protocol_amd import _lib
    path = os.path.join(os.path.dirname(_lib.SO_PATH), "libvsp_hip_diag.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    for name in ("vsp_create", "vsp_destroy", "vsp_bases_from_device_g1", "vsp_bases_free", "vsp_msm_resident", "vsp_diag_clock", "vsp_last_error"):
        fn = getattr(lib, name); fn.restype, fn.argtypes = _lib.PROTOTYPES[name]
    ctx = lib.vsp_create(dev_index)
    if not ctx:
        return None
    try:
        b = lib.vsp_bases_from_device_g1(ctx, C.c_void_p(d_bases_canon), n)
        if not b:
            return None
        out = np.zeros(12, np.uint64); inf = C.c_int(0)
        run = lambda: lib.vsp_msm_resident(ctx, b, 0, n, C.c_void_p(d_scalars_ptr), out.ctypes.data_as(C.c_void_p), C.byref(inf))
        for _ in range(3):
            run()
        ghz, waves = C.c_double(0), C.c_double(0)
        lib.vsp_diag_clock(ctx, 1, C.byref(ghz), C.byref(waves))          # reset after the warm-up
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds:
            if run() != 0:
                return None
            k += 1
        rc = lib.vsp_diag_clock(ctx, 0, C.byref(ghz), C.byref(waves))
        lib.vsp_bases_free(ctx, b)
        info = {"ghz": ghz.value, "waves_stamped": waves.value, "msms": k, "seconds": time.perf_counter() - t0} if rc == 0 and ghz.value > 0 else None
        # the same for the passes of the 2^22 transform (k_ntt29_pass stamps a whole pass per wave): 1 s of back-to-back forward transforms
        try:
            for name in ("vsp_dmalloc", "vsp_dfree", "vsp_ntt_fr_device", "vsp_diag_clock_ntt"):
                fn = getattr(lib, name); fn.restype, fn.argtypes = _lib.PROTOTYPES[name]
            lg = 22
            d_a = lib.vsp_dmalloc(ctx, (1 << lg) * 32)
            if d_a and info is not None:
                for _ in range(2):
                    lib.vsp_ntt_fr_device(ctx, C.c_void_p(d_a), lg, 0, None)
                g2, w2 = C.c_double(0), C.c_double(0)
                lib.vsp_diag_clock_ntt(ctx, 1, C.byref(g2), C.byref(w2))
                t1, k2 = time.perf_counter(), 0
                while time.perf_counter() - t1 < seconds / 2:
                    lib.vsp_ntt_fr_device(ctx, C.c_void_p(d_a), lg, 0, None); k2 += 1
                if lib.vsp_diag_clock_ntt(ctx, 0, C.byref(g2), C.byref(w2)) == 0 and g2.value > 0:
                    info["ntt_ghz"] = g2.value; info["ntt_waves_stamped"] = w2.value; info["ntt_transforms"] = k2
                lib.vsp_dfree(ctx, C.c_void_p(d_a))
        except Exception as e:                                   # a diagnostic: never take the line down
            print("bench.py: NTT clock leg failed: %r" % (e,), file=sys.stderr)
        return info
    finally:
        lib.vsp_destroy(ctx)


def host_cores():
    """CPU cores this process may really use: the cgroup quota where one is set (a GPU box gives a share of its host), else the affinity mask"""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return min(n, 64)


def shard_bounds(total, world, rank):
    """contiguous point chunk of rank `rank`: [lo, hi) -- the package's rule (vote_saver_protocol_amd/sharded.py)"""
    from vote_saver_protocol_amd.sharded import shard_bounds as sb
    return sb(total, world, rank)


def visible_gpu_count():
    """GPUs this process could use, WITHOUT initialising the HIP runtime (the launcher must not touch the GPU before it starts its
    ranks): the KFD topology nodes that have SIMDs, cut by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None when the topology is
    not readable (the ranks then check for themselves)."""
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return 0 if not os.path.exists("/dev/kfd") else None
    n = 0
    for f in nodes:
        try:
            for line in open(f):
                k, _, val = line.partition(" ")
                if k == "simd_count" and int(val) > 0:
                    n += 1
        except OSError:
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        if os.environ.get(var, "") != "":
            n = min(n, len([x for x in os.environ[var].split(",") if x.strip() != ""]))
    return n


def spawn_ranks(args, argv):
    """`bench.py --gpus N` started as a plain process: run the N ranks as a child `torch.distributed.run` (one process per GPU, RCCL
    rendezvous on 127.0.0.1) and return its exit code.  This process has not imported torch and never touches HIP: nothing that has
    initialised the GPU is ever replaced or forked.  Rank 0 of the child writes the JSON line to the stdout it inherits from here."""
    have = visible_gpu_count()
    if have is not None and have < args.gpus and not args.allow_shared_gpu:
        print(f"bench.py: --gpus {args.gpus} but {have} GPU(s) are visible on this machine; nothing was launched "
              "(--allow-shared-gpu --backend gloo rehearses several ranks on one GPU)", file=sys.stderr)
        return 3
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def dot_mod_r_device(torch, k64, s4):
    """sum_i k_i * s_i mod r for 64-bit k (int64 bit patterns, [n]) and 256-bit s ([n,4] int64 limbs), on the device: both are cut in
    16-bit pieces, the 64 piece-by-piece sums stay below 2^58 for n <= 2^26, python ints put them together.  Checker arithmetic."""
    kp = [(k64 >> (16 * i)) & 0xFFFF for i in range(4)]
    total = 0
    for j in range(16):
        sj = (s4[:, j // 4] >> (16 * (j % 4))) & 0xFFFF
        for i in range(4):
            total += int((kp[i] * sj).sum().item()) << (16 * (i + j))
    return total % R_MOD


class ShardProblem:
    """Synthetic input of one MSM problem of `total` points of group 1 / 2 split by contiguous chunk over the ranks; every rank builds
    and keeps only its chunk (bases k_i * generator with 64-bit k_i -- full-size curve points; the small k only makes the checker's sum
    cheap -- and uniform 254-bit scalars, both generated on the device).  The multi-exponentiation over it is the package's
    vote_saver_protocol_amd.sharded.ShardedMsm."""

    def __init__(self, ctx, v, torch, dev, group, total, world, rank, seed, sample=0):
        self.ctx, self.v, self.group, self.total, self.world = ctx, v, group, total, world
        lo, hi = shard_bounds(total, world, rank)
        self.n = n = hi - lo
        g = torch.Generator(device=dev); g.manual_seed(seed * 1000003 + rank)
        k64 = torch.randint(-(1 << 63), (1 << 63) - 1, (n,), dtype=torch.int64, device=dev, generator=g)
        self.d_s = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=dev, generator=g)
        self.d_s[:, 3] &= 0x3FFFFFFFFFFFFFFF                       # < 2^254 < r: canonical
        k4 = torch.zeros((n, 4), dtype=torch.int64, device=dev); k4[:, 0] = k64
        torch.cuda.synchronize()
        d_b = v.fixed_base_mul(ctx, k4, n, group)
        del k4
        self.bases = ctx.bases_from_device(d_b, n, group)
        self.sample = None
        if rank == 0 and sample:                                   # a bounded host copy of the first points: the CPU baseline's input
            m = min(n, sample)
            hb = np.zeros((m, 12 * group), np.uint64); ctx.d2h(hb, d_b)
            self.sample = (hb, self.d_s[:m].cpu().numpy().view(np.uint64).copy())
        ctx.dfree(d_b)
        self.e_local = dot_mod_r_device(torch, k64, self.d_s)
        del k64

    def free(self):
        self.bases.free(); self.d_s = None


def run_sharded(problem, steps, warmup, exchange, barrier, depth):
    """`steps` MSMs over the rank's chunk + exchange, `depth` in flight (vote_saver_protocol_amd.sharded.ShardedMsm.run: the exchange
    of step k completes while step k + 1's multi-exponentiation is awaited); returns (seconds for `steps`, last folded result)"""
    from vote_saver_protocol_amd.sharded import ShardedMsm
    job = ShardedMsm(problem.bases, exchange)
    if depth > 1:
        # set-up, not a step: the first multi-exponentiation on a work slot allocates that slot's workspaces, pinned buffers and stream
        # (15-20 ms), and W warm-up steps touch only min(W, depth - 1) + ... of the `depth` slots (W = 3, depth = 4: the fourth slot's first
        # use fell inside the timed region -- 5.06 instead of 3.16 ms per step over 10 steps through the nccl path)
        job.run(problem.d_s, depth, depth)
    if warmup:
        job.run(problem.d_s, warmup, depth)
    barrier(); t0 = time.perf_counter()
    res = job.run(problem.d_s, steps, depth)
    barrier()
    return time.perf_counter() - t0, res


def replica_leg(args, v, torch, dist, coll_dev, rank, world, dev_index, barrier, allmax, cref, o):
    """bench_prove_replicas wired to torch.distributed; rank 0 adds the pairing check of the proof and the CPU prover's time on a bounded sample"""
    comm = {"rank": rank, "world": world, "barrier": barrier, "allmax": allmax,
            "bcast": lambda arrays: bcast_arrays(dist, torch, coll_dev, rank, arrays),
            "gather": lambda blob: gather_bytes(dist, torch, coll_dev, world, blob)}
    ni = 30
    nc = (1 << args.prove_log_n) - ni - 2

    def instance():
        gen = o.splitmix64(5)
        cs, wit = cref.R1CS.synth(nc, ni, 4, ballot=(25, 7))
        tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
        r = limbs(o.rand_fr(gen), 4); s_ = limbs(o.rand_fr(gen), 4)
        trip = cs.export(); nv = cs.num_vars; cs.free()
        return nc, ni, nv, trip, wit, tox, r, s_

    res, proof, pub, parts = bench_prove_replicas(lambda: v.Context(dev_index), v, comm, instance, args.prove_log_n)
    if rank == 0:
        try:
            import pairing as pg
            vk = dict(alpha_g1=o.g1_from_limbs(parts["alpha_g1"][0]), beta_g2=o.g2_from_limbs(parts["beta_g2"][0]), gamma_g2=o.g2_from_limbs(parts["gamma_g2"][0]),
                      delta_g2=o.g2_from_limbs(parts["delta_g2"][0]), gamma_ABC_g1=[o.g1_from_limbs(x) for x in parts["gamma_ABC_g1"]])
            res["pairing_verified"] = bool(pg.groth16_verify(vk, [int(x) for x in to_ints(pub).tolist()],
                                                             (o.g1_from_limbs(proof[0]), o.g2_from_limbs(proof[1]), o.g1_from_limbs(proof[2]))))
        except Exception as e:                             # the checker must not take the line down
            res["pairing_error"] = repr(e)
        if not args.no_cpu_baseline:
            # the reference's prover on this box's host, bounded sample (2^16 constraints, the real circuit's likely size), and this library on the same instance
            try:
                c0 = v.Context(dev_index)
                gen = o.splitmix64(5)
                tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
                r = limbs(o.rand_fr(gen), 4); s_ = limbs(o.rand_fr(gen), 4)
                leg = cpu_prove_leg(c0, v, cref, o, 16, tox, r, s_)
                c0.close()
                res["cpu_baseline"] = {"value": 1.0 / leg["prove_2p16_cpu_oracle_s"], "unit": "proofs/s", "cores": 1, "kind": "port",
                                       "sample": "one proof of a 2^16-constraint instance by the oracle's serial r1cs_gg_ppzksnark prover (oracle/vsp_ref.c), %.2f s; "
                                                 "the same instance on one GPU: %.2f ms, bit-exact: %s" % (leg["prove_2p16_cpu_oracle_s"], leg["prove_2p16_gpu_ms"], leg["prove_2p16_bit_exact_vs_cpu"]),
                                       **leg}
            except Exception as e:
                res["cpu_baseline_error"] = repr(e)
    barrier()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20, help="weak scaling: log2 of the points per GPU (BASELINE config 2: 20)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="strong: ONE problem of 2^total-log-n points of --group split over the ranks (BASELINE config 5)")
    ap.add_argument("--total-log-n", type=int, default=0, help="strong scaling: log2 of the whole problem (default 26 for g1, 24 for g2)")
    ap.add_argument("--group", choices=("g1", "g2"), default="g1")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 leg (2^26 G1 + 2^24 G2 over the ranks) of the default run")
    ap.add_argument("--config5-log-g1", type=int, default=26)
    ap.add_argument("--config5-log-g2", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary NTT / G2 / resident-key measurements")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--pipeline-depth", type=int, default=4, help="MSMs in flight (work slots with their own streams)")
    ap.add_argument("--hw-queues", type=int, default=8, help="GPU_MAX_HW_QUEUES for this process unless already in the environment; 0 = leave the runtime's default (4)")
    ap.add_argument("--no-pipeline", action="store_true", help="blocking MSM calls (one in flight): for clean per-kernel profiles")
    ap.add_argument("--precompute", action="store_true", help="headline over a resident key with window multiples (16x table) instead of plain bases")
    ap.add_argument("--precompute-split", action="store_true", help="with --precompute: the table carries the endomorphism rows (vsp_bases_precompute_split)")
    ap.add_argument("--prove-h-first", type=int, default=-1, help="prover queue order (library option prove_h_first); -1 = library default")
    ap.add_argument("--no-prove", action="store_true", help="skip the secondary full-prover measurement (config 4)")
    ap.add_argument("--prove-log-n", type=int, default=20, help="log2 of the synthetic R1CS domain for the prover measurement")
    ap.add_argument("--no-diag-clock", action="store_true", help="skip the 2 s in-kernel clock leg (diagnostic build of the library)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="collective backend of the exchange step: nccl = RCCL over xGMI (device buffers); gloo = host buffers (rehearsals)")
    ap.add_argument("--force-replicas", action="store_true",
                    help="rehearsal: run the N > 1 replica-proving leg at N = 1 too (under torch.distributed.run: its broadcasts and gathers then go through RCCL with one rank)")
    ap.add_argument("--allow-shared-gpu", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: rank g uses GPU g mod (GPUs present); forces --backend gloo (RCCL refuses two ranks on one GPU)")
    args = ap.parse_args()
    if args.allow_shared_gpu:
        args.backend = "gloo"
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    # N > 1 started as a plain process (no RANK in the environment): start the ranks as a child process BEFORE anything touches the GPU
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    # stdout carries exactly one JSON line: whatever libraries print there (RCCL's version banner, for one) is sent to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # The HIP runtime maps every stream of the process onto GPU_MAX_HW_QUEUES hardware queues (default 4) per priority.  Results never
    # depend on it, and a single proof's latency does not either (the prover's two chains get their queues first, capi.hip vsp_create);
    # multi-exponentiations IN FLIGHT TOGETHER overlap better with eight: 2.86-2.91 ms per 2^20-point step against 3.03-3.11 with four
    # (round 3, depth 3 / 4, same box; slot streams spread over the priority pools did not change the four-queue figure, so it is not two
    # slots sharing a queue).  A deployment knob like NCCL_*: set here unless the caller set a value or passed --hw-queues 0, reported in
    # the JSON line (config.runtime_env), documented in include/vsp.h.  It has to be in the environment before the runtime initialises,
    # i.e. before torch is imported.
    if os.environ.get("VSP_BENCH_HW_QUEUES"):                 # older spelling of --hw-queues
        os.environ["GPU_MAX_HW_QUEUES"] = os.environ["VSP_BENCH_HW_QUEUES"]
    elif args.hw_queues > 0:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(args.hw_queues))
    import torch
    import torch.distributed as dist
    import vote_saver_protocol_amd as v

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    n_dev = torch.cuda.device_count()                    # counting devices does not initialise the runtime
    if n_dev < 1 or (n_dev < world and not args.allow_shared_gpu):
        if rank == 0:
            print(f"bench.py: {world} rank(s) need {world} GPU(s), {n_dev} visible (this path has no CPU fallback)", file=sys.stderr)
        sys.exit(3)
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = "RANK" in os.environ                     # launched by torch.distributed.run (also with one rank: same code path)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")      # where the bookkeeping collectives (max of timings, checker sums) live

    ctx = v.Context(dev_index)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    if args.window_bits:
        ctx.set_option("msm_window_bits", args.window_bits)
    if os.environ.get("VSP_MSM_SPLIT"):
        ctx.set_option("msm_split", int(os.environ["VSP_MSM_SPLIT"]))      # experiment knob: points per bucket part
    if os.environ.get("VSP_MSM_GLV"):
        ctx.set_option("msm_glv", int(os.environ["VSP_MSM_GLV"]))          # experiment knob: 2 forces the endomorphism split at any size
    for kv in [x for x in os.environ.get("VSP_OPTS", "").split(",") if x]:      # experiment knob: VSP_OPTS="name=value,name=value"
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if args.prove_h_first >= 0:
        ctx.set_option("prove_h_first", args.prove_h_first)
    depth = 1 if args.no_pipeline else args.pipeline_depth

    # exchange step (vote_saver_protocol_amd/sharded.py): one Jacobian record per rank (144 B G1 / 288 B G2), all-gathered over RCCL from
    # device buffers on a stream of its own, folded locally.  Two buffers per group: the all-gather of step k is in flight while the host
    # waits for the multi-exponentiation of step k + 1, so its latency -- a small kernel that has to find a wave slot on a GPU full of
    # accumulation waves, ~0.2 ms -- stays off the step time
    exchange = v.TorchExchange(ctx, dev) if use_dist else v.LocalExchange(ctx)
    ranks_seen = exchange.ranks_seen()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_e(e_local):
        if not use_dist:
            return e_local % R_MOD
        e_dev = torch.from_numpy(limbs(e_local, 4).view(np.int64)).to(coll_dev)
        parts = [torch.zeros(4, dtype=torch.int64, device=coll_dev) for _ in range(world)]
        dist.all_gather(parts, e_dev)
        rows = torch.stack(parts).cpu().numpy().view(np.uint64).reshape(world, 4)
        return sum(int(x) for x in to_ints(rows).tolist()) % R_MOD

    cref = o = gens = None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import cref                                          # checker / CPU baseline only
        import bls12_381 as o
        gens = {1: np.array(o.g1_to_limbs(o.G1.gen), dtype=np.uint64), 2: np.array(o.g2_to_limbs(o.G2.gen), dtype=np.uint64)}

    def expected_point(group, e_tot):                        # rank 0: (sum_i k_i s_i) * generator by the oracle's scalar multiplication
        return (cref.g1_mul if group == 1 else cref.g2_mul)(gens[group], limbs(e_tot, 4))

    def cpu_baseline_legs(group, host_b, host_s, gpu_result=None):
        """rank 0, any N: the reference algorithm on the host cores of this box -- the oracle's serial BDLO12 (1 thread, as the reference
        builds), then the same work cut in `cores` chunks (multiexp's chunks argument), one thread per core.  Bounded sample."""
        from concurrent.futures import ThreadPoolExecutor
        m = host_b.shape[0]
        fn = cref.msm_g1 if group == 1 else cref.msm_g2
        G = o.G1 if group == 1 else o.G2
        to_l, from_l = (o.g1_to_limbs, o.g1_from_limbs) if group == 1 else (o.g2_to_limbs, o.g2_from_limbs)
        name = "G1" if group == 1 else "G2"
        tc = time.perf_counter()
        ref = fn(host_b, host_s)
        dt = time.perf_counter() - tc
        one = {"value": m / dt, "unit": "points/s", "cores": 1, "kind": "port",
               "sample": f"one {m}-point {name} MSM (serial BDLO12 restatement, oracle/vsp_ref.c), {dt:.1f} s",
               "matches_gpu_result": bool(np.array_equal(ref, gpu_result)) if gpu_result is not None else None}
        cores = host_cores()
        cuts = [shard_bounds(m, cores, i) for i in range(cores)]
        tc = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:                 # the C call releases the GIL: one chunk per core
            parts = list(ex.map(lambda ab: fn(host_b[ab[0]:ab[1]], host_s[ab[0]:ab[1]]), cuts))
        acc = None
        for q in parts:
            acc = q if acc is None else np.array(to_l(G.add(from_l(acc), from_l(q))), dtype=np.uint64)
        dta = time.perf_counter() - tc
        allc = {"value": m / dta, "unit": "points/s", "cores": cores, "kind": "port",
                "sample": f"the same {m}-point {name} MSM in {cores} chunks, one thread per host core, partial sums added ({dta:.2f} s)",
                "matches_serial_result": bool(np.array_equal(acc, ref))}
        return one, allc

    def accum_stats():
        ms, launches = ctx.stat("msm_accum_ms"), ctx.stat("msm_accum_launches")
        return (ms / launches) if launches else float("nan")

    def roofline_of(group, n_points, windows, excl_ms, excl_step_ms, pipe_ms, split):
        """HBM line (the contract's) and the v_mad_u64_u32 issue line (the bound that applies) for one accumulation launch;
        split: the endomorphism split was on (2n half-length scalars over `windows` windows)"""
        bytes_alg = n_points * BYTES_PER_PAIR[group]
        achieved = bytes_alg / (excl_ms * 1e-3) / 1e9
        traffic, src = None, None
        if group == 1 and n_points == 1 << 20:
            key = "k_accum_G1_2p20_precomputed" if args.precompute else "k_accum_G1_2p20_plain"
            for f in PMC_FILES:
                try:
                    with open(os.path.join(ROOT, "profiles", f)) as fh:
                        j = json.load(fh)
                    use = key if key in j else "k_accum_G1_2p20"
                    traffic = j[use]["traffic_bytes_per_launch"]; src = "profiles/" + f + ":" + use
                    break
                except Exception:
                    continue
        mads = n_points * (2 if split else 1) * windows * MADS_PER_ADD[group]
        rl = {"bound": "hbm", "kernel": "k_accum28 (bucket accumulation on 14 x 28-bit limbs)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
              "traffic_over_algorithmic": (traffic / bytes_alg) if traffic else None,
              "algorithmic_bytes_per_launch": bytes_alg, "avg_launch_ms": excl_ms, "ms_per_step_same_mode": excl_step_ms,
              "avg_launch_ms_pipelined": pipe_ms,
              "note": "avg_launch_ms: HIP events around the kernel with ONE multi-exponentiation in flight (nothing else on the GPU), so it is <= "
                      "ms_per_step_same_mode; avg_launch_ms_pipelined is the same bracket inside the timed region, where neighbouring steps' kernels share "
                      "the GPU.  The kernel is integer-issue bound (about 160 Montgomery products per point), not HBM bound: see roofline_valu.  traffic is "
                      "not measurable in-process (PMC passes need rocprofv3): it is read from traffic_source"}
        # the bound that applies: VALU issue.  One iteration of the generated loop = one mixed addition of a whole wave; its instruction mix is
        # counted from the generator's own list, priced with the SIMD's measured issue cost per instruction class (shader cycles, >= 3 waves
        # per SIMD) -- that is the peak: cycles a SIMD cannot go below for this instruction stream.  The kernel's cycles come from its
        # duration and the clock the chip HELD inside the loop (diagnostic build, same run), so the fraction cannot exceed 1 by a clock or
        # a denominator artefact.  G2 (lane pairs, compiler-allocated kernel): the multiply-adds alone at the same issue cost.
        adds = n_points * (2 if split else 1) * windows
        costs, cost_src = issue_costs()
        wave_adds_per_simd = adds * (2 if group == 2 else 1) / 64.0 / N_SIMD          # G2: two lanes per point
        if group == 1:
            mix = accum_instr_mix()
            model_cycles = mix["v_mad_u64_u32"] * costs["mad"] + mix["valu_other"] * costs["simple"]
            seen_cycles, seen_src = accum_loop_measured_cycles()
            ideal_cycles = min(model_cycles, seen_cycles) if seen_cycles else model_cycles      # a ceiling the kernel cannot beat: the lower of model and measurement
        else:
            mix = {"v_mad_u64_u32": MADS_PER_ADD[2] // 2, "valu_other": None, "note": "per lane of a pair; the rest of the compiler-allocated loop is not priced (a looser peak)"}
            ideal_cycles = model_cycles = mix["v_mad_u64_u32"] * costs["mad"]
            seen_cycles, seen_src = None, None
        clk = clock_info["ghz"] if clock_info else None
        kernel_cycles = (excl_ms * 1e-3 * clk * 1e9 / wave_adds_per_simd) if clk else None
        rv = {"bound": "VALU issue (one wave-instruction per SIMD every 4.6 cycles for v_mad_u64_u32, 2.9 for simple 32-bit ops)", "kernel": rl["kernel"],
              "unit": "shader cycles of a SIMD per mixed addition of a wave", "peak": ideal_cycles, "achieved": kernel_cycles,
              "peak_instruction_mix_model": model_cycles, "peak_measured_loop_three_waves": seen_cycles, "peak_measured_source": seen_src,
              "frac": (ideal_cycles / kernel_cycles) if kernel_cycles else None,
              "instr_mix": mix, "issue_cycles_per_wave_instruction": costs, "issue_cost_source": cost_src,
              "clock_ghz_in_kernel": clk, "clock_source": ("libvsp_hip_diag.so: delta s_memtime / delta s_memrealtime x 100 MHz around the loop, summed over %d waves of %d "
                                                           "back-to-back multi-exponentiations in this run" % (clock_info["waves_stamped"], clock_info["msms"])) if clock_info else
                                                          "diagnostic build absent: no in-kernel clock, no fraction",
              "mixed_additions_per_launch": adds, "wave_additions_per_simd": wave_adds_per_simd, "endomorphism_split": bool(split),
              "mads_per_mixed_addition": MADS_PER_ADD[group],
              "sq_counters": "profiles/r3_sq_counters.json (rocprofv3 --pmc, tools/gpu_sq.sh): SQ_INSTS_VALU per launch / 1024 SIMDs x cycles = the same figure from the hardware's own count"}
        return rl, rv

    extras = {}
    cpu_baseline = cpu_all = None
    verified = None
    clock_info = None                                        # in-kernel clock of the G1 accumulation (diagnostic build), headline run only

    # =============================================================== strong scaling: ONE problem split over the ranks (BASELINE config 5)
    if args.scaling == "strong":
        group = 1 if args.group == "g1" else 2
        lg_total = args.total_log_n or (26 if group == 1 else 24)
        total = 1 << lg_total
        prob = ShardProblem(ctx, v, torch, dev, group, total, world, rank, seed=77 + group,
                            sample=0 if args.no_cpu_baseline else (1 << 20 if group == 1 else 1 << 17))
        ctx.stats_reset()
        elapsed, result = run_sharded(prob, args.steps, max(args.warmup, depth), exchange, barrier, depth)      # one warm-up per work slot: their multi-GB workspaces are allocated on first use
        elapsed = allmax(elapsed)
        pipe_ms = accum_stats()
        main_c, main_w, main_split = int(ctx.stat("msm_window_bits")), int(ctx.stat("msm_windows")), int(ctx.stat("msm_endomorphism_split"))
        ctx.stats_reset()
        ex_steps = max(2, args.steps // 4)
        ex_elapsed, _ = run_sharded(prob, ex_steps, 1, exchange, barrier, 1)
        ex_elapsed = allmax(ex_elapsed)
        excl_ms = accum_stats()
        e_tot = gather_e(prob.e_local)
        if rank == 0:
            verified = bool(np.array_equal(result, expected_point(group, e_tot)))
            if prob.sample is not None:                      # every N: the CPU path timed on this box's host cores in the same run
                cpu_baseline, cpu_all = cpu_baseline_legs(group, prob.sample[0], prob.sample[1])
        barrier()
        rl, rv = roofline_of(group, prob.n, main_w, excl_ms, ex_elapsed / ex_steps * 1e3, pipe_ms, main_split)
        out = {"metric": f"{args.group.upper()} MSM points/sec, one 2^{lg_total}-point problem sharded over the GPUs (BASELINE config 5)",
               "value": total * args.steps / elapsed, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
               "config": {"workload": f"one 2^{lg_total}-point BLS12-381 {args.group.upper()} Pippenger MSM, plain bases k_i*G (64-bit k_i), uniform 254-bit scalars, "
                                      f"resident in HBM; rank g holds the contiguous chunk g of {world}; one all-gather ({exchange.backend}) of {144 * group}-byte Jacobian records + local fold per MSM",
                          "total_points": total, "points_per_gpu": prob.n, "window_bits": main_c, "windows": main_w, "msms_in_flight": depth,
                          "exchange_backend": exchange.backend, "ranks_seen_by_collective": ranks_seen, "gpus_visible": n_dev,
                          "runtime_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}},
               "verified_bit_exact": verified, "latency_ms_one_in_flight": ex_elapsed / ex_steps * 1e3, "roofline": rl, "roofline_valu": rv,
               "cpu_baseline": cpu_baseline, "cpu_baseline_all_cores": cpu_all}
        if rank == 0:
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        prob.free(); ctx.close()
        if use_dist:
            dist.destroy_process_group()
        return

    # =============================================================== headline (weak): 2^log_n G1 points per GPU, plain bases
    n = 1 << args.log_n
    # synthetic inputs (SURVEY.md 8(d) cfg 2): bases k_i * G built on the GPU, uniform scalars; all resident
    ks = rand_fr(n, seed=1000 + rank)
    ss = rand_fr(n, seed=2000 + rank)
    d_k = torch.from_numpy(ks.view(np.int64)).to(dev)
    d_s = torch.from_numpy(ss.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    d_bases_canon = v.fixed_base_mul(ctx, d_k, n, 1)
    del d_k

    class Head:                                              # the run_sharded interface over the headline inputs
        group = 1
    head = Head(); head.d_s = d_s; head.n = n
    head.bases = ctx.bases_from_device(d_bases_canon, n, 1)
    precompute_s = None
    if args.precompute:
        t_pre = time.perf_counter(); head.bases.precompute(int(os.environ.get("VSP_BENCH_PRE_C", "16")), split=args.precompute_split); precompute_s = time.perf_counter() - t_pre

    ctx.stats_reset()
    elapsed, result = run_sharded(head, args.steps, args.warmup, exchange, barrier, depth)
    elapsed = allmax(elapsed)
    pipe_ms = accum_stats()
    main_c, main_w, main_split = int(ctx.stat("msm_window_bits")), int(ctx.stat("msm_windows")), int(ctx.stat("msm_endomorphism_split"))
    # the same MSM with ONE in flight: latency of a single multi-exponentiation and the exclusive duration of its accumulation kernel
    ctx.stats_reset()
    ex_steps = max(4, args.steps // 2)
    ex_elapsed, _ = run_sharded(head, ex_steps, 1, exchange, barrier, 1)
    ex_elapsed = allmax(ex_elapsed)
    excl_ms = accum_stats()

    if rank == 0 and not args.no_diag_clock and not args.precompute:
        try:
            clock_info = diag_clock_ghz(dev_index, d_bases_canon, n, d_s.data_ptr())
        except Exception as e:                               # a diagnostic: never take the bench line down
            print("bench.py: diagnostic clock leg failed: %r" % (e,), file=sys.stderr)
    # ---- untimed verification: sum over all ranks of (sum_i k_i s_i) * G must equal the folded result
    e_tot = gather_e(int(sum((to_ints(ks) * to_ints(ss)).tolist()) % R_MOD))
    if rank == 0:
        verified = bool(np.array_equal(result, expected_point(1, e_tot)))

        if not args.no_cpu_baseline:
            # every N (north_star: throughput at 1/2/4/8 GPUs next to the CPU path timed on the same box in the same run): the reference
            # algorithm on the host, rank 0's own 2^log_n bases and scalars -- serial BDLO12 (c = 16 at 2^20), 1 thread, then all cores
            host_b = np.zeros((n, 12), np.uint64)
            ctx.d2h(host_b, d_bases_canon)
            m = min(n, 1 << 20)
            own = None
            if world == 1 and m == n:
                own = result
            elif m == n:                                      # N > 1: rank 0's own partial sum, through the blocking single-GPU entry point
                own, _ = head.bases.msm(d_s)
            cpu_baseline, cpu_all = cpu_baseline_legs(1, host_b[:m], ss[:m], own)
            del host_b

        if world == 1 and not args.no_extras and not args.precompute:
            # a resident proving key may keep the window multiples 2^(16 w) * P (16x the memory, built once): every MSM over it shares one bucket set
            pre = Head(); pre.d_s = d_s; pre.n = n
            pre.bases = ctx.bases_from_device(d_bases_canon, n, 1)
            t_pre = time.perf_counter(); pre.bases.precompute(16); pre_s = time.perf_counter() - t_pre
            ctx.stats_reset()
            k2 = max(4, args.steps // 2)
            el2, res2 = run_sharded(pre, k2, 2, exchange, barrier, depth)
            extras["resident_key_window_multiples"] = {"ms_per_step": el2 / k2 * 1e3, "points_per_s": n * k2 / el2, "table_memory_factor": 16, "build_once_s": pre_s,
                                                       "k_accum28_avg_ms_pipelined": accum_stats(), "same_result": bool(np.array_equal(res2, result))}
            pre.bases.free()
            # the same key for DENSE scalars: window multiples WITH their endomorphism images (vsp_bases_precompute_split), 19-bit windows:
            # 7 windows of a split scalar over one bucket set -- 12.5 % fewer additions than the headline's 8, 14 rows of 128 bytes per point
            pre.bases = ctx.bases_from_device(d_bases_canon, n, 1)
            t_pre = time.perf_counter(); pre.bases.precompute(19, split=True); pre_s = time.perf_counter() - t_pre
            ctx.stats_reset()
            el3, res3 = run_sharded(pre, k2, 2, exchange, barrier, depth)
            extras["resident_key_window_multiples_split_19bit"] = {"ms_per_step": el3 / k2 * 1e3, "points_per_s": n * k2 / el3, "rows_per_point": 14, "build_once_s": pre_s,
                                                                   "windows": int(ctx.stat("msm_windows")), "k_accum28_avg_ms_pipelined": accum_stats(),
                                                                   "same_result": bool(np.array_equal(res3, result))}
            pre.bases.free()

        if world == 1 and not args.no_extras:
            # secondary numbers (not the headline): BASELINE config 3 (2^22 NTT) and a 2^18 G2 MSM
            lg = 22
            a = torch.from_numpy(rand_fr(1 << lg, 7).view(np.int64)).to(dev)
            dom = v.EvaluationDomain(ctx, 1 << lg)
            dom.fft_device(a); ctx.synchronize()
            reps = 10
            tn = time.perf_counter()
            for _ in range(reps):
                dom.fft_device(a)
            ctx.synchronize()
            dtn = (time.perf_counter() - tn) / reps
            extras["ntt_2p22_ms"] = dtn * 1e3
            extras["ntt_2p22_elements_per_s"] = (1 << lg) / dtn
            tn = time.perf_counter()
            for _ in range(reps):
                dom.fft_device(a, inverse=True)
            ctx.synchronize()
            extras["ntt_2p22_inverse_ms"] = (time.perf_counter() - tn) / reps * 1e3
            ntt_clk = (clock_info or {}).get("ntt_ghz") or (clock_info["ghz"] if clock_info else 2.1)
            ntt_clk_src = ("libvsp_hip_diag.so: s_memtime / s_memrealtime around every pass of k_ntt29_pass, %d waves of %d transforms in this run" % (clock_info["ntt_waves_stamped"], clock_info["ntt_transforms"])
                           if (clock_info or {}).get("ntt_ghz") else "the accumulation loop's clock of this run (the transform's own stamp is absent)")
            if not args.no_cpu_baseline:
                # BASELINE config 3 beside its CPU figure: the oracle's serial radix-2 transform (libfqfft-lineage basic_radix2_domain, oracle/vsp_ref.c) on the same input
                a_host = rand_fr(1 << lg, 7)
                tc = time.perf_counter(); ref_fft = cref.ntt_fr(a_host); cpu_ntt_s = time.perf_counter() - tc
                d_chk = torch.from_numpy(a_host.view(np.int64)).to(dev)
                dom.fft_device(d_chk); ctx.synchronize()
                extras["ntt_2p22_cpu_s"] = cpu_ntt_s
                extras["ntt_2p22_cpu_baseline"] = {"value": (1 << lg) / cpu_ntt_s, "unit": "elements/s", "cores": 1, "kind": "port",
                                                   "sample": "one 2^22 forward transform by the oracle's serial radix-2 FFT, %.2f s" % cpu_ntt_s,
                                                   "matches_gpu_result": bool(np.array_equal(d_chk.cpu().numpy().view(np.uint64).reshape(-1, 4), ref_fft))}
                del a_host, ref_fft, d_chk
            # algorithmic bytes = 64 B per element for the whole transform (one ideal read + write); the kernel makes
            # `passes` round trips through HBM: the first reads 32 B and writes 36 B per element (lazy 9 x 29-bit limbs), the middle ones
            # read and write 36 B, the last reads 36 B and writes 32 B
            npass = int(ctx.stat("ntt_passes")); f29 = bool(ctx.stat("ntt_fr29"))
            moved = (1 << lg) * ((32 + 36) * 2 + 72 * (npass - 2)) if (f29 and npass > 1) else npass * (1 << lg) * 64
            # VALU line: 11 products of 162 v_mad_u64_u32 per element (plus one to leave the lazy domain) against the measured issue peak
            mads = (1 << lg) * (lg / 2.0 + 1) * 162 if f29 else (1 << lg) * (lg / 2.0) * 128
            extras["roofline_ntt_2p22"] = {"bound": "hbm", "kernel": ("k_ntt29_pass" if f29 else "k_ntt_pass") + " (all passes of one forward transform)",
                                           "achieved": (1 << lg) * 64 / dtn / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": (1 << lg) * 64 / dtn / 1e9 / 8000.0,
                                           "passes": npass, "actual_bytes_moved": moved, "butterflies_on_29_bit_limbs": f29,
                                           "valu_mads_per_element": mads / (1 << lg),
                                           "valu_frac_of_mad_issue_peak": (mads / 64.0 / N_SIMD * issue_costs()[0]["mad"]) / (dtn * ntt_clk * 1e9),
                                           "clock_ghz_in_kernel": ntt_clk, "clock_source": ntt_clk_src,
                                           "valu_frac_note": "multiply-adds alone at the SIMD's issue cost (profiles/r3_ubench_issue.txt) over the transform's cycles at clock_ghz_in_kernel",
                                           # the whole instruction stream (disassembly of k_ntt29_pass): a radix-4 butterfly = 4 products of 162 mads + 46
                                           # simple instructions, + ~400 simple ones around them; per element of the last pass one more product + ~150; ~100 per element and pass
                                           "valu_frac_of_instruction_mix_peak": ((((1 << lg) / 4.0) * (lg / 2.0) * (4 * (162 * issue_costs()[0]["mad"] + 46 * issue_costs()[0]["simple"]) + 400 * issue_costs()[0]["simple"])
                                                                                  + (1 << lg) * (162 * issue_costs()[0]["mad"] + (46 + 150 + 100 * npass) * issue_costs()[0]["simple"])) / 64.0 / N_SIMD)
                                                                                / (dtn * ntt_clk * 1e9) if f29 else None,
                                           "knock_out_timings_ms": {"as_built": 0.60, "products_replaced_by_additions": 0.37, "no_twiddle_loads": 0.52, "no_step_barriers": 0.59,
                                                                    "note": "2^22 forward, round 3 (variant builds, not shipped): the time is memory side + products, not their maximum -- DESIGN.md 3.2"},
                                           "note": "integer-VALU bound: 11 Fr products per element (22 stages as radix-4 steps); the mads alone are a third of the "
                                                   "instructions of a butterfly; see DESIGN.md 3.2"}
            del a
            n2 = 1 << 18
            k2, s2 = rand_fr(n2, 11), rand_fr(n2, 12)
            d_k2 = torch.from_numpy(k2.view(np.int64)).to(dev)
            d_b2 = v.fixed_base_mul(ctx, d_k2, n2, 2)
            b2 = ctx.bases_from_device(d_b2, n2, 2)
            ctx.dfree(d_b2)
            d_s2 = torch.from_numpy(s2.view(np.int64)).to(dev)
            res2, _ = b2.msm(d_s2)
            tg = time.perf_counter()
            for _ in range(3):
                b2.msm(d_s2)
            dtg = (time.perf_counter() - tg) / 3
            e2 = int(sum((to_ints(k2) * to_ints(s2)).tolist()) % R_MOD)
            extras["g2_msm_2p18_ms"] = dtg * 1e3
            extras["g2_msm_2p18_points_per_s"] = n2 / dtg
            g2p = Head(); g2p.group = 2; g2p.d_s = d_s2; g2p.n = n2; g2p.bases = b2
            elg, resg = run_sharded(g2p, 9, 2, exchange, barrier, depth)          # the same MSM with `depth` in flight, like the headline
            extras["g2_msm_2p18_pipelined_ms"] = elg / 9 * 1e3
            extras["g2_msm_2p18_pipelined_points_per_s"] = n2 * 9 / elg
            extras["g2_msm_2p18_pipelined_same_result"] = bool(np.array_equal(resg, res2))
            extras["g2_msm_2p18_verified"] = bool(np.array_equal(res2, expected_point(2, e2)))
            b2.free()
            del d_k2, d_s2
            # host-buffer entry point (vsp_msm_g1): bases and scalars cross PCIe on every call -- never `value`
            host_b = np.zeros((n, 12), np.uint64)
            ctx.d2h(host_b, d_bases_canon)
            v.multiexp(ctx, host_b, ss, 1)
            th = time.perf_counter()
            res_h = v.multiexp(ctx, host_b, ss, 1)
            dth = time.perf_counter() - th
            extras["msm_host_buffers_2p20_ms"] = dth * 1e3
            extras["msm_host_buffers_2p20_points_per_s"] = n / dth
            extras["msm_host_buffers_matches"] = bool(np.array_equal(res_h, result))
            del host_b

        if world == 1 and not args.no_prove:
            extras.update(bench_prove(ctx, v, cref, o, dev, torch, args.prove_log_n, precompute=True))
            extras.update(bench_prove_step_domain(ctx, v, cref, o, precompute=True))
            try:
                extras.update(bench_prove_small(ctx, v, cref, o, 16, precompute=True))
            except Exception as e:                         # secondary measurement: never take the bench line down
                extras["prove_2p16_error"] = repr(e)

    ctx.dfree(d_bases_canon)
    head.bases.free()
    del d_s

    # =============================================================== proofs/s at N > 1: replica proving, every rank its own resident key
    # (N = 1: bench_prove above measures the same ring mode -- ONE host thread, three contexts, packed witness -- and much more)
    replicas = None
    if (world > 1 or (args.force_replicas and use_dist)) and not args.no_prove:
        replicas = replica_leg(args, v, torch, dist, coll_dev, rank, world, dev_index, barrier, allmax, cref, o)

    # =============================================================== BASELINE config 5 in the same run: 2^26 G1 + 2^24 G2 over the ranks
    if not args.no_config5 and not (world == 1 and args.no_extras):
        c5 = {"scaling": "strong", "n_gpus": world, "msms_in_flight": depth}
        for group, lg in ((1, args.config5_log_g1), (2, args.config5_log_g2)):
            prob = ShardProblem(ctx, v, torch, dev, group, 1 << lg, world, rank, seed=55 + group)
            ctx.stats_reset()
            k5 = 2 * depth
            el5, res5 = run_sharded(prob, k5, depth, exchange, barrier, depth)      # warm-up = one MSM per work slot: their multi-GB workspaces are allocated on first use
            el5 = allmax(el5)
            e5 = gather_e(prob.e_local)
            tag = "g1" if group == 1 else "g2"
            c5[tag] = {"total_log_n": lg, "points_per_gpu": prob.n, "ms_per_msm": el5 / k5 * 1e3, "points_per_s": (1 << lg) * k5 / el5,
                       "k_accum28_avg_ms_pipelined": accum_stats(), "window_bits": int(ctx.stat("msm_window_bits")),
                       "verified": bool(np.array_equal(res5, expected_point(group, e5))) if rank == 0 else None}
            prob.free()
        extras["config5"] = c5
        c5["exchange_backend"] = exchange.backend
        c5["ranks_seen_by_collective"] = ranks_seen

    out_rl_inputs = (1, n, main_w, excl_ms, ex_elapsed / ex_steps * 1e3, pipe_ms, main_split)
    rl, rv = roofline_of(*out_rl_inputs)
    out = {
        "metric": "G1 MSM points/sec at 2^20 (Groth16 prover hot path)",
        "value": n * world * args.steps / elapsed,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",          # 32-bit registers: 28-bit limbs, 32x32->64-bit multiply-add: exact integer Montgomery arithmetic (381-bit Fp, 255-bit Fr)
        "data": "synthetic",
        "config": {"workload": (f"2^{args.log_n}-point BLS12-381 G1 Pippenger MSM per GPU, " +
                                ("resident key WITH the 16x table of window multiples, " if args.precompute else
                                 "PLAIN bases k_i*G (no table of window multiples; the library's 28-bit-limb copy keeps phi(P) = (beta x, y) beside every P for the endomorphism split), ") +
                                f"uniform scalars, bases and scalars resident in HBM; N ranks = one 2^{args.log_n}*N-point MSM sharded by contiguous chunk, "
                                "RCCL all-gather of Jacobian partial sums + fold"),
                   "points_per_gpu": n, "window_bits": main_c, "windows": main_w, "endomorphism_split": bool(main_split), "msms_in_flight": depth,
                   "bases_precomputed_window_multiples": bool(args.precompute),
                   "bases_memory_bytes_per_point": (16 * (96 + 128)) if args.precompute else (96 + 2 * 128),
                   "precompute_once_s": precompute_s,
                   "runtime_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}},
        "verified_bit_exact": verified,
        "latency_ms_one_in_flight": ex_elapsed / ex_steps * 1e3,
        "roofline": rl,
        "roofline_valu": rv,
        "cpu_baseline": cpu_baseline,
        "cpu_baseline_all_cores": cpu_all,
        "exchange": {"backend": exchange.backend, "ranks_seen_by_collective": ranks_seen, "gpus_visible": n_dev,
                     "records": "one Jacobian partial sum per rank and MSM (144 B G1 / 288 B G2), all-gathered, folded locally (vote_saver_protocol_amd/sharded.py)"},
    }
    if "config5" in extras:
        # BASELINE config 5 at the top level: ONE 2^26-point G1 and ONE 2^24-point G2 problem split over the N ranks of this run -- the driver's
        # N = 1, 2, 4, 8 lines give the strong-scaling curve (north_star: >= 6x at 8 GPUs) from these two figures
        c5 = extras["config5"]
        out["scaling_strong"] = {"g1_points_per_s": c5["g1"]["points_per_s"], "g2_points_per_s": c5["g2"]["points_per_s"],
                                 "g1_total_log_n": c5["g1"]["total_log_n"], "g2_total_log_n": c5["g2"]["total_log_n"],
                                 "g1_ms_per_msm": c5["g1"]["ms_per_msm"], "g2_ms_per_msm": c5["g2"]["ms_per_msm"],
                                 "verified": bool(c5["g1"]["verified"] and c5["g2"]["verified"]) if rank == 0 else None,
                                 "ranks_seen_by_rccl": ranks_seen if exchange.backend == "nccl" else None,
                                 "ranks_seen_by_collective": ranks_seen, "backend": exchange.backend, "n_gpus": world}
    if replicas is not None and world == 1:          # --force-replicas rehearsal: reported beside the N = 1 figures, not instead of them
        extras["prove_replicas_rehearsal"] = replicas
        replicas = None
    if replicas is not None:                         # N > 1: the other half of BASELINE.json's metric from the replica leg
        out["secondary"] = {"metric": "Groth16 proofs/sec at 2^%d constraints (synthetic SAVER-shaped R1CS, pairing-verified), all GPUs" % args.prove_log_n,
                            "value": replicas["proofs_per_s"], "unit": "proofs/s", "n_gpus": world, "mode": replicas["mode"],
                            "per_gpu_proofs_per_s": replicas["proofs_per_s"] / world, "every_rank_same_proof_bytes": replicas["every_rank_same_proof_bytes"],
                            "pairing_verified": replicas.get("pairing_verified"), "cpu_baseline": replicas.get("cpu_baseline")}
        extras["prove_replicas"] = replicas
    if extras:
        out["extras"] = extras
        if replicas is None and "prove_2p20_proofs_per_s" in extras:      # the other half of BASELINE.json's metric, same run
            ot = extras.get("prove_2p20_one_thread_pipelined", {}).get("packed")
            out["secondary"] = {"metric": "Groth16 proofs/sec at 2^20 constraints (synthetic SAVER-shaped R1CS, pairing-verified)",
                                "value": ot["proofs_per_s"] if ot else extras.get("prove_2p20_two_contexts_proofs_per_s", extras["prove_2p20_proofs_per_s"]), "unit": "proofs/s",
                                "mode": ("ONE host thread, %d proofs in flight (vsp_groth16_prove_launch / _finish, one context each) over one resident key, packed witness" % ot["contexts"]) if ot
                                        else "two host threads / contexts proving concurrently over one resident key",
                                "two_threads_two_contexts_proofs_per_s": extras.get("prove_2p20_two_contexts_proofs_per_s"),
                                "single_context_proofs_per_s": extras["prove_2p20_proofs_per_s"], "single_proof_latency_ms": extras["prove_2p20_ms"],
                                "plain_key_proofs_per_s": extras.get("prove_2p20_plain_key_proofs_per_s"), "plain_key_bytes": extras.get("prove_2p20_plain_key_bytes"),
                                "table_key_bytes": extras.get("prove_2p20_key_bytes"),
                                "real_circuit_size_2p16": {"single_proof_ms": extras.get("prove_2p16_ms"), "batched_proofs_per_s_one_context": extras.get("prove_2p16_batched_best_proofs_per_s"),
                                                           "batched_proofs_per_s_several_contexts": extras.get("prove_2p16_batched_multi_context_proofs_per_s"),
                                                           "four_contexts_proofs_per_s": extras.get("prove_2p16_4_contexts_proofs_per_s"),
                                                           "cpu_oracle_proof_s": extras.get("prove_2p16_cpu_oracle_s")}}
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
