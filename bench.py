#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Groth16 prover hot path.

Metric (BASELINE.json): G1 MSM points/sec at 2^20 (config[1]: 2^20-point BLS12-381 G1 Pippenger MSM, random
scalars/points, bit-exact vs multiexp), whole job, inputs resident in HBM.  One "step" = one full MSM of the
rank's resident 2^20-point shard (sort + bucket accumulation + bucket reduction + host Horner), then -- for
N > 1 -- the exchange step of the sharded MSM: an RCCL all-gather of the 144-byte Jacobian partial sums and a
local fold.  Per-GPU work is fixed as N grows ("weak": N ranks = one 2^20*N-point MSM).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (bucket accumulation, k_accum) with HIP
events recorded on the launch stream inside the timed region; `cpu_baseline` times the oracle's serial BDLO12
(the reference algorithm, 1 thread like the reference build) on the same inputs -- a reported baseline only.
Nothing here reads /root/reference.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
BYTES_PER_PAIR_G1 = 128        # 96 B affine base + 32 B scalar


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


def bench_prove(ctx, v, cref, o, dev, torch, log_m, precompute=True):
    """BASELINE config 4: full r1cs_gg_ppzksnark prove on a synthetic satisfiable R1CS filling a 2^log_m domain
    (90 % boolean wires, 30 public inputs).  The proving key is built on the GPU (generator batch
    exponentiation), the proof is verified with the oracle's pairing, the CPU leg is the oracle's serial prover."""
    import pairing as pg
    ni = 30
    nc = (1 << log_m) - ni - 2
    gen = o.splitmix64(5)
    cs, wit = cref.R1CS.synth(nc, ni, 4)
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
    pa, pb, pc, proof = v.groth16_prove(ctx, dcs, pk, wit, r, s_)          # warm-up (twiddles, workspaces)
    reps = 5
    ctx.stats_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        pa, pb, pc, proof = v.groth16_prove(ctx, dcs, pk, wit, r, s_)
    dt = (time.perf_counter() - t0) / reps
    phases = {k: ctx.stat("prove_" + k + "_ms") / reps for k in ("launch", "host_overlap", "wait", "assembly")}
    # throughput mode: contexts are independent and keys / constraint systems are plain device resources, so two host threads
    # with a context each prove concurrently over ONE resident key (the tails of one proof overlap the bulk of the other)
    import threading
    n_ctx = int(os.environ.get("VSP_BENCH_PROVE_CONTEXTS", "2"))
    extra_ctx = [v.Context(ctx.device) for _ in range(n_ctx - 1)]
    per_thread = 6
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
    for c in extra_ctx:
        c.close()
    vk = dict(alpha_g1=o.g1_from_limbs(alpha_g1), beta_g2=o.g2_from_limbs(beta_g2), gamma_g2=o.g2_from_limbs(gamma_g2),
              delta_g2=o.g2_from_limbs(delta_g2), gamma_ABC_g1=[o.g1_from_limbs(x) for x in gamma_abc])
    pub = [int(x) for x in to_ints(wit[:ni]).tolist()]
    ok = pg.groth16_verify(vk, pub, (o.g1_from_limbs(pa), o.g2_from_limbs(pb), o.g1_from_limbs(pc)))
    out = {f"prove_2p{log_m}_ms": dt * 1e3, f"prove_2p{log_m}_proofs_per_s": 1.0 / dt, f"prove_2p{log_m}_pairing_verified": bool(ok),
           f"prove_2p{log_m}_constraints": nc, f"generate_2p{log_m}_gpu_s": setup_s, f"prove_2p{log_m}_key_precomputed": bool(precompute),
           f"prove_2p{log_m}_phase_ms": phases,
           f"prove_2p{log_m}_two_contexts_ms_per_proof": dt2 * 1e3, f"prove_2p{log_m}_two_contexts_proofs_per_s": 1.0 / dt2,
           f"prove_2p{log_m}_two_contexts_same_proof": bool(same)}
    kp.free(); dcs.free(); cs.free()
    # CPU leg on a bounded sample: the oracle's serial generator + prover at 2^14, and the GPU on the same instance
    lg_s = 14
    nc_s = (1 << lg_s) - ni - 2
    cs2, wit2 = cref.R1CS.synth(nc_s, ni, 4)
    kp = cref.Keypair(cs2, tox)
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
    t0 = time.perf_counter()
    gA, gB, gC, _ = v.groth16_prove(ctx, dcs2, pk2, wit2, r, s_)
    gpu_dt = time.perf_counter() - t0
    out.update({f"prove_2p{lg_s}_cpu_oracle_s": cpu_dt, f"prove_2p{lg_s}_gpu_ms": gpu_dt * 1e3,
                f"prove_2p{lg_s}_bit_exact_vs_cpu": bool(np.array_equal(gA, eA) and np.array_equal(gB, eB) and np.array_equal(gC, eC))})
    pk2.free(); dcs2.free(); [q.free() for q in q2]; kp.free(); cs2.free()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20, help="log2 of the points per GPU (BASELINE config: 20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary NTT / G2 measurements")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--pipeline-depth", type=int, default=3, help="MSMs in flight (work slots with their own streams)")
    ap.add_argument("--no-pipeline", action="store_true", help="blocking MSM calls (one in flight): for clean per-kernel profiles")
    ap.add_argument("--no-precompute", action="store_true", help="do not precompute the window multiples of the resident bases")
    ap.add_argument("--prove-h-first", type=int, default=-1, help="prover queue order (library option prove_h_first); -1 = library default")
    ap.add_argument("--no-prove", action="store_true", help="skip the secondary full-prover measurement (config 4)")
    ap.add_argument("--prove-log-n", type=int, default=20, help="log2 of the synthetic R1CS domain for the prover measurement")
    args = ap.parse_args()
    # stdout carries exactly one JSON line: whatever libraries print there (RCCL's version banner, for one) is sent to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import vote_saver_protocol_amd as v

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ                     # launched by torch.distributed.run (also with one rank: same code path)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    ctx = v.Context(local_rank)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    if args.window_bits:
        ctx.set_option("msm_window_bits", args.window_bits)
    if os.environ.get("VSP_MSM_SPLIT"):
        ctx.set_option("msm_split", int(os.environ["VSP_MSM_SPLIT"]))      # experiment knob: points per bucket part
    if args.prove_h_first >= 0:
        ctx.set_option("prove_h_first", args.prove_h_first)

    n = 1 << args.log_n
    # ---- synthetic inputs (SURVEY.md 8(d) cfg 2): bases k_i * G built on the GPU, uniform scalars; all resident
    ks = rand_fr(n, seed=1000 + rank)
    ss = rand_fr(n, seed=2000 + rank)
    d_k = torch.from_numpy(ks.view(np.int64)).to(dev)
    d_s = torch.from_numpy(ss.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    d_bases_canon = v.fixed_base_mul(ctx, d_k, n, 1)
    bases = ctx.bases_from_device(d_bases_canon, n, 1)
    del d_k
    if args.log_n > 21:
        args.no_precompute = True      # measured: the precomputed table pays up to ~2^21 points per GPU (fixed tails); equal beyond
    if not args.no_precompute:
        # once per resident key: 2^(16 w) * P for the 16 windows (16x the bases' memory); every MSM then shares one bucket set
        t_pre = time.perf_counter()
        bases.precompute(16)
        precompute_s = time.perf_counter() - t_pre
    else:
        precompute_s = None

    rec_dev = torch.zeros(18, dtype=torch.int64, device=dev)
    all_dev = torch.zeros(18 * world, dtype=torch.int64, device=dev)

    def exchange(rec):
        if use_dist:                                        # exchange step: 144-byte Jacobian record per rank
            rec_dev.copy_(torch.from_numpy(rec.view(np.int64)))
            dist.all_gather_into_tensor(all_dev, rec_dev)
            recs = all_dev.cpu().numpy().view(np.uint64).reshape(world, 18)
        else:
            recs = rec.reshape(1, 18)
        return v.fold_jacobian(ctx, recs, 1)                # local fold + affine normalisation

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(k_steps):
        """k_steps full MSMs, software-pipelined over work slots with their own streams (default three deep; measured 1: 5.38 ms, 2: 4.02, 3: 3.72, 4: 3.78): the sort and
        bucket accumulation of step k+1 overlap the latency-bound bucket reduction and host Horner of step k."""
        if args.no_pipeline:
            for _ in range(k_steps):
                res = exchange(bases.msm_jacobian(d_s))
            return res
        res = None
        sl = (1, 2, 4, 5)[:max(1, min(4, args.pipeline_depth))]   # work slots with streams of equal priority
        D = len(sl)
        for k in range(min(D - 1, k_steps)):
            bases.msm_launch(sl[k % D], d_s)
        for k in range(k_steps):
            if k + D - 1 < k_steps:
                bases.msm_launch(sl[(k + D - 1) % D], d_s)
            res = exchange(bases.msm_finish_jacobian(sl[k % D]))
        return res

    if args.warmup:
        result = run_steps(args.warmup)
    ctx.stats_reset()
    barrier()
    t0 = time.perf_counter()
    result = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    accum_ms = ctx.stat("msm_accum_ms")
    accum_launches = ctx.stat("msm_accum_launches")
    main_c, main_w = int(ctx.stat("msm_window_bits")), int(ctx.stat("msm_windows"))
    # transparency: the same pipelined loop over PLAIN resident bases (no precomputed window multiples), not part of `value`
    plain_ms = None
    if not args.no_precompute and not args.no_extras:
        plain = ctx.bases_from_device(d_bases_canon, n, 1)
        keep = bases
        bases = plain
        run_steps(2)
        barrier(); tp = time.perf_counter()
        run_steps(max(4, args.steps // 2))
        barrier(); plain_ms = (time.perf_counter() - tp) / max(4, args.steps // 2) * 1e3
        bases = keep
        plain.free()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- untimed verification: sum over all ranks of (sum_i k_i s_i) * G must equal the folded result
    e_local = int(sum((to_ints(ks) * to_ints(ss)).tolist()) % R_MOD)
    e_dev = torch.from_numpy(limbs(e_local, 4).view(np.int64)).to(dev)
    if use_dist:
        e_all = torch.zeros(4 * world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(e_all, e_dev)
        e_rows = e_all.cpu().numpy().view(np.uint64).reshape(world, 4)
    else:
        e_rows = limbs(e_local, 4).reshape(1, 4)
    verified = None
    cpu_baseline = None
    extras = {}
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import cref                                          # checker / CPU baseline only
        import bls12_381 as o
        e_tot = sum(int(x) for x in to_ints(e_rows).tolist()) % R_MOD
        expect = cref.g1_mul(np.array(o.g1_to_limbs(o.G1.gen), dtype=np.uint64), limbs(e_tot, 4))
        verified = bool(np.array_equal(result, expect))

        if world == 1 and not args.no_cpu_baseline:
            # reference algorithm on the host: serial BDLO12 (c = 16 at 2^20), same bases and scalars, 1 thread
            host_b = np.zeros((n, 12), np.uint64)
            ctx.d2h(host_b, d_bases_canon)
            m = min(n, 1 << 20)
            tc = time.perf_counter()
            ref = cref.msm_g1(host_b[:m], ss[:m])
            dt = time.perf_counter() - tc
            ok = bool(np.array_equal(ref, result)) if m == n else None
            cpu_baseline = {"value": m / dt, "unit": "points/s", "cores": 1, "kind": "port",
                            "sample": f"one {m}-point G1 MSM (serial BDLO12 restatement, oracle/vsp_ref.c), {dt:.1f} s",
                            "matches_gpu_result": ok}

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
            extras["ntt_2p22_algorithmic_GBs"] = (1 << lg) * 64 / dtn / 1e9
            # algorithmic bytes = 64 B per element for the whole transform (one ideal read + write); the kernel makes
            # `passes` round trips through HBM, each reading and writing every element once
            npass = int(ctx.stat("ntt_passes"))
            extras["roofline_ntt_2p22"] = {"bound": "hbm", "kernel": "k_ntt_pass (all passes of one transform)", "achieved": (1 << lg) * 64 / dtn / 1e9,
                                           "peak": 8000.0, "unit": "GB/s", "frac": (1 << lg) * 64 / dtn / 1e9 / 8000.0, "passes": npass,
                                           "actual_bytes_moved": npass * (1 << lg) * 64, "per_pass_GBs": (1 << lg) * 64 / (dtn / npass) / 1e9,
                                           "note": "integer-VALU bound: 11 Fr products per element; see DESIGN.md 3.2 for the measured split"}
            del a
            # G2: 2^18 points, and 2^21 = the per-GPU shard of BASELINE config 5 (2^24 G2 points over 8 GPUs); verified through the
            # discrete-log identity sum_i s_i (k_i G2) = (sum_i k_i s_i) G2 against the oracle's scalar multiplication
            g2_gen = np.array(o.g2_to_limbs(o.G2.gen), dtype=np.uint64)
            for lg2 in (18, 21):
                n2 = 1 << lg2
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
                extras[f"g2_msm_2p{lg2}_ms"] = dtg * 1e3
                extras[f"g2_msm_2p{lg2}_points_per_s"] = n2 / dtg
                extras[f"g2_msm_2p{lg2}_verified"] = bool(np.array_equal(res2, cref.g2_mul(g2_gen, limbs(e2, 4))))
                b2.free()
                del d_k2, d_s2

        if world == 1 and not args.no_extras:
            # host-buffer entry point (vsp_msm_g1): bases and scalars cross PCIe on every call -- never `value`
            host_b = np.zeros((n, 12), np.uint64)
            ctx.d2h(host_b, d_bases_canon)
            v.multiexp(ctx, host_b, ss, 1)
            th = time.perf_counter()
            res_h = v.multiexp(ctx, host_b, ss, 1)
            dth = time.perf_counter() - th
            extras["msm_host_buffers_2p20_ms"] = dth * 1e3
            extras["msm_host_buffers_2p20_points_per_s"] = n / dth
            extras["msm_host_buffers_matches"] = bool(np.array_equal(res_h, result)) if world == 1 else None
            del host_b

        if world == 1 and not args.no_prove:
            extras.update(bench_prove(ctx, v, cref, o, dev, torch, args.prove_log_n, precompute=not args.no_precompute))
            extras.update(bench_prove_step_domain(ctx, v, cref, o, precompute=not args.no_precompute))

    ctx.dfree(d_bases_canon)
    total_points = n * world * args.steps
    value = total_points / elapsed
    accum_avg_s = (accum_ms / accum_launches) * 1e-3 if accum_launches else float("nan")
    precomputed = bool(not args.no_precompute)       # (log_n > 21 switches precomputation off above)
    mads_per_add = (8 * 392 + 588) if True else 0      # 28-bit-limb accumulation (plain and precomputed G1 bases): 8 products + 1 dual product
    achieved = n * BYTES_PER_PAIR_G1 / accum_avg_s / 1e9 if accum_launches else float("nan")
    # HBM-side traffic of the dominant kernel cannot be read inside this process (PMC passes need rocprofv3): it is taken
    # from the committed summary of the same command, profiles/r1_h_pmc_hbm_traffic.json (separate FETCH_SIZE / WRITE_SIZE passes)
    traffic = None
    try:
        if args.log_n == 20:
            with open(os.path.join(ROOT, "profiles", "r1_h_pmc_hbm_traffic.json")) as f:
                traffic = json.load(f)["k_accum_G1_2p20"]["traffic_bytes_per_launch"]
    except Exception:
        traffic = None
    out = {
        "metric": "G1 MSM points/sec at 2^20 (Groth16 prover hot path)",
        "value": value,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",          # 32-bit limbs, 32x32->64-bit multiply-add: exact integer Montgomery arithmetic (381-bit Fp, 255-bit Fr)
        "data": "synthetic",
        "config": {"workload": f"2^{args.log_n}-point BLS12-381 G1 Pippenger MSM per GPU, bases k_i*G, uniform scalars, "
                               f"resident in HBM; N ranks = one 2^{args.log_n}*N-point MSM sharded by contiguous chunk, "
                               "RCCL all-gather of Jacobian partial sums + fold",
                   "points_per_gpu": n, "window_bits": main_c, "windows": main_w,
                   "bases_precomputed_window_multiples": not args.no_precompute, "bases_memory_factor": 1 if args.no_precompute else main_w,
                   "precompute_once_s": precompute_s,
                   "plain_bases_ms_per_step": plain_ms, "plain_bases_points_per_s": (n * world / (plain_ms * 1e-3)) if plain_ms else None},
        "verified_bit_exact": verified,
        "roofline": {"bound": "hbm", "kernel": "k_accum28 (bucket accumulation, 28-bit limbs)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "algorithmic_bytes_per_launch": n * BYTES_PER_PAIR_G1, "avg_launch_ms": accum_avg_s * 1e3,
                     "note": "integer-VALU bound (about 160 Montgomery products per point), not HBM bound; launch time is measured while the "
                             "neighbouring step's kernels share the GPU (three MSMs in flight); traffic = 2*FETCH_SIZE + WRITE_SIZE from "
                             "profiles/r1_h_pmc_hbm_traffic.json: every base is gathered once per window (16 rows of 112 B from the 28-bit-limb table), see DESIGN.md"},
        # the bound that actually applies: 32x32->64-bit multiply-add issue.  One mixed addition = 10 Montgomery products; with
        # the accumulation runs on 14 x 28-bit limbs: 2 * 14 * 14 = 392 v_mad_u64_u32 per product, carry-free, and one dual product
        # a*b + c*d with a single reduction (588) -- 3724 per mixed addition.  peak = measured v_mad_u64_u32 issue rate
        # (profiles/r1_ubench_valu.txt: 1.46 G wave-instructions/s/CU x 64 lanes x 256 CUs).  Same launch time as above.
        "roofline_valu": {"bound": "v_mad_u64_u32 issue", "kernel": "k_accum28 (bucket accumulation, 14 x 28-bit limbs)",
                          "achieved": n * main_w * mads_per_add / accum_avg_s / 1e12 if accum_launches else None,
                          "peak": 1.46e9 * 64 * 256 / 1e12, "unit": "T v_mad_u64_u32 lane-ops/s",
                          "frac": (n * main_w * mads_per_add / accum_avg_s) / (1.46e9 * 64 * 256) if accum_launches else None},
        "cpu_baseline": cpu_baseline,
    }
    if extras:
        out["extras"] = extras
        if "prove_2p20_proofs_per_s" in extras:      # the other half of BASELINE.json's metric, same run
            out["secondary"] = {"metric": "Groth16 proofs/sec at 2^20 constraints (synthetic SAVER-shaped R1CS, pairing-verified)",
                                "value": extras.get("prove_2p20_two_contexts_proofs_per_s", extras["prove_2p20_proofs_per_s"]), "unit": "proofs/s",
                                "mode": "two host threads / contexts proving concurrently over one resident key",
                                "single_context_proofs_per_s": extras["prove_2p20_proofs_per_s"], "single_proof_latency_ms": extras["prove_2p20_ms"]}
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    bases.free()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
