"""Randomised parity run of the multi-exponentiation against the C oracle: sizes, duplicated / negated / infinite bases, scalar kinds, window
sizes 5..22, sort modes, bucket splits, precomputed tables, the endomorphism split on / off / forced, sub-ranges of a resident key.
   python tools/fuzz_msm.py [seconds] [seed]  -- prints the failing configuration and exits 1 on the first mismatch.
tests/test_gpu_fuzz.py runs a bounded number of configurations of fixed seeds inside `pytest -m gpu`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def fuzz(budget, seed, max_cases=None, pool1_size=None, pool2_size=None, log=print):
    """-> (configurations run, counts per class); raises AssertionError with the failing configuration on a mismatch"""
    import vote_saver_protocol_amd as v
    import cref, bls12_381 as o
    from conftest import rand_fr_array, L, g1_limbs, g2_limbs
    rng = np.random.default_rng(seed)
    ctx = v.Context(0)
    POOL1 = pool1_size or int(os.environ.get("POOL1", "30000")); POOL2 = pool2_size or int(os.environ.get("POOL2", "3000"))
    pool1 = cref.g1_batch_mul_gen(rand_fr_array(POOL1, seed=1000 + seed))
    pool2 = cref.g2_batch_mul_gen(rand_fr_array(POOL2, seed=2000 + seed))
    t0 = time.time(); it = 0; stats = {}
    last = time.time()
    try:
        while time.time() - t0 < budget and (max_cases is None or it < max_cases):
            it += 1
            if time.time() - last > 60:
                log("... %d cases, %.0f s" % (it, time.time() - t0)); last = time.time()
            group = 1 if rng.random() < 0.75 else 2
            pool = pool1 if group == 1 else pool2
            n = int(np.exp(rng.uniform(0, np.log(len(pool))))) or 1
            idx = rng.integers(0, len(pool), size=n)
            dup = rng.random()
            if dup < 0.15:
                idx = idx[rng.integers(0, max(1, n // 50), size=n)]                # few distinct points: equal-x pairs everywhere
            bases = pool[idx].copy()
            if rng.random() < 0.3 and n > 3:
                for k in rng.integers(0, n, size=max(1, n // 100)):
                    bases[k] = 0                                                     # infinities
            if rng.random() < 0.3:
                for k in rng.integers(0, n, size=max(1, n // 10)):                   # negatives of other entries
                    src = int(rng.integers(0, n))
                    P = (o.g1_from_limbs if group == 1 else o.g2_from_limbs)(bases[src])
                    G = o.G1 if group == 1 else o.G2
                    bases[k] = (g1_limbs if group == 1 else g2_limbs)(G.neg(P)) if P is not None else 0
            kind = rng.choice(["uniform", "boolean", "small", "equal", "edges"])
            ss = rand_fr_array(n, seed=int(rng.integers(1, 1 << 30)))
            if kind == "boolean":
                m = rng.random(n) < 0.9; ss[m] = 0; ss[m, 0] = rng.integers(0, 2, size=int(m.sum()), dtype=np.uint64)
            elif kind == "small":
                ss[:] = 0; ss[:, 0] = rng.integers(0, 70000, size=n, dtype=np.uint64)
            elif kind == "equal":
                ss[:] = ss[0]
            elif kind == "edges":
                for k in range(0, n, 7): ss[k] = L(o.R - 1 - int(rng.integers(0, 3)), 4)
                for k in range(3, n, 11): ss[k] = 0
            wb = int(rng.choice([0, 0, 0, 5, 8, 11, 13, 16, 17, 18, 19, 20, 21, 22]))
            sort_mode = int(rng.choice([0, 1, 2])); ctx.set_option("msm_sort", sort_mode)      # policy / never staged / staged from c = 12 on
            pre = bool(rng.random() < 0.5)
            ctx.set_option("msm_window_bits", wb)
            ctx.set_option("msm_split", int(rng.choice([0, 0, 0, 16, 64])))
            glv = int(rng.choice([1, 1, 0, 2])); ctx.set_option("msm_glv", glv)      # endomorphism split: policy / off / forced
            for kv in [x for x in os.environ.get("FUZZ_OPTS", "").split(",") if x]:   # e.g. FUZZ_OPTS=msm_dimsum_lanes=16 to bisect a mismatch
                ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
            exp = (cref.msm_g1 if group == 1 else cref.msm_g2)(bases, ss, mixed=True)
            B = ctx.upload_bases(bases, group)
            try:
                if pre:
                    B.precompute(wb if 8 <= wb <= 22 else 0)
                d_s = ctx.to_device(ss)
                first = int(rng.integers(0, n)) if rng.random() < 0.3 else 0
                cnt = n - first
                got, _ = B.msm(d_s + 32 * first, n=cnt, first=first)
                if first:
                    exp = (cref.msm_g1 if group == 1 else cref.msm_g2)(bases[first:], ss[first:], mixed=True)
                ok = np.array_equal(got, exp)
                ctx.dfree(d_s)
            finally:
                B.free()
            key = (group, pre, kind, glv); stats[key] = stats.get(key, 0) + 1
            assert ok, "MISMATCH %r" % (dict(it=it, seed=seed, group=group, n=n, kind=kind, wb=wb, sort_mode=sort_mode, pre=pre, glv=glv, first=first, dup=dup < 0.15),)
    finally:
        ctx.close()
    return it, stats


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    try:
        it, stats = fuzz(budget, seed, log=lambda m: print(m, flush=True))
    except AssertionError as e:
        print(e); sys.exit(1)
    print("fuzz ok: %d configurations in %.0f s" % (it, time.time() - t0), {str(k): c for k, c in sorted(stats.items())})
