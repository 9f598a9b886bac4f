"""GPU parity: vsp_msm_g1 / vsp_msm_g2 / resident-bases MSM (HIP) vs the oracle -- identical affine points.
Edge cases follow what multiexp's callers can feed it: empty input, single point, zero / one / r-1 scalars,
duplicate bases, cancelling pairs, infinity bases, the 0/1-heavy witness distribution of
multiexp_with_mixed_addition, every-scalar-equal (one bucket per window)."""
import numpy as np
import pytest

import bls12_381 as o
from conftest import I, L, dec1, dec2, fr_array, fr_ints_fast, g1_limbs, g2_limbs, load_golden, rand_fr_array

import vote_saver_protocol_amd as v

pytestmark = pytest.mark.gpu


def test_msm_golden_vectors(ctx):
    for case in load_golden("msm.json"):
        scal = fr_array([int(s, 16) for s in case["scalars"]])
        if case["group"] == "g1":
            bases = np.stack([g1_limbs(dec1(p)) for p in case["bases"]])
            assert o.g1_from_limbs(v.multiexp(ctx, bases, scal, 1)) == dec1(case["result"]), case["n"]
            assert o.g1_from_limbs(v.multiexp_with_mixed_addition(ctx, bases, scal, 1)) == dec1(case["result"])
        else:
            bases = np.stack([g2_limbs(dec2(p)) for p in case["bases"]])
            assert o.g2_from_limbs(v.multiexp(ctx, bases, scal, 2)) == dec2(case["result"]), case["n"]


def test_msm_empty_and_degenerate(ctx):
    assert not v.multiexp(ctx, np.zeros((0, 12), np.uint64), np.zeros((0, 4), np.uint64), 1).any()
    assert not v.multiexp(ctx, np.zeros((0, 24), np.uint64), np.zeros((0, 4), np.uint64), 2).any()
    G = g1_limbs(o.G1.gen).reshape(1, 12)
    assert not v.multiexp(ctx, G, fr_array([0]), 1).any()                      # 0 * G
    assert o.g1_from_limbs(v.multiexp(ctx, G, fr_array([1]), 1)) == o.G1.gen
    assert o.g1_from_limbs(v.multiexp(ctx, G, fr_array([o.R - 1]), 1)) == o.G1.neg(o.G1.gen)
    two = np.concatenate([G, g1_limbs(o.G1.neg(o.G1.gen)).reshape(1, 12)])
    assert not v.multiexp(ctx, two, fr_array([12345, 12345]), 1).any()          # P - P
    assert not v.multiexp(ctx, np.zeros((3, 12), np.uint64), fr_array([5, 6, 7]), 1).any()   # all-infinity bases
    with pytest.raises(ValueError):
        v.multiexp(ctx, G, fr_array([1, 2]), 1)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 256, 1000, 4096])
@pytest.mark.parametrize("group", [1, 2])
def test_msm_vs_c_oracle_random(ctx, cref, n, group):
    if group == 2 and n > 1000:
        pytest.skip("G2 oracle time")
    ks = rand_fr_array(n, seed=7 * n + group)
    ss = rand_fr_array(n, seed=11 * n + group)
    bases = cref.g1_batch_mul_gen(ks) if group == 1 else cref.g2_batch_mul_gen(ks)
    exp = cref.msm_g1(bases, ss) if group == 1 else cref.msm_g2(bases, ss)
    assert np.array_equal(v.multiexp(ctx, bases, ss, group), exp)


@pytest.mark.parametrize("window_bits", [3, 4, 5, 7, 10, 13, 15, 16, 17, 18, 19, 20, 21, 22, 23])
def test_msm_every_window_size(ctx, cref, window_bits):
    n = 700
    bases = cref.g1_batch_mul_gen(rand_fr_array(n, seed=1))
    ss = rand_fr_array(n, seed=2)
    ss[0] = L(o.R - 1, 4); ss[1] = L(1, 4); ss[2] = 0
    ss[3] = L((o.R - 1) // 2, 4); ss[4] = L((o.R + 1) // 2, 4); ss[5] = L(o.R - 2, 4)      # either side of the fold k -> min(k, r - k) (c = 3, 5, 15, 17)
    ss[6] = L((1 << 254) - 1, 4); ss[7] = L(1 << 254, 4); ss[8] = L(o.R - (1 << 254), 4)
    exp = cref.msm_g1(bases, ss)
    ctx.set_option("msm_window_bits", window_bits)
    try:
        assert np.array_equal(v.multiexp(ctx, bases, ss, 1), exp)
        assert ctx.stat("msm_window_bits") == window_bits
        assert ctx.stat("msm_windows") == (254 if 255 % window_bits == 0 else 255) // window_bits + 1
        if 255 % window_bits == 0:                                 # the same without the fold: one window more, same point
            ctx.set_option("msm_fold", 0)
            assert np.array_equal(v.multiexp(ctx, bases, ss, 1), exp)
    finally:
        ctx.set_option("msm_window_bits", 0); ctx.set_option("msm_fold", 1)


@pytest.mark.parametrize("sort_mode", [1, 2])
@pytest.mark.parametrize("window_bits", [17, 18, 19, 20, 21, 22, 23])
def test_msm_windows_wider_than_16_bits_through_the_two_pass_sort(ctx, cref, window_bits, sort_mode):
    """c = 17 .. 23: the LDS counting sort over the high 15 bits of the bucket index, k_segment_sort over the low
    c - 16, two-digit bucket reduction with digits of up to 2^11 values through k_dimbits -- G1 (plain, with and without the endomorphism
    split, and over a table of window multiples sharing one bucket set) and G2, against the C oracle; skewed scalars (zeros, ones, one
    value repeated: a segment of thousands of entries for one wave, and of more than 8192 for the staged sort's third pass) included"""
    n = 40000 if window_bits < 22 else 34000
    ks, ss = rand_fr_array(n, seed=100 + window_bits), rand_fr_array(n, seed=200 + window_bits)
    ss[0] = L(o.R - 1, 4); ss[1] = L(1, 4); ss[2] = 0
    ss[100:4000] = ss[99]                                           # one value 3900 times: every window has one crowded bucket
    ss[5000:9000] = 0; ss[5000:9000:2, 0] = 1                       # zeros and ones
    ss[10000:19400] = ss[9999]                                      # 9400 times: longer than one wave orders (MS_SEG_LONG): the third pass, in every window
    ss[20000:29000] = 0; ss[20000:29000, 0] = 1 + (np.arange(9000) % 3 == 0)      # ones and twos: one long segment, two buckets in it (c >= 18)
    b1 = cref.g1_batch_mul_gen(ks)
    want = cref.msm_g1(b1, ss)
    # sort_mode 1: LDS counting sort over the high 15 bits + k_segment_sort; 2: the staged sort of large problems (8 + 8 bits through LDS
    # tiles, the rest inside LDS: k_ms_*), forced here at a size the oracle can follow
    ctx.set_option("msm_window_bits", window_bits); ctx.set_option("msm_sort", sort_mode)
    try:
        for glv in (1, 0):
            ctx.set_option("msm_glv", 2 * glv)
            B = ctx.upload_bases(b1, 1); d_s = ctx.to_device(ss)
            got, _ = B.msm(d_s)
            assert np.array_equal(got, want), (window_bits, glv)
            assert ctx.stat("msm_window_bits") == window_bits and ctx.stat("msm_endomorphism_split") == glv
            if not glv and window_bits in (17, 20):
                part, _ = B.msm(d_s + 777 * 32, 35000 - 777, 777)   # a sub-range of the resident bases
                assert np.array_equal(part, cref.msm_g1(b1[777:35000], ss[777:35000]))
            B.free(); ctx.dfree(d_s)
        if window_bits in (17, 20, 22):                             # window multiples: all windows share ONE set of 2^(c-1) buckets
            ctx.set_option("msm_window_bits", 0)
            m1 = 33000
            B = ctx.upload_bases(b1[:m1], 1).precompute(window_bits); d_s = ctx.to_device(ss[:m1])
            got, _ = B.msm(d_s)
            assert np.array_equal(got, cref.msm_g1(b1[:m1], ss[:m1])) and ctx.stat("msm_window_bits") == window_bits and ctx.stat("msm_bucket_sets") == 1
            B.free(); ctx.dfree(d_s)
            ctx.set_option("msm_window_bits", window_bits)
        if window_bits in (17, 19, 21):
            m2 = 33000
            b2 = cref.g2_batch_mul_gen(ks[:m2])
            B = ctx.upload_bases(b2, 2); d_s = ctx.to_device(ss[:m2])
            got, _ = B.msm(d_s)
            assert np.array_equal(got, cref.msm_g2(b2, ss[:m2]))
            B.free(); ctx.dfree(d_s)
    finally:
        ctx.set_option("msm_window_bits", 0); ctx.set_option("msm_glv", 1); ctx.set_option("msm_sort", 0)


@pytest.mark.parametrize("kind", ["all_zero", "all_one", "boolean_90", "all_equal", "small"])
def test_msm_skewed_scalars(ctx, cref, kind):
    """the stress variants of SURVEY.md 8(d): heavy buckets are split and merged by workgroups"""
    n = 6000
    bases = cref.g1_batch_mul_gen(rand_fr_array(n, seed=5))
    rng = np.random.default_rng(9)
    ss = np.zeros((n, 4), np.uint64)
    if kind == "all_one":
        ss[:, 0] = 1
    elif kind == "boolean_90":
        ss = rand_fr_array(n, seed=6)
        mask = rng.random(n) < 0.9
        ss[mask] = 0
        ss[mask, 0] = rng.integers(0, 2, size=int(mask.sum()), dtype=np.uint64)
    elif kind == "all_equal":
        ss[:] = rand_fr_array(1, seed=8)[0]
    elif kind == "small":
        ss[:, 0] = rng.integers(0, 1000, size=n, dtype=np.uint64)
    exp = cref.msm_g1(bases, ss, mixed=True)
    ctx.set_option("msm_split", 64)               # force many bucket parts even at this size
    try:
        assert np.array_equal(v.multiexp_with_mixed_addition(ctx, bases, ss, 1), exp)
    finally:
        ctx.set_option("msm_split", 0)
    assert np.array_equal(v.multiexp(ctx, bases, ss, 1), exp)


def test_msm_g2_skewed(ctx, cref):
    n = 900
    bases = cref.g2_batch_mul_gen(rand_fr_array(n, seed=15))
    ss = rand_fr_array(n, seed=16)
    ss[: n // 2] = 0
    ss[: n // 2, 0] = 1
    ctx.set_option("msm_split", 32)
    try:
        assert np.array_equal(v.multiexp(ctx, bases, ss, 2), cref.msm_g2(bases, ss, mixed=True))
    finally:
        ctx.set_option("msm_split", 0)


def test_msm_resident_bases_subranges_and_jacobian_fold(ctx, cref):
    """resident proving-key slices, sub-range MSM, and the Jacobian partial-sum records of the sharded MSM"""
    n = 3000
    bases = cref.g1_batch_mul_gen(rand_fr_array(n, seed=21))
    ss = rand_fr_array(n, seed=22)
    B = ctx.upload_bases(bases, 1)
    d_s = ctx.to_device(ss)
    try:
        full, inf = B.msm(d_s)
        assert not inf and np.array_equal(full, cref.msm_g1(bases, ss))
        # shard into 3 uneven chunks, fold the Jacobian records
        cuts = [0, 1000, 1001, n]
        recs = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            recs.append(B.msm_jacobian(d_s + 32 * a, n=b - a, first=a))
            part, _ = B.msm(d_s + 32 * a, n=b - a, first=a)
            assert np.array_equal(part, cref.msm_g1(bases[a:b], ss[a:b]))
        assert np.array_equal(v.fold_jacobian(ctx, np.stack(recs), 1), full)
        with pytest.raises(v.VspError):
            B.msm(d_s, n=10, first=n - 5)          # range outside the resident bases
    finally:
        ctx.dfree(d_s); B.free()


def test_fixed_base_mul_vs_oracle(ctx, cref):
    n = 300
    ks = rand_fr_array(n, seed=31)
    ks[0] = 0; ks[1] = L(1, 4); ks[2] = L(o.R - 1, 4)
    d_k = ctx.to_device(ks)
    for group, width, ref in ((1, 12, cref.g1_batch_mul_gen), (2, 24, cref.g2_batch_mul_gen)):
        d_out = v.fixed_base_mul(ctx, d_k, n, group)
        out = np.zeros((n, width), np.uint64)
        ctx.d2h(out, d_out)
        assert np.array_equal(out, ref(ks))
        ctx.dfree(d_out)
    ctx.dfree(d_k)


def test_msm_full_size_2p20_discrete_log_identity(ctx, cref):
    """BASELINE config 2 (2^20 points, uniform scalars): bases k_i*G are generated on the GPU, so the MSM must
    equal (sum k_i s_i mod r) * G -- one scalar multiplication on the oracle side.  Plus a 2^17 slice checked
    bit for bit against the C oracle's BDLO12."""
    n = 1 << 20
    ks = rand_fr_array(n, seed=1)
    ss = rand_fr_array(n, seed=2)
    d_k = ctx.to_device(ks); d_s = ctx.to_device(ss)
    d_b = v.fixed_base_mul(ctx, d_k, n, 1)
    B = ctx.bases_from_device(d_b, n, 1)
    try:
        got, inf = B.msm(d_s)
        k_int = fr_ints_fast(ks); s_int = fr_ints_fast(ss)
        e = sum(a * b for a, b in zip(k_int, s_int)) % o.R
        assert not inf and np.array_equal(got, cref.g1_mul(g1_limbs(o.G1.gen), L(e, 4)))
        m = 1 << 15
        host_b = np.zeros((m, 12), np.uint64); ctx.d2h(host_b, d_b + 96 * 777)
        part, _ = B.msm(d_s + 32 * 777, n=m, first=777)
        assert np.array_equal(part, cref.msm_g1(host_b, ss[777:777 + m]))
    finally:
        B.free(); ctx.dfree(d_b); ctx.dfree(d_k); ctx.dfree(d_s)


@pytest.mark.parametrize("precompute", [False, True])
def test_msm_g2_2p18_discrete_log_identity(ctx, cref, precompute):
    """G2 at a size that fills the GPU (the lane-pair kernels with every part count, merges, the three-digit bucket reduction),
    plain bases and precomputed window multiples: sum s_i (k_i G2) = (sum k_i s_i) G2, and a slice bit for bit vs the C oracle."""
    n = 1 << 18
    ks = rand_fr_array(n, seed=31)
    ss = rand_fr_array(n, seed=32)
    ss[:20000] = 0; ss[:10000, 0] = 1                      # zeros and ones among the scalars (multiexp_with_mixed_addition's cases)
    d_k = ctx.to_device(ks); d_s = ctx.to_device(ss)
    d_b = v.fixed_base_mul(ctx, d_k, n, 2)
    B = ctx.bases_from_device(d_b, n, 2)
    if precompute:
        B.precompute(0)
    try:
        got, inf = B.msm(d_s)
        e = sum(a * b for a, b in zip(fr_ints_fast(ks), fr_ints_fast(ss))) % o.R
        assert not inf and np.array_equal(got, cref.g2_mul(g2_limbs(o.G2.gen), L(e, 4)))
        m = 1 << 13
        host_b = np.zeros((m, 24), np.uint64); ctx.d2h(host_b, d_b + 192 * 5555)
        part, _ = B.msm(d_s + 32 * 5555, n=m, first=5555)
        assert np.array_equal(part, cref.msm_g2(host_b, ss[5555:5555 + m], mixed=True))
    finally:
        B.free(); ctx.dfree(d_b); ctx.dfree(d_k); ctx.dfree(d_s)


@pytest.mark.gpu
@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("precompute", [False, True])
def test_msm_equal_partial_sums_in_the_28bit_merges(ctx, cref, group, precompute):
    """Every base is the same point P (or -P) and every scalar is one: the all-ones bucket is cut into parts whose sums are
    IDENTICAL multiples of P, so the merges and the bucket reduction -- which run on the 28-bit form (fp28.h xyzz_add over XYZZ<Fp28> /
    XYZZ<Fp28L>) -- meet equal x at every level: doublings with (P, P, ...), cancellations with (P, -P, ...) blocks.  Both are the
    cold path that goes through the generic formulas."""
    gen = (cref.g1_batch_mul_gen if group == 1 else cref.g2_batch_mul_gen)(rand_fr_array(1, seed=171))[0]
    Pt = (o.g1_from_limbs if group == 1 else o.g2_from_limbs)(gen)
    G = o.G1 if group == 1 else o.G2
    neg = np.array((g1_limbs if group == 1 else g2_limbs)(G.neg(Pt)), dtype=np.uint64)
    ref = cref.msm_g1 if group == 1 else cref.msm_g2
    for n, pattern in ((5000, "same"), (40000, "same"), (40000, "blocks"), (4096, "alternate")):
        bases = np.tile(gen, (n, 1))
        if pattern == "blocks":
            bases[(np.arange(n) // 16) % 2 == 1] = neg            # parts of 16 points: +16 P, -16 P, ... -> cancellations in the merge tree
        elif pattern == "alternate":
            bases[1::2] = neg
        ss = np.zeros((n, 4), dtype=np.uint64); ss[:, 0] = 1
        ss[::97] = rand_fr_array(len(ss[::97]), seed=172)          # a few dense scalars so that every window has buckets
        exp = ref(bases, ss, mixed=True)
        B = ctx.upload_bases(bases, group)
        if precompute:
            B.precompute(0)
        d_s = ctx.to_device(ss)
        try:
            got, _ = B.msm(d_s)
            assert np.array_equal(got, exp), (n, pattern)
        finally:
            ctx.dfree(d_s); B.free()


@pytest.mark.parametrize("precompute", [False, True])
def test_msm_duplicate_bases_take_the_equal_x_path(ctx, cref, precompute):
    """Only ten distinct points (and their negatives) among 4000 bases: buckets keep meeting equal x -- doublings and
    cancellations -- which the 28-bit-limb accumulation hands back to the generic kernel part by part."""
    n = 4000
    few = cref.g1_batch_mul_gen(rand_fr_array(10, seed=71))
    neg = few.copy()
    for i in range(10):
        neg[i] = g1_limbs(o.G1.neg(o.g1_from_limbs(few[i])))
    rng = np.random.default_rng(72)
    pick = rng.integers(0, 20, size=n)
    bases = np.stack([few[k] if k < 10 else neg[k - 10] for k in pick])
    bases[17] = 0                                              # an infinity among them
    ss = rand_fr_array(n, seed=73)
    ss[:500] = 0; ss[:500, 0] = rng.integers(0, 3, size=500, dtype=np.uint64)
    exp = cref.msm_g1(bases, ss, mixed=True)
    B = ctx.upload_bases(bases, 1)
    if precompute:
        B.precompute(0)
    d_s = ctx.to_device(ss)
    try:
        got, _ = B.msm(d_s)
        assert np.array_equal(got, exp)
        part, _ = B.msm(d_s + 32 * 100, n=3000, first=100)
        assert np.array_equal(part, cref.msm_g1(bases[100:3100], ss[100:3100], mixed=True))
    finally:
        ctx.dfree(d_s); B.free()


@pytest.mark.parametrize("group", [1, 2])
def test_msm_one_point_many_times_in_the_bucket_reduction(ctx, cref, group):
    """Every base the SAME point, uniform scalars, narrow windows: the bucket sums are small multiples of one point, so the subset sums of
    the bucket reduction (k_dimsum_mixed, k_dimbits / k_dimweight) keep adding EQUAL points -- the doubling path of the full addition --
    and opposite ones.  The configuration with which tools/fuzz_msm.py (seed 41) showed a prefetching variant of k_dimsum_mixed to be
    wrong while the rest of the suite passed; several sizes and window widths around it."""
    one = (cref.g1_batch_mul_gen if group == 1 else cref.g2_batch_mul_gen)(rand_fr_array(1, seed=91))
    fn = cref.msm_g1 if group == 1 else cref.msm_g2
    try:
        for n, wb in ((34, 8), (34, 0), (200, 5), (200, 11), (1500, 8), (1500, 13), (5000, 0)):
            bases = np.repeat(one, n, axis=0)
            ss = rand_fr_array(n, seed=92 + n)
            if n > 100:
                bases[n // 2:] = (g1_limbs(o.G1.neg(o.g1_from_limbs(one[0]))) if group == 1 else g2_limbs(o.G2.neg(o.g2_from_limbs(one[0]))))
            want = fn(bases, ss, mixed=True)
            ctx.set_option("msm_window_bits", wb)
            for pre in (False, True):
                B = ctx.upload_bases(bases, group); d_s = ctx.to_device(ss)
                try:
                    if pre:
                        B.precompute(wb if 8 <= wb <= 16 else 0)
                    got, _ = B.msm(d_s)
                    assert np.array_equal(got, want), (n, wb, pre)
                finally:
                    B.free(); ctx.dfree(d_s)
    finally:
        ctx.set_option("msm_window_bits", 0)


def test_msm_pipelined_slots_and_shared_streams(ctx, cref):
    """vsp_msm_launch / vsp_msm_finish_jacobian: several multi-exponentiations in flight on their own streams (G1 and G2
    mixed), finished out of order; results identical to the blocking calls and to the oracle."""
    n = 5000
    b1 = cref.g1_batch_mul_gen(rand_fr_array(n, seed=41)); b2 = cref.g2_batch_mul_gen(rand_fr_array(600, seed=42))
    sa, sb, sc = rand_fr_array(n, seed=43), rand_fr_array(n, seed=44), rand_fr_array(600, seed=45)
    sb[::2] = 0; sb[::2, 0] = 1                        # half of them equal to one: split-bucket path
    B1 = ctx.upload_bases(b1, 1); B2 = ctx.upload_bases(b2, 2)
    d_a, d_b, d_c = ctx.to_device(sa), ctx.to_device(sb), ctx.to_device(sc)
    try:
        for rounds in range(2):                          # second round reuses warm workspaces
            B1.msm_launch(1, d_a); B2.msm_launch(2, d_c); B1.msm_launch(3, d_b, n=n - 7, first=7); B1.msm_launch(0, d_a)
            r3 = B1.msm_finish_jacobian(3); r0 = B1.msm_finish_jacobian(0); r2 = B2.msm_finish_jacobian(2); r1 = B1.msm_finish_jacobian(1)
            assert np.array_equal(v.fold_jacobian(ctx, r1[None], 1), cref.msm_g1(b1, sa))
            assert np.array_equal(v.fold_jacobian(ctx, r0[None], 1), cref.msm_g1(b1, sa))
            assert np.array_equal(v.fold_jacobian(ctx, r3[None], 1), cref.msm_g1(b1[7:], sb[:n - 7], mixed=True))
            assert np.array_equal(v.fold_jacobian(ctx, r2[None], 2), cref.msm_g2(b2, sc))
        with pytest.raises(v.VspError):
            B1.msm_finish_jacobian(4)                    # finish without launch
        with pytest.raises(v.VspError):
            B1.msm_launch(9, d_a)                        # no such slot
    finally:
        for d in (d_a, d_b, d_c):
            ctx.dfree(d)
        B1.free(); B2.free()


@pytest.mark.parametrize("group,n", [(1, 3000), (2, 500)])
@pytest.mark.parametrize("window_bits", [8, 13, 16])
def test_msm_precomputed_window_multiples(ctx, cref, group, n, window_bits):
    """vsp_bases_precompute: all windows share one bucket set; same results for random, boolean-heavy and all-equal
    scalars, for sub-ranges, and for the pipelined form."""
    ks = rand_fr_array(n, seed=50 + group)
    bases = cref.g1_batch_mul_gen(ks) if group == 1 else cref.g2_batch_mul_gen(ks)
    bases[5] = 0                                            # an infinity base
    ref = cref.msm_g1 if group == 1 else cref.msm_g2
    B = ctx.upload_bases(bases, group).precompute(window_bits)
    B.precompute(window_bits)                               # idempotent
    with pytest.raises(v.VspError):
        B.precompute(9 if window_bits != 9 else 10)          # a different window size is refused
    rng = np.random.default_rng(3)
    cases = {"random": rand_fr_array(n, seed=60)}
    sk = rand_fr_array(n, seed=61); m = rng.random(n) < 0.9; sk[m] = 0; sk[m, 0] = rng.integers(0, 2, size=int(m.sum()), dtype=np.uint64)
    cases["boolean_90"] = sk
    eq = np.zeros((n, 4), np.uint64); eq[:] = rand_fr_array(1, seed=62)[0]
    cases["all_equal"] = eq
    edge = rand_fr_array(n, seed=63); edge[0] = L(o.R - 1, 4); edge[1] = L(1, 4); edge[2] = 0
    cases["edge"] = edge
    try:
        for name, ss in cases.items():
            d_s = ctx.to_device(ss)
            got, _ = B.msm(d_s)
            assert np.array_equal(got, ref(bases, ss, mixed=True)), name
            a, b = 17, n - 100
            part, _ = B.msm(d_s + 32 * a, n=b - a, first=a)
            assert np.array_equal(part, ref(bases[a:b], ss[a:b], mixed=True)), name
            B.msm_launch(2, d_s); B.msm_launch(1, d_s + 32 * a, n=b - a, first=a)
            r1 = B.msm_finish_jacobian(1); r2 = B.msm_finish_jacobian(2)
            assert np.array_equal(v.fold_jacobian(ctx, r2[None], group), got) and np.array_equal(v.fold_jacobian(ctx, r1[None], group), part)
            ctx.dfree(d_s)
        assert ctx.stat("msm_bucket_sets") == 1 and ctx.stat("msm_window_bits") == window_bits
    finally:
        B.free()


@pytest.mark.parametrize("precompute", [False, True])
def test_msm_lds_sort_path_ragged_sizes(ctx, cref, precompute):
    """n above the LDS-counting-sort threshold and not a power of two (ragged last chunk), dense and boolean-heavy scalars,
    whole range and an unaligned sub-range, plain and precomputed bases -- all bit-exact vs the oracle's BDLO12."""
    n = 50001
    bases = cref.g1_batch_mul_gen(rand_fr_array(n, seed=70))
    B = ctx.upload_bases(bases, 1)
    if precompute:
        B.precompute(0)
    rng = np.random.default_rng(8)
    dense = rand_fr_array(n, seed=71)
    sparse = rand_fr_array(n, seed=72); m = rng.random(n) < 0.9; sparse[m] = 0; sparse[m, 0] = rng.integers(0, 2, size=int(m.sum()), dtype=np.uint64)
    try:
        for name, ss in (("dense", dense), ("sparse", sparse)):
            d_s = ctx.to_device(ss)
            got, inf = B.msm(d_s)
            assert not inf and np.array_equal(got, cref.msm_g1(bases, ss, mixed=True)), name
            a, b = 333, 333 + 40000
            part, _ = B.msm(d_s + 32 * a, n=b - a, first=a)
            assert np.array_equal(part, cref.msm_g1(bases[a:b], ss[a:b], mixed=True)), name
            ctx.dfree(d_s)
    finally:
        B.free()


def test_msm_g2_lds_sort_path(ctx, cref):
    """G2 at a size that takes the LDS counting sort (the benchmark's G2 sizes are otherwise unchecked)"""
    n = 33000
    bases = cref.g2_batch_mul_gen(rand_fr_array(n, seed=80))
    ss = rand_fr_array(n, seed=81)
    ss[:5000] = 0; ss[:5000, 0] = 1
    exp = cref.msm_g2(bases, ss, mixed=True)
    assert np.array_equal(v.multiexp(ctx, bases, ss, 2), exp)
    B = ctx.upload_bases(bases, 2).precompute(0)
    d_s = ctx.to_device(ss)
    try:
        got, _ = B.msm(d_s)
        assert np.array_equal(got, exp)
    finally:
        ctx.dfree(d_s); B.free()


def test_msm_g2_sharded_jacobian_fold(ctx, cref):
    """the sharded MSM's exchange record for G2 (288-byte Jacobian) and its fold (BASELINE config 5 shards G2 as well)"""
    n = 1200
    bases = cref.g2_batch_mul_gen(rand_fr_array(n, seed=90))
    ss = rand_fr_array(n, seed=91)
    B = ctx.upload_bases(bases, 2)
    d_s = ctx.to_device(ss)
    try:
        cuts = [0, 400, 401, n]
        recs = [B.msm_jacobian(d_s + 32 * a, n=b - a, first=a) for a, b in zip(cuts[:-1], cuts[1:])]
        assert recs[0].shape == (36,)
        assert np.array_equal(v.fold_jacobian(ctx, np.stack(recs), 2), cref.msm_g2(bases, ss))
        assert not v.fold_jacobian(ctx, np.zeros((0, 36), np.uint64), 2).any()          # empty fold = infinity
    finally:
        ctx.dfree(d_s); B.free()


def test_msm_randomized_configurations(ctx, cref):
    """30 seeded random configurations across the code paths: size (global-atomic and LDS sort paths, ragged), scalar
    distribution, forced window size, plain / precomputed bases, whole range / sub-range -- each bit-exact vs the oracle."""
    rng = np.random.default_rng(20260101)
    nmax = 70000
    all_bases = cref.g1_batch_mul_gen(rand_fr_array(nmax, seed=123))
    for case in range(30):
        n = int(rng.choice([1, 2, 63, 64, 65, 1000, 4095, 4096, 4097, 32767, 32768, 33001, int(rng.integers(2, nmax))]))
        kind = rng.choice(["dense", "boolean", "small", "equal", "mostly_zero"])
        ss = rand_fr_array(n, seed=1000 + case)
        if kind == "boolean":
            m = rng.random(n) < 0.85; ss[m] = 0; ss[m, 0] = rng.integers(0, 2, size=int(m.sum()), dtype=np.uint64)
        elif kind == "small":
            ss[:] = 0; ss[:, 0] = rng.integers(0, 70000, size=n, dtype=np.uint64)
        elif kind == "equal":
            ss[:] = ss[0]
        elif kind == "mostly_zero":
            m = rng.random(n) < 0.97; ss[m] = 0
        pre = bool(rng.integers(0, 2))
        first = int(rng.integers(0, nmax - n + 1))
        bases = all_bases[first:first + n]
        B = ctx.upload_bases(all_bases, 1)
        forced = 0
        try:
            if pre:
                B.precompute(int(rng.choice([0, 8, 11, 16])))
            else:
                forced = int(rng.choice([0, 0, 5, 9, 12, 16]))
                ctx.set_option("msm_window_bits", forced)
            d_s = ctx.to_device(ss)
            got, _ = B.msm(d_s, n=n, first=first)
            ctx.dfree(d_s)
            exp = cref.msm_g1(bases, ss, mixed=True)
            assert np.array_equal(got, exp), dict(case=case, n=n, kind=str(kind), pre=pre, first=first, forced=forced)
        finally:
            ctx.set_option("msm_window_bits", 0)
            B.free()


def _dlog_identity(ks, ss):
    """sum_i k_i s_i mod r through numpy object arrays (8 M terms in a few seconds)"""
    from conftest import fr_ints_fast
    import numpy as np
    k = np.array(fr_ints_fast(ks), dtype=object); s = np.array(fr_ints_fast(ss), dtype=object)
    return int((k * s).sum() % o.R)


def test_msm_g1_2p23_config5_shard_plain_bases(ctx, cref):
    """BASELINE config 5, the per-GPU share of the G1 half: 2^26 points over 8 GPUs = 2^23 points per rank, PLAIN resident bases
    (no window-multiple table).  sum_i s_i (k_i G) = (sum_i k_i s_i) G on the whole shard, a 2^15-point slice bit for bit against
    the C oracle's BDLO12, and the Jacobian record the rank would send folds to the same point."""
    n = 1 << 23
    ks = rand_fr_array(n, seed=101); ss = rand_fr_array(n, seed=102)
    ss[1000:3000] = 0; ss[2000:3000, 0] = 1
    d_k = ctx.to_device(ks); d_s = ctx.to_device(ss)
    d_b = v.fixed_base_mul(ctx, d_k, n, 1)
    ctx.dfree(d_k)
    B = ctx.bases_from_device(d_b, n, 1)
    try:
        got, inf = B.msm(d_s)
        e = _dlog_identity(ks, ss)
        assert not inf and np.array_equal(got, cref.g1_mul(g1_limbs(o.G1.gen), L(e, 4)))
        assert ctx.stat("msm_bucket_sets") == ctx.stat("msm_windows")            # plain bases: one bucket set per window
        rec = B.msm_jacobian(d_s)
        assert np.array_equal(v.fold_jacobian(ctx, rec[None], 1), got)
        m, at = 1 << 15, (1 << 22) + 12345
        host_b = np.zeros((m, 12), np.uint64); ctx.d2h(host_b, d_b + 96 * at)
        part, _ = B.msm(d_s + 32 * at, n=m, first=at)
        assert np.array_equal(part, cref.msm_g1(host_b, ss[at:at + m]))
    finally:
        B.free(); ctx.dfree(d_b); ctx.dfree(d_s)


@pytest.mark.parametrize("group,n", [(1, 6000), (2, 3000)])
def test_msm_window_multiples_with_the_endomorphism_rows(ctx, cref, group, n):
    """vsp_bases_precompute_split: the table of window multiples for DENSE scalars -- (2^(cw) P, phi(2^(cw) P)) for the ceil(128 / c) windows
    of a split scalar, all windows sharing one bucket set.  Against the C oracle for several window sizes, with a sub-range of the
    resident bases, crowded buckets, zeros and ones; bases outside the subgroup keep the ordinary table and the exact sum."""
    ks, ss = rand_fr_array(n, seed=7), rand_fr_array(n, seed=8)
    ss[0] = L(o.R - 1, 4); ss[1] = L(1, 4); ss[2] = 0; ss[100:400] = ss[99]; ss[500:900] = 0; ss[500:900:2, 0] = 1
    b = cref.g1_batch_mul_gen(ks) if group == 1 else cref.g2_batch_mul_gen(ks)
    fn = cref.msm_g1 if group == 1 else cref.msm_g2
    want = fn(b, ss)
    for wb in (8, 13, 16, 19, 0):
        B = ctx.upload_bases(b, group).precompute(wb, split=True); d_s = ctx.to_device(ss)
        try:
            got, _ = B.msm(d_s)
            assert np.array_equal(got, want), wb
            assert ctx.stat("msm_endomorphism_split") == 1 and ctx.stat("msm_bucket_sets") == 1
            if wb:
                assert ctx.stat("msm_windows") == (128 + wb - 1) // wb
            part, _ = B.msm(d_s + 32 * 123, n - 500 - 123, 123)
            assert np.array_equal(part, fn(b[123:n - 500], ss[123:n - 500])), wb
        finally:
            B.free(); ctx.dfree(d_s)
    if group == 1:                                                 # a point with a cofactor component: no endomorphism rows, same exact sum
        from test_gpu_subgroup import curve_point_outside_g1
        b2 = b.copy(); b2[5] = g1_limbs(curve_point_outside_g1(7))
        B = ctx.upload_bases(b2, 1).precompute(13, split=True); d_s = ctx.to_device(ss)
        try:
            got, _ = B.msm(d_s)
            assert ctx.stat("msm_endomorphism_split") == 0 and np.array_equal(got, cref.msm_g1(b2, ss))
        finally:
            B.free(); ctx.dfree(d_s)


@pytest.mark.parametrize("window_bits", [0, 20, 22])
def test_msm_staged_sort_at_scale_with_crowded_buckets(ctx, cref, window_bits):
    """2^22 points through the staged sort (k_ms_*) with scalars that crowd it: half of them ONE value (a bucket of 2^21 entries in every
    window: hundreds of pieces in the second and third splitting passes, thousands of parts in the heavy-bucket merge), a quarter below
    2^6 (the low window's first segment holds 2^20 entries in a few buckets), zeros, ones, the rest uniform.  The discrete-log identity
    on the whole problem; the policy's window (17 bits, scalars folded: 15 windows) and two wider ones."""
    n = 1 << 22
    ks = rand_fr_array(n, seed=301); ss = rand_fr_array(n, seed=302)
    ss[: n // 2] = ss[n // 2]
    ss[n // 2: 3 * n // 4] = 0; ss[n // 2: 3 * n // 4, 0] = np.arange(n // 4, dtype=np.uint64) % 64
    ss[3 * n // 4: 3 * n // 4 + 65536] = 0; ss[3 * n // 4 + 32768: 3 * n // 4 + 65536, 0] = 1
    d_k = ctx.to_device(ks); d_s = ctx.to_device(ss)
    d_b = v.fixed_base_mul(ctx, d_k, n, 1)
    ctx.dfree(d_k)
    ctx.set_option("msm_census_sync", 1)                          # plan from this vector's own 0 / 1 census
    B = ctx.bases_from_device(d_b, n, 1)
    ctx.set_option("msm_window_bits", window_bits)
    try:
        got, inf = B.msm(d_s)
        e = _dlog_identity(ks, ss)
        assert not inf and np.array_equal(got, cref.g1_mul(g1_limbs(o.G1.gen), L(e, 4)))
        assert ctx.stat("msm_window_bits") == (window_bits or 17)
        if not window_bits:
            assert ctx.stat("msm_windows") == 15
    finally:
        ctx.set_option("msm_window_bits", 0); ctx.set_option("msm_census_sync", 0)
        B.free(); ctx.dfree(d_b); ctx.dfree(d_s)


def test_msm_g2_2p21_config5_shard_plain_bases(ctx, cref):
    """BASELINE config 5, the per-GPU share of the G2 half: 2^24 points over 8 GPUs = 2^21 per rank, plain bases; the identity on
    the whole shard, a 2^12-point slice against the C oracle, and the 288-byte Jacobian record."""
    n = 1 << 21
    ks = rand_fr_array(n, seed=111); ss = rand_fr_array(n, seed=112)
    d_k = ctx.to_device(ks); d_s = ctx.to_device(ss)
    d_b = v.fixed_base_mul(ctx, d_k, n, 2)
    ctx.dfree(d_k)
    B = ctx.bases_from_device(d_b, n, 2)
    try:
        got, inf = B.msm(d_s)
        e = _dlog_identity(ks, ss)
        assert not inf and np.array_equal(got, cref.g2_mul(g2_limbs(o.G2.gen), L(e, 4)))
        rec = B.msm_jacobian(d_s)
        assert rec.shape == (36,) and np.array_equal(v.fold_jacobian(ctx, rec[None], 2), got)
        m, at = 1 << 12, (1 << 20) + 777
        host_b = np.zeros((m, 24), np.uint64); ctx.d2h(host_b, d_b + 192 * at)
        part, _ = B.msm(d_s + 32 * at, n=m, first=at)
        assert np.array_equal(part, cref.msm_g2(host_b, ss[at:at + m]))
    finally:
        B.free(); ctx.dfree(d_b); ctx.dfree(d_s)


@pytest.mark.gpu
@pytest.mark.parametrize("option,value", [("msm_dimbits", 0), ("msm_dimbits", 1), ("msm_dimsum_lanes", 8), ("msm_dimsum_lanes", 16),
                                          ("msm_dimsum_lanes", 32), ("msm_dimsum_lanes", 64), ("msm_glv", 0), ("msm_glv", 2), ("msm_fp28", 0),
                                          ("msm_dimsum_prefetch", 1)])
def test_msm_same_result_under_every_kernel_variant_option(cref, option, value):
    """the tuning options of include/vsp.h pick kernel variants (lanes per bucket-digit sum, bit-decomposed or weighted last step of the
    bucket reduction, endomorphism split, 28-bit or 12 x 32-bit accumulation): every variant gives the oracle's point, on plain and on
    precomputed bases, dense and boolean-heavy scalars, both groups"""
    c = v.Context(0)
    try:
        c.set_option(option, value)
        for group, n in ((1, 3000), (1, 40000), (2, 2500), (2, 33000)):
            if (group, n) not in _VARIANT_CASES:               # inputs and the oracle's answers once for all options
                ks = rand_fr_array(n, seed=600 + n)
                bases = (cref.g1_batch_mul_gen if group == 1 else cref.g2_batch_mul_gen)(ks)
                dense = rand_fr_array(n, seed=700 + n)
                sparse = dense.copy(); m = np.arange(n) % 10 != 0; sparse[m] = 0; sparse[m, 0] = (np.arange(n)[m] & 1).astype(np.uint64)
                ref = cref.msm_g1 if group == 1 else cref.msm_g2
                _VARIANT_CASES[(group, n)] = (bases, [(ss, ref(bases, ss, mixed=True)) for ss in (dense, sparse)])
            bases, cases = _VARIANT_CASES[(group, n)]
            for precompute in (False, True):
                B = c.upload_bases(bases, group)
                if precompute:
                    B.precompute(0)
                for ss, exp in cases:
                    d_s = c.to_device(ss)
                    got, _ = B.msm(d_s)
                    c.dfree(d_s)
                    assert np.array_equal(got, exp), (option, value, group, n, precompute)
                B.free()
    finally:
        c.close()


_VARIANT_CASES = {}


@pytest.mark.parametrize("group,n,K,kinds", [(1, 1, 3, "uniform"), (1, 300, 4, "mixed"), (1, 3000, 5, "mixed"), (1, 40000, 8, "mixed"), (1, 70000, 3, "boolean"),
                                             (2, 257, 3, "mixed"), (2, 2500, 4, "mixed"), (1, 2000, 16, "boolean")])
def test_batch_of_scalar_vectors_over_one_set_of_bases(ctx, cref, group, n, K, kinds):
    """vsp_msm_resident_batch (round 4): K scalar vectors over the same resident bases in ONE pass -- one sort, one accumulation, one bucket
    reduction over K x windows bucket sets -- must give, vector by vector, what K separate multi-exponentiations give (the C oracle's
    multiexp_with_mixed_addition): uniform, 90 % boolean, all-equal, all-zero and edge-valued vectors mixed in one batch, a stride
    larger than n, with and without the endomorphism split, a sub-range of the bases"""
    bases = (cref.g1_batch_mul_gen if group == 1 else cref.g2_batch_mul_gen)(rand_fr_array(n + 5, 500 + n))
    msm = cref.msm_g1 if group == 1 else cref.msm_g2
    stride = n + 7
    rng = np.random.default_rng(n + K)
    vecs = np.zeros((K, stride, 4), np.uint64)
    for k in range(K):
        ss = rand_fr_array(n, 600 + 10 * n + k)
        kind = kinds if kinds != "mixed" else ("uniform", "boolean", "equal", "zero", "edges")[k % 5]
        if kind == "boolean":
            m = rng.random(n) < 0.9; ss[m] = 0; ss[m, 0] = rng.integers(0, 2, size=int(m.sum()), dtype=np.uint64)
        elif kind == "equal":
            ss[:] = ss[0]
        elif kind == "zero":
            ss[:] = 0
        elif kind == "edges":
            for i in range(0, n, 3): ss[i] = L(o.R - 1 - (i % 3), 4)
            for i in range(1, n, 5): ss[i] = 0
        vecs[k, :n] = ss
        vecs[k, n:] = rand_fr_array(stride - n, 7)            # what lies between two vectors must not matter
    d_s = ctx.to_device(vecs.reshape(-1, 4))
    for glv in (1, 0):
        ctx.set_option("msm_glv", glv)
        B = ctx.upload_bases(bases, group)
        for first in (0, 3):
            cnt = n if first == 0 else n - 1
            got, inf = B.msm_batch(d_s, K, n=cnt, first=first, stride=stride)
            for k in range(K):
                exp = msm(bases[first:first + cnt], vecs[k, :cnt], mixed=True)
                assert np.array_equal(got[k], exp), (group, n, K, glv, first, k)
                assert bool(inf[k]) == (not exp.any())
        # the same batch over the table of window multiples (end of round 4): ONE bucket set per vector, slices of the table per window
        if n >= 257:
            for wb in (12, 16) if n >= 2000 else (9,):
                B.precompute(wb, split=bool(glv))
                for first in (0, 3):
                    cnt = n if first == 0 else n - 1
                    got, inf = B.msm_batch(d_s, K, n=cnt, first=first, stride=stride)
                    for k in range(K):
                        exp = msm(bases[first:first + cnt], vecs[k, :cnt], mixed=True)
                        assert np.array_equal(got[k], exp), (group, n, K, glv, first, k, "table", wb)
                        assert bool(inf[k]) == (not exp.any())
                B.free(); B = ctx.upload_bases(bases, group)
        B.free()
    ctx.set_option("msm_glv", 1)
    ctx.dfree(d_s)


@pytest.mark.parametrize("fp28", [1, 0])
def test_prefetching_bucket_reduction_on_the_inputs_that_exposed_the_round3_finding(cref, fp28):
    """option msm_dimsum_prefetch = 1 (k_dimsum_mixed with the next bucket record in flight): the loop round 3 measured WRONG on inputs whose
    additions take the doubling path -- one point many times, two points, 34 / 200 / 300 entries, 5..13-bit windows (tools/dimsum_prefetch_probe.py)
    -- because the toolchain's machine scheduler corrupted the compiler's liveness.  Built without that scheduler it must
    give the oracle's point on exactly those inputs, on the generic and on the 28-bit form (n >= 1024 gets a 28-bit table)."""
    c = v.Context(0)
    try:
        c.set_option("msm_dimsum_prefetch", 1); c.set_option("msm_fp28", fp28)
        one = cref.g1_batch_mul_gen(rand_fr_array(1, seed=91))
        many = cref.g1_batch_mul_gen(rand_fr_array(300, seed=93))
        makers = {"one point": lambda n: np.repeat(one, n, axis=0), "distinct": lambda n: np.concatenate([many] * (n // 300 + 1))[:n].copy(),
                  "two points": lambda n: np.concatenate([np.repeat(one, n // 2, axis=0), np.repeat(many[:1], n - n // 2, axis=0)])}
        for label, mk in makers.items():
            for n, wb in ((34, 8), (34, 5), (200, 8), (200, 11), (300, 13), (1500, 8), (1500, 11), (4096, 9)):
                bases = mk(n); ss = rand_fr_array(n, seed=92 + n)
                c.set_option("msm_window_bits", wb)
                B = c.upload_bases(bases, 1); d_s = c.to_device(ss)
                got, _ = B.msm(d_s)
                assert np.array_equal(got, cref.msm_g1(bases, ss, mixed=True)), (label, n, wb, fp28)
                B.free(); c.dfree(d_s)
    finally:
        c.close()
