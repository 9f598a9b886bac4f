"""GPU: the endomorphism split's precondition.  phi(P) = lambda P holds only on the order-r subgroup; a raw-limb C
ABI can be handed curve points with a cofactor component, for which a split multi-exponentiation would return a point different from
sum k_i P_i -- what the reference's generic multiexp (and this library's plain pipeline) computes.  Uploads therefore check
phi(P) = lambda P per point wherever the split would be used, and bases that fail keep the plain layout: the result is the same point
either way.  Checked against the naive big-integer sum and the C oracle's generic BDLO12."""
import numpy as np
import pytest

import bls12_381 as o
from conftest import fr_ints, g1_limbs, g2_limbs, rand_fr_array

import vote_saver_protocol_amd as v

pytestmark = pytest.mark.gpu


def curve_point_outside_g1(seed):
    x = seed
    while True:
        x += 1
        y = o.fp_sqrt((x * x * x + 4) % o.P)
        if y is not None and not o.G1.in_subgroup((x, y)):
            return (x, y)


def curve_point_outside_g2(seed):
    x0 = seed
    while True:
        x0 += 1
        x = (x0, 1)
        rhs = o.Fp2Ops.add(o.Fp2Ops.mul(o.Fp2Ops.sqr(x), x), (4, 4))
        y = o.fp2_sqrt(rhs)
        if y is not None and not o.G2.in_subgroup((x, y)):
            return (x, y)


def test_curve_points_outside_the_subgroup_give_the_same_sum_with_and_without_the_split(cref):
    n = 2000
    ks, ss = rand_fr_array(n, 61), rand_fr_array(n, 62)
    for group in (1, 2):
        m = n if group == 1 else 1100
        bases = (cref.g1_batch_mul_gen if group == 1 else cref.g2_batch_mul_gen)(ks[:m])
        good = bases.copy()
        outside = [curve_point_outside_g1(7), curve_point_outside_g1(1000)] if group == 1 else [curve_point_outside_g2(7), curve_point_outside_g2(1000)]
        for idx, pt in zip((5, m - 3), outside):
            bases[idx] = g1_limbs(pt) if group == 1 else g2_limbs(pt)
        ref = cref.msm_g1 if group == 1 else cref.msm_g2
        want = ref(bases, ss[:m])
        # the generic oracle against the definition, on the two foreign points and two ordinary ones (the rest is covered elsewhere)
        Cv, from_l, to_l = (o.G1, o.g1_from_limbs, g1_limbs) if group == 1 else (o.G2, o.g2_from_limbs, g2_limbs)
        sel = [5, m - 3, 0, 9]
        naive = Cv.msm_naive([from_l(bases[i]) for i in sel], fr_ints(ss[sel]))
        assert np.array_equal(ref(bases[sel], ss[sel]), to_l(naive))
        results = {}
        for mode in ("default", "glv_off", "check_off", "strict", "forced_on_good", "host_buffers"):
            with v.Context(0) as c:
                if mode == "glv_off":
                    c.set_option("msm_glv", 0)
                if mode == "check_off":
                    c.set_option("bases_check_subgroup", 0)
                if mode == "strict":
                    c.set_option("bases_check_subgroup", 2)
                    with pytest.raises(v.VspError, match="subgroup"):
                        c.upload_bases(bases, group)
                    B = c.upload_bases(good, group)                    # the same call accepts subgroup points
                    assert c.stat("bases_outside_subgroup") == 1 and c.stat("bases_subgroup_checks") == 2
                    B.free()
                    continue
                if mode == "host_buffers":                             # vsp_msm_g1 / _g2: one call's bases, plain layout, no check
                    results[mode] = v.multiexp(c, bases, ss[:m], group)
                    assert c.stat("bases_subgroup_checks") == 0 and c.stat("msm_endomorphism_split") == 0
                    continue
                if mode == "forced_on_good":                           # "msm_glv" = 2: the caller vouches, no check -- right for subgroup points
                    c.set_option("msm_glv", 2); c.set_option("bases_check_subgroup", 0)
                    B = c.upload_bases(good, group); d_s = c.to_device(ss[:m])
                    got, _ = B.msm(d_s)
                    assert c.stat("msm_endomorphism_split") == 1 and c.stat("bases_subgroup_checks") == 0
                    assert np.array_equal(got, ref(good, ss[:m]))
                    c.dfree(d_s); B.free()
                    continue
                B = c.upload_bases(bases, group); d_s = c.to_device(ss[:m])
                results[mode], _ = B.msm(d_s)
                split = c.stat("msm_endomorphism_split")
                if mode == "default":                                  # checked, found foreign points, kept the plain layout
                    assert c.stat("bases_subgroup_checks") == 1 and c.stat("bases_outside_subgroup") == 1 and split == 0
                    G = c.upload_bases(good, group)                    # subgroup points pass the check and get the split
                    got, _ = G.msm(d_s)
                    assert c.stat("bases_subgroup_checks") == 2 and c.stat("bases_outside_subgroup") == 1 and c.stat("msm_endomorphism_split") == 1
                    assert np.array_equal(got, ref(good, ss[:m]))
                    G.free()
                else:
                    assert c.stat("bases_subgroup_checks") == 0 and split == 0
                c.dfree(d_s); B.free()
        for mode, got in results.items():
            assert np.array_equal(got, want), (group, mode)


def test_known_answer_check_runs_on_library_points_not_on_caller_data(cref):
    """a context whose FIRST upload holds curve points outside the subgroup still passes the kernels' known-answer check (it runs over
    multiples of the generator the library computes itself) and keeps the 28-bit kernels"""
    n = 1500
    bases = cref.g1_batch_mul_gen(rand_fr_array(n, 71))
    for i in range(0, n, 100):
        bases[i] = g1_limbs(curve_point_outside_g1(50 + i))
    ss = rand_fr_array(n, 72)
    with v.Context(0) as c:
        B = c.upload_bases(bases, 1); d_s = c.to_device(ss)
        got, _ = B.msm(d_s)
        assert c.stat("msm_fp28_selfcheck_g1") == 1.0 and c.stat("bases_outside_subgroup") == 1
        assert np.array_equal(got, cref.msm_g1(bases, ss))
        c.dfree(d_s); B.free()
