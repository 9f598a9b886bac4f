"""round 4 diagnostic: which leg of the context-time known-answer check of the 28-bit kernels differs (stat msm_fp28_selfcheck_detail_*)"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vote_saver_protocol_amd as v, cref
from conftest import rand_fr_array
ctx = v.Context(0)
b1 = cref.g1_batch_mul_gen(rand_fr_array(2048, seed=5)); b2 = cref.g2_batch_mul_gen(rand_fr_array(1100, seed=6))
B1 = ctx.upload_bases(b1, 1); B2 = ctx.upload_bases(b2, 2)
for g in (1, 2):
    print("group", g, "selfcheck", ctx.stat("msm_fp28_selfcheck_g%d" % g), "detail", ctx.stat("msm_fp28_selfcheck_detail_g%d" % g))
