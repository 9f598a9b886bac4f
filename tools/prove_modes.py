"""prove latency at 2^20 constraints for different per-query precompute masks (diagnostic):  python tools/prove_modes.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cref, bls12_381 as o
import vote_saver_protocol_amd as v
lg = int(os.environ.get("LOG_M", "20"))
ni = 30; nc = (1 << lg) - ni - 2
ctx = v.Context(0)
if os.environ.get("VSP_MSM_GLV"):
    ctx.set_option("msm_glv", int(os.environ["VSP_MSM_GLV"]))
gen = o.splitmix64(5)
cs, wit = cref.R1CS.synth(nc, ni, 4, ballot=(25, 7))
tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
r = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64); s = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64)
witp = ctx.host_register(np.ascontiguousarray(wit))
ref = None
for name, mask in (("recommended (all but H)", 1), ("plain", 0), ("all five", 2 | 4 | 8 | 16 | 32)):
    kp = v.Keypair(ctx, dcs, tox, precompute=mask)
    for pf in (1, 0):
        ctx.set_option("prove_plan_first", pf)
        w = witp
        out = v.groth16_prove(ctx, dcs, kp.pk, w, r, s)
        ctx.stats_reset()
        t0 = time.perf_counter()
        for _ in range(6):
            out = v.groth16_prove(ctx, dcs, kp.pk, w, r, s)
        dt = (time.perf_counter() - t0) / 6 * 1e3
        ph = {k: round(ctx.stat("prove_" + k + "_ms") / 6, 2) for k in ("launch", "host_overlap", "wait", "assembly")}
        ref = ref or out[3]
        print("%-24s plan_first=%d: %.2f ms  %s  same proof: %s" % (name, pf, dt, ph, out[3] == ref), flush=True)
    kp.free()
