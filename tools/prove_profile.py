"""20 proofs at 2^LOG_M constraints over a resident key (recommended precompute set), for rocprofv3 --kernel-trace --stats:
   cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 $GRAFT_REPO_ROOT/tools/prove_profile.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cref, bls12_381 as o
import vote_saver_protocol_amd as v
lg = int(os.environ.get("LOG_M", "20")); reps = int(os.environ.get("REPS", "20"))
ni = 30; nc = (1 << lg) - ni - 2
ctx = v.Context(0)
for kv in [x for x in os.environ.get("OPTS", "").split(",") if x]:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
gen = o.splitmix64(5)
cs, wit = cref.R1CS.synth(nc, ni, 4, ballot=(25, 7))
tox = np.array([o.int_to_limbs(o.rand_fr(gen), 4) for _ in range(5)], dtype=np.uint64)
dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
r = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64); s = np.array(o.int_to_limbs(o.rand_fr(gen), 4), np.uint64)
if os.environ.get("ZERO_ONES"):                      # experiment: how long does a proof take without the all-ones bucket? (not a valid witness)
    ones = (wit[:, 0] == 1) & (wit[:, 1] == 0) & (wit[:, 2] == 0) & (wit[:, 3] == 0)
    wit[ones] = 0
wit = ctx.host_register(np.ascontiguousarray(wit))
kp = v.Keypair(ctx, dcs, tox, precompute=int(os.environ.get("PRE", "1")))
v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
t0 = time.perf_counter()
for _ in range(reps):
    v.groth16_prove(ctx, dcs, kp.pk, wit, r, s)
print("%.2f ms per proof" % ((time.perf_counter() - t0) / reps * 1e3))
