"""BASELINE config 1 counterpart.  The reference's `cli --mode encrypted_input` times one vote phase --
circuit synthesis is not rebuilt here, so the stand-in is one Groth16 proof over a synthetic SAVER-shaped R1CS (30 public inputs,
90 % boolean wires) -- and prints `Vote Phase Time_execution: <n>ms` (bin/cli/src/main.cpp:449-456).  This prints the same line for
the CPU restatement of the reference's prover (oracle, one thread) and, when a GPU is present, for vsp_groth16_prove on the same
instance, checking that the two proofs are identical.   python tools/vote_phase_time.py [--log-constraints 14]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cref
from conftest import rand_fr_array

ap = argparse.ArgumentParser()
ap.add_argument("--log-constraints", type=int, default=14)
args = ap.parse_args()
ni = 30
nc = (1 << args.log_constraints) - ni - 2
cs, wit = cref.R1CS.synth(nc, ni, 4)
tox = rand_fr_array(5, seed=5)
r, s = rand_fr_array(2, seed=6)
kp = cref.Keypair(cs, tox)
t0 = time.perf_counter()
ea, eb, ec = kp.prove(wit, r, s)
print("[CPU restatement, 1 thread, %d constraints] Vote Phase Time_execution: %dms" % (nc, round((time.perf_counter() - t0) * 1e3)))
try:
    import vote_saver_protocol_amd as v
    ctx = v.Context(0)
except Exception as e:                                   # no GPU / library: the CPU line is all there is
    print("(no GPU path here: %s)" % e)
    sys.exit(0)
dcs = v.R1CS(ctx, nc, ni, cs.num_vars, *cs.export())
q = [ctx.upload_bases(kp.part(n), g).precompute(0) for n, g in (("A_query", 1), ("B_query_g1", 1), ("B_query_g2", 2), ("H_query", 1), ("L_query", 1))]
pk = v.ProvingKey(ctx, kp.part("alpha_g1")[0], kp.part("beta_g1")[0], kp.part("beta_g2")[0], kp.part("delta_g1")[0], kp.part("delta_g2")[0], *q)
v.groth16_prove(ctx, dcs, pk, wit, r, s)                 # first call builds twiddles and workspaces
t0 = time.perf_counter()
pa, pb, pc, _ = v.groth16_prove(ctx, dcs, pk, wit, r, s)
dt = time.perf_counter() - t0
print("[MI355X, vsp_groth16_prove, same instance] Vote Phase Time_execution: %dms  (%.2f ms)" % (round(dt * 1e3), dt * 1e3))
print("proofs identical:", bool(np.array_equal(pa, ea) and np.array_equal(pb, eb) and np.array_equal(pc, ec)))
