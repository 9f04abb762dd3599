"""Where the limb launch of the real secp256k1 circuit spends its time: the quotient stage with one component of k_quotient_limbs compiled
out at a time (-DLIMB_NO_CMP / _NO_HEADS / _NO_RP / _NO_ACC builds of prover.hip; the proofs of those builds are of course invalid --
timing only).  The circuit is built once and handed over through a hand-off file."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky2_lib_amd as glp
path = "/tmp/ecdsa10.glpc"
if not os.path.exists(path):
    from plonky2_lib_amd import gadgets_ecdsa as E
    c = E.ecdsa_circuit(E.random_signatures(10, seed=1))
    glp.write_circuit_file(path, c)
ctx = glp.Context(0)
with glp.CircuitFile(path) as cf:
    gc = glp.Circuit(ctx, cf.desc)
    w = np.ascontiguousarray(cf.desc.wires)
    d = ctx.dev_alloc(w.nbytes); ctx.dev_upload(d, w)
    gc.prove_device(d)
    ctx.set_profiling(True); ctx.stage_reset()
    for _ in range(3): gc.prove_device(d)
    ctx.synchronize()
    q = [ms for k, ms, _ in ctx.stages() if k == "quotient_eval"]
    print("%-10s quotient_eval %.2f ms" % (sys.argv[1] if len(sys.argv) > 1 else "base", sum(q) / len(q)), flush=True)
