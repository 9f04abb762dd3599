"""Prove the real secp256k1 circuit (argv[1] signatures) a few times: the workload for rocprofv3 counter runs on the quotient kernels."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky2_lib_amd as glp
from plonky2_lib_amd import gadgets_ecdsa as E
nsig = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = glp.Context(0)
c = E.ecdsa_circuit(E.random_signatures(nsig, seed=1))
gc = glp.Circuit(ctx, c)
w = np.ascontiguousarray(c.wires)
d = ctx.dev_alloc(w.nbytes); ctx.dev_upload(d, w)
for _ in range(reps):
    p = gc.prove_device(d)
print("verified", gc.verify(p))
