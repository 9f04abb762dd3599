"""The real secp256k1 ECDSA circuit (plonky2-lib_amd/gadgets_ecdsa.py) against the gate-mix stand-in of the same size: per-stage times of
one proof with a resident witness.  argv[1] = signatures (1 -> 2^17 rows, 2 -> 2^18, 4 -> 2^19, 10 -> 2^20)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky2_lib_amd as glp, plonky2_lib_amd.synth as synth
from plonky2_lib_amd import gadgets_ecdsa as E
nsig = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = glp.Context(0)
t = time.perf_counter()
real = E.ecdsa_circuit(E.random_signatures(nsig, seed=1))
print("built the %d-signature circuit in %.1f s: 2^%d rows (%d gate rows), %d gate kinds, %d selectors" %
      (nsig, time.perf_counter() - t, real.degree_bits, real.gadget_rows, len(real.gates), real.num_selectors), flush=True)
for k, v in real.gate_rows.items():
    print("    %-58s %8d rows" % (k, v))
for name, desc in (("real circuit", real), ("gate-mix stand-in (synth.ecdsa_shape_circuit)", synth.ecdsa_shape_circuit(real.degree_bits))):
    gc = glp.Circuit(ctx, desc)
    w = np.ascontiguousarray(desc.wires)
    d = ctx.dev_alloc(w.nbytes); ctx.dev_upload(d, w)
    p = gc.prove_device(d)
    t = time.perf_counter(); n = 5
    for _ in range(n): p = gc.prove_device(d)
    dt = (time.perf_counter() - t) / n
    print("%-48s %.2f ms per proof, verified %s" % (name, dt * 1e3, gc.verify(p)), flush=True)
    ctx.set_profiling(True); ctx.stage_reset(); gc.prove_device(d); ctx.synchronize()
    print("   ", {k: round(ms, 2) for k, ms, _ in ctx.stages() if ms > 0.3})
    ctx.set_profiling(False)
    ctx.dev_free(d); gc.free()
