"""Configs 2 and 4 (Keccak-shaped 2^15 rows, SMT-shaped 2^12 rows): single-proof latency vs glp_prove_batch throughput on one GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky2_lib_amd as glp, plonky2_lib_amd.synth as synth
ctx = glp.Context(0)
for name, desc, Ks in (("zkdsa 2^3", synth.zkdsa_circuit(3), (1, 64, 256)), ("SMT gate mix 2^12 (config 4)", synth.smt_shape_circuit(12), (1, 8, 32, 64)),
                       ("Keccak / u32 gate set 2^15 (config 2)", synth.keccak_shape_circuit(15), (1, 4, 8, 16))):
    gc = glp.Circuit(ctx, desc)
    gc.prove()
    t = time.perf_counter(); n = 10
    for _ in range(n): gc.prove()
    single = (time.perf_counter() - t) / n
    print("%-40s single proof %.2f ms (%.0f proofs/s)" % (name, single * 1e3, 1 / single))
    for K in Ks:
        w = np.ascontiguousarray(np.stack([desc.wires] * K)); pis = np.stack([desc.public_inputs] * K)
        d = ctx.dev_alloc(w.nbytes); ctx.dev_upload(d, w)
        out = gc.prove_batch_device(d, K, pis)
        t = time.perf_counter(); n = 5
        for _ in range(n): out = gc.prove_batch_device(d, K, pis)
        dt = (time.perf_counter() - t) / n
        ok = all(gc.verify(p) for p in out[:2])
        print("    batch of %3d (witnesses resident): %.2f ms per batch, %.0f proofs/s, verified %s" % (K, dt * 1e3, K / dt, ok))
        ctx.dev_free(d)
    gc.free()
