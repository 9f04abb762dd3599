"""BASELINE config 2 as the real circuit (plonky2-lib_amd/gadgets.py): the reference's Keccak-256 gadget, 1 rate block (2^13 rows) and
4 rate blocks (2^15 rows), PoseidonGoldilocksConfig and KeccakGoldilocksConfig: single-proof latency (witness resident) and
glp_prove_batch throughput on one GPU.  Every proof verified; public inputs = the digests of [REF src/hash/keccak256.rs:196-212,256-277]."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky2_lib_amd as glp
from plonky2_lib_amd import gadgets
ctx = glp.Context(0)
rng = np.random.default_rng(3)
for blocks, Ks in ((1, (1, 8, 32)), (4, (1, 4, 16))):
    t = time.perf_counter()
    base = gadgets.keccak256_circuit(bytes(rng.integers(0, 256, 100 * blocks, dtype=np.uint8)), blocks_num=blocks)
    tb = time.perf_counter() - t
    print("%d-block circuit: 2^%d rows (%d gate rows), built with its witness in %.1f s of Python; operations %s"
          % (blocks, base.degree_bits, base.gadget_rows, tb, base.gate_ops), flush=True)
    for h, hname in ((0, "PoseidonGoldilocksConfig"), (1, "KeccakGoldilocksConfig")):
        base.hasher, base.circuit_digest = h, None
        gc = glp.Circuit(ctx, base)
        w = np.ascontiguousarray(base.wires)
        d = ctx.dev_alloc(w.nbytes); ctx.dev_upload(d, w)
        p = gc.prove_device(d, base.public_inputs)
        t = time.perf_counter(); n = 10
        for _ in range(n): p = gc.prove_device(d, base.public_inputs)
        dt = (time.perf_counter() - t) / n
        print("  %-26s single proof %.2f ms (%.0f proofs/s), verified %s" % (hname, dt * 1e3, 1 / dt, gc.verify(p)), flush=True)
        ctx.dev_free(d)
        for K in Ks[1:]:
            ws = np.ascontiguousarray(np.stack([base.wires] * K)); pis = np.stack([base.public_inputs] * K)
            dd = ctx.dev_alloc(ws.nbytes); ctx.dev_upload(dd, ws)
            out = gc.prove_batch_device(dd, K, pis)
            t = time.perf_counter(); n = 3
            for _ in range(n): out = gc.prove_batch_device(dd, K, pis)
            dt = (time.perf_counter() - t) / n
            print("      batch of %2d: %.1f ms, %.0f proofs/s, verified %s" % (K, dt * 1e3, K / dt, all(gc.verify(q) for q in out[:2])), flush=True)
            ctx.dev_free(dd)
        gc.free()
