"""KeccakGoldilocksConfig vs PoseidonGoldilocksConfig on one GPU: the Keccak-256 circuit's gate set at 2^15 rows (the circuit the reference
proves under this config [REF src/hash/keccak256.rs:281]) and the headline 2^20 secp256k1 shape, single proofs with resident witnesses;
plus the commitment alone (135 columns, 2^20 rows, rate 8).  Every proof is checked by glp_verify."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonky2_lib_amd as glp, plonky2_lib_amd.synth as synth
ctx = glp.Context(0)
rng = np.random.default_rng(1)
vals = rng.integers(0, glp.P, (135, 1 << 20), dtype=np.uint64)
for h, name in ((0, "Poseidon"), (1, "KeccakHash<25>")):
    b = ctx.batch_from_values(vals, 3, 4, hasher=h); b.free()
    t = time.perf_counter()
    for _ in range(3):
        b = ctx.batch_from_values(vals, 3, 4, hasher=h); b.free()
    print("commit 135 x 2^20 (host values, rate 8)  %-15s %.1f ms" % (name, (time.perf_counter() - t) / 3 * 1e3), flush=True)
del vals
for cname, make, reps in (("Keccak-256 gate set 2^15", lambda: synth.keccak_shape_circuit(15), 10),
                          ("secp256k1 shape 2^20", lambda: synth.ecdsa_shape_circuit(20), 4)):
    for h, name in ((0, "Poseidon"), (1, "KeccakHash<25>")):
        desc = make()
        if h:
            desc.hasher, desc.circuit_digest = 1, None
        gc = glp.Circuit(ctx, desc)
        w = np.ascontiguousarray(desc.wires)
        d = ctx.dev_alloc(w.nbytes); ctx.dev_upload(d, w)
        p = gc.prove_device(d)
        t = time.perf_counter()
        for _ in range(reps):
            p = gc.prove_device(d)
        dt = (time.perf_counter() - t) / reps
        print("%-26s %-15s %.2f ms per proof, verified %s" % (cname, name, dt * 1e3, gc.verify(p)), flush=True)
        ctx.dev_free(d); gc.free()
