"""Single-proof latency of the small reference circuits: glp_prove (host transcript, a round trip per Fiat-Shamir step) against the lock-step batch path
with K = 1 (transcripts on the device, one copy back).  usage: python profiles/single_vs_batch1.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plonky2_lib_amd as glp
import plonky2_lib_amd.synth as synth
from plonky2_lib_amd import gadgets
ctx = glp.Context(0)
tree = gadgets.SparseMerkleTree()
for k, v in ((1, 2), (12, 1), (5, 51)):
    tree.insert(gadgets.hash_out_from_u128(k), gadgets.hash_out_from_u128(v))
cases = {"zkdsa 2^3": synth.zkdsa_circuit(3), "smt inclusion 2^7": gadgets.smt_inclusion_circuit(tree, gadgets.hash_out_from_u128(5)),
         "keccak256 1 block 2^13": gadgets.keccak256_circuit(b"abc"), "arith 2^12": synth.arith_circuit(12, synth.Config.standard_recursion_config(), seed=2),
         "arith 2^16 x 136": synth.arith_circuit(16, synth.Config.standard_ecc_config(), seed=16)}
for name, d in cases.items():
    gc = glp.Circuit(ctx, d)
    w = np.ascontiguousarray(d.wires)[None]
    pi = np.ascontiguousarray(d.public_inputs)[None] if len(d.public_inputs) else None
    p1 = gc.prove(); pb = gc.prove_batch(w, pi)
    assert (pb[0] == p1).all()
    reps = 30
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): gc.prove()
    t1 = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): gc.prove_batch(w, pi)
    t2 = (time.perf_counter() - t0) / reps
    print("%-24s glp_prove %7.3f ms | glp_prove_batch(K = 1) %7.3f ms" % (name, t1 * 1e3, t2 * 1e3), flush=True)
    gc.free()
