#!/usr/bin/env python3
"""Randomised parity campaign (a tool, not collected by pytest): random CircuitConfig / FriConfig fields, trace sizes,
gate mixes and public inputs; for each case the HIP library's proof must equal the CPU oracle's word for word, both
verifiers must accept it and reject a tampered copy; (round 2) the row-local witness generators on the GPU must rebuild a scrambled
witness exactly as the oracle's do, and with two challenges a two-proof glp_prove_batch (round 3: its transcripts on the device) must return the
single proofs; (round 3) glp_verify_batch must give glp_verify's verdicts and reasons, glp_proof_to_bytes the oracle serializer's bytes, and a witness staged
from page-locked memory with its advice columns derived on the GPU the proof of the oracle-derived witness.  Usage: python tests/fuzz_parity.py [cases] [seed]   (FUZZ_MIN_LG / FUZZ_MAX_LG bound the trace length, default 5..11)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import plonky2_lib_amd as glp
import plonky2_lib_amd.synth as synth
from oracle import oracle


def random_case(rng):
    rate_bits = int(rng.choice([1, 2, 3, 3, 3, 4]))
    qdf = int(rng.choice([q for q in (4, 8, 16) if q <= (1 << rate_bits)] or [1 << rate_bits]))
    if qdf < 4:
        rate_bits, qdf = 2, 4                      # the gate set has degree-4 range products
    lg = int(rng.integers(int(os.environ.get("FUZZ_MIN_LG", "5")), int(os.environ.get("FUZZ_MAX_LG", "11")) + 1))
    kw = dict(num_challenges=int(rng.choice([1, 2, 2, 2, 3, 4])), max_quotient_degree_factor=qdf, rate_bits=rate_bits,
              cap_height=int(rng.integers(0, min(6, lg + rate_bits) + 1)), proof_of_work_bits=int(rng.choice([0, 1, 8, 16, 18])),
              num_query_rounds=int(rng.integers(1, 30)), arity_bits=int(rng.integers(1, 5)), final_poly_bits=int(rng.integers(0, 6)))
    wide = bool(rng.random() < 0.7)
    config = synth.Config(136 if wide else 135, 80, **kw)
    if sum(config.reduction_arity_bits(lg)) > lg:      # plonky2 asserts degree_bits >= arity_bits in every reduction
        return random_case(rng)
    npi = int(rng.choice([0, 0, 1, 3, 8, 13]))
    pi = [int(x) for x in oracle.rand_field(rng, (npi,))]
    gate_rows, subset = 0, None
    if wide and lg >= 7 and qdf >= 8 and rng.random() < 0.6:     # the u32 / comparison gates need quotient degree factor 8
        gate_rows = int(rng.integers(1, 3))
    family = "arith"
    seed = int(rng.integers(1, 1 << 30))
    if qdf >= 8 and rng.random() < 0.75:       # PoseidonGate (degree 7) and the reference's u32 gates need quotient degree factor 8
        family = str(rng.choice(["poseidon_chain", "u32", "zkdsa", "keccak", "smt", "real_smt", "real_keccak", "real_curve"]))
    if family.startswith("real_"):
        # the reference's own circuits rebuilt gadget for gadget (gadgets.py / gadgets_ecdsa.py) on random inputs, under the random FRI / challenge
        # configuration drawn above; 135 wires for the hash circuits, 136 for the curve gadgets (U32RangeCheckGate of 8 limbs)
        from plonky2_lib_amd import gadgets, gadgets_ecdsa
        if family == "real_smt":
            tree = gadgets.SparseMerkleTree()
            keys = [tuple(int(x) for x in rng.integers(0, 1 << 32, 4)) for _ in range(int(rng.integers(0, 40)))]
            for k in keys:
                tree.insert(k, tuple(int(x) for x in rng.integers(1, 1 << 32, 4)))
            key = keys[int(rng.integers(0, len(keys)))] if keys and rng.random() < 0.6 else tuple(int(x) for x in rng.integers(0, 1 << 32, 4))
            desc = gadgets.smt_inclusion_circuit(tree, key, config=synth.Config(135, 80, **kw), public=bool(rng.random() < 0.5),
                                                 enabled=bool(rng.random() < 0.9))
        elif family == "real_keccak":
            desc = gadgets.keccak256_circuit(bytes(rng.integers(0, 256, int(rng.integers(0, 136)), dtype=np.uint8)), config=synth.Config(135, 80, **kw))
        else:
            eb = gadgets_ecdsa.EcdsaBuilder(synth.Config(136, 80, **kw))
            a, b = int(rng.integers(1, 1 << 62)), int(rng.integers(1, 1 << 62))
            p1, p2 = eb.virtual_affine_point(gadgets_ecdsa.pt_mul(a, gadgets_ecdsa.G)), eb.virtual_affine_point(gadgets_ecdsa.pt_mul(a + b, gadgets_ecdsa.G))
            eb.curve_assert_valid(p1)
            s = eb.curve_conditional_add(eb.curve_double(p1), p2, eb.target(int(rng.integers(0, 2))))
            eb.random_access_curve_points(eb.target(int(rng.integers(0, 16))), [s, p1, p2, eb.curve_add(p1, p2)] * 4)
            desc = eb.build()
        if sum(desc.reduction_arity_bits) > desc.degree_bits:
            return random_case(rng)
        hasher = int(rng.random() < 0.3)
        if hasher:
            desc.hasher, desc.circuit_digest = 1, None
        return desc, dict(family=family, lg=int(desc.degree_bits), wide=True, npi=len(desc.public_inputs), gate_rows=0, hasher=hasher, **kw)
    if family == "poseidon_chain":
        desc = synth.poseidon_chain_circuit(max(lg, 5), config, seed=seed)
    elif family == "u32":
        desc = synth.u32_circuit(max(lg, 6), config, seed=seed)
    elif family == "zkdsa":
        desc = synth.zkdsa_circuit(3, config, seed=seed)
    elif family == "keccak":
        desc = synth.keccak_shape_circuit(max(lg, 6), seed=seed)
    elif family == "smt":
        desc = synth.smt_shape_circuit(max(lg, 5), config, seed=seed)
    else:
        desc = synth.arith_circuit(lg, config, seed=seed, public_inputs=pi, pi_hash=oracle.hash_no_pad(pi) if npi else None,
                                   ecdsa_gate_rows=gate_rows, ecdsa_gate_subset=subset)
    if sum(desc.reduction_arity_bits) > desc.degree_bits:      # family fixed its own trace length: same plonky2 assertion
        return random_case(rng)
    hasher = int(rng.random() < 0.3)                            # KeccakGoldilocksConfig: Merkle hasher, transcript permutation, PoW
    if hasher:
        desc.hasher, desc.circuit_digest = 1, None
    return desc, dict(family=family, lg=int(desc.degree_bits), wide=wide, npi=len(desc.public_inputs), gate_rows=gate_rows, hasher=hasher, **kw)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    rng = np.random.default_rng(seed)
    oracle.build()
    ctx = glp.Context(0)
    t0 = time.time()
    for i in range(cases):
        desc, info = random_case(rng)
        oc = oracle.OracleCircuit(desc)
        gc = glp.Circuit(ctx, desc)
        rc, ref = oc.prove()
        got = gc.prove()
        ok = rc == 0 and (got == ref).all() and oc.verify(got) == 0 and gc.verify(got)
        bad = got.copy()
        pos = int(rng.integers(0, len(bad)))
        bad[pos] = (int(bad[pos]) ^ 1) if info["hasher"] else (int(bad[pos]) + 1) % glp.P      # Keccak digests are bytes, not field elements
        ok = ok and (not gc.verify(bad)) and oc.verify(bad) != 0
        # row-local witness generation: scramble what the generators derive, regenerate on the GPU and with the oracle
        from test_oracle_witness import scramble_derived
        w, _ = scramble_derived(desc, rng)
        w = np.ascontiguousarray(w)
        dptr = ctx.dev_alloc(w.nbytes)
        ctx.dev_upload(dptr, w)
        gc.witness_fill(dptr)
        back = np.empty_like(w)
        ctx.dev_download(dptr, back)
        ctx.dev_free(dptr)
        ok = ok and (back == oc.witness_fill(w)).all()
        if desc.num_challenges == 2:             # the batch path: two proofs (the witness and its regenerated twin) in lock step
            both = gc.prove_batch(np.stack([desc.wires, back]), np.stack([desc.public_inputs, desc.public_inputs]) if len(desc.public_inputs) else None)
            ok = ok and (both[0] == got).all() and (both[1] == gc.prove(wires=back)).all()
        # round 3: the batch verifier (query rounds on the GPU) against the host verifier, verdict and reason; the oracle's own proof serializer;
        # a witness staged from page-locked host memory, routed columns only (advice columns derived on the GPU)
        vok, why = gc.verify_batch(np.stack([got, bad, got]), reasons=True)
        host_ok = gc.verify(bad)
        ok = ok and list(vok) == [True, False, True] and not host_ok and why[1] == glp.load_library().glp_last_error().decode()
        ok = ok and gc.proof_to_bytes(got) == oc.proof_to_bytes(got)
        nr_ = desc.num_routed_wires
        pinned = ctx.host_alloc((nr_, 1 << desc.degree_bits))
        pinned[:] = desc.wires[:nr_]
        st = gc.stage_witness(pinned, routed_only=True)
        staged = gc.prove_staged(st)
        st.free()
        ctx.host_free(pinned)
        w0 = desc.wires.copy()
        w0[nr_:] = 0
        ok = ok and (staged == gc.prove(wires=oc.witness_fill(w0, only_advice=True))).all()
        print("case %3d %s  %s  (%.0f s)" % (i, "ok  " if ok else "FAIL", info, time.time() - t0), flush=True)
        if not ok:
            if rc == 0 and not (got == ref).all():
                print("   first differing proof word:", int(np.argmax(got != ref)), "of", len(got))
            sys.exit(1)
        gc.free()
    ctx.close()
    print("fuzz ok: %d cases, seed %d" % (cases, seed))


if __name__ == "__main__":
    main()
