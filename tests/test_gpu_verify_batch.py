"""glp_verify_batch (the verifier for batch self-checking, query rounds on the GPU): on batches with tampered
members its accept / reject verdicts and its rejection reasons must equal glp_verify's (host code) proof by proof, and both must agree
with the oracle verifier -- section by section of the proof (caps, openings, FRI caps, query leaves, Merkle paths, fold evaluations,
final polynomial, proof-of-work witness, public inputs).  Mirrors what every reference driver does after proving:
`data.verify(proof)` [REF src/zkdsa/circuits/mod.rs:341-347, src/ecdsa/gadgets/ecdsa.rs:349-352]."""
import numpy as np
import pytest

import plonky2_lib_amd as glp
import plonky2_lib_amd.synth as synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = glp.Context(0)
    yield c
    c.close()


def _sections(desc):
    """word ranges of a proof by section, from the documented layout (include/glp.h)"""
    nch, capn = desc.num_challenges, 1 << desc.cap_height
    cols = [desc.num_constants + desc.num_routed_wires, desc.num_wires, nch * (1 + desc.num_partial_products), nch * desc.quotient_degree_factor]
    nopen = sum(cols) + nch
    depth0 = desc.degree_bits + desc.rate_bits - desc.cap_height
    o = {}
    o["wires_cap"] = (0, 4 * capn)
    o["zs_cap"] = (4 * capn, 8 * capn)
    o["quotient_cap"] = (8 * capn, 12 * capn)
    o["openings"] = (12 * capn, 12 * capn + 2 * nopen)
    p = 12 * capn + 2 * nopen
    nred = len(desc.reduction_arity_bits)
    o["fri_caps"] = (p, p + 4 * capn * nred)
    p += 4 * capn * nred
    q0 = p
    at = q0
    for k in range(4):
        o["q0_leaf%d" % k] = (at, at + cols[k]); at += cols[k]
        o["q0_path%d" % k] = (at, at + 4 * depth0); at += 4 * depth0
    lg = desc.degree_bits + desc.rate_bits
    for r, ab in enumerate(desc.reduction_arity_bits):
        lg -= ab
        o["q0_step%d_evals" % r] = (at, at + (2 << ab)); at += 2 << ab
        o["q0_step%d_path" % r] = (at, at + 4 * (lg - desc.cap_height)); at += 4 * (lg - desc.cap_height)
    stride = at - q0
    last = q0 + (desc.num_query_rounds - 1) * stride
    o["qlast_leaf1"] = (last + cols[0] + 4 * depth0, last + cols[0] + 4 * depth0 + cols[1])
    p = q0 + desc.num_query_rounds * stride
    fl = 1 << (desc.degree_bits - sum(desc.reduction_arity_bits))
    o["final_poly"] = (p, p + 2 * fl)
    o["pow"] = (p + 2 * fl, p + 2 * fl + 1)
    o["public_inputs"] = (p + 2 * fl + 1, p + 2 * fl + 1 + len(desc.public_inputs))
    return {k: v for k, v in o.items() if v[1] > v[0]}


def _tampered_batch(proof, desc, rng):
    sec = _sections(desc)
    assert max(v[1] for v in sec.values()) == len(proof)
    names, batch = ["untouched"], [proof.copy()]
    for name, (lo, hi) in sec.items():
        bad = proof.copy()
        pos = int(rng.integers(lo, hi))
        bad[pos] = np.uint64((int(bad[pos]) + 1) % glp.P)
        names.append(name)
        batch.append(bad)
    nc = proof.copy()
    nc[sec["openings"][0] + 3] = np.uint64(glp.P + 1)               # not a canonical field element
    names.append("non_canonical"); batch.append(nc)
    names.append("untouched_again"); batch.append(proof.copy())
    return names, np.stack(batch)


@pytest.mark.parametrize("which", ["zkdsa", "ecdsa", "smt", "arith_rec"])
def test_verdicts_and_reasons_equal_the_host_verifier(ctx, oracle, which):
    desc = {"zkdsa": lambda: synth.zkdsa_circuit(3), "ecdsa": lambda: synth.ecdsa_shape_circuit(7, seed=8),
            "smt": lambda: synth.smt_shape_circuit(10, seed=9),                      # two FRI reductions
            "arith_rec": lambda: synth.arith_circuit(12, synth.Config.standard_recursion_config(), seed=2)}[which]()
    gc = glp.Circuit(ctx, desc)
    oc = oracle.OracleCircuit(desc)
    proof = gc.prove()
    names, batch = _tampered_batch(proof, desc, np.random.default_rng(7))
    ok, reasons = gc.verify_batch(batch, reasons=True)
    L = glp.load_library()
    for k, name in enumerate(names):
        host_ok = gc.verify(batch[k])
        host_reason = "" if host_ok else L.glp_last_error().decode()
        assert bool(ok[k]) == host_ok, (name, reasons[k], host_reason)
        assert reasons[k] == host_reason, (name, reasons[k], host_reason)
        if name != "non_canonical":
            assert (oc.verify(batch[k]) == 0) == host_ok, name
        assert bool(ok[k]) == name.startswith("untouched"), (name, reasons[k])
    # the reasons name the stage that caught it
    by = dict(zip(names, reasons))
    assert "Merkle proof" in by["q0_path1"] and "initial tree 1" in by["q0_path1"]
    assert "proof of work" in by["pow"].lower()
    assert "canonical" in by["non_canonical"]
    gc.free()


def test_host_and_device_transcripts_of_the_verifier_agree(ctx):
    """PoseidonGoldilocksConfig runs the K transcripts on the device (the prover's kernels over the uploaded proofs); GLP_VERIFY_HOST_TRANSCRIPT=1
    keeps them on host threads (what KeccakGoldilocksConfig always does): same verdicts, same reasons, on a batch tampered section by section."""
    import os
    desc = synth.smt_shape_circuit(10, seed=9)
    gc = glp.Circuit(ctx, desc)
    names, batch = _tampered_batch(gc.prove(), desc, np.random.default_rng(23))
    dev_ok, dev_why = gc.verify_batch(batch, reasons=True)
    os.environ["GLP_VERIFY_HOST_TRANSCRIPT"] = "1"
    try:
        host_ok, host_why = gc.verify_batch(batch, reasons=True)
    finally:
        del os.environ["GLP_VERIFY_HOST_TRANSCRIPT"]
    assert list(dev_ok) == list(host_ok) and list(dev_why) == list(host_why), list(zip(names, dev_why, host_why))
    assert dev_ok[0] and dev_ok[-1] and dev_ok.sum() == 2
    gc.free()


def test_keccak_config_batch(ctx, oracle):
    """KeccakGoldilocksConfig: KeccakHash<25> paths and caps on the device side of the verifier"""
    desc = synth.zkdsa_circuit(3)
    desc.hasher, desc.circuit_digest = 1, None
    gc = glp.Circuit(ctx, desc)
    proof = gc.prove()
    names, batch = _tampered_batch(proof, desc, np.random.default_rng(11))
    ok, reasons = gc.verify_batch(batch, reasons=True)
    L = glp.load_library()
    for k, name in enumerate(names):
        host_ok = gc.verify(batch[k])
        assert bool(ok[k]) == host_ok and reasons[k] == ("" if host_ok else L.glp_last_error().decode()), (name, reasons[k])
    assert ok[0] and ok[-1] and ok.sum() == 2
    gc.free()


def test_a_batch_of_different_proofs(ctx, oracle):
    """256 zkdsa proofs from glp_prove_batch, two of them damaged: exactly those two are rejected"""
    rng = np.random.default_rng(500)
    K = 256
    descs = [synth.zkdsa_circuit(3, seed=5, private_key=synth.gl.rand(rng, 4), message=synth.gl.rand(rng, 4)) for _ in range(K)]
    gc = glp.Circuit(ctx, descs[0])
    proofs = gc.prove_batch(np.stack([d.wires for d in descs]), np.stack([d.public_inputs for d in descs]))
    assert gc.verify_batch(proofs).all()
    proofs[17, 40] ^= np.uint64(1)
    proofs[200, gc.proof_words - 20] ^= np.uint64(1)
    ok = gc.verify_batch(proofs)
    assert not ok[17] and not ok[200] and ok.sum() == K - 2
    with pytest.raises(glp.GlpError):
        gc.verify_batch(proofs[:, :-1])
    gc.free()


def test_headline_shape_proof(ctx, oracle):
    """one proof of the 2^16-row, 136-wire shape (depth-15 paths, arity-16 folds): device verdict = host verdict = oracle verdict"""
    desc = synth.arith_circuit(16, synth.Config.standard_ecc_config(), seed=16)
    gc = glp.Circuit(ctx, desc)
    proof = gc.prove()
    bad = proof.copy(); bad[len(bad) // 2] ^= np.uint64(1)
    ok, reasons = gc.verify_batch(np.stack([proof, bad, proof]), reasons=True)
    assert list(ok) == [True, False, True] and gc.verify(proof) and not gc.verify(bad)
    assert reasons[1] == glp.load_library().glp_last_error().decode()
    gc.free()
