"""glp_prove_batch (BASELINE config 5: many independent proofs of one circuit per launch): every proof of the batch must be
word for word the proof glp_prove / the oracle prover produce for the same witness."""
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


def _zkdsa_batch(K, log_n=3):
    """K signatures under ONE circuit: same gates / wiring (same builder seed), different private keys and messages
    [REF src/zkdsa/circuits/mod.rs:322-339: one circuit, many witnesses]."""
    rng = np.random.default_rng(500)
    descs = [synth.zkdsa_circuit(log_n, seed=5, private_key=synth.gl.rand(rng, 4), message=synth.gl.rand(rng, 4)) for _ in range(K)]
    for d in descs[1:]:
        assert (d.constants == descs[0].constants).all() and (d.sigmas == descs[0].sigmas).all()
    wires = np.stack([d.wires for d in descs])
    pis = np.stack([d.public_inputs for d in descs])
    return descs, wires, pis


def test_zkdsa_batch_equals_single_proofs_and_oracle(ctx, oracle):
    K = 12
    descs, wires, pis = _zkdsa_batch(K)
    gc = glp.Circuit(ctx, descs[0])
    oc = oracle.OracleCircuit(descs[0])
    proofs = gc.prove_batch(wires, pis)
    assert proofs.shape == (K, gc.proof_words)
    for k in range(K):
        single = gc.prove(wires=wires[k], public_inputs=pis[k])
        assert (proofs[k] == single).all(), "proof %d: first mismatch at word %d" % (k, int(np.argmax(proofs[k] != single)))
        assert gc.verify(proofs[k])
    for k in (0, 5, K - 1):
        rc, ref = oc.prove(wires=wires[k], public_inputs=pis[k])
        assert rc == 0 and (proofs[k] == ref).all()
        assert oc.verify(proofs[k]) == 0
    assert len({p.tobytes() for p in proofs}) == K              # K different proofs
    gc.free()


@pytest.mark.parametrize("make,K", [
    (lambda s: synth.ecdsa_shape_circuit(7, seed=s), 3),           # limb-gate launch + Comparison + light gates, 1 FRI reduction
    (lambda s: synth.keccak_shape_circuit(6, seed=s), 2),          # interleave gates (single-gate launches)
    (lambda s: synth.smt_shape_circuit(10, seed=s), 5),            # Poseidon + BaseSum<2>, 2 FRI reductions
    (lambda s: synth.arith_circuit(12, synth.Config.standard_recursion_config(), seed=s), 4),
    (lambda s: synth.zkdsa_circuit(3), 1),                         # a batch of one
])
def test_batch_other_circuit_families(ctx, oracle, make, K):
    """Same circuit (constants, sigmas), K different witnesses: the synthetic builders draw circuit and witness from one seed,
    so the K witnesses are made by re-randomising the free cells of one circuit's witness and regenerating the derived ones."""
    desc = make(3)
    oc = oracle.OracleCircuit(desc)
    gc = glp.Circuit(ctx, desc)
    rng = np.random.default_rng(K)
    ws = [desc.wires]
    for _ in range(K - 1):
        w = desc.wires.copy()
        # NoopGate rows are unconstrained and uncopied: any values there give another valid witness
        noop = next(i for i, g in enumerate(desc.gates) if g["type"] == synth.GATE_NOOP)
        rows = np.nonzero(desc.constants[desc.gates[noop]["selector_index"]] == np.uint64(noop))[0]
        w[:, rows] = synth.gl.rand(rng, (w.shape[0], len(rows)))
        ws.append(w)
    wires = np.stack(ws)
    pis = np.tile(np.asarray(desc.public_inputs, np.uint64), (K, 1))
    proofs = gc.prove_batch(wires, pis)
    for k in range(K):
        assert (proofs[k] == gc.prove(wires=wires[k])).all(), k
        assert gc.verify(proofs[k])
    rc, ref = oc.prove(wires=wires[K - 1])
    assert (proofs[K - 1] == ref).all()
    gc.free()


def test_batch_of_256_zkdsa_proofs(ctx, oracle):
    """The BASELINE configuration itself: 256 proofs in one call, from a device-resident witness block; all verify, spot checks
    against single proofs."""
    K = 256
    descs, wires, pis = _zkdsa_batch(K)
    gc = glp.Circuit(ctx, descs[0])
    w = np.ascontiguousarray(wires)
    dptr = ctx.dev_alloc(w.nbytes)
    ctx.dev_upload(dptr, w)
    proofs = gc.prove_batch_device(dptr, K, pis)
    ctx.dev_free(dptr)
    assert all(gc.verify(p) for p in proofs)
    for k in (0, 100, 255):
        assert (proofs[k] == gc.prove(wires=wires[k], public_inputs=pis[k])).all()
    oc = oracle.OracleCircuit(descs[0])
    assert oc.verify(proofs[77]) == 0
    gc.free()


def test_batch_of_600_zkdsa_proofs(ctx):
    """More proofs than a proof-of-work workgroup has threads, and not a multiple of it: the search ranks the open proofs
    t, t + 256, t + 512 per thread.  Every witness must still be the smallest one, i.e. every proof the single proof."""
    K = 600
    descs, wires, pis = _zkdsa_batch(K)
    gc = glp.Circuit(ctx, descs[0])
    proofs = gc.prove_batch(wires, pis)
    assert gc.verify_batch(proofs).all()
    for k in (0, 255, 256, 300, 511, 512, 599):
        assert (proofs[k] == gc.prove(wires=wires[k], public_inputs=pis[k])).all()
    gc.free()


def test_batch_argument_errors(ctx):
    desc = synth.zkdsa_circuit(3)
    gc = glp.Circuit(ctx, desc)
    with pytest.raises(glp.GlpError):
        gc.prove_batch(desc.wires)                                  # not [K][num_wires][n]
    with pytest.raises(glp.GlpError):
        gc.prove_batch(desc.wires[None], np.zeros((1, 3), np.uint64))
    c3 = synth.arith_circuit(5, synth.Config(135, 80, num_challenges=3), seed=1)
    g3 = glp.Circuit(ctx, c3)
    with pytest.raises(glp.GlpError) as e:
        g3.prove_batch(c3.wires[None])
    assert "num_challenges" in str(e.value)
    gc.free(); g3.free()


def test_host_and_device_transcripts_agree(ctx, oracle):
    """glp_prove_batch keeps the K Fiat-Shamir transcripts on the GPU (one 16-lane group per proof and transcript step); with
    GLP_BATCH_HOST_TRANSCRIPT=1 they run on host threads with a round trip per step (the round-2 path, still what
    KeccakGoldilocksConfig takes).  Same words either way, with public inputs (zkdsa: 12) and without, one and two FRI reductions."""
    import os
    K = 9
    descs, wires, pis = _zkdsa_batch(K)
    gc = glp.Circuit(ctx, descs[0])
    dev = gc.prove_batch(wires, pis)
    os.environ["GLP_BATCH_HOST_TRANSCRIPT"] = "1"
    try:
        host = gc.prove_batch(wires, pis)
    finally:
        del os.environ["GLP_BATCH_HOST_TRANSCRIPT"]
    assert (dev == host).all(), "first mismatch at proof %d word %d" % tuple(int(x) for x in np.argwhere(dev != host)[0])
    assert gc.verify_batch(dev).all()
    gc.free()
    desc = synth.smt_shape_circuit(10, seed=3)                       # no public inputs, two reductions
    gc = glp.Circuit(ctx, desc)
    w = np.stack([desc.wires, desc.wires])
    dev = gc.prove_batch(w)
    os.environ["GLP_BATCH_HOST_TRANSCRIPT"] = "1"
    try:
        host = gc.prove_batch(w)
    finally:
        del os.environ["GLP_BATCH_HOST_TRANSCRIPT"]
    assert (dev == host).all() and (dev[0] == gc.prove()).all()
    gc.free()
