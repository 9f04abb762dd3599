"""BASELINE config 4 on the GPU as a REAL circuit: the reference's sparse-Merkle inclusion proof (plonky2-lib_amd/gadgets.py, 16 levels,
witnesses from the native tree) proved by the HIP library -- mirrors `test_verify_inclusion_proof_by_plonky2`
[REF src/smt/gadgets/verify/mod.rs:3-52]: build once, set the witness of `tree.find(key)`, prove, verify."""
import numpy as np
import pytest

import plonky2_lib_amd as glp
from plonky2_lib_amd import gadgets as G
from test_smt_circuit import H, reference_tree

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = glp.Context(0)
    yield c
    c.close()


def test_reference_scenario(ctx, oracle):
    t = reference_tree()
    c = G.smt_inclusion_circuit(t, H(5))
    gc = glp.Circuit(ctx, c)
    oc = oracle.OracleCircuit(c)
    proof = gc.prove()
    rc, ref = oc.prove()
    assert rc == 0 and (proof == ref).all(), "first mismatch at word %d" % int(np.argmax(proof != ref))
    assert gc.verify(proof) and oc.verify(proof) == 0
    # the same circuit on other witnesses: inclusion of the other keys, non-inclusion, the disabled proof
    others = [G.smt_inclusion_circuit(t, H(1)), G.smt_inclusion_circuit(t, H(12)), G.smt_inclusion_circuit(t, H(7)),
              G.smt_inclusion_circuit(t, H(1 << 100)), G.smt_inclusion_circuit(t, H(5), enabled=False)]
    batch = gc.prove_batch(np.stack([o.wires for o in others]))
    for o, p in zip(others, batch):
        assert gc.verify(p) and (p == gc.prove(wires=o.wires)).all()
    rc, ref = oc.prove(wires=others[2].wires)
    assert rc == 0 and (batch[2] == ref).all()
    # GPU witness generation on the circuit: the Poseidon S-box traces, the arithmetic outputs and the key bits from the routed inputs
    from test_oracle_witness import scramble_derived
    w, _ = scramble_derived(c, np.random.default_rng(2))
    w = np.ascontiguousarray(w)
    dptr = ctx.dev_alloc(w.nbytes)
    ctx.dev_upload(dptr, w)
    gc.witness_fill(dptr)
    back = np.empty_like(w)
    ctx.dev_download(dptr, back)
    assert (back == oc.witness_fill(w)).all()
    assert gc.verify(gc.prove_device(dptr))
    ctx.dev_free(dptr)
    gc.free()


def test_many_inclusion_proofs_of_a_larger_tree(ctx, oracle):
    """300 random keys; 48 membership and 16 non-membership proofs of one root in one lock-step batch, root / key / value public."""
    rng = np.random.default_rng(11)
    t = G.SparseMerkleTree()
    keys = [tuple(int(x) for x in rng.integers(0, 1 << 32, 4)) for _ in range(300)]
    for k in keys:
        t.insert(k, tuple(int(x) for x in rng.integers(1, 1 << 32, 4)))
    absent = [tuple(int(x) for x in rng.integers(0, 1 << 32, 4)) for _ in range(16)]
    descs = [G.smt_inclusion_circuit(t, k, public=True) for k in keys[:48] + absent]
    depth = max(len(d.smt_witness["siblings"]) for d in descs)
    assert 8 <= depth < 16
    gc = glp.Circuit(ctx, descs[0])
    proofs = gc.prove_batch(np.stack([d.wires for d in descs]), np.stack([d.public_inputs for d in descs]))
    oc = oracle.OracleCircuit(descs[0])
    for i, (d, p) in enumerate(zip(descs, proofs)):
        assert gc.verify(p), i
        assert [int(x) for x in p[-12:-8]] == list(t.root)
        assert d.smt_witness["found"] == (i < 48)
    for i in (0, 47, 48, 63):
        assert oc.verify(proofs[i]) == 0
        rc, ref = oc.prove(wires=descs[i].wires, public_inputs=descs[i].public_inputs)
        assert rc == 0 and (proofs[i] == ref).all()
    # public inputs bind the statement: the proof for key i does not verify with key j's public inputs
    bad = proofs[0].copy()
    bad[-8:] = proofs[1][-8:]
    assert not gc.verify(bad)
    gc.free()


def test_process_proofs_on_the_gpu(ctx, oracle):
    """`test_verify_process_proof_by_plonky2` [REF src/smt/gadgets/process/mod.rs:4-82]: insert / update / remove / no-op witnesses of ONE circuit in one
    lock-step batch, each word-equal to the single proof, the first also to the checker's"""
    from test_smt_circuit import process_sequence
    seq = list(process_sequence())
    c0 = seq[0][2]
    gc, oc = glp.Circuit(ctx, c0), oracle.OracleCircuit(c0)
    batch = gc.prove_batch(np.stack([c.wires for _, _, c in seq]), np.stack([c.public_inputs for _, _, c in seq]))
    for (name, proof, c), p in zip(seq, batch):
        assert gc.verify(p) and oc.verify(p) == 0, name
        assert (p == gc.prove(wires=c.wires, public_inputs=c.public_inputs)).all(), name
    rc, ref = oc.prove()
    assert rc == 0 and (batch[0] == ref).all()
    gc.free()
