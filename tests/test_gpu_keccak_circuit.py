"""BASELINE config 2 on the GPU as a REAL circuit: the reference's Keccak-256 gadget rebuilt in plonky2-lib_amd/gadgets.py, proved by
the HIP library through the C ABI.  Mirrors `test_keccak256_short` / `test_keccak256_long` [REF src/hash/keccak256.rs:193-252,254-337]:
build the circuit once, prove each message, public inputs = the digest, verify -- the long test under KeccakGoldilocksConfig."""
import numpy as np
import pytest

import plonky2_lib_amd as glp
from plonky2_lib_amd import gadgets
from test_oracle_keccak import LONG_IN, LONG_OUT, SHORT
from test_oracle_witness import scramble_derived

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = glp.Context(0)
    yield c
    c.close()


def _pi_hex(proof, n=8):
    return b"".join(int(v).to_bytes(4, "little") for v in proof[-n:]).hex()


def test_keccak256_short_circuit(ctx, oracle):
    descs = [gadgets.keccak256_circuit(bytes.fromhex(m)) for m, _ in SHORT]
    gc = glp.Circuit(ctx, descs[0])                           # "build circuit once"
    oc = oracle.OracleCircuit(descs[0])
    proofs = []
    for d, (_, dig) in zip(descs, SHORT):
        proof = gc.prove(wires=d.wires, public_inputs=d.public_inputs)
        assert _pi_hex(proof) == dig                          # `proof.public_inputs` is the digest [REF src/hash/keccak256.rs:316-334]
        assert gc.verify(proof) and oc.verify(proof) == 0
        proofs.append(proof)
    # word-for-word against the checker's prover on one of them
    rc, ref = oc.prove(wires=descs[1].wires, public_inputs=descs[1].public_inputs)
    assert rc == 0 and (proofs[1] == ref).all(), "first mismatch at word %d" % int(np.argmax(proofs[1] != ref))
    # a proof for one message does not pass as a proof for another digest
    bad = proofs[0].copy()
    bad[-8:] = proofs[1][-8:]
    assert not gc.verify(bad) and oc.verify(bad) != 0
    # the three witnesses in one lock-step batch
    both = gc.prove_batch(np.stack([d.wires for d in descs]), np.stack([d.public_inputs for d in descs]))
    for k in range(3):
        assert (both[k] == proofs[k]).all()
    # GPU witness generation: every advice and output cell of the circuit's generators, from the routed inputs
    d = descs[2]
    w, _ = scramble_derived(d, np.random.default_rng(4))
    w = np.ascontiguousarray(w)
    dptr = ctx.dev_alloc(w.nbytes)
    ctx.dev_upload(dptr, w)
    gc.witness_fill(dptr)
    p3 = gc.prove_device(dptr, d.public_inputs)               # cells outside every gate's wires keep the scramble: another valid proof
    assert gc.verify(p3) and oc.verify(p3) == 0 and _pi_hex(p3) == SHORT[2][1]
    back = np.empty_like(w)
    ctx.dev_download(dptr, back)
    ctx.dev_free(dptr)
    assert (back == oc.witness_fill(w)).all()
    assert not gc.verify(gc.prove(wires=w, public_inputs=d.public_inputs))        # the scrambled witness itself proves nothing
    gc.free()


@pytest.mark.parametrize("hasher", [1, 0], ids=["KeccakGoldilocksConfig", "PoseidonGoldilocksConfig"])
def test_keccak256_long_circuit(ctx, oracle, hasher):
    """Four rate blocks, 2^15 rows; the reference runs this one with `type C = KeccakGoldilocksConfig` [REF src/hash/keccak256.rs:281]."""
    d = gadgets.keccak256_circuit(bytes.fromhex(LONG_IN), blocks_num=4)
    d.hasher, d.circuit_digest = hasher, None
    assert d.degree_bits == 15
    gc = glp.Circuit(ctx, d)
    proof = gc.prove()
    assert _pi_hex(proof) == LONG_OUT
    assert gc.verify(proof)
    oc = oracle.OracleCircuit(d, cs_cap=gc.constants_sigmas_cap())
    assert oc.verify(proof) == 0
    bad = proof.copy()
    bad[len(bad) // 2] ^= np.uint64(1)
    assert not gc.verify(bad) and oc.verify(bad) != 0
    # a short message through the same circuit (block flags off)
    s = gadgets.keccak256_circuit(bytes.fromhex(SHORT[2][0]), blocks_num=4)
    p2 = gc.prove(wires=s.wires, public_inputs=s.public_inputs)
    assert _pi_hex(p2) == SHORT[2][1] and gc.verify(p2)
    gc.free()


def test_b32_gadget_circuit(ctx, oracle):
    """BASELINE config 1 (the u32 / b32 gadget circuit, tests/test_keccak_circuit.py::b32_gadget_circuit) on the GPU: word-equal to the checker's proof"""
    from test_keccak_circuit import b32_gadget_circuit
    c = b32_gadget_circuit(seed=9)
    gc, oc = glp.Circuit(ctx, c), oracle.OracleCircuit(c)
    proof = gc.prove()
    rc, ref = oc.prove()
    assert rc == 0 and (proof == ref).all() and gc.verify(proof) and oc.verify(proof) == 0
    gc.free()
