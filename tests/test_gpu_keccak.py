"""KeccakGoldilocksConfig on the GPU, commitment half: the Keccak-256 primitive against the reference's known-answer pairs
[REF src/hash/keccak256.rs:196-212,256-277] and the `KeccakHash<25>` Merkle commitment (leaves, every digest, cap, paths) against the
oracle, through the C ABI."""
import numpy as np
import pytest

import plonky2_lib_amd as glp
from test_oracle_keccak import LONG_IN, LONG_OUT, SHORT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = glp.Context(0)
    yield c
    c.close()


def test_keccak256_reference_vectors_on_gpu(ctx, oracle):
    for msg, dig in SHORT + [(LONG_IN, LONG_OUT)]:
        assert ctx.keccak256([bytes.fromhex(msg)])[0].hex() == dig
    rng = np.random.default_rng(2)
    for n in (0, 1, 7, 8, 9, 135, 136, 137, 271, 272, 273, 1000):
        msgs = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for _ in range(70)]
        got = ctx.keccak256(msgs)
        assert got == [oracle.keccak256(m) for m in msgs], n


@pytest.mark.parametrize("ncols,lg,rb,ch", [
    (1, 3, 3, 2), (3, 4, 3, 4),            # hash_or_noop copies up to 3 elements (24 bytes fit a 25-byte digest) ...
    (4, 4, 3, 3),                          # ... and hashes 4
    (16, 5, 3, 2), (17, 5, 3, 2), (18, 5, 2, 0),       # around one 136-byte rate block (17 lanes): padding block of its own at 17
    (34, 6, 3, 4), (135, 8, 3, 4), (136, 10, 3, 4), (20, 12, 3, 4), (5, 14, 1, 3),
])
def test_keccak_batch_parity(ctx, oracle, ncols, lg, rb, ch):
    rng = np.random.default_rng(100 * ncols + lg)
    vals = oracle.rand_field(rng, (ncols, 1 << lg))
    ref = oracle.batch_from_values(vals, rb, ch, hasher=1)
    b = ctx.batch_from_values(vals, rb, ch, hasher=1)
    assert (b.coeffs() == ref.coeffs).all()
    assert (b.digests() == ref.digests).all()
    assert (b.cap() == ref.cap).all()
    nl = ref.leaves.shape[0]
    for j in (0, 1, nl // 3, nl - 1):
        assert (b.leaf(j) == ref.leaves[j]).all()
        sib = b.prove(j)
        assert (sib == ref.prove(j)).all()
        assert oracle.merkle_verify(ref.leaves[j], j, ref.cap, sib, hasher=1)
    # every digest is 25 bytes: the top 7 bytes of its last word are zero
    assert (b.digests()[:, 3] <= 0xFF).all()
    # and the Poseidon commitment of the same values is a different tree
    p = ctx.batch_from_values(vals, rb, ch)
    assert not (p.cap() == b.cap()).all()
    b.free(); p.free()


def test_keccak_batch_from_coeffs(ctx, oracle):
    rng = np.random.default_rng(9)
    co = oracle.rand_field(rng, (7, 1 << 9))
    ref = oracle.batch_from_coeffs(co, 3, 4, hasher=1)
    b = ctx.batch_from_coeffs(co, 3, 4, hasher=1)
    assert (b.cap() == ref.cap).all() and (b.digests() == ref.digests).all()
    b.free()
    with pytest.raises(glp.GlpError):
        ctx.batch_from_values(co, 3, 4, hasher=7)


# ---------------------------------------------------------------------------------------------------------------------------
# KeccakGoldilocksConfig, whole proofs: `desc.hasher = 1` switches the Merkle hasher (wires / Z / quotient / FRI trees, circuit
# digest), the transcript permutation (KeccakPermutation) and the PoW search; the public-input hash stays Poseidon (InnerHasher).
# [REF src/hash/keccak256.rs:281-295: `type C = KeccakGoldilocksConfig; ... data.prove(pw); data.verify(proof)`]
import plonky2_lib_amd.synth as synth


def _keccak_desc(make):
    d = make()
    d.hasher = 1
    d.circuit_digest = None            # derived (KeccakHash<25>) by the library and by the oracle, each on its own
    return d


KECCAK_CASES = [
    ("arith 2^6 + public inputs", lambda: synth.arith_circuit(6, synth.Config.standard_recursion_config(), seed=3, public_inputs=[5, 6, 7],
                                                              pi_hash=None)),
    ("arith 2^12 ecc config", lambda: synth.arith_circuit(12, synth.Config.standard_ecc_config(), seed=4)),
    ("zkdsa 2^3", lambda: synth.zkdsa_circuit(3)),
    ("keccak256 circuit shape 2^7", lambda: synth.keccak_shape_circuit(7, seed=8)),     # the circuit the reference proves under this config
    ("ecdsa shape 2^7", lambda: synth.ecdsa_shape_circuit(7, seed=2)),
    ("smt shape 2^9", lambda: synth.smt_shape_circuit(9, seed=6)),
]


@pytest.mark.parametrize("name,make", KECCAK_CASES, ids=[c[0] for c in KECCAK_CASES])
def test_keccak_config_proof_parity(ctx, oracle, name, make):
    if "public inputs" in name:
        pi = [5, 6, 7]
        desc = synth.arith_circuit(6, synth.Config.standard_recursion_config(), seed=3, public_inputs=pi, pi_hash=oracle.hash_no_pad(pi))
        desc.hasher, desc.circuit_digest = 1, None
    else:
        desc = _keccak_desc(make)
    oc = oracle.OracleCircuit(desc)
    gc = glp.Circuit(ctx, desc)
    assert (gc.digest() == np.asarray(desc.circuit_digest, np.uint64)).all()        # OracleCircuit filled desc.circuit_digest
    assert (gc.constants_sigmas_cap() == oc.cs_cap).all()
    proof = gc.prove()
    rc, ref = oc.prove()
    assert rc == 0
    assert (proof == ref).all(), "first mismatch at word %d of %d" % (int(np.argmax(proof != ref)), len(ref))
    assert gc.verify(proof) and oc.verify(proof) == 0
    # a Poseidon-config circuit over the same gates gives a different proof and does not accept this one
    desc0 = make() if "public inputs" not in name else None
    if desc0 is not None:
        g0 = glp.Circuit(ctx, desc0)
        assert not (g0.prove() == proof).all()
        assert not g0.verify(proof)
        g0.free()
    # tampering: a cap byte, an opening, a Merkle sibling, the PoW witness
    rng = np.random.default_rng(1)
    for pos in (0, 3, int(rng.integers(0, len(proof))), int(rng.integers(0, len(proof))), len(proof) - 1 - len(desc.public_inputs)):
        bad = proof.copy()
        bad[pos] ^= np.uint64(1)
        assert not gc.verify(bad), pos
        assert oc.verify(bad) != 0, pos
    gc.free()


def test_keccak_config_proof_bytes(ctx, oracle):
    """`ProofWithPublicInputs::to_bytes` under KeccakGoldilocksConfig: a digest is 25 bytes on the wire (a Poseidon HashOut 32)."""
    desc = _keccak_desc(lambda: synth.zkdsa_circuit(3))
    gc = glp.Circuit(ctx, desc)
    proof = gc.prove()
    data = gc.proof_to_bytes(proof)
    g0 = glp.Circuit(ctx, synth.zkdsa_circuit(3))
    p0 = g0.prove()
    d0 = g0.proof_to_bytes(p0)
    oc = oracle.OracleCircuit(desc)
    assert data == oc.proof_to_bytes(proof)        # oracle/gl_proof_bytes.c: the independent restatement of the Rust writer (25-byte digests)
    # count digests from the layout: every 32-byte digest of the Poseidon encoding shrinks by 7 bytes
    ndig = (len(d0) - len(data)) // 7
    assert (len(d0) - len(data)) % 7 == 0 and ndig > 3 * (1 << desc.cap_height)
    assert data[:25] == b"".join(int(w).to_bytes(8, "little") for w in proof[:4])[:25]
    assert (gc.proof_from_bytes(data) == proof).all()
    assert gc.verify(gc.proof_from_bytes(data))
    with pytest.raises(glp.GlpError):
        gc.proof_from_bytes(d0)                     # a Poseidon-config encoding has another length
    # a digest word may exceed p under Keccak (bytes, not field elements) -- from_bytes must take it; an opening may not
    raw = bytearray(data)
    raw[0:8] = (glp.P + 5).to_bytes(8, "little")
    assert int(gc.proof_from_bytes(bytes(raw))[0]) == glp.P + 5
    o = 25 * 3 * (1 << desc.cap_height)               # first opening: right after the three caps
    raw = bytearray(data)
    raw[o:o + 8] = (glp.P + 5).to_bytes(8, "little")
    with pytest.raises(glp.GlpError):
        gc.proof_from_bytes(bytes(raw))
    # the verifier refuses a digest slot holding more than 25 bytes
    bad = proof.copy()
    bad[3] |= np.uint64(1 << 20)
    assert not gc.verify(bad)
    gc.free(); g0.free()


def test_keccak_config_batch_and_session(ctx, oracle):
    """glp_prove_batch and the stepped session under KeccakGoldilocksConfig return the one-shot proof."""
    import ctypes
    rng = np.random.default_rng(500)
    descs = [synth.zkdsa_circuit(3, seed=5, private_key=synth.gl.rand(rng, 4), message=synth.gl.rand(rng, 4)) for _ in range(5)]
    for d in descs:
        d.hasher, d.circuit_digest = 1, None
    gc = glp.Circuit(ctx, descs[0])
    oc = oracle.OracleCircuit(descs[0])
    wires = np.stack([d.wires for d in descs]); pis = np.stack([d.public_inputs for d in descs])
    proofs = gc.prove_batch(wires, pis)
    for k in range(len(descs)):
        single = gc.prove(wires=wires[k], public_inputs=pis[k])
        assert (proofs[k] == single).all(), (k, int(np.argmax(proofs[k] != single)))
        assert gc.verify(proofs[k])
    rc, ref = oc.prove(wires=wires[2], public_inputs=pis[2])
    assert rc == 0 and (proofs[2] == ref).all() and oc.verify(proofs[2]) == 0
    # stepped session with a caller-side Keccak transcript (the oracle's Challenger standing in for the Rust one)
    desc = descs[0]
    nch, n_red = desc.num_challenges, len(desc.reduction_arity_bits)
    s = glp.Session(gc)
    ch = oracle.Challenger(hasher=1)
    ch.observe_hashes(np.asarray(desc.circuit_digest, np.uint64))
    ch.observe(s.public_inputs_hash)               # InnerHasher = Poseidon: 4 field elements
    ch.observe_hashes(s.wires_cap)
    betas, gammas = ch.get_n(nch), ch.get_n(nch)
    ch.observe_hashes(s.partial_products(betas, gammas))
    ch.observe_hashes(s.quotient(ch.get_n(nch)))
    op = s.open(ch.get_ext()).reshape(-1)
    nc_nr, nw = desc.num_constants + desc.num_routed_wires, desc.num_wires
    npp, qdf = desc.num_partial_products, desc.quotient_degree_factor
    o, parts = 0, {}
    for name, cnt in (("cs", nc_nr), ("w", nw), ("zs", nch), ("zn", nch), ("pp", nch * npp), ("q", nch * qdf)):
        parts[name] = op[o:o + 2 * cnt]; o += 2 * cnt
    for name in ("cs", "w", "zs", "pp", "q", "zn"):
        ch.observe(parts[name])
    s.fri_combine(ch.get_ext())
    for _ in range(n_red):
        ch.observe_hashes(s.fri_commit())
        s.fri_fold(ch.get_ext())
    ch.observe(s.fri_final_poly())
    raw = np.frombuffer(ctypes.string_at(ch._buf, 8 * 21), dtype=np.uint64)
    nin = int(np.frombuffer(ctypes.string_at(ctypes.addressof(ch._buf) + 8 * 20, 4), dtype=np.int32)[0])
    w = s.pow_search(raw[:12], raw[12:12 + nin], desc.proof_of_work_bits)
    ch.observe([w])
    assert ch.get() >> (64 - desc.proof_of_work_bits) == 0
    N = 1 << (desc.degree_bits + desc.rate_bits)
    s.queries(w, [ch.get() % N for _ in range(desc.num_query_rounds)])
    got = s.proof()
    s.end()
    assert (got == proofs[0]).all(), int(np.argmax(got != proofs[0]))
    gc.free()


def test_keccak_config_rejects_unknown_hasher(ctx):
    desc = synth.zkdsa_circuit(3)
    desc.hasher = 2
    with pytest.raises(glp.GlpError):
        glp.Circuit(ctx, desc)


def test_keccak_config_circuit_file(ctx, oracle, tmp_path):
    """The hand-off file carries the hasher: a Keccak-config circuit written, re-read and proved gives the same proof."""
    desc = _keccak_desc(lambda: synth.zkdsa_circuit(3))
    gc = glp.Circuit(ctx, desc)
    proof = gc.prove()
    path = str(tmp_path / "k.glpc")
    glp.write_circuit_file(path, desc)
    with glp.CircuitFile(path) as cf:
        assert cf.desc.hasher == 1
        g2 = glp.Circuit(ctx, cf.desc)
        assert (g2.prove() == proof).all()
        g2.free()
    gc.free()
