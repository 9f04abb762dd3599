"""GPU parity tests for the whole prove() path (partial products, quotient, openings, FRI commit,
proof of work, queries): every word of the HIP library's proof must equal the CPU oracle's, and the
oracle's verifier must accept it."""
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


def _sections(oc, desc):
    """name -> slice of the proof words (layout: include/glp.h)."""
    cap = 4 << desc.cap_height
    nch = desc.num_challenges
    nopen = (desc.num_constants + desc.num_routed_wires + desc.num_wires + 2 * nch + nch * desc.num_partial_products +
             nch * desc.quotient_degree_factor)
    o = 0
    out = {}
    for name, ln in (("wires_cap", cap), ("zs_pp_cap", cap), ("quotient_cap", cap), ("openings", 2 * nopen),
                     ("fri_caps", cap * len(desc.reduction_arity_bits))):
        out[name] = slice(o, o + ln); o += ln
    out["rest"] = slice(o, None)
    return out


def _check(ctx, oracle, desc):
    oc = oracle.OracleCircuit(desc)
    gc = glp.Circuit(ctx, desc)
    assert (gc.constants_sigmas_cap() == oc.cs_cap).all()
    assert (gc.digest() == np.asarray(desc.circuit_digest, np.uint64)).all()
    assert gc.proof_words == oc.proof_words
    rc, ref = oc.prove()
    assert rc == 0 and oc.verify(ref) == 0
    got = gc.prove()
    sec = _sections(oc, desc)
    for name, sl in sec.items():
        assert (got[sl] == ref[sl]).all(), "first mismatch in section %s at word %d" % (
            name, sl.start + int(np.argmax(got[sl] != ref[sl])))
    assert oc.verify(got) == 0
    assert gc.verify(got) and gc.verify(ref)          # the library's own verifier accepts both provers' proofs
    return oc, gc, got


@pytest.mark.parametrize("lg,cfg", [(5, "ecc"), (6, "rec"), (9, "ecc"), (12, "rec"), (13, "ecc")])
def test_prove_parity_arith(ctx, oracle, lg, cfg):
    config = synth.Config.standard_ecc_config() if cfg == "ecc" else synth.Config.standard_recursion_config()
    desc = synth.arith_circuit(lg, config, seed=100 + lg)
    _check(ctx, oracle, desc)


@pytest.mark.parametrize("kw", [
    dict(num_challenges=1),                                    # monolithic quotient kernel, one challenge
    dict(num_challenges=3),                                    # ... three challenges
    dict(max_quotient_degree_factor=4),                        # quotient on every second LDE plane (step 2), 19 partial products
    dict(rate_bits=2, max_quotient_degree_factor=4),           # LDE x4
    dict(rate_bits=4, cap_height=2, num_query_rounds=5),       # LDE x16, small cap
    dict(proof_of_work_bits=0, num_query_rounds=1, cap_height=0),
    dict(arity_bits=3, final_poly_bits=2, proof_of_work_bits=8),   # arity-8 FRI, more reductions
    dict(arity_bits=1, final_poly_bits=3, num_query_rounds=3),     # arity-2 FRI: 4-element leaves are copied, not hashed
])
def test_prove_parity_config_sweep(ctx, oracle, kw):
    """CircuitConfig / FriConfig fields away from the presets: every field the prover reads is honoured."""
    config = synth.Config(135, 80, **kw)
    desc = synth.arith_circuit(8, config, seed=31)
    _check(ctx, oracle, desc)


def test_prove_parity_wide_ecc_config(ctx, oracle):
    """`CircuitConfig::wide_ecc_config` (234 wires) [REF src/ecdsa/gadgets/ecdsa.rs:489]: 30 sponge permutations per wires leaf."""
    desc = synth.arith_circuit(8, synth.Config(234, 80), seed=52)
    _check(ctx, oracle, desc)


def test_prove_parity_non_geometric_coset_shifts(ctx, oracle):
    """k_is that are NOT 1, g, g^2, ... take the generic path of the permutation kernel (the chained multiply-by-g
    shortcut only applies to plonky2's own choice of shifts)."""
    config = synth.Config.standard_recursion_config()
    config.scramble_k_is = True
    desc = synth.arith_circuit(7, config, seed=41)
    assert int(desc.k_is[0]) != 1
    _check(ctx, oracle, desc)


def test_prove_parity_public_inputs(ctx, oracle):
    pi = np.array([5, 6, 7, 8, 9], np.uint64)
    desc = synth.arith_circuit(7, seed=9, public_inputs=pi, pi_hash=oracle.hash_no_pad(pi))
    oc, gc, got = _check(ctx, oracle, desc)
    assert [int(x) for x in got[-5:]] == [5, 6, 7, 8, 9]


def test_prove_parity_reference_u32_gates(ctx, oracle):
    # U32InterleaveGate / UninterleaveToU32Gate / UninterleaveToB32Gate [REF src/u32/gates/*.rs]
    desc = synth.u32_circuit(6)
    _check(ctx, oracle, desc)
    desc = synth.u32_circuit(8, seed=5)
    _check(ctx, oracle, desc)


def test_prove_parity_ecdsa_gate_set(ctx, oracle):
    """BASELINE config 3 gate set: all 11 gate types of the secp256k1 circuit [REF src/ecdsa/gadgets/ecdsa.rs:72-96]
    in three selector groups, standard_ecc_config."""
    desc = synth.ecdsa_shape_circuit(7)
    assert desc.num_selectors == 3 and desc.num_gate_constraints == 136 and len(desc.gates) == 11
    _check(ctx, oracle, desc)
    _check(ctx, oracle, synth.ecdsa_shape_circuit(10, seed=8, rows_per_gate=5))


def test_prove_parity_keccak_gate_set(ctx, oracle):
    """BASELINE configs 1/2 gate set (u32 arithmetic gates + the reference's interleave gates), 135 wires."""
    _check(ctx, oracle, synth.keccak_shape_circuit(7))


def test_keccak_shape_2_15_verifies(ctx, oracle):
    desc = synth.keccak_shape_circuit(15)
    gc = glp.Circuit(ctx, desc)
    proof = gc.prove()
    desc.circuit_digest = gc.digest()
    assert oracle.OracleCircuit(desc, cs_cap=gc.constants_sigmas_cap()).verify(proof) == 0
    gc.free()


def test_prove_parity_zkdsa_circuit(ctx, oracle):
    """BASELINE config 5: the simple-signature circuit, 2^3 rows, 4 PoseidonGate rows, 12 public inputs, two
    selector groups (PoseidonGate has degree 7)."""
    desc = synth.zkdsa_circuit(3)
    assert desc.num_selectors == 2
    # public inputs = message, public_key, signature [REF src/zkdsa/circuits/mod.rs:34-36]
    pk = oracle.two_to_one(desc.wires[0:4, 1], desc.wires[4:8, 1])
    assert (desc.public_inputs[4:8] == pk).all()
    oc, gc, got = _check(ctx, oracle, desc)
    assert (got[-12:] == desc.public_inputs).all()


def test_prove_parity_poseidon_chain(ctx, oracle):
    """BASELINE config 4 shape: a 2^5-row chain of PoseidonGate rows (SMT path walk)."""
    _check(ctx, oracle, synth.poseidon_chain_circuit(5))


@pytest.mark.parametrize("lg", [5, 6, 7, 8])
def test_prove_parity_smt_gate_mix(ctx, oracle, lg):
    """BASELINE config 4 gate mix [REF src/smt/gadgets/verify/verify_smt.rs:214-307, src/smt/gadgets/common.rs:87-112]:
    PoseidonGate chains, BaseSumGate<2> with 63 limbs (`split_le(key, 64)`), ArithmeticGate rows, Constant / PublicInput /
    Noop, two selector groups.  Word-for-word against the oracle prover, both verifiers accept."""
    desc = synth.smt_shape_circuit(lg, seed=70 + lg)
    kinds = {(g["type"], g["p0"], g["p1"]) for g in desc.gates}
    assert (synth.GATE_BASE_SUM, 63, 2) in kinds and (synth.GATE_POSEIDON, 0, 0) in kinds and (synth.GATE_ARITHMETIC, 20, 0) in kinds
    _check(ctx, oracle, desc)


def test_smt_shape_2_12_verifies(ctx, oracle):
    """Config 4 at the size BASELINE.md lists (2^12 rows): proof accepted by both verifiers, a broken BaseSum<2> limb (2 is
    not a bit) and a tampered proof word are rejected."""
    desc = synth.smt_shape_circuit(12, seed=4)
    gc = glp.Circuit(ctx, desc)
    proof = gc.prove()
    desc.circuit_digest = gc.digest()
    oc = oracle.OracleCircuit(desc, cs_cap=gc.constants_sigmas_cap())
    assert oc.verify(proof) == 0 and gc.verify(proof)
    bad = proof.copy(); bad[777] = np.uint64((int(bad[777]) + 1) % glp.P)
    assert oc.verify(bad) != 0 and not gc.verify(bad)
    row_b = next(r for r in range(1 << 12) if int(desc.constants[0][r]) == next(i for i, g in enumerate(desc.gates) if g["type"] == synth.GATE_BASE_SUM))
    w = desc.wires.copy()
    w[5, row_b] = 2
    assert not gc.verify(gc.prove(wires=w))
    gc.free()


def test_wrong_shapes_are_errors_not_overreads(ctx):
    """The C ABI takes bare pointers; the binding checks every array against the circuit before the call."""
    desc = synth.arith_circuit(5, synth.Config.standard_recursion_config(), seed=2)
    gc = glp.Circuit(ctx, desc)
    proof = gc.prove()
    with pytest.raises(glp.GlpError):
        gc.prove(wires=desc.wires[:, :16])
    with pytest.raises(glp.GlpError):
        gc.prove(public_inputs=[1, 2, 3])
    with pytest.raises(glp.GlpError):
        gc.verify(proof[:-1])                       # glp_verify_n: a truncated proof is an argument error
    with pytest.raises(glp.GlpError):
        gc.proof_to_bytes(proof[:100])
    bad = synth.arith_circuit(5, synth.Config.standard_recursion_config(), seed=2)
    bad.sigmas = bad.sigmas[:, :8]
    with pytest.raises(glp.GlpError):
        glp.Circuit(ctx, bad)
    d = ctx.dev_alloc(4096)
    with pytest.raises(glp.GlpError):
        ctx.dev_upload(d, np.zeros(1024, np.uint64))          # 8192 bytes into a 4096-byte block
    with pytest.raises(glp.GlpError):
        ctx.dev_download(d + 4000, np.zeros(64, np.uint64))
    ctx.dev_upload(d + 2048, np.zeros(256, np.uint64))        # inside the block at an offset: fine
    ctx.dev_free(d)
    with pytest.raises(glp.GlpError):
        glp.load_library() and ctx.lde(np.zeros((1, 8), np.uint64), rate_bits=9)


def test_unsatisfied_witness_fails_verification(ctx, oracle):
    desc = synth.arith_circuit(7, seed=11)
    oc = oracle.OracleCircuit(desc)
    gc = glp.Circuit(ctx, desc)
    w = desc.wires.copy()
    w[7, 50] = np.uint64((int(w[7, 50]) + 1) % glp.P)
    proof = gc.prove(wires=w)
    assert oc.verify(proof) != 0
    rc, ref = oc.prove(wires=w)
    assert (proof == ref).all()       # same (invalid) transcript on both sides


@pytest.mark.parametrize("which", ["ecdsa", "keccak", "zkdsa", "poseidon_chain"])
def test_prove_parity_on_boundary_valued_wires(ctx, oracle, which):
    """An (unsatisfying) witness whose wires are boundary values of the field and of its 32-bit limbs: the transcript and
    every proof word must still equal the oracle prover's.  Exercises the lazy arithmetic of the quotient kernels
    (non-canonical products, carry-free accumulators, base-4 limb sums) where wrap-around cases are likeliest."""
    desc = {"ecdsa": lambda: synth.ecdsa_shape_circuit(6, seed=9), "keccak": lambda: synth.keccak_shape_circuit(6, seed=9),
            "zkdsa": lambda: synth.zkdsa_circuit(3), "poseidon_chain": lambda: synth.poseidon_chain_circuit(5)}[which]()
    special = np.array([0, 1, 2, 3, glp.P - 1, glp.P - 2, glp.P - 3, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, 0xFFFFFFFF00000000,
                        0xFFFFFFFE00000001, 0xFFFFFFFEFFFFFFFF, 1 << 63, (1 << 63) - 1, 0xFFFFFFFF], dtype=np.uint64)
    rng = np.random.default_rng(99)
    w = special[rng.integers(0, len(special), size=desc.wires.shape)]
    keep = rng.random(desc.wires.shape) < 0.3
    w[keep] = desc.wires[keep]
    oc = oracle.OracleCircuit(desc)
    gc = glp.Circuit(ctx, desc)
    rc, ref = oc.prove(wires=w)
    got = gc.prove(wires=w)
    assert (got == ref).all(), "first mismatch at word %d" % int(np.argmax(got != ref))
    assert not gc.verify(got) and oc.verify(got) != 0


def test_prove_properties_2_16(ctx, oracle):
    """2^16 rows x 136 wires: too slow for the oracle PROVER in a test, so check through the oracle
    VERIFIER (the relation the reference's tests assert) plus determinism."""
    desc = synth.arith_circuit(16, synth.Config.standard_ecc_config(), seed=16)
    oc = oracle.OracleCircuit(desc)
    gc = glp.Circuit(ctx, desc)
    p1 = gc.prove()
    assert oc.verify(p1) == 0
    assert (gc.prove() == p1).all()
    bad = p1.copy(); bad[1000] = np.uint64((int(bad[1000]) + 1) % glp.P)
    assert oc.verify(bad) != 0


def test_stand_in_2_20_verifies(ctx, oracle):
    """The gate-mix stand-in of the headline shape (synth.ecdsa_shape_circuit(20): 2^20 rows x 136 wires, the 11 gate kinds of the
    secp256k1 circuit with ONE U32AddMany parameter set, the direct two-pass 2^20 transforms) -- what rounds 1-2 timed and what
    bench.py still reports as variants.gate_mix_stand_in.  The circuit bench.py times by default (the real 17-gate secp256k1
    circuit, 10 signatures) has its own test: tests/test_gpu_ecdsa_circuit.py::test_headline_ten_signatures_2_20.  The proof must
    satisfy BOTH verifiers (the oracle's and the library's own), a tampered word must be rejected by both, and proving is
    deterministic [REF src/bin/perf.rs:7-9, src/ecdsa/gadgets/ecdsa.rs:349-352: prove, then verify]."""
    desc = synth.ecdsa_shape_circuit(20, seed=0x5EED0003)
    assert len(desc.gates) == 11 and desc.num_wires == 136
    gc = glp.Circuit(ctx, desc)
    proof = gc.prove()
    desc.circuit_digest = gc.digest()
    oc = oracle.OracleCircuit(desc, cs_cap=gc.constants_sigmas_cap())
    assert oc.verify(proof) == 0
    assert gc.verify(proof)
    w = np.ascontiguousarray(desc.wires)
    dptr = ctx.dev_alloc(w.nbytes)
    ctx.dev_upload(dptr, w)
    assert (gc.prove_device(dptr) == proof).all()        # the entry point the bench times
    ctx.dev_free(dptr)
    rng = np.random.default_rng(20)
    for pos in [int(x) for x in rng.integers(0, len(proof), 4)] + [0, len(proof) - 1]:
        bad = proof.copy(); bad[pos] = np.uint64((int(bad[pos]) + 1) % glp.P)
        assert oc.verify(bad) != 0 and not gc.verify(bad), pos
    # an unsatisfied witness (one ArithmeticGate output off by one in the middle of the trace) must not verify
    w2 = desc.wires.copy()
    w2[3, 1 << 19] = np.uint64((int(w2[3, 1 << 19]) + 1) % glp.P)
    assert not gc.verify(gc.prove(wires=w2))
    gc.free()


def test_prove_2_21_rows_verifies(ctx, oracle):
    """2^21 rows x 135 wires, 5 FRI reductions: the transforms are two passes with the 512-row LDS tiles of k_strided32 (three passes
    until round 3; the real 20-signature circuit of this size: tests/test_gpu_ecdsa_circuit.py::test_perf_rs_twenty_signatures_2_21); the
    proof must satisfy the oracle verifier (constants/sigmas cap taken from the circuit object)."""
    desc = synth.arith_circuit(21, synth.Config.standard_recursion_config(), seed=21)
    gc = glp.Circuit(ctx, desc)
    proof = gc.prove()
    desc.circuit_digest = gc.digest()
    oc = oracle.OracleCircuit(desc, cs_cap=gc.constants_sigmas_cap())
    assert oc.verify(proof) == 0
    bad = proof.copy(); bad[5000] = np.uint64((int(bad[5000]) + 1) % glp.P)
    assert oc.verify(bad) != 0
    gc.free()


def test_proof_bytes_roundtrip(ctx, oracle):
    desc = synth.arith_circuit(9, seed=21)
    gc = glp.Circuit(ctx, desc)
    words = gc.prove()
    data = gc.proof_to_bytes(words)
    depth0 = desc.degree_bits + desc.rate_bits - desc.cap_height
    n_paths = desc.num_query_rounds * (4 + len(desc.reduction_arity_bits))
    assert len(data) == 8 * len(words) + n_paths
    assert data[:8] == int(words[0]).to_bytes(8, "little")
    # first Merkle path of the first query: one length byte, then depth0 digests
    nch = desc.num_challenges
    nopen = (desc.num_constants + desc.num_routed_wires + desc.num_wires + 2 * nch + nch * desc.num_partial_products +
             nch * desc.quotient_degree_factor)
    cap = 4 << desc.cap_height
    q0 = 8 * (3 * cap + 2 * nopen + cap * len(desc.reduction_arity_bits)) + 8 * (desc.num_constants + desc.num_routed_wires)
    assert data[q0] == depth0
    assert (gc.proof_from_bytes(data) == words).all()
    bad = bytearray(data); bad[q0] ^= 1
    with pytest.raises(glp.GlpError):
        gc.proof_from_bytes(bytes(bad))


@pytest.mark.parametrize("which", ["arith_rec", "ecdsa", "zkdsa", "poseidon_chain"])
def test_proof_bytes_equal_the_oracle_serializer(ctx, oracle, which):
    """glp_proof_to_bytes against oracle/gl_proof_bytes.c -- an independent restatement of `Buffer::write_proof_with_public_inputs`
    that follows the Rust writer's call tree instead of the product's flat piece list -- byte for byte, on circuits with
    different column counts, FRI arities and public-input counts.  Both are RECALLED from plonky2 0.1.4 (the reference holds no
    proof bytes): agreement is self-consistency, parity with the fork stays unpinned."""
    pi = np.array([5, 6, 7], np.uint64)
    desc = {"arith_rec": lambda: synth.arith_circuit(9, synth.Config.standard_recursion_config(), seed=21, public_inputs=pi, pi_hash=oracle.hash_no_pad(pi)),
            "ecdsa": lambda: synth.ecdsa_shape_circuit(7, seed=4), "zkdsa": lambda: synth.zkdsa_circuit(3),
            "poseidon_chain": lambda: synth.poseidon_chain_circuit(5)}[which]()
    gc = glp.Circuit(ctx, desc)
    oc = oracle.OracleCircuit(desc)
    words = gc.prove()
    data = gc.proof_to_bytes(words)
    assert data == oc.proof_to_bytes(words)
    assert (gc.proof_from_bytes(oc.proof_to_bytes(words)) == words).all()
    gc.free()


def test_malformed_gate_shapes_are_rejected_on_the_host(ctx):
    """A description whose gates would read past the wire columns must fail in glp_circuit_create, not on the GPU."""
    desc = synth.ecdsa_shape_circuit(7)
    rc_gate = next(g for g in desc.gates if g["type"] == synth.GATE_U32_RANGE_CHECK)
    rc_gate["p0"] = 9; rc_gate["num_constraints"] = 9 * 17          # 153 wires > 136
    with pytest.raises(glp.GlpError) as e:
        glp.Circuit(ctx, desc)
    assert e.value.code == -1 and "wires" in str(e.value)
    desc = synth.ecdsa_shape_circuit(7)
    desc.gates[3]["num_constraints"] += 1
    with pytest.raises(glp.GlpError):
        glp.Circuit(ctx, desc)


def test_unsupported_gate_is_reported(ctx):
    desc = synth.arith_circuit(5, seed=1)
    desc.gates[0]["type"] = 99         # not a gate this build knows
    with pytest.raises(glp.GlpError) as e:
        glp.Circuit(ctx, desc)
    assert e.value.code == -3


def _stepped_proof(gc, oracle, desc, use_gpu_pow=True):
    """Drive glp_session_* with an EXTERNAL transcript (the oracle's Challenger, standing in for the Rust prover's) in
    the order of plonky2's prove_with_partition_witness / fri_proof."""
    import ctypes
    nch, n_red = desc.num_challenges, len(desc.reduction_arity_bits)
    s = glp.Session(gc)
    ch = oracle.Challenger()
    ch.observe(np.asarray(desc.circuit_digest, np.uint64))
    ch.observe(s.public_inputs_hash)
    ch.observe(s.wires_cap)
    betas, gammas = ch.get_n(nch), ch.get_n(nch)
    ch.observe(s.partial_products(betas, gammas))
    ch.observe(s.quotient(ch.get_n(nch)))
    op = s.open(ch.get_ext()).reshape(-1)
    nc_nr, nw = desc.num_constants + desc.num_routed_wires, desc.num_wires
    npp, qdf = desc.num_partial_products, desc.quotient_degree_factor
    o = 0
    parts = {}
    for name, cnt in (("cs", nc_nr), ("w", nw), ("zs", nch), ("zn", nch), ("pp", nch * npp), ("q", nch * qdf)):
        parts[name] = op[o:o + 2 * cnt]; o += 2 * cnt
    for name in ("cs", "w", "zs", "pp", "q", "zn"):          # OpeningSet::to_fri_openings: zeta batch, then zeta*g batch
        ch.observe(parts[name])
    s.fri_combine(ch.get_ext())
    for _ in range(n_red):
        ch.observe(s.fri_commit())
        s.fri_fold(ch.get_ext())
    ch.observe(s.fri_final_poly())
    # proof of work on the transcript's own sponge: state + pending inputs (oracle struct: st[12], in[8], nin)
    raw = np.frombuffer(ctypes.string_at(ch._buf, 8 * 21), dtype=np.uint64)
    nin = int(np.frombuffer(ctypes.string_at(ctypes.addressof(ch._buf) + 8 * 20, 4), dtype=np.int32)[0])
    w = s.pow_search(raw[:12], raw[12:12 + nin], desc.proof_of_work_bits)
    ch.observe([w])
    resp = ch.get()
    assert resp >> (64 - desc.proof_of_work_bits) == 0
    N = 1 << (desc.degree_bits + desc.rate_bits)
    s.queries(w, [ch.get() % N for _ in range(desc.num_query_rounds)])
    proof = s.proof()
    s.end()
    return proof


@pytest.mark.parametrize("lg,cfg", [(6, "rec"), (12, "ecc")])
def test_stepped_session_equals_prove(ctx, oracle, lg, cfg):
    """The stage-level C ABI driven by a caller-side transcript yields the identical proof as glp_prove (whose words
    the other tests pin against the oracle prover), and the oracle verifier accepts it."""
    config = synth.Config.standard_ecc_config() if cfg == "ecc" else synth.Config.standard_recursion_config()
    pi = [3, 1 << 40, 0xFFFFFFFF00000000]
    desc = synth.arith_circuit(lg, config, seed=77, public_inputs=pi, pi_hash=oracle.hash_no_pad(pi))
    oc = oracle.OracleCircuit(desc)
    gc = glp.Circuit(ctx, desc)
    ref = gc.prove()
    got = _stepped_proof(gc, oracle, desc)
    assert (got == ref).all(), "first mismatch at word %d" % int(np.argmax(got != ref))
    assert oc.verify(got) == 0


def test_golden_proof_fixture(ctx):
    """The committed zkdsa proof (tests/golden/proof_zkdsa_2_3.npz) is reproduced word for word by the HIP prover and
    accepted by glp_verify -- no oracle in the loop."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "proof_zkdsa_2_3.npz"))
    desc = synth.zkdsa_circuit(3)
    gc = glp.Circuit(ctx, desc)
    assert (gc.digest() == g["circuit_digest"]).all()
    assert (gc.constants_sigmas_cap() == g["constants_sigmas_cap"]).all()
    assert (gc.prove() == g["proof"]).all()
    assert gc.verify(g["proof"])


def test_circuit_file_to_proof(ctx, tmp_path):
    """The hand-off path end to end, no oracle in the loop: the committed sample file (circuit + witness, as a Rust machine
    would ship them) -> glp_circuit_create -> glp_prove reproduces the committed proof word for word; a file written from a
    larger synthetic circuit proves identically to the in-process descriptor."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "proof_zkdsa_2_3.npz"))
    with glp.CircuitFile(os.path.join(os.path.dirname(__file__), "golden", "zkdsa_2_3.glpc")) as cf:
        gc = glp.Circuit(ctx, cf.desc)
        assert (gc.digest() == g["circuit_digest"]).all()
        proof = gc.prove(wires=cf.desc.wires, public_inputs=cf.desc.public_inputs)
        assert (proof == g["proof"]).all() and gc.verify(proof)
        gc.free()
    desc = synth.ecdsa_shape_circuit(10, seed=33)
    ref = glp.Circuit(ctx, desc)
    want = ref.prove()
    path = str(tmp_path / "ecdsa10.glpc")
    glp.write_circuit_file(path, desc)
    with glp.CircuitFile(path) as cf:
        gc = glp.Circuit(ctx, cf.desc)
        assert (gc.digest() == ref.digest()).all()
        assert (gc.prove(wires=cf.desc.wires, public_inputs=cf.desc.public_inputs) == want).all()
        gc.free()
    ref.free()


def test_verifier_rejects_tampered_proofs(ctx, oracle):
    """glp_verify (host-side restatement of plonk/verifier.rs + fri/verifier.rs, independent of the oracle) must reject a
    proof with any single word changed, section by section, and agree with the oracle's verifier on each."""
    desc = synth.ecdsa_shape_circuit(6, seed=5)
    oc = oracle.OracleCircuit(desc)
    gc = glp.Circuit(ctx, desc)
    proof = gc.prove()
    assert gc.verify(proof) and oc.verify(proof) == 0
    sec = _sections(oc, desc)
    rng = np.random.default_rng(7)
    L = len(proof)
    picks = [sl.start + int(rng.integers(0, max(1, (sl.stop or L) - sl.start))) for sl in sec.values()]
    picks += [L - 1 - len(desc.public_inputs) if len(desc.public_inputs) else L - 1, L - 2]     # pow witness region / tail
    picks += [int(x) for x in rng.integers(0, L, 24)]
    for pos in picks:
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % 0xFFFFFFFF00000001
        assert not gc.verify(bad), "tampered word %d accepted" % pos
        assert oc.verify(bad) != 0
    bad = proof.copy()
    bad[0] = 0xFFFFFFFF00000001          # not canonical
    assert not gc.verify(bad)


def test_verifier_rejects_unsatisfied_witness(ctx, oracle):
    desc = synth.arith_circuit(7, seed=11)
    w = desc.wires.copy()
    w[7, 50] = np.uint64((int(w[7, 50]) + 1) % glp.P)        # break one ArithmeticGate row
    gc = glp.Circuit(ctx, desc)
    assert not gc.verify(gc.prove(wires=w))
    assert gc.verify(gc.prove())


def test_stepped_session_with_device_resident_wires(ctx, oracle):
    """glp_session_begin(wires_on_device = 1): the witness already sits in HBM (what bench.py times through
    glp_prove_device); same proof as from host wires."""
    desc = synth.ecdsa_shape_circuit(7, seed=21)
    gc = glp.Circuit(ctx, desc)
    ref = gc.prove()
    w = np.ascontiguousarray(desc.wires)
    dptr = ctx.dev_alloc(w.nbytes)                           # glp_dev_alloc / glp_dev_upload: no HIP runtime on the caller's side
    ctx.dev_upload(dptr, w)
    back = np.empty_like(w)
    ctx.dev_download(dptr, back)
    assert (back == w).all()
    assert (gc.prove_device(dptr) == ref).all()
    s = glp.Session(gc, dev_wires_ptr=dptr)
    ch = oracle.Challenger()
    ch.observe(gc.digest())
    ch.observe(s.public_inputs_hash)
    ch.observe(s.wires_cap)
    nch = desc.num_challenges
    betas, gammas = ch.get_n(nch), ch.get_n(nch)
    zs_cap = s.partial_products(betas, gammas)
    cap = 4 << desc.cap_height
    assert (s.wires_cap.reshape(-1) == ref[:cap]).all() and (zs_cap.reshape(-1) == ref[cap:2 * cap]).all()
    s.end()
    ctx.dev_free(dptr)
    with pytest.raises(glp.GlpError):
        ctx.dev_free(dptr)                                   # not (any longer) a live allocation of this context


def test_pow_search_is_the_smallest_witness(ctx, oracle):
    """glp_pow_search on an arbitrary sponge state: the returned witness satisfies the leading-zero condition under the oracle's
    transcript and no smaller one does."""
    import ctypes
    desc = synth.arith_circuit(5, synth.Config.standard_recursion_config(), seed=8)
    gc = glp.Circuit(ctx, desc)
    s = glp.Session(gc)
    ch = oracle.Challenger()
    ch.observe(np.arange(1, 12, dtype=np.uint64))            # 11 elements: one duplex, three pending inputs
    raw = np.frombuffer(ctypes.string_at(ch._buf, 8 * 21), dtype=np.uint64)
    nin = int(np.frombuffer(ctypes.string_at(ctypes.addressof(ch._buf) + 8 * 20, 4), dtype=np.int32)[0])
    assert nin == 3
    bits = 10
    w = s.pow_search(raw[:12], raw[12:12 + nin], bits)
    s.end()

    def response(cand):
        c2 = oracle.Challenger()
        ctypes.memmove(c2._buf, ch._buf, len(ch._buf))
        c2.observe([cand])
        return c2.get()
    assert response(w) >> (64 - bits) == 0
    assert all(response(c) >> (64 - bits) != 0 for c in range(w))


def test_stepped_session_enforces_order(ctx):
    desc = synth.arith_circuit(5, synth.Config.standard_recursion_config(), seed=3)
    gc = glp.Circuit(ctx, desc)
    s = glp.Session(gc)
    with pytest.raises(glp.GlpError):
        s.quotient([1] * desc.num_challenges)          # partial_products must come first
    with pytest.raises(glp.GlpError):
        s.proof()                                      # nothing to hand out yet
    with pytest.raises(glp.GlpError):
        s.partial_products([0xFFFFFFFF00000001] * desc.num_challenges, [7] * desc.num_challenges)   # p itself: not canonical
    s.partial_products([5] * desc.num_challenges, [7] * desc.num_challenges)
    with pytest.raises(glp.GlpError):
        s.fri_commit()
    s.end()
