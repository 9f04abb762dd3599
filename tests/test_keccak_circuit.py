"""The reference's Keccak-256 circuit rebuilt gadget for gadget (plonky2-lib_amd/gadgets.py: CircuitBuilderB32
[REF src/u32/interleaved_u32.rs] + hash_keccak256 [REF src/hash/keccak256.rs:79-165]) -- BASELINE config 2 as a real circuit.
What the reference's own tests pin at this boundary [REF src/hash/keccak256.rs:196-252,256-334]: the circuit's eight public inputs
are the little-endian u32 limbs of keccak256(message), and verify(prove(witness)) is Ok.  CPU half: the digests, the structure,
the checker's prover and verifier on the circuit.  (GPU half: tests/test_gpu_keccak_circuit.py.)"""
import numpy as np
import pytest

from plonky2_lib_amd import gadgets
import plonky2_lib_amd.synth as synth
from test_oracle_keccak import LONG_IN, LONG_OUT, SHORT
from test_oracle_witness import scramble_derived


@pytest.fixture(scope="module")
def short_circuits():
    return [gadgets.keccak256_circuit(bytes.fromhex(m)) for m, _ in SHORT]


def test_public_inputs_are_the_reference_digests(short_circuits):
    for c, (_, dig) in zip(short_circuits, SHORT):
        assert c.digest_bytes.hex() == dig                   # computed THROUGH the gates' semantics (interleave / add / uninterleave)
        assert len(c.public_inputs) == 8 and int(c.public_inputs.max()) < 1 << 32
    # "build circuit once" [REF src/hash/keccak256.rs:214-231]: gates, constants and wiring do not depend on the message
    a = short_circuits[0]
    for c in short_circuits[1:]:
        assert c.gates == a.gates and (c.constants == a.constants).all() and (c.sigmas == a.sigmas).all()
        assert not (c.wires == a.wires).all()


def test_operation_counts_follow_the_gadget(short_circuits):
    """One Keccak-f[1600] through [REF src/hash/keccak256.rs:79-128]: per round theta = 10 five-way XORs (5 interleaves, 2 three-term
    adds, one uninterleave of each kind), 5 rotations (3 U32Arithmetic operations each) and 30 64-bit XORs; rho/pi = 24 rotations; chi =
    25 x (not, and, xor) on 64 bits; iota = 1 XOR; a 32-bit AND/XOR = 2 interleaves + 1 add + 1 uninterleave."""
    ops = short_circuits[0].gate_ops
    per_round = dict(U32InterleaveGate=50 + 120 + 200 + 4, UninterleaveToU32Gate=10 + 60 + 100 + 2, UninterleaveToB32Gate=10,
                     U32ArithmeticGate=3 * 5 + 3 * 24, U32SubtractionGate=50, ArithmeticGate=40 + 60 + 100 + 2)
    for k, v in per_round.items():
        assert ops[k] == 24 * v, k
    c = short_circuits[0]
    assert c.degree_bits == 13 and c.gadget_rows == 6341
    kinds = {g["type"] for g in c.gates}
    assert kinds == {synth.GATE_NOOP, synth.GATE_CONSTANT, synth.GATE_PUBLIC_INPUT, synth.GATE_ARITHMETIC, synth.GATE_POSEIDON,
                     synth.GATE_U32_INTERLEAVE, synth.GATE_UNINTERLEAVE_U32, synth.GATE_UNINTERLEAVE_B32, synth.GATE_U32_ARITHMETIC,
                     synth.GATE_U32_SUBTRACTION}             # SURVEY.md section 8 row Q, cfg 2, minus U32AddMany (the gadget never adds three u32)


def test_checker_proves_and_verifies_the_circuit(oracle, short_circuits):
    c = short_circuits[2]
    assert (oracle.hash_no_pad([int(x) for x in c.public_inputs]) == c.pi_hash).all()
    oc = oracle.OracleCircuit(c)
    rc, proof = oc.prove()
    assert rc == 0 and oc.verify(proof) == 0
    assert (proof[-8:] == c.public_inputs).all()
    bad = proof.copy()
    bad[-1] ^= np.uint64(1)                                   # claim another digest
    assert oc.verify(bad) != 0
    # a witness that breaks one XOR (an uninterleave output flipped, its copies left alone) is not provable
    gi = next(i for i, g in enumerate(c.gates) if g["type"] == synth.GATE_UNINTERLEAVE_U32)
    row = int(np.nonzero(c.constants[c.gates[gi]["selector_index"]] == np.uint64(gi))[0][100])
    w = c.wires.copy()
    w[2, row] ^= np.uint64(1)
    rc, p2 = oc.prove(wires=w)
    assert rc != 0 or oc.verify(p2) != 0


def test_row_local_generators_rebuild_the_advice(oracle, short_circuits):
    c = short_circuits[1]
    oc = oracle.OracleCircuit(c)
    w, touched = scramble_derived(c, np.random.default_rng(8))
    assert touched.sum() > 500_000
    filled = oc.witness_fill(w)
    rewritten = filled != w                                  # cells no generator writes (columns past a gate's last wire) keep the scramble
    assert rewritten.sum() > 400_000 and (filled[rewritten] == c.wires[rewritten]).all()
    # everything a constraint reads is back: the cells still different from the built witness lie outside every gate's wires
    used_cols = {synth.GATE_U32_INTERLEAVE: 2 * 3 + 32 * 3, synth.GATE_UNINTERLEAVE_U32: 3 * 2 + 64 * 2, synth.GATE_UNINTERLEAVE_B32: 3 * 2 + 64 * 2,
                 synth.GATE_U32_ARITHMETIC: 6 * 3 + 32 * 3, synth.GATE_U32_SUBTRACTION: 5 * 6 + 16 * 6, synth.GATE_ARITHMETIC: 80,
                 synth.GATE_CONSTANT: 2, synth.GATE_POSEIDON: 135}
    for gi, g in enumerate(c.gates):
        if g["type"] in used_cols:
            rows = np.nonzero(c.constants[g["selector_index"]] == np.uint64(gi))[0]
            k = used_cols[g["type"]]
            assert (filled[:k][:, rows] == c.wires[:k][:, rows]).all(), g


def test_four_block_circuit_long_vector():
    """`test_keccak256_long` [REF src/hash/keccak256.rs:279-334]: `add_virtual_hash_input_target(4, KECCAK256_R)`, a 532-byte message
    (4 rate blocks), public inputs hex = the digest."""
    msg = bytes.fromhex(LONG_IN)
    assert len(msg) == 532
    c = gadgets.keccak256_circuit(msg, blocks_num=4)
    assert c.digest_bytes.hex() == LONG_OUT
    assert c.degree_bits == 15                                # BASELINE config 2's size
    # a one-block message through the same four-block circuit: the block flags switch the later permutations off
    d = gadgets.keccak256_circuit(bytes.fromhex(SHORT[2][0]), blocks_num=4)
    assert d.digest_bytes.hex() == SHORT[2][1]
    assert d.gates == c.gates and (d.sigmas == c.sigmas).all() and (d.constants == c.constants).all()
    with pytest.raises(ValueError):
        gadgets.keccak256_circuit(msg, blocks_num=3)


def b32_gadget_circuit(seed=5):
    """BASELINE config 1: every method of the reference's `CircuitBuilderB32` [REF src/u32/interleaved_u32.rs:19-54] on random u32 / u64
    inputs in one circuit, each result checked against Python integers while wiring."""
    rng = np.random.default_rng(seed)
    gb = gadgets.GadgetBuilder()
    M = (1 << 32) - 1
    r32 = lambda: int(rng.integers(0, 1 << 32))
    rot = lambda v, n, w: ((v << n) | (v >> (w - n))) & ((1 << w) - 1) if n % w else v
    x, y, z = r32(), r32(), r32()
    X, Y, Z = gb.target(x), gb.target(y), gb.target(z)
    v = gb.val
    assert v[gb.not_u32(X)] == x ^ M
    assert v[gb.xor_u32(X, Y)] == x ^ y and v[gb.and_u32(X, Y)] == x & y
    for n in (1, 7, 13, 31):
        assert v[gb.lsh_u32(X, n)] == (x << n) & M and v[gb.rsh_u32(X, n)] == x >> n
        assert v[gb.lrot_u32(X, n)] == rot(x, n, 32) and v[gb.rrot_u32(X, n)] == rot(x, 32 - n, 32)
    assert v[gb.rsh_u32(X, 0)] == x
    for bit in (0, 1):
        assert v[gb.conditional_u32(X, Y, gb.target(bit))] == (x if bit else y)
    a, b = gb.and_xor_u32(X, Y)
    assert v[a] == gadgets._interleave(x & y) and v[b] == gadgets._interleave(x ^ y)
    for k in range(0, 8):                                     # unsafe_xor_many_u32: every branch of its case split
        vals = [r32() for _ in range(k)]
        want = 0
        for t in vals:
            want ^= t
        assert v[gb.unsafe_xor_many_u32([gb.target(t) for t in vals])] == want
    lo, hi = r32(), r32()
    w64 = lo | (hi << 32)
    W = [gb.target(lo), gb.target(hi)]
    for n in (1, 31, 33, 44, 62):
        out = gb.lrot_u64(W, n)
        assert v[out[0]] | (v[out[1]] << 32) == rot(w64, n, 64)
    o = gb.not_u64(W)
    assert v[o[0]] | (v[o[1]] << 32) == w64 ^ ((1 << 64) - 1)
    V = [gb.target(y), gb.target(z)]
    o = gb.xor_u64(W, V)
    assert v[o[0]] == lo ^ y and v[o[1]] == hi ^ z
    o = gb.and_u64(W, V)
    assert v[o[0]] == lo & y and v[o[1]] == hi & z
    o = gb.conditional_u64(W, V, gb.target(1))
    assert (v[o[0]], v[o[1]]) == (lo, hi)
    for t in (a, b, o[0]):
        gb.register_public_input(t)
    return gb.build()


def test_b32_gadgets(oracle):
    c = b32_gadget_circuit()
    assert len(c.public_inputs) == 3
    oc = oracle.OracleCircuit(c)
    rc, proof = oc.prove()
    assert rc == 0 and oc.verify(proof) == 0
