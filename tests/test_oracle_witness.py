"""Row-local witness generators (oracle/gl_witness_oracle.c, the checker of glp_witness_fill): on every synthetic circuit
family, scrambling everything a row's generators are supposed to derive and running them again must give back the witness
that plonky2-lib_amd/synth.py built independently in Python, and the gate constraints must accept it."""
import numpy as np
import pytest

import plonky2_lib_amd.synth as synth

S = synth


def generator_inputs(g, desc):
    """Wire columns a gate's generators READ (their `dependencies()`), restated here from the gate layouts."""
    t, p0, p1 = g["type"], g["p0"], g["p1"]
    if t == S.GATE_ARITHMETIC:
        return [4 * i + k for i in range(p0) for k in range(3)]
    if t == S.GATE_POSEIDON:
        return list(range(12)) + [24]
    if t == S.GATE_U32_INTERLEAVE:
        return [2 * i for i in range(p0)]
    if t in (S.GATE_UNINTERLEAVE_U32, S.GATE_UNINTERLEAVE_B32):
        return [3 * i for i in range(p0)]
    if t == S.GATE_U32_ARITHMETIC:
        return [6 * i + k for i in range(p0) for k in range(3)]
    if t == S.GATE_U32_ADD_MANY:
        return [(p0 + 3) * i + k for i in range(p1) for k in range(p0 + 1)]
    if t == S.GATE_U32_SUBTRACTION:
        return [5 * i + k for i in range(p0) for k in range(3)]
    if t == S.GATE_U32_RANGE_CHECK:
        return list(range(p0))
    if t == S.GATE_COMPARISON:
        return [0, 1]
    if t == S.GATE_BASE_SUM:
        return [0]
    if t == S.GATE_RANDOM_ACCESS:
        bits, copies = p0, p1 & 0xFFFF
        vs = 1 << bits
        return [(2 + vs) * c + k for c in range(copies) for k in [0] + list(range(2, 2 + vs))]
    if t == S.GATE_CONSTANT:
        return []
    return None          # no generator: Noop, PublicInput


def rows_of_gate(desc, gi):
    g = desc.gates[gi]
    return np.nonzero(desc.constants[g["selector_index"]] == np.uint64(gi))[0]


def scramble_derived(desc, rng, only_advice=False):
    """Random field elements in every column of a generator row that is not an input of its generators."""
    w = desc.wires.copy()
    touched = np.zeros(w.shape, bool)
    for gi, g in enumerate(desc.gates):
        ins = generator_inputs(g, desc)
        if ins is None:
            continue
        rows = rows_of_gate(desc, gi)
        cols = np.array([c for c in range(desc.num_wires) if c not in set(ins) and (not only_advice or c >= desc.num_routed_wires)], dtype=np.int64)
        if len(rows) and len(cols):
            w[np.ix_(cols, rows)] = synth.gl.rand(rng, (len(cols), len(rows)))
            touched[np.ix_(cols, rows)] = True
    return w, touched


FAMILIES = {
    "ecdsa": lambda: synth.ecdsa_shape_circuit(7, seed=11),
    "keccak": lambda: synth.keccak_shape_circuit(7, seed=12),
    "smt": lambda: synth.smt_shape_circuit(6, seed=13),
    "zkdsa": lambda: synth.zkdsa_circuit(3),
    "u32": lambda: synth.u32_circuit(6, seed=14),
}


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_generators_rebuild_the_synthetic_witness(oracle, family):
    desc = FAMILIES[family]()
    oc = oracle.OracleCircuit(desc)
    assert (oc.witness_fill(desc.wires) == desc.wires).all()            # idempotent on a complete witness
    rng = np.random.default_rng(5)
    w, touched = scramble_derived(desc, rng)
    assert touched.any()
    filled = oc.witness_fill(w)
    # every cell the generators write is back at the value the Python builder computed; cells no generator writes keep the scramble
    rewritten = filled != w
    assert (filled[rewritten] == desc.wires[rewritten]).all()
    assert rewritten.sum() > 0
    rc, proof = oc.prove(wires=filled)
    assert rc == 0 and oc.verify(proof) == 0                            # the gate constraints accept the generated rows
    rc, bad = oc.prove(wires=w)
    assert oc.verify(bad) != 0                                          # ... and do not accept the scrambled ones


def test_only_advice_leaves_routed_wires_alone(oracle):
    desc = synth.ecdsa_shape_circuit(7, seed=21)
    oc = oracle.OracleCircuit(desc)
    rng = np.random.default_rng(6)
    w, touched = scramble_derived(desc, rng, only_advice=True)
    assert not touched[:desc.num_routed_wires].any()
    filled = oc.witness_fill(w, only_advice=True)
    assert (filled[:desc.num_routed_wires] == desc.wires[:desc.num_routed_wires]).all()
    rc, proof = oc.prove(wires=filled)
    assert rc == 0 and oc.verify(proof) == 0
    # the limb columns of the plonky2_u32 gates are all advice: 56 of 136 columns in this configuration
    assert desc.num_wires - desc.num_routed_wires == 56


def test_reference_generator_semantics(oracle):
    """The three generators read from the reference, on hand-made rows: big-endian bit order and the even / odd split
    [REF src/u32/gates/interleave_u32.rs:289-318, uninterleave_to_u32.rs:332-369, uninterleave_to_b32.rs:335-372]."""
    desc = synth.u32_circuit(6, seed=3)
    oc = oracle.OracleCircuit(desc)
    w = desc.wires.copy()
    gi = next(i for i, g in enumerate(desc.gates) if g["type"] == S.GATE_U32_INTERLEAVE)
    row = int(rows_of_gate(desc, gi)[0])
    nops = desc.gates[gi]["p0"]
    w[0, row] = 0x80000001                                               # x of op 0
    f = oc.witness_fill(w)
    bits = [int(b) for b in f[2 * nops:2 * nops + 32, row]]
    assert bits == [1] + [0] * 30 + [1]                                  # bit wire k = bit (31 - k)
    assert int(f[1, row]) == (1 << 62) | 1                               # interleaved: bit i -> position 2 i
    gu = next(i for i, g in enumerate(desc.gates) if g["type"] == S.GATE_UNINTERLEAVE_U32)
    gb = next(i for i, g in enumerate(desc.gates) if g["type"] == S.GATE_UNINTERLEAVE_B32)
    for gidx, b32 in ((gu, False), (gb, True)):
        row = int(rows_of_gate(desc, gidx)[0])
        w[0, row] = 0b1001
        f = oc.witness_fill(w)
        ev, od = int(f[1, row]), int(f[2, row])
        # bit pairs from the top: pair j = (bit at shift + 1 -> evens, bit at shift -> odds); the last two pairs are (1,0) and (0,1)
        assert (ev, od) == ((0b10, 0b01) if not b32 else (0b0100, 0b0001))
