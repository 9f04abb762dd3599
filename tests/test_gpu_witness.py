"""GPU parity tests for glp_witness_fill (row-local witness generation): the HIP kernel against
the oracle's per-generator restatement, bit for bit, and end to end: a witness whose derived columns were produced on the
GPU proves and verifies."""
import numpy as np
import pytest

import plonky2_lib_amd as glp
import plonky2_lib_amd.synth as synth
from test_oracle_witness import FAMILIES, rows_of_gate, scramble_derived

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = glp.Context(0)
    yield c
    c.close()


def _fill_on_gpu(ctx, gc, w, only_advice=False):
    w = np.ascontiguousarray(w)
    d = ctx.dev_alloc(w.nbytes)
    ctx.dev_upload(d, w)
    gc.witness_fill(d, only_advice=only_advice)
    out = np.empty_like(w)
    ctx.dev_download(d, out)
    return out, d


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_witness_fill_parity(ctx, oracle, family):
    desc = FAMILIES[family]()
    oc = oracle.OracleCircuit(desc)
    gc = glp.Circuit(ctx, desc)
    rng = np.random.default_rng(31)
    w, touched = scramble_derived(desc, rng)
    ref = oc.witness_fill(w)
    got, dptr = _fill_on_gpu(ctx, gc, w)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, "first mismatch at (column, row) %s" % bad[0].tolist()
    # the cells the kernel wrote are exactly the role-1 columns of each row's gate
    for gi, g in enumerate(desc.gates):
        role = gc.witness_columns(gi)
        rows = rows_of_gate(desc, gi)
        if len(rows) == 0:
            continue
        changed = (got != w)[:, rows].any(axis=1)
        assert not (changed & (role != 1)).any(), "gate %d: a column outside its output set was written" % gi
        assert (got[np.ix_(role == 1, rows)] == desc.wires[np.ix_(role == 1, rows)]).all()
    # the GPU-generated witness, still in HBM, goes straight into the prover
    proof = gc.prove_device(dptr)
    assert gc.verify(proof) and oc.verify(proof) == 0
    rc, ref_proof = oc.prove(wires=ref)
    assert (proof == ref_proof).all()
    ctx.dev_free(dptr)
    gc.free()


def test_keccak_shape_with_gpu_generated_bits_and_limbs(ctx, oracle):
    """BASELINE config 2 shape at 2^13 rows: every bit / limb / u32-result column is ERASED (zeroed) and produced on the GPU;
    the completed witness equals the builder's and its proof verifies."""
    desc = synth.keccak_shape_circuit(13, seed=2)
    gc = glp.Circuit(ctx, desc)
    w = desc.wires.copy()
    erased = 0
    for gi, g in enumerate(desc.gates):
        role = gc.witness_columns(gi)
        rows = rows_of_gate(desc, gi)
        if len(rows) and (role == 1).any():
            w[np.ix_(role == 1, rows)] = 0
            erased += int((role == 1).sum()) * len(rows)
    assert erased > 0 and (w != desc.wires).any()
    got, dptr = _fill_on_gpu(ctx, gc, w)
    assert (got == desc.wires).all()
    proof = gc.prove_device(dptr)
    desc.circuit_digest = gc.digest()
    oc = oracle.OracleCircuit(desc, cs_cap=gc.constants_sigmas_cap())
    assert gc.verify(proof) and oc.verify(proof) == 0
    assert (proof == gc.prove()).all()                       # same proof as from the builder's complete witness
    ctx.dev_free(dptr)
    gc.free()


def test_only_advice_mode(ctx, oracle):
    desc = synth.ecdsa_shape_circuit(8, seed=41)
    oc = oracle.OracleCircuit(desc)
    gc = glp.Circuit(ctx, desc)
    rng = np.random.default_rng(8)
    w, touched = scramble_derived(desc, rng, only_advice=True)
    got, dptr = _fill_on_gpu(ctx, gc, w, only_advice=True)
    assert (got == oc.witness_fill(w, only_advice=True)).all()
    assert (got[:desc.num_routed_wires] == w[:desc.num_routed_wires]).all()       # routed columns untouched
    # and a scrambled ROUTED output stays scrambled in this mode (it is the CPU pass's job)
    w2 = w.copy()
    gi = next(i for i, g in enumerate(desc.gates) if g["type"] == synth.GATE_U32_ARITHMETIC)
    row = int(rows_of_gate(desc, gi)[0])
    w2[3, row] = np.uint64(12345)
    ctx.dev_upload(dptr, np.ascontiguousarray(w2))
    gc.witness_fill(dptr, only_advice=True)
    back = np.empty_like(w2)
    ctx.dev_download(dptr, back)
    assert int(back[3, row]) == 12345
    ctx.dev_free(dptr)
    gc.free()


def test_headline_shape_2_16(ctx, oracle):
    """136 wires x 2^16 rows, the secp256k1 gate set with many rows per gate: GPU fill == oracle fill on a scrambled witness."""
    desc = synth.ecdsa_shape_circuit(16, seed=77, rows_per_gate=300)
    oc = oracle.OracleCircuit(desc)
    gc = glp.Circuit(ctx, desc)
    rng = np.random.default_rng(9)
    w, touched = scramble_derived(desc, rng)
    got, dptr = _fill_on_gpu(ctx, gc, w)
    assert (got == oc.witness_fill(w)).all()
    assert gc.verify(gc.prove_device(dptr))
    ctx.dev_free(dptr)
    gc.free()
