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


@pytest.mark.parametrize("family", ["ecdsa", "zkdsa", "keccak"])
def test_staged_witness_routed_columns_only(ctx, oracle, family):
    """glp_witness_stage(GLP_WITNESS_ROUTED_ONLY) + glp_prove_staged: only the routed columns cross PCIe (from page-locked host
    memory, on the copy stream), the advice columns are zero-filled and derived on the GPU.  The proof must be word for word the
    proof of the witness the ORACLE's generators derive from the same routed columns, both verifiers accept it, and a complete
    staged witness (all columns) gives glp_prove's proof."""
    desc = FAMILIES[family]()
    oc = oracle.OracleCircuit(desc)
    gc = glp.Circuit(ctx, desc)
    nr, n = desc.num_routed_wires, 1 << desc.degree_bits
    pinned = ctx.host_alloc((nr, n))
    pinned[:] = desc.wires[:nr]
    st = gc.stage_witness(pinned, routed_only=True)
    st2 = gc.stage_witness(pinned, routed_only=True)          # two uploads in flight, as in a pipeline
    proof = gc.prove_staged(st)
    assert (gc.prove_staged(st2) == proof).all()
    st.free(); st2.free()
    w0 = desc.wires.copy()
    w0[nr:] = 0
    want_w = oc.witness_fill(w0, only_advice=True)
    rc, ref = oc.prove(wires=want_w)
    assert rc == 0 and (proof == ref).all(), "first mismatch at word %d" % int(np.argmax(proof != ref))
    if (want_w == desc.wires).all():                           # every advice wire of this witness is generated or zero
        assert gc.verify(proof) and oc.verify(proof) == 0
    full = gc.stage_witness(np.ascontiguousarray(desc.wires))  # pageable host memory, every column
    assert (gc.prove_staged(full) == gc.prove()).all()
    full.free()
    with pytest.raises(glp.GlpError):
        gc.stage_witness(pinned[: nr - 1], routed_only=True)
    ctx.host_free(pinned)
    gc.free()


def test_staged_witness_real_secp256k1_circuit(ctx, oracle):
    """The real one-signature circuit (2^17 rows): 80 of 136 columns uploaded, 56 derived on the GPU -- the same proof words as from
    the complete host witness (every advice wire of the reference's generators is row-local)."""
    from plonky2_lib_amd import gadgets_ecdsa as E
    c = E.ecdsa_circuit(E.random_signatures(1, seed=5))
    gc = glp.Circuit(ctx, c)
    want = gc.prove()
    pinned = ctx.host_alloc((c.num_routed_wires, 1 << c.degree_bits))
    pinned[:] = c.wires[:c.num_routed_wires]
    st = gc.stage_witness(pinned, routed_only=True)
    got = gc.prove_staged(st)
    st.free()
    ctx.host_free(pinned)
    assert (got == want).all() and gc.verify(got)
    gc.free()
