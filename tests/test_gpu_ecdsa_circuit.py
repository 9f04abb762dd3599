"""BASELINE config 3 on the GPU as the REAL circuit: the reference's `verify_message_circuit` [REF src/ecdsa/gadgets/ecdsa.rs:136-159] rebuilt in
plonky2-lib_amd/gadgets_ecdsa.py (98 687 rows per signature: nonnative secp256k1 arithmetic on u32 limbs, 4-bit windowed fixed-base
multiplication, GLV + 2-bit windowed double-scalar multiplication), proved by the HIP library through the C ABI -- mirrors
`test_batch_ecdsa_circuit_with_config` [REF src/ecdsa/gadgets/ecdsa.rs:214-353]: prove, then verify."""
import numpy as np
import pytest

import plonky2_lib_amd as glp
from plonky2_lib_amd import gadgets_ecdsa as E
from test_oracle_witness import scramble_derived

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = glp.Context(0)
    yield c
    c.close()


def test_one_signature(ctx, oracle):
    sigs = E.random_signatures(2, seed=11)
    c = E.ecdsa_circuit(sigs[:1])
    assert c.degree_bits == 17
    gc = glp.Circuit(ctx, c)
    oc = oracle.OracleCircuit(c)
    proof = gc.prove()
    assert gc.verify(proof) and oc.verify(proof) == 0
    rc, ref = oc.prove()                                       # the checker's prover on the same 2^17-row witness
    assert rc == 0 and (proof == ref).all(), "first mismatch at word %d" % int(np.argmax(proof != ref))
    bad = proof.copy()
    bad[len(bad) // 3] ^= np.uint64(1)
    assert not gc.verify(bad) and oc.verify(bad) != 0
    # "build once": another signature is another witness of the same circuit
    d = E.ecdsa_circuit(sigs[1:])
    assert d.gates == c.gates and (d.sigmas == c.sigmas).all() and (d.constants == c.constants).all()
    p2 = gc.prove(wires=d.wires)
    assert gc.verify(p2) and not (p2 == proof).all()
    # a witness with one limb of a product changed proves nothing
    w = c.wires.copy()
    import plonky2_lib_amd.synth as synth
    gi = next(i for i, g in enumerate(c.gates) if g["type"] == synth.GATE_U32_ARITHMETIC)
    row = int(np.nonzero(c.constants[c.gates[gi]["selector_index"]] == np.uint64(gi))[0][1000])
    w[3, row] ^= np.uint64(1)
    assert not gc.verify(gc.prove(wires=w))
    # GPU witness generation: all advice columns (2-bit limbs, comparison chunks, access bits, inverses) from the routed wires
    ws, _ = scramble_derived(c, np.random.default_rng(3), only_advice=True)
    ws = np.ascontiguousarray(ws)
    dptr = ctx.dev_alloc(ws.nbytes)
    ctx.dev_upload(dptr, ws)
    gc.witness_fill(dptr, only_advice=True)
    p3 = gc.prove_device(dptr)
    ctx.dev_free(dptr)
    assert gc.verify(p3)
    gc.free()


def test_two_signatures_batch_circuit(ctx, oracle):
    """`batch_verify_message_circuit` [REF src/ecdsa/gadgets/ecdsa.rs:161-191] on two signatures: 2^18 rows (the fixed-base tables are shared)"""
    c = E.ecdsa_circuit(E.random_signatures(2, seed=12))
    assert c.degree_bits == 18 and 0 <= 2 * 98687 + 7714 + 1 - c.gadget_rows < 16        # partly filled rows are shared between the two
    gc = glp.Circuit(ctx, c)
    proof = gc.prove()
    assert gc.verify(proof)
    oc = oracle.OracleCircuit(c, cs_cap=gc.constants_sigmas_cap())
    assert oc.verify(proof) == 0
    gc.free()


def _gate_row(c, gtype, which=0, nth=100):
    """(gate index, a trace row) of the `which`-th gate of type `gtype`: rows carry their gate's index in the selector column"""
    import plonky2_lib_amd.synth as synth  # noqa: F401
    gi = [i for i, g in enumerate(c.gates) if g["type"] == gtype][which]
    rows = np.nonzero(c.constants[c.gates[gi]["selector_index"]] == np.uint64(gi))[0]
    return gi, int(rows[min(nth, len(rows) - 1)])


def _bump(w, col, row):
    w[col, row] = np.uint64((int(w[col, row]) + 1) % glp.P)


def test_headline_ten_signatures_2_20(ctx, oracle):
    """THE workload `bench.py` times by default: `gadgets_ecdsa.ecdsa_circuit(random_signatures(10, ...), min_log_n=20)` -- ten
    secp256k1 signatures in one 2^20-row x 136-wire trace, 17 gates in 4 selector groups (ten base-4 limb gates: the two
    k_quotient_limbs groups), `test_batch_ecdsa_circuit_with_config` [REF src/ecdsa/gadgets/ecdsa.rs:214-353; driver
    REF src/bin/perf.rs:7-9]: prove, then verify.  Both verifiers (the library's and the oracle's) must accept; six tampered words,
    a U32AddMany limb off by one and a ComparisonGate chunk off by one must be rejected; `glp_prove_device` (what the bench
    calls) and `glp_prove` (host witness) must return the same words."""
    import plonky2_lib_amd.synth as synth
    SEED = 0x5EED0003                                          # bench.py's seed for rank 0
    c = E.ecdsa_circuit(E.random_signatures(10, seed=SEED), min_log_n=20)
    assert c.degree_bits == 20 and c.num_wires == 136 and len(c.gates) == 17
    assert sum(1 for g in c.gates if g["type"] in (synth.GATE_U32_ARITHMETIC, synth.GATE_U32_ADD_MANY, synth.GATE_U32_SUBTRACTION,
                                                    synth.GATE_U32_RANGE_CHECK)) == 10
    gc = glp.Circuit(ctx, c)
    proof = gc.prove()                                         # glp_prove: witness from host memory
    c.circuit_digest = gc.digest()
    oc = oracle.OracleCircuit(c, cs_cap=gc.constants_sigmas_cap())
    assert gc.verify(proof) and oc.verify(proof) == 0
    w = np.ascontiguousarray(c.wires)
    dptr = ctx.dev_alloc(w.nbytes)
    ctx.dev_upload(dptr, w)
    assert (gc.prove_device(dptr) == proof).all()              # the entry point the bench times
    rng = np.random.default_rng(20)
    for pos in [int(x) for x in rng.integers(0, len(proof), 4)] + [0, len(proof) - 1]:
        bad = proof.copy(); bad[pos] = np.uint64((int(bad[pos]) + 1) % glp.P)
        assert not gc.verify(bad) and oc.verify(bad) != 0, pos
    # one base-4 limb of a U32AddMany sum (limb columns start at (num_addends + 3) * num_ops) and one chunk of a ComparisonGate
    # (first-input chunks start at wire 4): the witness no longer satisfies the gate, no verifier may accept the proof
    for gtype, col_of in ((synth.GATE_U32_ADD_MANY, lambda g: (g["p0"] + 3) * g["p1"]), (synth.GATE_COMPARISON, lambda g: 4)):
        gi, row = _gate_row(c, gtype, which=-1)
        _bump(w, col_of(c.gates[gi]), row)
        ctx.dev_upload(dptr, w)
        badp = gc.prove_device(dptr)
        assert not gc.verify(badp) and oc.verify(badp) != 0, (gtype, gi, row)
        w[:, row] = c.wires[:, row]
    ctx.dev_free(dptr)
    gc.free()


def test_perf_rs_twenty_signatures_2_21(ctx, oracle):
    """The literal `perf` workload: `test_batch_ecdsa_circuit_with_config(20, standard_ecc_config)` [REF src/bin/perf.rs:7-9]
    proves TWENTY signatures in one circuit.  At this repository's gate density (98 687 rows per signature; row count vs
    plonky2's builder: parity unpinned) that is a 2^21-row trace -- past the 2^20 tile of the contiguous pass, so the strided pass
    runs on 512-row LDS tiles.  Prove, then verify with both verifiers; a tampered word and a broken limb must be rejected."""
    import plonky2_lib_amd.synth as synth
    c = E.ecdsa_circuit(E.random_signatures(20, seed=0x5EED0003))
    assert c.degree_bits == 21 and len(c.gates) == 17
    gc = glp.Circuit(ctx, c)
    proof = gc.prove()
    c.circuit_digest = gc.digest()
    oc = oracle.OracleCircuit(c, cs_cap=gc.constants_sigmas_cap())
    assert gc.verify(proof) and oc.verify(proof) == 0
    bad = proof.copy(); bad[len(bad) // 2] ^= np.uint64(1)
    assert not gc.verify(bad) and oc.verify(bad) != 0
    w = c.wires.copy()
    gi, row = _gate_row(c, synth.GATE_U32_ARITHMETIC, nth=(1 << 20) // 8)
    _bump(w, 6 * c.gates[gi]["p0"], row)                      # first base-4 limb of op 0's low output word
    assert not gc.verify(gc.prove(wires=w))
    gc.free()


@pytest.mark.parametrize("which", ["fixed_base", "glv", "msm", "bitwise", "windowed"])
def test_scalar_multiplication_gadgets(ctx, oracle, which):
    """The three scalar-multiplication circuits the verification is made of, each as the reference tests it on its own:
    `test_fixed_base` [REF src/ecdsa/gadgets/curve_fixed_base.rs:88-117], `test_glv_gadget` [REF src/ecdsa/gadgets/glv.rs:195-224],
    `test_curve_msm` [REF src/ecdsa/gadgets/curve_msm.rs:98-137] (full 256-bit scalars there: four 2-bit windows per limb)."""
    rng = np.random.default_rng({"fixed_base": 1, "glv": 2, "msm": 3, "bitwise": 4, "windowed": 5}[which])
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (E.FN - 1) + 1
    eb = E.EcdsaBuilder()
    if which == "fixed_base":
        n = rnd()
        got = eb.fixed_base_curve_mul(E.G, eb.virtual_nonnative(n))
        want = E.pt_mul(n, E.G)
    elif which == "glv":
        p, k = E.pt_mul(rnd(), E.G), rnd()
        got = eb.glv_mul(eb.constant_affine_point(p), eb.constant_biguint(k) + [eb.zero_u32()] * (8 - len(E._u32_digits(k))))
        want = E.pt_mul(k, p)
    elif which in ("bitwise", "windowed"):
        # `curve_scalar_mul` [REF src/ecdsa/gadgets/curve.rs:211-251, test_curve_mul :335-363] (BaseSumGate<2> bit splits, 2^18 rows) and
        # `curve_scalar_mul_windowed` [REF src/ecdsa/gadgets/curve_windowed_mul.rs:133-177, test :222-254]: not on the ECDSA path
        p, n = E.pt_mul(rnd(), E.G), rnd()
        mul = eb.curve_scalar_mul if which == "bitwise" else eb.curve_scalar_mul_windowed
        got = mul(eb.constant_affine_point(p), eb.virtual_nonnative(n))
        want = E.pt_mul(n, p)
    else:
        p, q, n, m = E.pt_mul(rnd(), E.G), E.pt_mul(rnd(), E.G), rnd(), rnd()
        got = eb.curve_msm(eb.constant_affine_point(p), eb.constant_affine_point(q), eb.virtual_nonnative(n), eb.virtual_nonnative(m))
        want = E.pt_add(E.pt_mul(n, p), E.pt_mul(m, q))
    assert eb.point_value(got) == want
    eb.curve_assert_valid(got)
    expected = eb.constant_affine_point(want)
    eb.connect_biguint(got[0], expected[0]); eb.connect_biguint(got[1], expected[1])
    c = eb.build()
    gc = glp.Circuit(ctx, c)
    proof = gc.prove()
    assert gc.verify(proof)
    oc = oracle.OracleCircuit(c, cs_cap=gc.constants_sigmas_cap())
    assert oc.verify(proof) == 0
    bad = proof.copy()
    bad[7] ^= np.uint64(1)
    assert not gc.verify(bad)
    gc.free()
