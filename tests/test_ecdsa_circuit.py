"""The reference's secp256k1 ECDSA-verification circuit (BASELINE config 3, the headline) rebuilt gadget for gadget in
plonky2-lib_amd/gadgets_ecdsa.py.  CPU half: the native side (signing, GLV decomposition), the gadget layers on small circuits that
the checker proves and verifies -- the shape of the reference's own unit tests [REF src/ecdsa/gadgets/nonnative.rs:855-1021,
src/ecdsa/gadgets/curve.rs:273-485, src/ecdsa/gadgets/biguint.rs:395-541] -- and the structure of the whole one-signature circuit."""
import numpy as np
import pytest

from plonky2_lib_amd import gadgets_ecdsa as E
import plonky2_lib_amd.synth as synth


def test_native_side():
    assert E.pt_mul(E.FN, E.G) is None and (E.GY * E.GY - E.GX ** 3 - 7) % E.FP == 0
    rng = np.random.default_rng(5)
    for _ in range(20):
        k = int.from_bytes(rng.bytes(32), "little") % E.FN
        k1, k2, n1, n2 = E.decompose_secp256k1_scalar(k)
        assert k1 < 1 << 128 and k2 < 1 << 128                # |k1|, |k2| < sqrt(n): four u32 limbs [REF src/ecdsa/gadgets/glv.rs:52-53]
        assert ((-k1 if n1 else k1) + E.GLV_S * (-k2 if n2 else k2)) % E.FN == k
    # the endomorphism: GLV_S * (x, y) = (GLV_BETA * x, y)
    p = E.pt_mul(12345, E.G)
    assert E.pt_mul(E.GLV_S, p) == (E.GLV_BETA * p[0] % E.FP, p[1])
    (msg, sig, pk), = E.random_signatures(1, seed=3)
    assert E.verify_message(msg, sig, pk) and not E.verify_message(msg + 1, sig, pk)
    assert E.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"     # [REF src/hash/keccak256.rs:198-200]


def _prove(oracle, c):
    oc = oracle.OracleCircuit(c)
    rc, proof = oc.prove()
    assert rc == 0 and oc.verify(proof) == 0
    return oc, proof


def test_biguint_and_nonnative_gadgets(oracle):
    """test_biguint_{add,sub,mul,cmp}, test_nonnative_{add,many_adds,sub,mul,neg,inv}: one circuit, results against Python integers"""
    rng = np.random.default_rng(7)
    rnd = lambda m: int.from_bytes(rng.bytes(40), "little") % m
    eb = E.EcdsaBuilder()
    x, y = rnd(E.FP), rnd(E.FP)
    xt, yt = eb.virtual_nonnative(x), eb.virtual_nonnative(y)
    assert eb.val_of(eb.add_biguint(xt, yt)) == x + y
    assert eb.val_of(eb.mul_biguint(xt, yt)) == x * y
    big, small = (xt, yt) if x >= y else (yt, xt)
    assert eb.val_of(eb.sub_biguint(big, small)) == abs(x - y)
    assert eb.val[eb.cmp_biguint(xt, yt)] == int(x <= y) and eb.val[eb.cmp_biguint(xt, xt)] == 1
    for m in (E.FP, E.FN):
        a, b = rnd(m), rnd(m)
        at, bt = eb.virtual_nonnative(a), eb.virtual_nonnative(b)
        assert eb.val_of(eb.add_nonnative(at, bt, m)) == (a + b) % m
        assert eb.val_of(eb.sub_nonnative(at, bt, m)) == (a - b) % m
        assert eb.val_of(eb.mul_nonnative(at, bt, m)) == a * b % m
        assert eb.val_of(eb.neg_nonnative(at, m)) == (-a) % m
        assert eb.val_of(eb.inv_nonnative(at, m)) == pow(a, -1, m)
        for flag in (0, 1):
            assert eb.val_of(eb.nonnative_conditional_neg(at, eb.target(flag), m)) == ((-a) % m if flag else a)
    div, rem = eb.div_rem_biguint(eb.mul_biguint(xt, yt), eb.constant_biguint(E.FN))          # test_biguint_div_rem
    assert eb.val_of(div) == x * y // E.FN and eb.val_of(rem) == x * y % E.FN
    assert eb.val_of(eb.reduce(eb.add_biguint(xt, yt), E.FP)) == (x + y) % E.FP
    four = eb.split_nonnative_to_4_bit_limbs(xt)
    assert sum(eb.val[t] << (4 * i) for i, t in enumerate(four)) == x and len(four) == 64
    c = eb.build()
    kinds = {g["type"] for g in c.gates}
    assert {synth.GATE_U32_ADD_MANY, synth.GATE_U32_RANGE_CHECK, synth.GATE_COMPARISON, synth.GATE_U32_SUBTRACTION, synth.GATE_U32_ARITHMETIC,
            synth.GATE_BASE_SUM} <= kinds
    oc, proof = _prove(oracle, c)
    # a limb of a product replaced in the witness (and nowhere else): unprovable
    gi = next(i for i, g in enumerate(c.gates) if g["type"] == synth.GATE_U32_ADD_MANY)
    row = int(np.nonzero(c.constants[c.gates[gi]["selector_index"]] == np.uint64(gi))[0][0])
    w = c.wires.copy()
    w[c.gates[gi]["p0"] + 1, row] ^= np.uint64(1)
    rc, p2 = oc.prove(wires=w)
    assert rc != 0 or oc.verify(p2) != 0
    # the generators' outputs are checked while wiring: a wrong remainder cannot be connected
    eb2 = E.EcdsaBuilder()
    with pytest.raises((ValueError, AssertionError)):
        t = eb2.virtual_nonnative(5)
        eb2.connect_biguint(eb2.mul_nonnative(t, t, E.FP), eb2.constant_biguint(26))


def test_curve_gadgets(oracle):
    """test_curve_point_is_valid / _neg / _double / _add / conditional add, random access into a table of points"""
    eb = E.EcdsaBuilder()
    p1, p2 = E.pt_mul(5, E.G), E.pt_mul(7, E.G)
    t1, t2 = eb.virtual_affine_point(p1), eb.virtual_affine_point(p2)
    eb.curve_assert_valid(t1)
    assert eb.point_value(eb.curve_add(t1, t2)) == E.pt_mul(12, E.G)
    assert eb.point_value(eb.curve_double(t1)) == E.pt_mul(10, E.G)
    assert eb.point_value(eb.curve_neg(t1)) == E.pt_neg(p1)
    assert eb.point_value(eb.curve_conditional_add(t1, t2, eb.target(0))) == p1
    assert eb.point_value(eb.curve_conditional_add(t1, t2, eb.target(1))) == E.pt_mul(12, E.G)
    table = [eb.constant_affine_point(E.pt_mul(i + 1, E.G)) for i in range(16)]
    assert eb.point_value(eb.random_access_curve_points(eb.target(9), table)) == E.pt_mul(10, E.G)
    c = eb.build()
    assert synth.GATE_RANDOM_ACCESS in {g["type"] for g in c.gates}
    _prove(oracle, c)
    # [REF src/ecdsa/gadgets/curve.rs:300-326 test_curve_point_is_not_valid]: a point off the curve cannot satisfy the circuit
    eb = E.EcdsaBuilder()
    with pytest.raises(ValueError):
        eb.curve_assert_valid(eb.virtual_affine_point((p1[0], (p1[1] + 1) % E.FP)))


def test_one_signature_circuit_structure():
    """`verify_message_circuit` on one signature: the size and gate list of the real circuit.  98 687 rows per signature + 7 714 shared
    ConstantGate rows (the fixed-base tables): 2^17 for one signature, 10 signatures fill 2^20 -- the headline trace."""
    (msg, sig, pk), = E.random_signatures(1, seed=1)
    c = E.ecdsa_circuit([(msg, sig, pk)])
    assert c.degree_bits == 17 and c.gadget_rows == 106402 and len(c.public_inputs) == 0
    ids = sorted(c.gate_rows)
    assert ids == sorted(["ArithmeticGate { num_ops: 20 }", "BaseSumGate { num_limbs: 16 } + Base: 4", "ComparisonGate { num_bits: 32, num_chunks: 16 }",
                          "ConstantGate { num_consts: 2 }", "PublicInputGate", "RandomAccessGate { bits: 4 }", "U32ArithmeticGate { num_ops: 3 }",
                          "U32RangeCheckGate { num_input_limbs: 8 }", "U32SubtractionGate { num_ops: 6 }"] +
                         ["U32AddManyGate { num_addends: %d, num_ops: %d }" % (a, o) for a, o in ((3, 5), (5, 5), (7, 4), (9, 4), (11, 4), (13, 4), (15, 3))])
    # the 11 gate TYPES are those SURVEY.md section 8 row Q lists for cfg 3; U32AddManyGate appears with 7 parameter sets
    assert {g["type"] for g in c.gates} == {synth.GATE_NOOP, synth.GATE_CONSTANT, synth.GATE_PUBLIC_INPUT, synth.GATE_ARITHMETIC, synth.GATE_BASE_SUM,
                                            synth.GATE_COMPARISON, synth.GATE_RANDOM_ACCESS, synth.GATE_U32_ARITHMETIC, synth.GATE_U32_ADD_MANY,
                                            synth.GATE_U32_RANGE_CHECK, synth.GATE_U32_SUBTRACTION}
    assert c.gate_rows["U32ArithmeticGate { num_ops: 3 }"] == 55616 and c.gate_rows["ComparisonGate { num_bits: 32, num_chunks: 16 }"] == 16912
    # a forged signature cannot be wired
    with pytest.raises(ValueError):
        E.ecdsa_circuit([(msg + 1, sig, pk)])
