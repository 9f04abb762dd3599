"""The reference's secp256k1 ECDSA-verification circuit (BASELINE config 3, the headline), gadget for gadget, in Python.

Restates, on `gadgets.GadgetBuilder`, the gadget stack of src/ecdsa/gadgets:
  biguint.rs        BigUintTarget arithmetic on u32 limbs                               [REF src/ecdsa/gadgets/biguint.rs:86-277]
  nonnative.rs      arithmetic modulo a foreign prime with generator-supplied results     [REF src/ecdsa/gadgets/nonnative.rs:128-460]
  split_nonnative   2-bit / 4-bit limb decompositions through BaseSumGate<4>              [REF src/ecdsa/gadgets/split_nonnative.rs:39-97]
  curve.rs          affine point validity / negation / doubling / addition               [REF src/ecdsa/gadgets/curve.rs:78-243]
  curve_windowed_mul.rs  `random_access_curve_points`                                    [REF src/ecdsa/gadgets/curve_windowed_mul.rs:73-118]
  curve_fixed_base.rs    4-bit windowed fixed-base multiplication                        [REF src/ecdsa/gadgets/curve_fixed_base.rs:22-77]
  curve_msm.rs      2-bit windowed double-scalar multiplication                          [REF src/ecdsa/gadgets/curve_msm.rs:21-83]
  glv.rs            GLV decomposition + endomorphism                                     [REF src/ecdsa/gadgets/glv.rs:38-100]
  ecdsa.rs          `verify_message_circuit` / `batch_verify_message_circuit`             [REF src/ecdsa/gadgets/ecdsa.rs:136-191]
and the native side that feeds the witness: secp256k1 arithmetic, signing [REF src/ecdsa/curve/ecdsa.rs], the GLV decomposition
[REF src/ecdsa/curve/glv.rs:11-77].  The plonky2_u32 helpers the stack calls (`add_many_u32`, `add_u32s_with_carry`,
`range_check_u32_circuit`, `list_le_u32_circuit`) and plonky2's `split_le_base`, `random_access` are absent from the reference tree and
are recalled.  As in gadgets.py: values are computed while building, gate placement is this builder's, and the statement of the
circuit -- every signature of the batch verifies -- is what the tests check (a signature that does not verify cannot be wired).

Nothing here touches the GPU or the oracle.
"""
import numpy as np

from . import synth
from .gadgets import GadgetBuilder, M32, P
from .synth import GATE_BASE_SUM, GATE_COMPARISON, GATE_RANDOM_ACCESS, GATE_U32_ADD_MANY, GATE_U32_RANGE_CHECK

# ---- secp256k1, native
FP = 2 ** 256 - 2 ** 32 - 977
FN = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A6855419 * 2 ** 64 + 0x9C47D08FFB10D4B8
G = (GX, GY)


def _limbs64(ws):
    return sum(int(w) << (64 * i) for i, w in enumerate(ws))


GLV_BETA = _limbs64([13923278643952681454, 11308619431505398165, 7954561588662645993, 8856726876819556112])   # [REF src/ecdsa/curve/glv.rs:11-16]
GLV_S = _limbs64([16069571880186789234, 1310022930574435960, 11900229862571533402, 6008836872998760672])     # [REF :18-23]
_A1 = _limbs64([16747920425669159701, 3496713202691238861])
_MINUS_B1 = _limbs64([8022177200260244675, 16448129721693014056])
_A2 = _limbs64([6323353552219852760, 1498098850674701302, 1])
_B2 = _A1


def pt_add(p, q):
    """affine addition; None is the point at infinity"""
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % FP == 0:
            return None
        lam = 3 * p[0] * p[0] * pow(2 * p[1], -1, FP) % FP
    else:
        lam = (q[1] - p[1]) * pow(q[0] - p[0], -1, FP) % FP
    x = (lam * lam - p[0] - q[0]) % FP
    return (x, (lam * (p[0] - x) - p[1]) % FP)


def pt_neg(p): return None if p is None else (p[0], (-p[1]) % FP)


def pt_mul(k, p):
    r = None
    while k:
        if k & 1:
            r = pt_add(r, p)
        p = pt_add(p, p)
        k >>= 1
    return r


def sign_message(msg, sk, k):
    """`sign_message` [REF src/ecdsa/curve/ecdsa.rs]: (r, s) with r = (k G).x mod n, s = (msg + r sk) / k"""
    r = pt_mul(k, G)[0] % FN
    return r, (msg + r * sk) * pow(k, -1, FN) % FN


def verify_message(msg, sig, pk):
    r, s = sig
    c = pow(s, -1, FN)
    pt = pt_add(pt_mul(msg * c % FN, G), pt_mul(r * c % FN, pk))
    return pt is not None and pt[0] % FN == r


def decompose_secp256k1_scalar(k):
    """[REF src/ecdsa/curve/glv.rs:38-77] -> (|k1|, |k2|, k1 < 0, k2 < 0) with k1 + GLV_S k2 = k (mod n)"""
    rnd = lambda a: (2 * a + FN) // (2 * FN)                 # Ratio::round for non-negative values
    c1, c2 = rnd(_B2 * k) % FN, rnd(_MINUS_B1 * k) % FN
    k1 = (k - c1 * _A1 - c2 * _A2) % FN
    k2 = (c1 * _MINUS_B1 - c2 * _B2) % FN
    assert (k1 + GLV_S * k2) % FN == k
    n1, n2 = k1 > FN // 2, k2 > FN // 2
    return (FN - k1 if n1 else k1), (FN - k2 if n2 else k2), n1, n2


def keccak256(data):
    """plain Keccak-256 (only for the `rando` constants the gadgets derive from `KeccakHash::<32>::hash_no_pad(&[F::ZERO])`)"""
    RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
          0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
          0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
          0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
    ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
    M = (1 << 64) - 1
    rol = lambda x, n: ((x << n) | (x >> (64 - n))) & M if n else x
    msg = bytearray(data) + b"\x01"
    msg += b"\0" * (-len(msg) % 136)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), 136):
        for i in range(17):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        for rc in RC:
            c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
            d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
            a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
            b = [[0] * 5 for _ in range(5)]
            for x in range(5):
                for y in range(5):
                    b[y][(2 * x + 3 * y) % 5] = rol(a[x][y], ROT[x][y])
            a = [[b[x][y] ^ (~b[(x + 1) % 5][y] & M & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
            a[0][0] ^= rc
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


_RANDO = None


def rando_point():
    """`(CurveScalar(hash_0_scalar) * G).to_affine()` with hash_0 = KeccakHash<32>::hash_no_pad(&[0]) [REF src/ecdsa/gadgets/curve_fixed_base.rs:38-43]"""
    global _RANDO
    if _RANDO is None:
        _RANDO = pt_mul(int.from_bytes(keccak256(b"\0" * 8), "little") % FN, G)
    return _RANDO


def _u32_digits(v):
    """BigUint::to_u32_digits: little-endian, no leading zero limbs, zero -> []"""
    out = []
    while v:
        out.append(v & M32)
        v >>= 32
    return out


class EcdsaBuilder(GadgetBuilder):
    """GadgetBuilder + the gates the ECDSA stack adds: U32AddManyGate, U32RangeCheckGate, ComparisonGate, BaseSumGate<4>,
    RandomAccessGate(4).  Default configuration: `standard_ecc_config` (136 wires) [REF src/ecdsa/gadgets/ecdsa.rs:476-483]."""

    def __init__(self, config=None):
        super().__init__(config or synth.Config.standard_ecc_config())
        self._addmany, self._range, self._cmp, self._ra = [], [], [], []

    def val_of(self, limbs): return sum(self.val[t] << (32 * i) for i, t in enumerate(limbs))

    # ---- plonky2_u32 (recalled)
    def _add_many_gate(self, addends, carry):
        na = len(addends)
        nops = min(self.cfg.num_routed_wires // (na + 3), self.cfg.num_wires // (na + 3 + 18))
        row, i = self._slot(("addmany", na), GATE_U32_ADD_MANY, na, nops, nops)
        o = (na + 3) * i
        for k, t in enumerate(addends):
            self._place(t, row, o + k)
        self._place(carry, row, o + na)
        tot = sum(self.val[t] for t in addends) + self.val[carry]
        assert tot >> 32 < 16
        self._addmany.append((row, (na + 3) * nops + 18 * i, tot))
        return self._wire(row, o + na + 1, tot & M32), self._wire(row, o + na + 2, tot >> 32)

    def add_many_u32(self, to_add):
        if len(to_add) == 0: return self.zero_u32(), self.zero_u32()
        if len(to_add) == 1: return to_add[0], self.zero_u32()
        if len(to_add) == 2: return self.add_u32(to_add[0], to_add[1])
        return self._add_many_gate(to_add, self.zero_u32())

    def add_u32s_with_carry(self, to_add, carry):
        if len(to_add) == 0: return carry, self.zero_u32()       # a column of a product with a zero-limb operand (the curve's A = 0): no gate
        if len(to_add) == 1: return self.add_u32(to_add[0], carry)
        return self._add_many_gate(to_add, carry)

    def range_check_u32(self, vals):
        """`range_check_u32_circuit`: one U32RangeCheckGate row with len(vals) inputs"""
        n = len(vals)
        if n == 0:
            return
        self.rows.append([GATE_U32_RANGE_CHECK, n, 0, (0, 0)])
        row = len(self.rows) - 1
        self.stats[GATE_U32_RANGE_CHECK] = self.stats.get(GATE_U32_RANGE_CHECK, 0) + 1
        for i, t in enumerate(vals):
            self._place(t, row, i)
            assert self.val[t] <= M32, "range check of a value that is not a u32: the witness is not satisfiable"
            self._range.append((row, n + 16 * i, self.val[t]))

    def _le_gate(self, a, b):
        """one ComparisonGate(32 bits, 16 chunks) row: result bit = (a <= b)"""
        self.rows.append([GATE_COMPARISON, 32, 16, (0, 0)])
        row = len(self.rows) - 1
        self.stats[GATE_COMPARISON] = self.stats.get(GATE_COMPARISON, 0) + 1
        self._place(a, row, 0)
        self._place(b, row, 1)
        assert self.val[a] <= M32 and self.val[b] <= M32
        self._cmp.append((row, self.val[a], self.val[b]))
        return self._wire(row, 2, 1 if self.val[a] <= self.val[b] else 0)

    def list_le_u32(self, a, b):
        """`list_le_circuit(a, b, 32)` (multiple_comparison.rs): a <= b as little-endian limb vectors"""
        one = self.one()
        result = one
        for x, y in zip(a, b):
            a_le_b, b_le_a = self._le_gate(x, y), self._le_gate(y, x)
            equal = self.mul(a_le_b, b_le_a)
            less = self.sub(one, b_le_a)
            result = self.mul_add(equal, result, less)
        return result

    def split_le_base4(self, x, num_limbs):
        """`split_le_base::<4>(x, num_limbs)`: one BaseSumGate<4> row"""
        self.rows.append([GATE_BASE_SUM, num_limbs, 4, (0, 0)])
        row = len(self.rows) - 1
        self.stats[GATE_BASE_SUM] = self.stats.get(GATE_BASE_SUM, 0) + 1
        self._place(x, row, 0)
        v = self.val[x]
        assert v < 4 ** num_limbs
        return [self._wire(row, 1 + i, (v >> (2 * i)) & 3) for i in range(num_limbs)]

    def random_access(self, index, v):
        """gadgets/random_access.rs: RandomAccessGate(bits = log2 len(v)); the claimed element is v[index]"""
        vs = len(v)
        bits = vs.bit_length() - 1
        assert vs == 1 << bits and bits == 4
        nw, nr = self.cfg.num_wires, self.cfg.num_routed_wires
        copies = min(nr // (2 + vs), nw // (2 + vs + bits))
        nextra = min(nr - copies * (2 + vs), self.cfg.num_constants)
        row, c = self._slot(("ra", bits), GATE_RANDOM_ACCESS, bits, copies | (nextra << 16), copies)
        o = (2 + vs) * c
        self._place(index, row, o)
        for k, t in enumerate(v):
            self._place(t, row, o + 2 + k)
        idx = self.val[index]
        assert idx < vs
        self._ra.append((row, (2 + vs) * copies + nextra + bits * c, idx))
        return self._wire(row, o + 1, self.val[v[idx]])

    # ---- biguint.rs
    def constant_biguint(self, value): return [self.constant_u32(l) for l in _u32_digits(value)]
    def virtual_biguint(self, value, num_limbs):
        assert 0 <= value < 1 << (32 * num_limbs)
        return [self.target((value >> (32 * i)) & M32) for i in range(num_limbs)]

    def connect_biguint(self, lhs, rhs):
        m = min(len(lhs), len(rhs))
        for i in range(m):
            self.connect(lhs[i], rhs[i])
        for t in lhs[m:] + rhs[m:]:
            self.assert_zero(t)

    def cmp_biguint(self, a, b):
        n = max(len(a), len(b))
        pad = lambda x: x + [self.zero_u32()] * (n - len(x))
        return self.list_le_u32(pad(a), pad(b))

    def add_biguint(self, a, b):
        out, carry = [], self.zero_u32()
        for i in range(max(len(a), len(b))):
            x = a[i] if i < len(a) else self.zero_u32()
            y = b[i] if i < len(b) else self.zero_u32()
            limb, carry = self.add_many_u32([carry, x, y])
            out.append(limb)
        return out + [carry]

    def sub_biguint(self, a, b):
        n = max(len(a), len(b))
        a, b = a + [self.zero_u32()] * (n - len(a)), b + [self.zero_u32()] * (n - len(b))
        out, borrow = [], self.zero_u32()
        for i in range(n):
            r, borrow = self.sub_u32(a[i], b[i], borrow)
            out.append(r)
        return out

    def mul_biguint(self, a, b):
        to_add = [[] for _ in range(len(a) + len(b))]
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                lo, hi = self.mul_u32(x, y)
                to_add[i + j].append(lo)
                to_add[i + j + 1].append(hi)
        out, carry = [], self.zero_u32()
        for summands in to_add:
            limb, carry = self.add_u32s_with_carry(summands, carry)
            out.append(limb)
        return out + [carry]

    def mul_biguint_by_bool(self, a, b): return [self.mul(l, b) for l in a]

    def div_rem_biguint(self, a, b):
        """[REF src/ecdsa/gadgets/biguint.rs:235-264]: quotient and remainder supplied by a generator, a = div b + rem and rem <= b enforced"""
        va, vb = self.val_of(a), self.val_of(b)
        div = self.virtual_biguint(va // vb, 0 if len(b) > len(a) + 1 else len(a) - len(b) + 1)
        rem = self.virtual_biguint(va % vb, len(b))
        self.connect_biguint(a, self.add_biguint(self.mul_biguint(div, b), rem))
        self.connect(self.cmp_biguint(rem, b), self.one())
        return div, rem

    def reduce(self, x, m):
        """`reduce` [REF src/ecdsa/gadgets/nonnative.rs:393-402]: x mod m through div_rem_biguint"""
        return self.div_rem_biguint(x, self.constant_biguint(m))[1]

    # ---- nonnative.rs (m = the foreign modulus; values are limb lists)
    def virtual_nonnative(self, value): return self.virtual_biguint(value, 8)

    def add_nonnative(self, a, b, m):
        va, vb = self.val_of(a), self.val_of(b)
        s = self.virtual_nonnative((va + vb) % m)
        overflow = self.target(1 if va + vb >= m else 0)
        sum_expected = self.add_biguint(a, b)
        modulus = self.constant_biguint(m)
        sum_actual = self.add_biguint(s, self.mul_biguint_by_bool(modulus, overflow))
        self.connect_biguint(sum_expected, sum_actual)
        self.connect(self.cmp_biguint(s, modulus), self.one())
        return s

    def sub_nonnative(self, a, b, m):
        va, vb = self.val_of(a), self.val_of(b)
        diff = self.virtual_nonnative((va - vb) % m)
        overflow = self.target(1 if va < vb else 0)
        self.range_check_u32(diff)
        self.assert_bool(overflow)
        diff_plus_b = self.add_biguint(diff, b)
        reduced = self.sub_biguint(diff_plus_b, self.mul_biguint_by_bool(self.constant_biguint(m), overflow))
        self.connect_biguint(a, reduced)
        return diff

    def mul_nonnative(self, a, b, m):
        va, vb = self.val_of(a), self.val_of(b)
        prod = self.virtual_nonnative(va * vb % m)
        modulus = self.constant_biguint(m)
        overflow = self.virtual_biguint(va * vb // m, len(a) + len(b) - len(modulus))
        self.range_check_u32(prod)
        self.range_check_u32(overflow)
        prod_expected = self.mul_biguint(a, b)
        prod_actual = self.add_biguint(prod, self.mul_biguint(modulus, overflow))
        self.connect_biguint(prod_expected, prod_actual)
        return prod

    def neg_nonnative(self, x, m): return self.sub_nonnative(self.constant_biguint(0), x, m)

    def inv_nonnative(self, x, m):
        vx = self.val_of(x)
        vi = pow(vx, -1, m)
        inv = self.virtual_biguint(vi, len(x))
        div = self.virtual_biguint((vx * vi - 1) // m, len(x))
        product = self.mul_biguint(x, inv)
        expected = self.add_biguint(self.mul_biguint(self.constant_biguint(m), div), self.constant_biguint(1))
        self.connect_biguint(product, expected)
        return inv

    def nonnative_conditional_neg(self, x, b, m):
        not_b = self.not_(b)
        neg = self.neg_nonnative(x, m)
        return self.add_nonnative(self.mul_biguint_by_bool(neg, b), self.mul_biguint_by_bool(x, not_b), m)

    # ---- split_nonnative.rs
    def split_u32_to_4_bit_limbs(self, limb):
        two = self.split_le_base4(limb, 16)
        four = self.constant(4)
        return [self.mul_add(two[2 * i + 1], four, two[2 * i]) for i in range(8)]

    def split_nonnative_to_4_bit_limbs(self, x): return [t for l in x for t in self.split_u32_to_4_bit_limbs(l)]
    def split_nonnative_to_2_bit_limbs(self, x): return [t for l in x for t in self.split_le_base4(l, 16)]

    # ---- curve.rs (points are (x limbs, y limbs); incomplete arithmetic, as in the reference)
    def constant_affine_point(self, p): return (self.constant_biguint(p[0]), self.constant_biguint(p[1]))
    def virtual_affine_point(self, p): return (self.virtual_nonnative(p[0]), self.virtual_nonnative(p[1]))
    def point_value(self, p): return (self.val_of(p[0]), self.val_of(p[1]))

    def curve_assert_valid(self, p):
        x, y = p
        b = self.constant_biguint(7)
        a = self.constant_biguint(0)
        y2 = self.mul_nonnative(y, y, FP)
        x2 = self.mul_nonnative(x, x, FP)
        x3 = self.mul_nonnative(x2, x, FP)
        ax = self.mul_nonnative(a, x, FP)
        rhs = self.add_nonnative(x3, self.add_nonnative(ax, b, FP), FP)
        self.connect_biguint(y2, rhs)

    def curve_neg(self, p): return (p[0], self.neg_nonnative(p[1], FP))
    def curve_conditional_neg(self, p, b): return (p[0], self.nonnative_conditional_neg(p[1], b, FP))

    def curve_double(self, p):
        x, y = p
        inv_2y = self.inv_nonnative(self.add_nonnative(y, y, FP), FP)
        x2 = self.mul_nonnative(x, x, FP)
        x2_3 = self.add_nonnative(self.add_nonnative(x2, x2, FP), x2, FP)
        lam = self.mul_nonnative(self.add_nonnative(x2_3, self.constant_biguint(0), FP), inv_2y, FP)
        lam2 = self.mul_nonnative(lam, lam, FP)
        x3 = self.sub_nonnative(lam2, self.add_nonnative(x, x, FP), FP)
        y3 = self.sub_nonnative(self.mul_nonnative(lam, self.sub_nonnative(x, x3, FP), FP), y, FP)
        return (x3, y3)

    def curve_repeated_double(self, p, n):
        for _ in range(n):
            p = self.curve_double(p)
        return p

    def curve_add(self, p1, p2):
        (x1, y1), (x2, y2) = p1, p2
        u = self.sub_nonnative(y2, y1, FP)
        v = self.sub_nonnative(x2, x1, FP)
        s = self.mul_nonnative(u, self.inv_nonnative(v, FP), FP)
        s2 = self.mul_nonnative(s, s, FP)
        x3 = self.sub_nonnative(s2, self.add_nonnative(x2, x1, FP), FP)
        y3 = self.sub_nonnative(self.mul_nonnative(s, self.sub_nonnative(x1, x3, FP), FP), y1, FP)
        return (x3, y3)

    def curve_conditional_add(self, p1, p2, b):
        not_b = self.not_(b)
        s = self.curve_add(p1, p2)
        x = self.add_nonnative(self.mul_biguint_by_bool(s[0], b), self.mul_biguint_by_bool(p1[0], not_b), FP)
        y = self.add_nonnative(self.mul_biguint_by_bool(s[1], b), self.mul_biguint_by_bool(p1[1], not_b), FP)
        return (x, y)

    def random_access_curve_points(self, index, pts):
        zero = self.zero_u32()
        limb = lambda big, i: big[i] if i < len(big) else zero
        xs = [self.random_access(index, [limb(p[0], i) for p in pts]) for i in range(8)]
        ys = [self.random_access(index, [limb(p[1], i) for p in pts]) for i in range(8)]
        return (xs, ys)

    # ---- the two general scalar multiplications the reference also defines (not on the ECDSA path, which uses fixed-base + GLV/MSM)
    def split_nonnative_to_bits(self, x):
        """[REF src/ecdsa/gadgets/nonnative.rs:420-436]: `split_le_base::<2>(limb, 32)` per limb"""
        out = []
        for limb in x:
            self.rows.append([GATE_BASE_SUM, 32, 2, (0, 0)])
            row = len(self.rows) - 1
            self.stats[GATE_BASE_SUM] = self.stats.get(GATE_BASE_SUM, 0) + 1
            self._place(limb, row, 0)
            v = self.val[limb]
            out += [self._wire(row, 1 + i, (v >> i) & 1) for i in range(32)]
        return out

    def curve_scalar_mul(self, p, n, rando=None):
        """bit-by-bit double-and-add [REF src/ecdsa/gadgets/curve.rs:211-251]; `rando`: the blinding point (random in the reference)"""
        bits = self.split_nonnative_to_bits(n)
        rando = rando or rando_point()
        randot = self.constant_affine_point(rando)
        result = self.virtual_affine_point(rando)
        self.connect_biguint(randot[0], result[0]); self.connect_biguint(randot[1], result[1])
        two_i_p = self.virtual_affine_point(self.point_value(p))
        self.connect_biguint(p[0], two_i_p[0]); self.connect_biguint(p[1], two_i_p[1])
        for bit in bits:
            not_bit = self.not_(bit)
            s = self.curve_add(result, two_i_p)
            x = self.add_nonnative(self.mul_biguint_by_bool(s[0], bit), self.mul_biguint_by_bool(result[0], not_bit), FP)
            y = self.add_nonnative(self.mul_biguint_by_bool(s[1], bit), self.mul_biguint_by_bool(result[1], not_bit), FP)
            result = (x, y)
            two_i_p = self.curve_double(two_i_p)
        return self.curve_add(result, self.curve_neg(randot))

    def precompute_window(self, p, g=None):
        """[REF src/ecdsa/gadgets/curve_windowed_mul.rs:50-71]: multiples[i] = i p (+ a blinding point g, removed again), i < 16"""
        g = g or pt_mul(0x1234567, G)
        neg = self.constant_affine_point(pt_neg(g))
        multiples = [self.constant_affine_point(g)]
        for i in range(1, 16):
            multiples.append(self.curve_add(p, multiples[i - 1]))
        for i in range(1, 16):
            multiples[i] = self.curve_add(neg, multiples[i])
        return multiples

    def curve_scalar_mul_windowed(self, p, n):
        """4-bit windowed multiplication [REF src/ecdsa/gadgets/curve_windowed_mul.rs:133-177]; the starting point comes from
        KeccakHash<25>::hash_no_pad(&[0]) there"""
        start = pt_mul(int.from_bytes(keccak256(b"\0" * 8)[:25], "little") % FN, G)
        start_multiplied = start
        for _ in range(256):
            start_multiplied = pt_add(start_multiplied, start_multiplied)
        result = self.constant_affine_point(start)
        pre = self.precompute_window(p)
        zero = self.zero()
        windows = self.split_nonnative_to_4_bit_limbs(n)
        for w in reversed(windows):
            result = self.curve_repeated_double(result, 4)
            to_add = self.random_access_curve_points(w, pre)
            should_add = self.not_(self.is_equal(w, zero))
            result = self.curve_conditional_add(result, to_add, should_add)
        return self.curve_add(result, self.curve_neg(self.constant_affine_point(start_multiplied)))

    # ---- curve_fixed_base.rs
    def fixed_base_curve_mul(self, base, scalar):
        limbs = self.split_nonnative_to_4_bit_limbs(scalar)
        rando = rando_point()
        zero = self.zero()
        result = self.constant_affine_point(rando)
        point = base
        for limb in limbs:
            muls, acc = [], None
            for _ in range(16):
                muls.append(acc)
                acc = pt_add(point, acc)
            muls_t = [self.constant_affine_point(q) for q in muls[1:]]
            muls_t.insert(0, muls_t[0])
            should_add = self.not_(self.is_equal(limb, zero))
            r = self.random_access_curve_points(limb, muls_t)
            result = self.curve_conditional_add(result, r, should_add)
            for _ in range(4):
                point = pt_add(point, point)
        return self.curve_add(result, self.constant_affine_point(pt_neg(rando)))

    # ---- curve_msm.rs
    def curve_msm(self, p, q, n, m):
        limbs_n, limbs_m = self.split_nonnative_to_2_bit_limbs(n), self.split_nonnative_to_2_bit_limbs(m)
        assert len(limbs_n) == len(limbs_m)
        rando = rando_point()
        rando_t, neg_rando = self.constant_affine_point(rando), self.constant_affine_point(pt_neg(rando))
        pre = [p] * 16
        cur_p, cur_q = rando_t, rando_t
        for i in range(4):
            pre[i], pre[4 * i] = cur_p, cur_q
            cur_p, cur_q = self.curve_add(cur_p, p), self.curve_add(cur_q, q)
        for i in range(1, 4):
            pre[i] = self.curve_add(pre[i], neg_rando)
            pre[4 * i] = self.curve_add(pre[4 * i], neg_rando)
        for i in range(1, 4):
            for j in range(1, 4):
                pre[i + 4 * j] = self.curve_add(pre[i], pre[4 * j])
        four, zero = self.constant(4), self.zero()
        result = rando_t
        for ln, lm in reversed(list(zip(limbs_n, limbs_m))):
            result = self.curve_repeated_double(result, 2)
            index = self.mul_add(four, lm, ln)
            r = self.random_access_curve_points(index, pre)
            should_add = self.not_(self.is_equal(index, zero))
            result = self.curve_conditional_add(result, r, should_add)
        start = rando
        for _ in range(2 * len(limbs_n)):
            start = pt_add(start, start)
        return self.curve_add(result, self.constant_affine_point(pt_neg(start)))

    # ---- glv.rs
    def glv_mul(self, p, k):
        vk = self.val_of(k)
        k1v, k2v, n1, n2 = decompose_secp256k1_scalar(vk)
        k1, k2 = self.virtual_biguint(k1v, 4), self.virtual_biguint(k2v, 4)
        k1_neg, k2_neg = self.target(int(n1)), self.target(int(n2))
        k1_raw = self.nonnative_conditional_neg(k1, k1_neg, FN)
        k2_raw = self.nonnative_conditional_neg(k2, k2_neg, FN)
        should_be_k = self.add_nonnative(self.mul_nonnative(self.constant_biguint(GLV_S), k2_raw, FN), k1_raw, FN)
        self.connect_biguint(should_be_k, k)
        sp = (self.mul_nonnative(self.constant_biguint(GLV_BETA), p[0], FP), p[1])
        return self.curve_msm(self.curve_conditional_neg(p, k1_neg), self.curve_conditional_neg(sp, k2_neg), k1, k2)

    # ---- ecdsa.rs
    def verify_message(self, msg, sig, pk):
        """`verify_message_circuit` [REF src/ecdsa/gadgets/ecdsa.rs:136-159]; msg, r, s: scalar limb lists; pk: point"""
        r, s = sig
        self.curve_assert_valid(pk)
        c = self.inv_nonnative(s, FN)
        u1 = self.mul_nonnative(msg, c, FN)
        u2 = self.mul_nonnative(r, c, FN)
        point = self.curve_add(self.fixed_base_curve_mul(G, u1), self.glv_mul(pk, u2))
        self.connect_biguint(r, point[0])

    # ---- build: the advice columns of the added gates, vectorised
    def build(self, min_log_n=0):
        c = super().build(min_log_n)
        w = c.wires
        U = np.uint64
        if self._addmany:
            rr = np.array([x[0] for x in self._addmany]); cc = np.array([x[1] for x in self._addmany])
            tot = np.array([x[2] for x in self._addmany], dtype=np.uint64)
            for k in range(16):
                w[cc + k, rr] = (tot >> U(2 * k)) & U(3)
            for k in range(2):
                w[cc + 16 + k, rr] = (tot >> U(32 + 2 * k)) & U(3)
        if self._range:
            rr = np.array([x[0] for x in self._range]); cc = np.array([x[1] for x in self._range])
            v = np.array([x[2] for x in self._range], dtype=np.uint64)
            for k in range(16):
                w[cc + k, rr] = (v >> U(2 * k)) & U(3)
        if self._ra:
            rr = np.array([x[0] for x in self._ra]); cc = np.array([x[1] for x in self._ra])
            idx = np.array([x[2] for x in self._ra], dtype=np.uint64)
            for k in range(4):
                w[cc + k, rr] = (idx >> U(k)) & U(1)
        if self._cmp:
            # ComparisonGate(32, 16) [plonky2_u32 gates/comparison.rs, layout as in synth.fill_ecdsa_gate_rows]: wires 0 first, 1 second,
            # 2 result, 3 most significant diff, then 16 chunks of each input, equality dummies, chunks-equal flags, intermediate
            # values, and the 3 bits of 2^2 + most_significant_diff
            rr = np.array([x[0] for x in self._cmp])
            a = np.array([x[1] for x in self._cmp], dtype=np.int64); b = np.array([x[2] for x in self._cmp], dtype=np.int64)
            inv = {d: pow(d % P, P - 2, P) for d in (-3, -2, -1, 1, 2, 3)}
            inv_tab = np.array([inv[-3], inv[-2], inv[-1], 1, inv[1], inv[2], inv[3]], dtype=np.uint64)       # diff 0 -> dummy 1
            msd = np.zeros(len(rr), dtype=np.int64)
            for i in range(16):
                ca, cb = (a >> (2 * i)) & 3, (b >> (2 * i)) & 3
                d = cb - ca
                eq = (d == 0)
                w[4 + i, rr] = ca.astype(np.uint64)
                w[4 + 16 + i, rr] = cb.astype(np.uint64)
                w[4 + 32 + i, rr] = inv_tab[d + 3]
                w[4 + 48 + i, rr] = eq.astype(np.uint64)
                inter = np.where(eq, msd, 0)
                w[4 + 64 + i, rr] = _to_field(inter)
                msd = np.where(eq, inter, d)
            w[3, rr] = _to_field(msd)
            top = 4 + msd                                     # 2^chunk_bits + msd, in 1..7
            for k in range(3):
                w[4 + 80 + k, rr] = ((top >> k) & 1).astype(np.uint64)
            assert ((top >> 2) & 1 == (a <= b)).all()
        c.gate_rows = {}
        for g, p0, p1, _ in self.rows:
            name = synth._GATE_META[g][1].format(p0=p0, p1=p1 & 0xFFFF).split("(PhantomData")[0]
            c.gate_rows[name] = c.gate_rows.get(name, 0) + 1
        return c


def _to_field(x):
    """int64 array with small negative entries -> canonical field elements"""
    out = x.astype(np.uint64)
    neg = x < 0
    out[neg] = (np.uint64(P) - (-x[neg]).astype(np.uint64))
    return out


def ecdsa_circuit(signatures, config=None, min_log_n=0):
    """`test_batch_ecdsa_circuit_with_config` [REF src/ecdsa/gadgets/ecdsa.rs:214-353]: one `batch_verify_message_circuit` over
    `signatures` = [(msg, (r, s), pk_point), ...]; no public inputs.  A signature that does not verify raises ValueError (its witness
    would violate the final copy constraint)."""
    eb = EcdsaBuilder(config)
    for msg, (r, s), pk in signatures:
        msg_t, r_t, s_t = eb.virtual_nonnative(msg), eb.virtual_nonnative(r), eb.virtual_nonnative(s)
        pk_t = eb.virtual_affine_point(pk)
        eb.verify_message(msg_t, (r_t, s_t), pk_t)
    return eb.build(min_log_n)


def random_signatures(count, seed=0):
    """`gen_batch_ecdsa_data` [REF src/ecdsa/gadgets/ecdsa.rs:193-212] with a seeded generator"""
    rng = np.random.default_rng(seed)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (FN - 1) + 1
    out = []
    for _ in range(count):
        msg, sk, k = rnd(), rnd(), rnd()
        sig = sign_message(msg, sk, k)
        pk = pt_mul(sk, G)
        assert verify_message(msg, sig, pk)
        out.append((msg, sig, pk))
    return out
