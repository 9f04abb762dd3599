"""Target-level circuit builder and the reference's Keccak-256 circuit, in Python (host side; numpy + integers).

The reference builds its circuits with Rust gadget traits on plonky2's `CircuitBuilder`; neither runs here.  This module
restates the part needed to obtain a REAL circuit of BASELINE config 2 for the GPU prover instead of a gate-mix stand-in:

  * `GadgetBuilder`: targets with copy constraints (union-find -> sigma cycles), `find_slot`-style packing of operations into
    gate rows (ArithmeticGate keyed by its two constants), ConstantGate cells, public inputs hashed in-circuit by PoseidonGate
    rows into the PublicInputGate -- what `CircuitBuilder` + `build()` do, on top of `synth.Builder`;
  * `CircuitBuilderU32` subset (`mul_add_u32`, `mul_u32`, `add_u32`, `sub_u32`; plonky2_u32, recalled) and the reference's own
    `CircuitBuilderB32` [REF src/u32/interleaved_u32.rs:56-269]: not / shifts / rotations through U32ArithmeticGate, AND / XOR
    through the interleaved representation and the reference's three gates [REF src/u32/gates/*.rs];
  * `hash_keccak256` [REF src/hash/keccak256.rs:79-165] with `add_virtual_hash_input_target` / `public_hash_output`
    [REF src/hash/types.rs:175-199] and the witness setter `set_keccak256_input_target` [REF src/hash/keccak256.rs:22-37].

Values are computed while building (the builder is handed the input), so one call yields circuit + satisfying witness; the
structure (gates, constants, sigmas) does not depend on the input, which tests assert.  Gate placement order is this builder's,
not plonky2's (proof bytes are not comparable with the Rust prover's; the circuit's PUBLIC INPUTS are: the reference's tests pin
them to the Keccak-256 digests of [REF src/hash/keccak256.rs:196-212,256-277]).

Nothing here touches the GPU or the oracle.
"""
import numpy as np

from . import gl_numpy as gl
from . import synth
from .synth import (GATE_ARITHMETIC, GATE_CONSTANT, GATE_POSEIDON, GATE_PUBLIC_INPUT, GATE_U32_ARITHMETIC, GATE_U32_INTERLEAVE,
                    GATE_U32_SUBTRACTION, GATE_UNINTERLEAVE_B32, GATE_UNINTERLEAVE_U32)

P = gl.P
M32 = (1 << 32) - 1


def _interleave(x):
    """bit i of x -> bit 2i [REF src/u32/gates/interleave_u32.rs:22-45]"""
    x &= M32
    x = (x | (x << 16)) & 0x0000FFFF0000FFFF
    x = (x | (x << 8)) & 0x00FF00FF00FF00FF
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0F
    x = (x | (x << 2)) & 0x3333333333333333
    x = (x | (x << 1)) & 0x5555555555555555
    return x


def _deinterleave(v):
    """bits 0, 2, 4, ... of v packed"""
    v &= 0x5555555555555555
    v = (v | (v >> 1)) & 0x3333333333333333
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0F
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FF
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFF
    v = (v | (v >> 16)) & 0x00000000FFFFFFFF
    return v


def _batch_inverse(values):
    """inverses mod p of non-zero integers with one exponentiation (Montgomery's trick)"""
    n = len(values)
    if n == 0:
        return np.zeros(0, np.uint64)
    pre, acc = [0] * n, 1
    for i, v in enumerate(values):
        pre[i] = acc
        acc = acc * v % P
    inv = pow(acc, P - 2, P)
    out = np.empty(n, np.uint64)
    for i in range(n - 1, -1, -1):
        out[i] = inv * pre[i] % P
        inv = inv * values[i] % P
    return out


class GadgetBuilder:
    """Targets are integer ids; `self.val[t]` is the value the witness assigns."""

    def __init__(self, config=None):
        self.cfg = config or synth.Config.standard_recursion_config()
        nw, nr = self.cfg.num_wires, self.cfg.num_routed_wires
        from array import array
        self.val, self._parent = array("Q"), array("q")             # 8-byte arrays, not lists of Python ints: a 2^20-row circuit has 6.5 M targets
        self._cell_t, self._cell_r, self._cell_c = array("q"), array("q"), array("q")      # every routed cell: (target, row, column)
        self.rows = []                 # per row: [gate type, p0, p1, (const0, const1)]
        self._open = {}                # slot key -> [row, next free op]
        self._ops = {t: [] for t in (GATE_U32_ARITHMETIC, GATE_U32_SUBTRACTION, GATE_U32_INTERLEAVE, GATE_UNINTERLEAVE_U32,
                                     GATE_UNINTERLEAVE_B32)}       # advice is filled from these records at build()
        self._consts = {}
        self._poseidon_rows = []       # (row, 12 input values): S-box traces are filled at build()
        self.public_inputs = []
        self.n_arith = nr // 4
        self.n_u32a = min(nr // 6, nw // 38)
        self.n_sub = min(nr // 5, nw // 21)
        self.n_il = min(nw // 34, nr // 2)
        self.n_ul = min(nw // 67, nr // 3)
        self.stats = {}

    # ---- targets and copy constraints
    def target(self, value):
        self.val.append(int(value) % P)
        self._parent.append(len(self.val) - 1)
        return len(self.val) - 1

    def _find(self, t):
        while self._parent[t] != t:
            self._parent[t] = self._parent[self._parent[t]]
            t = self._parent[t]
        return t

    def connect(self, a, b):
        if self.val[a] != self.val[b]:
            raise ValueError("connect() of targets with different values: the witness would violate a copy constraint")
        ra, rb = self._find(a), self._find(b)
        if ra != rb:
            self._parent[rb] = ra

    def _place(self, t, row, col):
        assert col < self.cfg.num_routed_wires
        self._cell_t.append(t); self._cell_r.append(row); self._cell_c.append(col)

    def _wire(self, row, col, value):
        """a fresh target living in cell (row, col)"""
        t = self.target(value)
        self._place(t, row, col)
        return t

    # ---- rows and slots (CircuitBuilder::find_slot)
    def _slot(self, key, gate, p0, p1, num_ops, consts=(0, 0)):
        cur = self._open.get(key)
        if cur is None or cur[1] == num_ops:
            self.rows.append([gate, p0, p1, consts])
            cur = self._open[key] = [len(self.rows) - 1, 0]
        cur[1] += 1
        self.stats[gate] = self.stats.get(gate, 0) + 1
        return cur[0], cur[1] - 1

    def constant(self, c):
        c = int(c) % P
        t = self._consts.get(c)
        if t is None:
            nc = self.cfg.num_constants
            row, i = self._slot(("const",), GATE_CONSTANT, nc, 0, nc, consts=None)
            if self.rows[row][3] is None:
                self.rows[row][3] = [0] * nc
            self.rows[row][3][i] = c
            t = self._consts[c] = self._wire(row, i, c)
        return t

    def zero(self): return self.constant(0)
    def one(self): return self.constant(1)

    # ---- ArithmeticGate: c0 x y + c1 z   (gadgets/arithmetic.rs `arithmetic`, without its constant-folding special cases)
    def arithmetic(self, c0, c1, x, y, z):
        c0, c1 = int(c0) % P, int(c1) % P
        row, j = self._slot(("arith", c0, c1), GATE_ARITHMETIC, self.n_arith, 0, self.n_arith, consts=(c0, c1))
        for k, t in enumerate((x, y, z)):
            self._place(t, row, 4 * j + k)
        return self._wire(row, 4 * j + 3, (c0 * self.val[x] * self.val[y] + c1 * self.val[z]) % P)

    def add(self, x, y): return self.arithmetic(1, 1, x, self.one(), y)
    def sub(self, x, y): return self.arithmetic(1, P - 1, x, self.one(), y)
    def mul(self, x, y): return self.arithmetic(1, 0, x, y, self.zero())

    def add_many(self, terms):
        acc = terms[0]
        for t in terms[1:]:
            acc = self.add(acc, t)
        return acc

    def not_(self, b):
        """BoolTarget negation: 1 - b"""
        return self.sub(self.one(), b)

    # ---- plonky2_u32 gadgets/arithmetic_u32.rs (recalled): one U32ArithmeticGate / U32SubtractionGate operation each
    def constant_u32(self, c): return self.constant(c & M32)
    def zero_u32(self): return self.zero()
    def one_u32(self): return self.one()

    def mul_add_u32(self, x, y, z):
        row, i = self._slot(("u32a",), GATE_U32_ARITHMETIC, self.n_u32a, 0, self.n_u32a)
        for k, t in enumerate((x, y, z)):
            self._place(t, row, 6 * i + k)
        prod = self.val[x] * self.val[y] + self.val[z]
        assert prod < 1 << 64 and max(self.val[x], self.val[y], self.val[z]) <= M32, "u32 operands out of range"
        self._ops[GATE_U32_ARITHMETIC].append((row, i, prod))
        return self._wire(row, 6 * i + 3, prod & M32), self._wire(row, 6 * i + 4, prod >> 32)

    def mul_u32(self, x, y): return self.mul_add_u32(x, y, self.zero_u32())
    def add_u32(self, a, b): return self.mul_add_u32(a, self.one_u32(), b)

    def sub_u32(self, x, y, borrow):
        row, i = self._slot(("u32s",), GATE_U32_SUBTRACTION, self.n_sub, 0, self.n_sub)
        for k, t in enumerate((x, y, borrow)):
            self._place(t, row, 5 * i + k)
        d = self.val[x] - self.val[y] - self.val[borrow]
        bo = 1 if d < 0 else 0
        res = d + (bo << 32)
        self._ops[GATE_U32_SUBTRACTION].append((row, i, res))
        return self._wire(row, 5 * i + 3, res), self._wire(row, 5 * i + 4, bo)

    def connect_u32(self, a, b): self.connect(a, b)

    # ---- the reference's CircuitBuilderB32 [REF src/u32/interleaved_u32.rs]
    def not_u32(self, a):                                   # [REF :59-63]
        return self.sub_u32(self.constant_u32(0xFFFFFFFF), a, self.zero_u32())[0]

    def lsh_u32(self, a, n): return self.mul_u32(a, self.constant_u32(1 << n))[0]                       # [REF :66-69]
    def rsh_u32(self, a, n): return a if n == 0 else self.mul_u32(a, self.constant_u32(1 << (32 - n)))[1]    # [REF :72-78]

    def lrot_u32(self, a, n):                               # [REF :81-85]
        lo, hi = self.mul_u32(a, self.constant_u32(1 << n))
        return self.add_u32(lo, hi)[0]

    def rrot_u32(self, a, n): return self.lrot_u32(a, 32 - n)

    def interleave_u32(self, x):                            # [REF :93-100]
        row, op = self._slot(("il",), GATE_U32_INTERLEAVE, self.n_il, 0, self.n_il)
        self._place(x, row, 2 * op)
        assert self.val[x] <= M32
        self._ops[GATE_U32_INTERLEAVE].append((row, op, self.val[x]))
        return self._wire(row, 2 * op + 1, _interleave(self.val[x]))

    def _uninterleave(self, gate, key, x_dirty):
        row, op = self._slot((key,), gate, self.n_ul, 0, self.n_ul)
        self._place(x_dirty, row, 3 * op)
        v = self.val[x_dirty]                                # < 2^64 by construction; as a field element it is < p
        self._ops[gate].append((row, op, v))
        ev, od = _deinterleave(v >> 1), _deinterleave(v)     # big-endian bit pairs: "evens" = the high bit of each pair
        if gate == GATE_UNINTERLEAVE_B32:
            ev, od = _interleave(ev), _interleave(od)
        return self._wire(row, 3 * op + 1, ev), self._wire(row, 3 * op + 2, od)

    def uninterleave_to_u32(self, x): return self._uninterleave(GATE_UNINTERLEAVE_U32, "ul32", x)      # [REF :102-116]
    def uninterleave_to_b32(self, x): return self._uninterleave(GATE_UNINTERLEAVE_B32, "ulb32", x)     # [REF :118-132]

    def and_xor_b32(self, x, y): return self.uninterleave_to_b32(self.add(x, y))                        # [REF :183-186]
    def and_xor_u32(self, x, y): return self.and_xor_b32(self.interleave_u32(x), self.interleave_u32(y))   # [REF :188-192]
    def and_xor_b32_to_u32(self, x, y): return self.uninterleave_to_u32(self.add(x, y))                 # [REF :194-197]
    def and_xor_u32_to_u32(self, x, y): return self.and_xor_b32_to_u32(self.interleave_u32(x), self.interleave_u32(y))
    def and_u32(self, x, y): return self.and_xor_u32_to_u32(x, y)[0]
    def xor_u32(self, x, y): return self.and_xor_u32_to_u32(x, y)[1]

    def unsafe_xor_many_u32(self, x):                       # [REF :148-181], same case split
        n = len(x)
        if n == 0: return self.zero_u32()
        if n == 1: return x[0]
        if n == 2: return self.xor_u32(x[0], x[1])
        if n == 3: return self.xor_u32(self.xor_u32(x[0], x[1]), x[2])
        r = self.interleave_u32(x[0])
        for i in range((n - 3) // 2):
            a, b = self.interleave_u32(x[1 + 2 * i]), self.interleave_u32(x[2 + 2 * i])
            r = self.uninterleave_to_b32(self.add_many([r, a, b]))[1]
        if n % 2 == 0:
            r = self.and_xor_b32(r, self.interleave_u32(x[n - 3]))[1]
        a, b = self.interleave_u32(x[n - 2]), self.interleave_u32(x[n - 1])
        return self.uninterleave_to_u32(self.add_many([r, a, b]))[1]

    def lrot_u64(self, a, n):                               # [REF :213-220]
        lo, hi = (a[0], a[1]) if n < 32 else (a[1], a[0])
        p2 = self.constant_u32(1 << (n % 32))
        lo0, hi0 = self.mul_u32(lo, p2)
        lo1, hi1 = self.mul_add_u32(hi, p2, hi0)
        return [self.add_u32(lo0, hi1)[0], lo1]

    def unsafe_xor_many_u64(self, xs): return [self.unsafe_xor_many_u32([e[0] for e in xs]), self.unsafe_xor_many_u32([e[1] for e in xs])]
    def xor_u64(self, x, y): return [self.xor_u32(x[0], y[0]), self.xor_u32(x[1], y[1])]
    def and_u64(self, x, y): return [self.and_u32(x[0], y[0]), self.and_u32(x[1], y[1])]
    def not_u64(self, x): return [self.not_u32(x[0]), self.not_u32(x[1])]

    def conditional_u32(self, x, y, z):                     # z ? x : y   [REF :248-252]
        not_z = self.not_(z)
        maybe_x = self.mul_u32(x, z)[0]
        return self.mul_add_u32(y, not_z, maybe_x)[0]

    def conditional_u64(self, x, y, z): return [self.conditional_u32(x[0], y[0], z), self.conditional_u32(x[1], y[1], z)]

    def register_public_input(self, t): self.public_inputs.append(t)

    # ---- plonky2 gadgets the SMT circuits use (recalled; semantics, not gate placement, is what matters here)
    def mul_add(self, x, y, z): return self.arithmetic(1, 1, x, y, z)
    def mul_sub(self, x, y, z): return self.arithmetic(1, P - 1, x, y, z)
    def mul_const_add(self, c, x, y): return self.arithmetic(c, 1, self.one(), x, y)
    def and_(self, a, b): return self.mul(a, b)
    def constant_bool(self, b): return self.one() if b else self.zero()
    def assert_zero(self, t): self.connect(t, self.zero())

    def assert_bool(self, b):                                # b (b - 1) = 0
        self.assert_zero(self.mul_sub(b, b, b))

    def add_virtual_bool_target_safe(self, value):
        b = self.target(1 if value else 0)
        self.assert_bool(b)
        return b

    def is_equal(self, x, y):
        """gadgets/arithmetic.rs `is_equal` with its EqualityGenerator: equal = (x == y), inv = 1 / (x - y) or 0"""
        d = (self.val[x] - self.val[y]) % P
        equal = self.target(1 if d == 0 else 0)
        not_equal = self.not_(equal)
        inv = self.target(pow(d, P - 2, P) if d else 0)
        diff = self.sub(x, y)
        self.assert_zero(self.mul(diff, equal))
        self.assert_zero(self.sub(self.mul(diff, inv), not_equal))
        return equal

    def permute(self, state):
        """One PoseidonGate row (`PoseidonHash::permute_swapped` with swap = false): 12 input targets -> 12 output targets."""
        from . import poseidon_py as pp
        self.rows.append([GATE_POSEIDON, 0, 0, (0, 0)])
        row = len(self.rows) - 1
        self.stats[GATE_POSEIDON] = self.stats.get(GATE_POSEIDON, 0) + 1
        vals = [self.val[t] for t in state]
        for k in range(12):
            self._place(state[k], row, k)
        self._place(self.zero(), row, 24)                    # swap flag
        self._poseidon_rows.append((row, vals))
        out = pp.permute_trace(vals)[0]
        return [self._wire(row, 12 + k, out[k]) for k in range(12)]

    def hash_n_to_hash_no_pad(self, inputs):
        """hashing.rs `hash_n_to_m_no_pad` in circuit: rate 8, overwrite mode, no padding; no input -> the zero state's first four."""
        state = [self.zero()] * 12
        for off in range(0, len(inputs), 8):
            chunk = inputs[off:off + 8]
            state = list(chunk) + state[len(chunk):]
            state = self.permute(state)
        return state[:4]

    def split_le(self, x, num_bits):
        """gadgets/split_base.rs `split_le`: BaseSumGate<2> rows of 63 limbs, surplus limbs tied to zero, the weighted sum tied to x."""
        nl = min(63, self.cfg.num_routed_wires - 1)
        k = -(-num_bits // nl)
        v = self.val[x]
        bits, sums = [], []
        for g in range(k):
            self.rows.append([synth.GATE_BASE_SUM, nl, 2, (0, 0)])
            row = len(self.rows) - 1
            self.stats[synth.GATE_BASE_SUM] = self.stats.get(synth.GATE_BASE_SUM, 0) + 1
            part = (v >> (nl * g)) & ((1 << nl) - 1)
            sums.append(self._wire(row, 0, part))
            bits += [self._wire(row, 1 + i, (part >> i) & 1) for i in range(nl)]
        for b in bits[num_bits:]:
            self.assert_zero(b)
        acc = self.zero()
        for t in reversed(sums):
            acc = self.mul_const_add(1 << nl, acc, t)
        self.connect(acc, x)
        return bits[:num_bits]

    # ---- build(): rows -> synth.Builder (selectors, sigmas) + the witness
    def build(self, min_log_n=0):
        from . import poseidon_py as pp
        cfg = self.cfg
        # public-input hash in circuit: hash_n_to_hash_no_pad over the public inputs, PoseidonGate rows (rate 8, overwrite
        # mode), result copy-constrained into the PublicInputGate's four wires [UPSTREAM plonk/circuit_builder.rs build()]
        pis = list(self.public_inputs)
        zero = self.zero()
        digest = self.hash_n_to_hash_no_pad(pis) if pis else [zero] * 4
        self.rows.append([GATE_PUBLIC_INPUT, 0, 0, (0, 0)])
        row_pi = len(self.rows) - 1
        for k in range(4):
            self._place(digest[k], row_pi, k)
        pi_hash = [self.val[t] for t in digest]
        poseidon_rows = self._poseidon_rows
        nrows = len(self.rows)
        log_n = max(min_log_n, 2, (nrows - 1).bit_length())
        # padding rows (NoopGate) stay all-zero, as in plonky2, where `PartitionWitness::full_witness` leaves every unset wire at zero:
        # then each advice wire of the witness is either a row-local generator's output or zero, which is what lets the routed
        # columns alone reproduce the whole witness on the GPU (glp_witness_stage with GLP_WITNESS_ROUTED_ONLY)
        b = synth.Builder(cfg, log_n, seed=0, random_from_row=1 << log_n)
        by_kind = {}
        for r, (g, p0, p1, consts) in enumerate(self.rows):
            by_kind.setdefault((g, p0, p1), []).append(r)
            if consts is not None:
                for k, c in enumerate(consts):
                    b.gate_consts[k, r] = c
        for (g, p0, p1), rr in by_kind.items():
            b.set_rows(np.array(rr), g, p0, p1)
        # routed cells of every target; one sigma cycle per connected class (all in numpy: a 2^20-row circuit has ~10^7 cells)
        ct = np.frombuffer(self._cell_t, dtype=np.int64)
        cr = np.frombuffer(self._cell_r, dtype=np.int64)
        cc = np.frombuffer(self._cell_c, dtype=np.int64)
        flat = cr * cfg.num_routed_wires + cc
        assert len(np.unique(flat)) == len(flat), "a routed cell was assigned twice"
        vals = np.frombuffer(self.val, dtype=np.uint64)
        b.wires[cc, cr] = vals[ct]
        par = np.frombuffer(self._parent, dtype=np.int64).copy()
        while True:                                           # pointer jumping: every target -> the root of its class
            nxt = par[par]
            if (nxt == par).all():
                break
            par = nxt
        root = par[ct]
        order = np.argsort(root, kind="stable")
        sr, sc, sroot = cr[order], cc[order], root[order]
        first = np.ones(len(order), dtype=bool)
        first[1:] = sroot[1:] != sroot[:-1]
        start = np.maximum.accumulate(np.where(first, np.arange(len(order)), 0))     # index of the first cell of each cell's class
        last = np.ones(len(order), dtype=bool)
        last[:-1] = first[1:]
        nxt_idx = np.where(last, start, np.arange(len(order)) + 1)                    # successor within the class, cyclically
        b.sig_row[sc, sr] = sr[nxt_idx]
        b.sig_col[sc, sr] = sc[nxt_idx]
        # advice wires (what the gates' row-local generators produce), vectorised per gate type
        w = b.wires

        def recs(gate):
            a = self._ops[gate]
            if not a:
                return None
            return (np.array([x[0] for x in a]), np.array([x[1] for x in a]), np.array([x[2] for x in a], dtype=np.uint64))
        # every U32ArithmeticGate slot (used or not) needs inverse = 1 / (2^32 - 1 - output_high)
        inv_free = pow(M32, P - 2, P)
        ua_rows = np.array([r for r, row in enumerate(self.rows) if row[0] == GATE_U32_ARITHMETIC], dtype=np.int64)
        for i in range(self.n_u32a):
            w[6 * i + 5, ua_rows] = inv_free
        rc = recs(GATE_U32_ARITHMETIC)
        if rc is not None:
            rr, ii, prod = rc
            hi = prod >> np.uint64(32)
            w[6 * ii + 5, rr] = _batch_inverse([M32 - int(h) for h in hi])
            for k in range(32):
                w[6 * self.n_u32a + 32 * ii + k, rr] = (prod >> np.uint64(2 * k)) & np.uint64(3)
        rc = recs(GATE_U32_SUBTRACTION)
        if rc is not None:
            rr, ii, res = rc
            for k in range(16):
                w[5 * self.n_sub + 16 * ii + k, rr] = (res >> np.uint64(2 * k)) & np.uint64(3)
        rc = recs(GATE_U32_INTERLEAVE)
        if rc is not None:
            rr, ii, x = rc
            for k in range(32):                              # big-endian bits
                w[2 * self.n_il + 32 * ii + k, rr] = (x >> np.uint64(31 - k)) & np.uint64(1)
        for gate in (GATE_UNINTERLEAVE_U32, GATE_UNINTERLEAVE_B32):
            rc = recs(gate)
            if rc is not None:
                rr, ii, v = rc
                for k in range(64):
                    w[3 * self.n_ul + 64 * ii + k, rr] = (v >> np.uint64(63 - k)) & np.uint64(1)
        for row, inputs in poseidon_rows:
            synth._fill_poseidon_row(b, row, inputs)
        b.public_inputs = np.array([self.val[t] for t in pis], dtype=np.uint64)
        c = b.build()
        c.pi_hash = np.array(pi_hash, dtype=np.uint64)
        c.gadget_rows = nrows
        c.gate_ops = {synth._GATE_META[g][1].split(" ")[0].split("(")[0]: n for g, n in sorted(self.stats.items())}
        return c


# ---------------------------------------------------------------------------------------------------------------------------
# Keccak-256 [REF src/hash/keccak256.rs]
KECCAK256_R = 1088
KECCAKF_ROTC = [1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44]
KECCAKF_PILN = [10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1]
KECCAKF_RNDC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
                0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
                0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
                0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]


def keccak_f1600(gb, s):
    """`_hash_keccak256_f1600` [REF src/hash/keccak256.rs:79-128]; s: 25 lanes of [lo, hi] U32 targets, updated in place."""
    rndc = [[gb.constant_u32(c & M32), gb.constant_u32(c >> 32)] for c in KECCAKF_RNDC]
    for rnd in range(24):
        bc = [gb.unsafe_xor_many_u64([s[i], s[i + 5], s[i + 10], s[i + 15], s[i + 20]]) for i in range(5)]       # theta
        for i in range(5):
            t2 = gb.xor_u64(bc[(i + 4) % 5], gb.lrot_u64(bc[(i + 1) % 5], 1))
            for j in range(5):
                s[5 * j + i] = gb.xor_u64(s[5 * j + i], t2)
        t = s[1]                                                                                                  # rho, pi
        for i in range(24):
            j = KECCAKF_PILN[i]
            s[j], t = gb.lrot_u64(t, KECCAKF_ROTC[i]), s[j]
        for j in range(5):                                                                                        # chi
            bc = [s[5 * j + i] for i in range(5)]
            for i in range(5):
                t2 = gb.and_u64(bc[(i + 2) % 5], gb.not_u64(bc[(i + 1) % 5]))
                s[5 * j + i] = gb.xor_u64(s[5 * j + i], t2)
        s[0] = gb.xor_u64(s[0], rndc[rnd])                                                                        # iota


def keccak256_circuit(message, blocks_num=1, config=None, min_log_n=0):
    """The circuit of `test_keccak256_short` / `_long` [REF src/hash/keccak256.rs:214-231,279-295]:
    `add_virtual_hash_input_target(blocks_num, KECCAK256_R)`, `hash_keccak256`, `public_hash_output`, with the witness
    `set_keccak256_input_target(message)` [REF :22-37].  Returns a synth.Circuit whose 8 public inputs are the little-endian u32
    limbs of keccak256(message) as the circuit computed them.  `c.hasher` is left at 0 (PoseidonGoldilocksConfig); the long
    test of the reference sets KeccakGoldilocksConfig (`c.hasher = 1`)."""
    message = bytes(message)
    num_actual = 1 + (8 * len(message)) // KECCAK256_R
    if num_actual > blocks_num:
        raise ValueError("message needs %d blocks, the circuit has %d" % (num_actual, blocks_num))
    padded = bytearray(blocks_num * (KECCAK256_R // 8))
    padded[:len(message)] = message
    padded[len(message)] |= 0x01                              # bit right after the end of the message
    padded[num_actual * (KECCAK256_R // 8) - 1] |= 0x80       # last bit of the last block
    gb = GadgetBuilder(config)
    limbs = [gb.target(int.from_bytes(padded[4 * i:4 * i + 4], "little")) for i in range(len(padded) // 4)]
    blocks = [gb.target(1 if i < num_actual - 1 else 0) for i in range(blocks_num - 1)]      # add_virtual_bool_target_unsafe
    chunks = KECCAK256_R // 64
    zero = gb.zero_u32()
    state = [[zero, zero] for _ in range(25)]
    for i in range(chunks):
        state[i] = [limbs[2 * i], limbs[2 * i + 1]]
    keccak_f1600(gb, state)
    for k, blk in enumerate(blocks):
        start = (k + 1) * chunks * 2
        nxt = [[gb.xor_u32(s[0], limbs[start + 2 * i]), gb.xor_u32(s[1], limbs[start + 2 * i + 1])] if i < chunks else list(s)
               for i, s in enumerate(state)]
        keccak_f1600(gb, nxt)
        state = [gb.conditional_u64(nxt[i], state[i], blk) for i in range(25)]
    out = [gb.target(gb.val[state[i // 2][i % 2]]) for i in range(8)]       # add_virtual_biguint_target(8)
    for i in range(8):
        gb.connect_u32(state[i // 2][i % 2], out[i])
        gb.register_public_input(out[i])
    c = gb.build(min_log_n)
    c.digest_bytes = b"".join(int(v).to_bytes(4, "little") for v in c.public_inputs)
    return c


# ---------------------------------------------------------------------------------------------------------------------------
# Sparse Merkle tree inclusion proof (BASELINE config 4) [REF src/smt/gadgets/verify/verify_smt.rs, src/smt/gadgets/common.rs]
def _poseidon_hash_no_pad(values):
    from . import poseidon_py as pp
    state = [0] * 12
    for off in range(0, len(values), 8):
        chunk = [int(v) % P for v in values[off:off + 8]]
        state = chunk + state[len(chunk):]
        state = pp.permute_trace(state)[0]
    return tuple(state[:4])


ZERO_HASH = (0, 0, 0, 0)


def hash_out_from_u128(v):
    """`GoldilocksHashOut::from_u128` [REF src/smt/goldilocks_poseidon/hash/mod.rs:254-267]: four little-endian u32 limbs"""
    return tuple((v >> (32 * i)) & M32 for i in range(4))


def _key_bits(key):
    """`KeyLike::to_bits` [REF src/smt/goldilocks_poseidon/mod.rs:27-48]: little-endian bits of the four elements' 8 bytes each"""
    return [(key[i // 64] >> (i % 64)) & 1 for i in range(256)]


class SparseMerkleTree:
    """The native tree the reference takes its witnesses from (`PoseidonSparseMerkleTreeMemory`), restated: `insert` / `find`
    [REF src/smt/tree.rs:253-372,589-676] with `PoseidonNodeHash` [REF src/smt/goldilocks_poseidon/mod.rs:160-184]:
    internal = two_to_one(left, right); leaf = hash_pad([key, value, 1]) = the un-padded hash of [key, value, 1, 1, 0, 1]."""

    def __init__(self):
        self.nodes, self.root = {}, ZERO_HASH

    @staticmethod
    def leaf_hash(key, value): return _poseidon_hash_no_pad(list(key) + list(value) + [1, 1, 0, 1])
    @staticmethod
    def internal_hash(left, right): return _poseidon_hash_no_pad(list(left) + list(right))

    def _put(self, node):
        h = self.leaf_hash(*node[1:]) if node[0] == "leaf" else self.internal_hash(*node[1:])
        self.nodes[h] = node
        return h

    def insert(self, key, value):
        if tuple(value) == ZERO_HASH:
            raise ValueError("value must be non-zero")
        bits = _key_bits(key)

        def ins(h, level):
            if h == ZERO_HASH:
                return self._put(("leaf", tuple(key), tuple(value)))
            node = self.nodes[h]
            if node[0] == "leaf":
                if node[1] == tuple(key):
                    raise ValueError("given key already exists")
                old_bits = _key_bits(node[1])
                lvl = level
                while old_bits[lvl] == bits[lvl]:            # both leaves move down while their paths agree
                    lvl += 1
                new = self._put(("leaf", tuple(key), tuple(value)))
                cur = self._put(("int", h, new) if bits[lvl] else ("int", new, h))
                for l in range(lvl - 1, level - 1, -1):
                    cur = self._put(("int", ZERO_HASH, cur) if bits[l] else ("int", cur, ZERO_HASH))
                return cur
            left, right = node[1], node[2]
            return self._put(("int", left, ins(right, level + 1)) if bits[level] else ("int", ins(left, level + 1), right))
        self.root = ins(self.root, 0)

    def find(self, key):
        """-> dict(root, found, siblings (root downwards), key, value, not_found_key, not_found_value, is_old0)"""
        bits, h, sib = _key_bits(key), self.root, []
        level = 0
        while True:
            if h == ZERO_HASH:
                return dict(root=self.root, found=False, siblings=sib, key=tuple(key), value=ZERO_HASH, not_found_key=ZERO_HASH,
                            not_found_value=ZERO_HASH, is_old0=True)
            node = self.nodes[h]
            if node[0] == "leaf":
                hit = node[1] == tuple(key)
                return dict(root=self.root, found=hit, siblings=sib, key=tuple(key), value=node[2] if hit else ZERO_HASH,
                            not_found_key=ZERO_HASH if hit else node[1], not_found_value=ZERO_HASH if hit else node[2], is_old0=False)
            sib.append(node[1] if bits[level] else node[2])
            h = node[2] if bits[level] else node[1]
            level += 1


def _logical_and_not(gb, x, y): return gb.arithmetic(P - 1, 1, x, y, x)                     # x (1 - y)  [REF src/smt/gadgets/common.rs:211-222]


def _conditionally_reverse(gb, x, y, cond):                                                    # [REF src/smt/gadgets/common.rs:128-156]
    left, right = [], []
    for xi, yi in zip(x, y):
        d = gb.sub(yi, xi)
        left.append(gb.arithmetic(1, 1, d, cond, xi))
        right.append(gb.arithmetic(P - 1, 1, d, cond, yi))
    return left, right


def _is_equal_hash_out(gb, l, r):                                                              # [REF src/smt/gadgets/common.rs:318-330]
    out = gb.constant_bool(True)
    for a, b in zip(l, r):
        out = gb.and_(out, gb.is_equal(a, b))
    return out


def _calc_leaf_hash(gb, key, value):                                                           # [REF src/smt/gadgets/common.rs:87-101]
    one, zero = gb.one(), gb.zero()
    return gb.hash_n_to_hash_no_pad(list(key) + list(value) + [one, one, zero, one])


def _calc_internal_hash(gb, child, sibling, swap):                                             # [REF src/smt/gadgets/common.rs:103-113,11-25]
    left, right = _conditionally_reverse(gb, child, sibling, swap)
    return gb.hash_n_to_hash_no_pad(left + right)


def _smt_lev_ins(gb, enabled, siblings):                                                       # [REF src/smt/gadgets/common.rs:372-431]
    n = len(siblings)
    zero4 = [gb.zero()] * 4
    is_zeros = [_is_equal_hash_out(gb, s, zero4) for s in siblings][::-1]
    gb.assert_zero(_logical_and_not(gb, enabled, is_zeros[0]))      # the last level's sibling must be zero
    lev_ins = [gb.not_(is_zeros[1])]
    done = [lev_ins[0]]
    for i in range(1, n - 1):
        lev_ins.append(_logical_and_not(gb, gb.not_(is_zeros[i + 1]), done[-1]))
        done.append(gb.add(lev_ins[-1], done[-1]))
    lev_ins.append(gb.not_(done[-1]))
    return lev_ins[::-1]


def _smt_verifier_sm(gb, is0, lev_ins, fnc, prev):                                             # [REF src/smt/gadgets/verify/verify_smt.rs:160-212]
    aux1 = gb.mul(prev["top"], lev_ins)
    aux2 = gb.mul(aux1, fnc)
    top = gb.sub(prev["top"], aux1)
    i_new = gb.sub(aux1, aux2)
    i_old = gb.mul(aux2, gb.sub(gb.constant_bool(True), is0))
    i0 = gb.mul(aux1, is0)
    na = gb.add(gb.add(gb.add(prev["na"], prev["i_new"]), prev["i_old"]), prev["i0"])
    return dict(top=top, i_new=i_new, i_old=i_old, i0=i0, na=na)


def _smt_verifier_level(gb, st, sibling, old1_leaf, new1_leaf, lr_bit, child):                 # [REF src/smt/gadgets/verify/verify_smt.rs:109-158]
    h = _calc_internal_hash(gb, child, sibling, lr_bit)
    out = []
    for a, b, c in zip(h, old1_leaf, new1_leaf):
        r = gb.add(gb.mul(a, st["top"]), gb.mul(b, st["i_old"]))
        out.append(gb.add(r, gb.mul(c, st["i_new"])))
    return out


def smt_inclusion_circuit(tree, key, n_levels=16, enabled=True, config=None, public=False, min_log_n=0):
    """`SparseMerkleInclusionProofTarget::add_virtual_to(builder, n_levels)` + `set_witness(tree.find(key), enabled)`
    [REF src/smt/gadgets/verify/verify_smt.rs:41-97,214-307; the driver is src/smt/gadgets/verify/mod.rs:3-52].  `public`: also
    register root, key and value as public inputs (the reference's driver registers none)."""
    w = tree.find(key)
    if len(w["siblings"]) >= n_levels:
        raise ValueError("the proof has %d siblings, the circuit %d levels" % (len(w["siblings"]), n_levels))
    gb = GadgetBuilder(config)
    h4 = lambda v: [gb.target(x) for x in v]
    siblings = [h4(s) for s in w["siblings"]] + [h4(ZERO_HASH) for _ in range(n_levels - len(w["siblings"]))]
    root, old_key, old_value = h4(w["root"]), h4(w["not_found_key"]), h4(w["not_found_value"])
    key_t, value_t = h4(w["key"]), h4(w["value"])
    enabled_t = gb.add_virtual_bool_target_safe(enabled)
    is_old0 = gb.add_virtual_bool_target_safe(w["is_old0"])
    fnc = gb.add_virtual_bool_target_safe(not w["found"])
    # verify_smt_inclusion_proof
    true_t, false_t = gb.constant_bool(True), gb.constant_bool(False)
    hash1_old = _calc_leaf_hash(gb, old_key, old_value)
    hash1_new = _calc_leaf_hash(gb, key_t, value_t)
    n2b_new = [b for i in range(4) for b in gb.split_le(key_t[i], 64)]
    lev_ins = _smt_lev_ins(gb, enabled_t, siblings)
    st = dict(top=enabled_t, i0=false_t, i_old=false_t, i_new=false_t, na=gb.sub(true_t, enabled_t))
    sm = []
    for i in range(n_levels):
        st = _smt_verifier_sm(gb, is_old0, lev_ins[i], fnc, st)
        sm.append(st)
    flag = gb.add(gb.add(gb.add(sm[-1]["na"], sm[-1]["i_old"]), sm[-1]["i_new"]), sm[-1]["i0"])
    gb.connect(flag, true_t)
    sm.reverse()
    levels = []
    for i in range(n_levels):
        child = [gb.zero()] * 4 if i == 0 else levels[i - 1]
        levels.append(_smt_verifier_level(gb, sm[i], siblings[n_levels - 1 - i], hash1_old, hash1_new, n2b_new[n_levels - 1 - i], child))
    levels.reverse()
    are_keys_equal = _is_equal_hash_out(gb, old_key, key_t)
    keys_ok = gb.and_(gb.and_(_logical_and_not(gb, fnc, is_old0), enabled_t), are_keys_equal)
    gb.connect(keys_ok, false_t)
    # enforce_equal_if_enabled(root, levels[0], enabled)   [REF src/smt/gadgets/common.rs:347-357]
    gb.connect(_logical_and_not(gb, enabled_t, _is_equal_hash_out(gb, root, levels[0])), false_t)
    if public:
        for t in root + key_t + value_t:
            gb.register_public_input(t)
    c = gb.build(min_log_n)
    c.smt_witness = w
    c.computed_root = tuple(gb.val[t] for t in levels[0])
    return c


# ---------------------------------------------------------------------------------------------------------------------------
# Sparse Merkle tree PROCESS proof (insert / update / remove / no-op) [REF src/smt/gadgets/process/process_smt.rs, process/utils.rs]
def smt_set(tree, key, value):
    """`SparseMerkleTree::set` [REF src/smt/tree.rs:139-150,561-586]: value 0 removes (or is a no-op), otherwise updates or inserts.
    Returns the process proof [REF src/smt/proof/process.rs] and leaves the tree in its new state.  Removal rebuilds the (canonical) tree
    from its remaining entries instead of restating `remove` [REF src/smt/tree.rs:373-534]; the proof fields are the same."""
    key, value = tuple(key), tuple(value)
    items = getattr(tree, "items", None)
    if items is None:
        items = tree.items = {n[1]: n[2] for n in tree.nodes.values() if n[0] == "leaf" and tree.find(n[1])["found"]}
    f = tree.find(key)
    old_root = tree.root
    trim = lambda sib: sib[:max((i + 1 for i, s in enumerate(sib) if s != ZERO_HASH), default=0)]
    if value == ZERO_HASH:
        if not f["found"]:                                   # ProcessNoOp
            return dict(old_root=old_root, new_root=old_root, old_key=key, old_value=ZERO_HASH, new_key=key, new_value=ZERO_HASH, siblings=[],
                        is_old0=True, fnc=(0, 0))
        del items[key]
        after = SparseMerkleTree()
        for k, v in items.items():
            after.insert(k, v)
        g = after.find(key)                                  # the insertion of (key, old value) into `after`, read backwards
        tree.nodes, tree.root = after.nodes, after.root
        return dict(old_root=old_root, new_root=after.root, old_key=key, old_value=f["value"], new_key=g["not_found_key"],
                    new_value=g["not_found_value"], siblings=trim(g["siblings"]), is_old0=g["is_old0"], fnc=(1, 1))
    if f["found"]:                                           # ProcessUpdate: the same path, another leaf
        items[key] = value
        bits = _key_bits(key)
        cur = tree._put(("leaf", key, value))
        for lvl in range(len(f["siblings"]) - 1, -1, -1):
            s = f["siblings"][lvl]
            cur = tree._put(("int", s, cur) if bits[lvl] else ("int", cur, s))
        tree.root = cur
        return dict(old_root=old_root, new_root=cur, old_key=key, old_value=f["value"], new_key=key, new_value=value, siblings=f["siblings"],
                    is_old0=False, fnc=(0, 1))
    items[key] = value                                       # ProcessInsert
    tree.insert(key, value)
    return dict(old_root=old_root, new_root=tree.root, old_key=f["not_found_key"], old_value=f["not_found_value"], new_key=key, new_value=value,
                siblings=trim(f["siblings"]), is_old0=f["is_old0"], fnc=(1, 0))


def _logical_or(gb, x, y): return gb.add(_logical_and_not(gb, x, y), y)                       # [REF src/smt/gadgets/common.rs:224-237]
def _logical_nor(gb, x, y): return _logical_and_not(gb, gb.not_(x), y)                        # [REF :250-259]
def _logical_xor(gb, x, y): return gb.sub(x, gb.arithmetic(2, P - 1, x, y, y))                # [REF :296-311]
def _conditionally_select(gb, x, y, cond): return _conditionally_reverse(gb, x, y, cond)[1]   # cond ? x : y   [REF :115-126]
def _element_wise_add(gb, x, y): return [gb.arithmetic(1, 1, a, gb.one(), b) for a, b in zip(x, y)]      # [REF :178-186]


def _enforce_equal_if_enabled(gb, left, right, enabled):                                      # [REF :347-357]
    gb.connect(_logical_and_not(gb, enabled, _is_equal_hash_out(gb, left, right)), gb.constant_bool(False))


def _process_role(gb, fnc):                                                                   # [REF src/smt/gadgets/process/utils.rs:28-57]
    return dict(is_no_op=_logical_nor(gb, fnc[0], fnc[1]), is_remove_op=gb.and_(fnc[0], fnc[1]), is_insert_or_remove_op=fnc[0],
                is_update_or_no_op=gb.not_(fnc[0]), is_not_no_op=_logical_or(gb, fnc[0], fnc[1]))


def _smt_processor_sm(gb, xor, is0, lev_ins, is_ins_or_rem, prev):                           # [REF src/smt/gadgets/process/process_smt.rs:386-431]
    aux1 = gb.and_(prev["top"], lev_ins)
    aux2 = gb.and_(aux1, is_ins_or_rem)
    top = _logical_and_not(gb, prev["top"], lev_ins)
    old0 = gb.and_(aux2, is0)
    t = _logical_or(gb, _logical_and_not(gb, aux2, is0), prev["bot"])
    new1 = gb.and_(t, xor)
    bot = _logical_and_not(gb, t, xor)
    upd = _logical_and_not(gb, aux1, is_ins_or_rem)
    na = _logical_or(gb, _logical_or(gb, _logical_or(gb, prev["new1"], prev["old0"]), prev["na"]), prev["upd"])
    return dict(top=top, old0=old0, new1=new1, bot=bot, na=na, upd=upd)


def _smt_processor_level(gb, st, sibling, old1_leaf, new1_leaf, new_lr_bit, old_child, new_child):      # [REF :292-384]
    zero4 = [gb.zero()] * 4
    old_hash = _calc_internal_hash(gb, old_child, sibling, new_lr_bit)
    b_n_u = gb.add(gb.add(st["bot"], st["new1"]), st["upd"])
    old_root = _element_wise_add(gb, _conditionally_select(gb, old_hash, zero4, st["top"]), _conditionally_select(gb, old1_leaf, zero4, b_n_u))
    t_b = gb.add(st["top"], st["bot"])
    new_left = _element_wise_add(gb, _conditionally_select(gb, new1_leaf, zero4, st["new1"]), _conditionally_select(gb, new_child, zero4, t_b))
    new_right = _element_wise_add(gb, _conditionally_select(gb, old1_leaf, zero4, st["new1"]), _conditionally_select(gb, sibling, zero4, st["top"]))
    new_hash = _calc_internal_hash(gb, new_left, new_right, new_lr_bit)
    t_b_n = gb.add(t_b, st["new1"])
    o_u = gb.add(st["old0"], st["upd"])
    new_root = _element_wise_add(gb, _conditionally_select(gb, new1_leaf, zero4, o_u), _conditionally_select(gb, new_hash, zero4, t_b_n))
    return old_root, new_root


def smt_process_circuit(proof, n_levels=16, config=None, public=True, min_log_n=0):
    """`SparseMerkleProcessProofTarget::add_virtual_to` + `set_witness(proof)` [REF src/smt/gadgets/process/process_smt.rs:41-118,120-290;
    driver: src/smt/gadgets/process/mod.rs:4-82, which registers keys, values and roots as public inputs].  `proof` = `smt_set(...)`."""
    if len(proof["siblings"]) >= n_levels:
        raise ValueError("siblings are too long")
    gb = GadgetBuilder(config)
    h4 = lambda v: [gb.target(x) for x in v]
    siblings = [h4(s) for s in proof["siblings"]] + [h4(ZERO_HASH) for _ in range(n_levels - len(proof["siblings"]))]
    old_root, old_key, old_value = h4(proof["old_root"]), h4(proof["old_key"]), h4(proof["old_value"])
    new_root, new_key, new_value = h4(proof["new_root"]), h4(proof["new_key"]), h4(proof["new_value"])
    is_old0 = gb.add_virtual_bool_target_safe(proof["is_old0"])
    fnc = [gb.add_virtual_bool_target_safe(proof["fnc"][0]), gb.add_virtual_bool_target_safe(proof["fnc"][1])]
    pis = old_key + old_value + new_key + new_value + old_root + new_root
    # verify_smt_process_proof
    true_t, false_t = gb.constant_bool(True), gb.constant_bool(False)
    zero4 = [gb.zero()] * 4
    is_remove = _process_role(gb, fnc)["is_remove_op"]
    # a removal is checked as the insertion it undoes: swap old and new, clear fnc[1]      [REF :144-158]
    tmp = gb.mul_sub(is_remove, fnc[1], fnc[1])
    fnc = [fnc[0], gb.mul_sub(is_remove, false_t, tmp)]                                    # builder._if(is_remove, false, fnc[1])
    old_key, new_key = _conditionally_reverse(gb, old_key, new_key, is_remove)
    old_value, new_value = _conditionally_reverse(gb, old_value, new_value, is_remove)
    old_root, new_root = _conditionally_reverse(gb, old_root, new_root, is_remove)
    role = _process_role(gb, fnc)
    enabled = role["is_not_no_op"]
    gb.connect(role["is_remove_op"], false_t)
    hash1_old = _calc_leaf_hash(gb, old_key, old_value)
    hash1_new = _calc_leaf_hash(gb, new_key, new_value)
    n2b_old = [b for e in old_key for b in gb.split_le(e, 64)]
    n2b_new = [b for e in new_key for b in gb.split_le(e, 64)]
    lev_ins = _smt_lev_ins(gb, enabled, siblings)
    xors = [_logical_xor(gb, a, b) for a, b in zip(n2b_old[:n_levels], n2b_new[:n_levels])]
    prev = dict(top=enabled, old0=false_t, new1=false_t, bot=false_t, na=gb.not_(enabled), upd=false_t)
    sm = []
    for i in range(n_levels):
        prev = _smt_processor_sm(gb, xors[i], is_old0, lev_ins[i], role["is_insert_or_remove_op"], prev)
        sm.append(prev)
    last = sm[-1]
    gb.connect(_logical_or(gb, _logical_or(gb, last["na"], last["new1"]), _logical_or(gb, last["old0"], last["upd"])), true_t)
    level = (zero4, zero4)
    for i in range(n_levels - 1, -1, -1):
        level = _smt_processor_level(gb, sm[i], siblings[i], hash1_old, hash1_new, n2b_new[i], level[0], level[1])
    _enforce_equal_if_enabled(gb, level[0], old_root, enabled)
    _enforce_equal_if_enabled(gb, level[1], new_root, enabled)
    _enforce_equal_if_enabled(gb, old_key, new_key, role["is_update_or_no_op"])
    _enforce_equal_if_enabled(gb, old_root, new_root, role["is_no_op"])
    _enforce_equal_if_enabled(gb, old_value, new_value, role["is_no_op"])
    if public:
        for t in pis:
            gb.register_public_input(t)
    c = gb.build(min_log_n)
    c.computed_roots = (tuple(gb.val[t] for t in level[0]), tuple(gb.val[t] for t in level[1]))
    return c
