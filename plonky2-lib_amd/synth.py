"""Synthetic plonky2 circuits + satisfying witnesses (host side, numpy).

The reference's circuits are built by Rust code (`CircuitBuilder` + the gadget traits of
src/{ecdsa,hash,smt,zkdsa}) that cannot run here, so tests and bench.py use stand-ins with the same
*shape*: `standard_ecc_config` (136 wires / 80 routed, [REF src/ecdsa/gadgets/ecdsa.rs:476-483]) or
`standard_recursion_config` (135 / 80), rows of ArithmeticGate / ConstantGate / PublicInputGate /
NoopGate and the reference's own three u32 gates [REF src/u32/gates/*.rs], copy constraints, and a
witness that satisfies every gate.  What `builder.build::<C>()` would produce for the prover is
restated: gates sorted by (degree, id), `selector_polynomials` grouping, k_is = 7^i, sigma values
k_is[col'] * w^row'.

Nothing here touches the GPU or the oracle.
"""
import numpy as np

from . import gl_numpy as gl

P = gl.P

GATE_NOOP, GATE_CONSTANT, GATE_PUBLIC_INPUT, GATE_ARITHMETIC, GATE_POSEIDON = 0, 1, 2, 3, 4
GATE_U32_INTERLEAVE, GATE_UNINTERLEAVE_U32, GATE_UNINTERLEAVE_B32 = 5, 6, 7
GATE_U32_ARITHMETIC, GATE_U32_ADD_MANY, GATE_U32_SUBTRACTION, GATE_U32_RANGE_CHECK = 8, 9, 10, 11
GATE_COMPARISON, GATE_BASE_SUM, GATE_RANDOM_ACCESS = 12, 13, 14

# (degree, id string as plonky2's `Gate::id` prints it) -- the build() sort key
_GATE_META = {
    GATE_NOOP: (0, "NoopGate"),
    GATE_CONSTANT: (1, "ConstantGate {{ num_consts: {p0} }}"),
    GATE_PUBLIC_INPUT: (1, "PublicInputGate"),
    GATE_ARITHMETIC: (3, "ArithmeticGate {{ num_ops: {p0} }}"),
    GATE_POSEIDON: (7, "PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>"),
    GATE_U32_INTERLEAVE: (2, "U32InterleaveGate {{ num_ops: {p0} }}"),
    GATE_UNINTERLEAVE_U32: (2, "UninterleaveToU32Gate {{ num_ops: {p0} }}"),
    GATE_UNINTERLEAVE_B32: (2, "UninterleaveToB32Gate {{ num_ops: {p0} }}"),
    GATE_U32_ARITHMETIC: (4, "U32ArithmeticGate {{ num_ops: {p0} }}"),
    GATE_U32_ADD_MANY: (4, "U32AddManyGate {{ num_addends: {p0}, num_ops: {p1} }}"),
    GATE_U32_SUBTRACTION: (4, "U32SubtractionGate {{ num_ops: {p0} }}"),
    GATE_U32_RANGE_CHECK: (4, "U32RangeCheckGate {{ num_input_limbs: {p0} }}"),
    GATE_COMPARISON: (4, "ComparisonGate {{ num_bits: {p0}, num_chunks: {p1} }}"),
    GATE_BASE_SUM: (4, "BaseSumGate {{ num_limbs: {p0} }} + Base: {p1}"),
    GATE_RANDOM_ACCESS: (5, "RandomAccessGate {{ bits: {p0} }}"),
}


def gate_degree(t, p0=0, p1=0):
    """`Gate::degree`: fixed per type except BaseSumGate<B> (the range product has B factors)."""
    return p1 if t == GATE_BASE_SUM else _GATE_META[t][0]


def gate_num_constraints(t, p0, p1=0):
    if t == GATE_U32_ADD_MANY:
        return p1 * 21
    if t == GATE_COMPARISON:
        cb = -(-p0 // p1)
        return 2 + 5 * p1 + 1 + (cb + 1) + 2
    if t == GATE_BASE_SUM:
        return 1 + p0
    if t == GATE_RANDOM_ACCESS:
        return (p1 & 0xFFFF) * (p0 + 2) + (p1 >> 16)
    return {GATE_NOOP: 0, GATE_CONSTANT: p0, GATE_PUBLIC_INPUT: 4, GATE_ARITHMETIC: p0, GATE_POSEIDON: 123,
            GATE_U32_INTERLEAVE: p0 * 34, GATE_UNINTERLEAVE_U32: p0 * 67, GATE_UNINTERLEAVE_B32: p0 * 67,
            GATE_U32_ARITHMETIC: p0 * 36, GATE_U32_SUBTRACTION: p0 * 19, GATE_U32_RANGE_CHECK: p0 * 17}[t]


class Config:
    """plonk/circuit_data.rs `CircuitConfig` presets the reference selects in code."""

    def __init__(self, num_wires, num_routed_wires, num_constants=2, num_challenges=2, max_quotient_degree_factor=8,
                 rate_bits=3, cap_height=4, proof_of_work_bits=16, num_query_rounds=28, arity_bits=4, final_poly_bits=5):
        self.num_wires, self.num_routed_wires, self.num_constants = num_wires, num_routed_wires, num_constants
        self.num_challenges, self.max_quotient_degree_factor = num_challenges, max_quotient_degree_factor
        self.rate_bits, self.cap_height, self.proof_of_work_bits = rate_bits, cap_height, proof_of_work_bits
        self.num_query_rounds, self.arity_bits, self.final_poly_bits = num_query_rounds, arity_bits, final_poly_bits

    @classmethod
    def standard_recursion_config(cls, **kw):
        return cls(135, 80, **kw)

    @classmethod
    def standard_ecc_config(cls, **kw):
        return cls(136, 80, **kw)

    def reduction_arity_bits(self, degree_bits):
        """fri/reduction_strategies.rs ConstantArityBits(arity_bits, final_poly_bits)."""
        out, d = [], degree_bits
        while d > self.final_poly_bits and d + self.rate_bits - self.arity_bits >= self.cap_height:
            out.append(self.arity_bits)
            d -= self.arity_bits
        return out


class Circuit:
    """Flat CommonCircuitData + ProverOnlyCircuitData (what prove() reads) and a witness."""
    pass


def _selector_groups(gates, max_degree):
    """gates/selectors.rs `selector_polynomials`: returns (selector_indices, groups)."""
    num_gates = len(gates)
    max_gate_degree = gates[-1][0]
    if max_gate_degree + num_gates - 1 <= max_degree:
        return [0] * num_gates, [(0, num_gates)]
    groups, start = [], 0
    while start < num_gates:
        size = 0
        while start + size < num_gates and size + gates[start + size][0] < max_degree:
            size += 1
        if size == 0:
            raise ValueError("gate of degree %d does not fit max_quotient_degree_factor %d (plonky2 panics here too)"
                             % (gates[start][0], max_degree - 1))
        groups.append((start, start + size))
        start += size
    sel = []
    for i in range(num_gates):
        sel.append(next(k for k, (a, b) in enumerate(groups) if a <= i < b))
    return sel, groups


class Builder:
    def __init__(self, config, log_n, seed=0, random_from_row=0):
        """random_from_row: rows below it start as zeros (a caller that fills them all saves the random draw)"""
        self.cfg, self.log_n, self.n = config, log_n, 1 << log_n
        self.rng = np.random.default_rng(seed)
        n, nw, nr = self.n, config.num_wires, config.num_routed_wires
        self.row_gate = np.zeros(n, dtype=np.int64)            # per row: key into self.gate_kinds
        self.gate_kinds = {}                                   # (type, p0) -> key
        self.gate_consts = np.zeros((config.num_constants, n), dtype=np.uint64)
        if random_from_row:                                    # unconstrained cells stay random
            self.wires = np.zeros((nw, n), dtype=np.uint64)
            if random_from_row < n:
                self.wires[:, random_from_row:] = gl.rand(self.rng, (nw, n - random_from_row))
        else:
            self.wires = gl.rand(self.rng, (nw, n))
        self.sig_row = np.tile(np.arange(n, dtype=np.int64), (nr, 1))
        self.sig_col = np.tile(np.arange(nr, dtype=np.int64)[:, None], (1, n))
        self.public_inputs = np.zeros(0, dtype=np.uint64)
        self._kind(GATE_NOOP, 0)

    def _kind(self, t, p0, p1=0):
        return self.gate_kinds.setdefault((t, p0, p1), len(self.gate_kinds))

    def set_rows(self, rows, t, p0, p1=0):
        self.row_gate[rows] = self._kind(t, p0, p1)

    def connect_pairs(self, rows_a, col_a, rows_b, col_b):
        """2-cycles between fresh cells (rows_a[i], col_a) <-> (rows_b[i], col_b)."""
        self.sig_row[col_a, rows_a], self.sig_col[col_a, rows_a] = rows_b, col_b
        self.sig_row[col_b, rows_b], self.sig_col[col_b, rows_b] = rows_a, col_a

    def connect_cycle(self, rows, cols):
        """One cycle through fresh cells (rows[i], cols[i]) in the given order."""
        rows, cols = np.asarray(rows), np.asarray(cols)
        self.sig_row[cols, rows] = np.roll(rows, -1)
        self.sig_col[cols, rows] = np.roll(cols, -1)

    def build(self):
        cfg, n, lg = self.cfg, self.n, self.log_n
        c = Circuit()
        kinds = sorted(self.gate_kinds.items(),
                       key=lambda kv: (gate_degree(*kv[0]), _GATE_META[kv[0][0]][1].format(p0=kv[0][1], p1=kv[0][2])))
        gates = [(gate_degree(t, p0, p1), t, p0, p1) for (t, p0, p1), _ in kinds]
        key_to_index = {key: i for i, (_, key) in enumerate(kinds)}
        sel_idx, groups = _selector_groups(gates, cfg.max_quotient_degree_factor + 1)
        num_selectors = len(groups)
        gate_index = np.array([key_to_index[k] for k in self.row_gate], dtype=np.int64)
        consts = np.zeros((num_selectors + cfg.num_constants, n), dtype=np.uint64)
        for g in range(num_selectors):
            a, b = groups[g]
            in_group = (gate_index >= a) & (gate_index < b)
            consts[g] = np.where(in_group, gate_index.astype(np.uint64), np.uint64(0xFFFFFFFF)) if num_selectors > 1 \
                else gate_index.astype(np.uint64)
        consts[num_selectors:] = self.gate_consts
        c.gates = [dict(type=t, p0=p0, p1=p1, selector_index=sel_idx[i], group_start=groups[sel_idx[i]][0],
                        group_end=groups[sel_idx[i]][1], row=i, num_constraints=gate_num_constraints(t, p0, p1))
                   for i, (_, t, p0, p1) in enumerate(gates)]
        c.degree_bits = lg
        c.num_wires, c.num_routed_wires = cfg.num_wires, cfg.num_routed_wires
        c.num_constants, c.num_selectors = consts.shape[0], num_selectors
        c.num_challenges, c.quotient_degree_factor = cfg.num_challenges, cfg.max_quotient_degree_factor
        c.num_partial_products = -(-cfg.num_routed_wires // c.quotient_degree_factor) - 1
        c.num_gate_constraints = max(g["num_constraints"] for g in c.gates)
        c.rate_bits, c.cap_height = cfg.rate_bits, cfg.cap_height
        c.proof_of_work_bits, c.num_query_rounds = cfg.proof_of_work_bits, cfg.num_query_rounds
        c.reduction_arity_bits = cfg.reduction_arity_bits(lg)
        c.k_is = gl.powers(7, cfg.num_routed_wires)          # get_unique_coset_shifts: powers of the generator
        if getattr(cfg, "scramble_k_is", False):               # test hook: any distinct coset shifts are valid
            c.k_is = np.ascontiguousarray(c.k_is[::-1])
        subgroup = gl.powers(gl.root_of_unity(lg), n)
        c.constants = consts
        c.sigmas = gl.mul(c.k_is[self.sig_col], subgroup[self.sig_row])
        c.wires = self.wires
        c.public_inputs = self.public_inputs
        c.circuit_digest = None     # filled by the prover library / the oracle (needs the constants+sigmas cap)
        return c


def _digits(v, base_bits, count):
    return [(v >> (base_bits * j)) & ((1 << base_bits) - 1) for j in range(count)]


def fill_ecdsa_gate_rows(b, first_row, rows_per_gate=2, only=None):
    """Rows of the seven remaining gate types of the secp256k1 circuit [REF src/ecdsa/gadgets/ecdsa.rs:72-96]
    with the parameters `standard_ecc_config` gives them (136 wires / 80 routed / 2 constants), each with a
    satisfying witness built from the gate's definition.  Returns the next free row."""
    cfg, w, rng = b.cfg, b.wires, b.rng
    nw, nr = cfg.num_wires, cfg.num_routed_wires
    M32 = (1 << 32) - 1
    ri = lambda hi: int(rng.integers(0, hi))
    row = first_row
    want = lambda t: only is None or t in only
    # U32ArithmeticGate: m0*m1 + addend = out_hi 2^32 + out_lo
    n_ops = min(nr // 6, nw // 38)
    for _ in range(rows_per_gate if want(GATE_U32_ARITHMETIC) else 0):
        b.set_rows(np.array([row]), GATE_U32_ARITHMETIC, n_ops)
        for i in range(n_ops):
            m0, m1, ad = ri(1 << 32), ri(1 << 32), ri(1 << 32)
            prod = m0 * m1 + ad
            lo, hi = prod & M32, prod >> 32
            w[6 * i:6 * i + 6, row] = [m0, m1, ad, lo, hi, pow((M32 - hi) % P, P - 2, P)]
            w[6 * n_ops + 32 * i:6 * n_ops + 32 * i + 32, row] = _digits(prod, 2, 32)
        row += 1
    # U32AddManyGate, 3 addends
    na = 3
    n_ops = min(nr // (na + 3), nw // (na + 3 + 18))
    for _ in range(rows_per_gate if want(GATE_U32_ADD_MANY) else 0):
        b.set_rows(np.array([row]), GATE_U32_ADD_MANY, na, n_ops)
        for i in range(n_ops):
            vals = [ri(1 << 32) for _ in range(na + 1)]
            tot = sum(vals)
            o = (na + 3) * i
            w[o:o + na + 1, row] = vals
            w[o + na + 1, row], w[o + na + 2, row] = tot & M32, tot >> 32
            lo = (na + 3) * n_ops + 18 * i
            w[lo:lo + 16, row] = _digits(tot & M32, 2, 16)
            w[lo + 16:lo + 18, row] = _digits(tot >> 32, 2, 2)
        row += 1
    # U32SubtractionGate
    n_ops = min(nr // 5, nw // 21)
    for _ in range(rows_per_gate if want(GATE_U32_SUBTRACTION) else 0):
        b.set_rows(np.array([row]), GATE_U32_SUBTRACTION, n_ops)
        for i in range(n_ops):
            x, y, bi = ri(1 << 32), ri(1 << 32), ri(2)
            d = x - y - bi
            bo = 1 if d < 0 else 0
            res = d + (bo << 32)
            w[5 * i:5 * i + 5, row] = [x, y, bi, res, bo]
            w[5 * n_ops + 16 * i:5 * n_ops + 16 * i + 16, row] = _digits(res, 2, 16)
        row += 1
    # U32RangeCheckGate, 8 limbs (exactly 136 wires)
    n_in = min(8, nw // 17)
    for _ in range(rows_per_gate if want(GATE_U32_RANGE_CHECK) else 0):
        b.set_rows(np.array([row]), GATE_U32_RANGE_CHECK, n_in)
        for i in range(n_in):
            v = ri(1 << 32)
            w[i, row] = v
            w[n_in + 16 * i:n_in + 16 * i + 16, row] = _digits(v, 2, 16)
        row += 1
    # ComparisonGate(32 bits, 16 chunks)
    nbits, nch = 32, 16
    cb = nbits // nch
    for k in range(rows_per_gate if want(GATE_COMPARISON) else 0):
        b.set_rows(np.array([row]), GATE_COMPARISON, nbits, nch)
        first, second = ri(1 << 32), ri(1 << 32)
        if k == 1:
            second = first                      # all chunks equal
        a, bb = _digits(first, cb, nch), _digits(second, cb, nch)
        ed, ce, iv, msd = [], [], [], 0
        for i in range(nch):
            diff = (bb[i] - a[i]) % P
            eq = 1 if diff == 0 else 0
            ed.append(1 if eq else pow(diff, P - 2, P))
            ce.append(eq)
            inter = eq * msd % P
            iv.append(inter)
            msd = (inter + (1 - eq) * diff) % P
        top = ((1 << cb) + msd) % P
        bits = _digits(top, 1, cb + 1)
        w[0, row], w[1, row], w[2, row], w[3, row] = first, second, bits[cb], msd
        o = 4
        for arr in (a, bb, ed, ce, iv, bits):
            w[o:o + len(arr), row] = arr
            o += len(arr)
        assert bits[cb] == (1 if first <= second else 0)
        row += 1
    # BaseSumGate<4>, 16 limbs
    for _ in range(rows_per_gate if want(GATE_BASE_SUM) else 0):
        b.set_rows(np.array([row]), GATE_BASE_SUM, 16, 4)
        v = ri(1 << 32)
        w[0, row] = v
        w[1:17, row] = _digits(v, 2, 16)
        row += 1
    # RandomAccessGate(bits = 4)
    bits_ra = 4
    vs = 1 << bits_ra
    copies = min(nr // (2 + vs), nw // (2 + vs + bits_ra))
    nextra = min(nr - copies * (2 + vs), cfg.num_constants)
    for _ in range(rows_per_gate if want(GATE_RANDOM_ACCESS) else 0):
        b.set_rows(np.array([row]), GATE_RANDOM_ACCESS, bits_ra, copies | (nextra << 16))
        for c in range(copies):
            idx = ri(vs)
            lst = [int(x) for x in gl.rand(rng, vs)]
            o = (2 + vs) * c
            w[o, row], w[o + 1, row] = idx, lst[idx]
            w[o + 2:o + 2 + vs, row] = lst
            w[(2 + vs) * copies + nextra + bits_ra * c:(2 + vs) * copies + nextra + bits_ra * (c + 1), row] = _digits(idx, 1, bits_ra)
        ex = [int(x) for x in gl.rand(rng, nextra)]
        b.gate_consts[:nextra, row] = ex
        w[(2 + vs) * copies:(2 + vs) * copies + nextra, row] = ex
        row += 1
    return row


def arith_circuit(log_n, config=None, seed=1, public_inputs=(), pi_hash=None, num_const_rows=4, num_noop_rows=3,
                  ecdsa_gate_rows=0, ecdsa_gate_subset=None, extra_rows=None):
    """ECDSA-shaped stand-in: one PublicInputGate row, a few ConstantGate rows, ArithmeticGate rows
    (20 ops wide for 80 routed wires) chained through copy constraints, NoopGate padding."""
    cfg = config or Config.standard_ecc_config()
    b = Builder(cfg, log_n, seed)
    n = b.n
    num_ops = cfg.num_routed_wires // 4
    pi = np.asarray(public_inputs, dtype=np.uint64)
    b.public_inputs = pi
    if pi_hash is None:
        if len(pi):
            raise ValueError("pass pi_hash = hash_no_pad(public_inputs) when public inputs are non-empty")
        pi_hash = np.zeros(4, np.uint64)
    need_const_rows = max(num_const_rows, 2)
    if n < 1 + need_const_rows + num_noop_rows + 2 + 10 * ecdsa_gate_rows:
        raise ValueError("log_n too small")
    row_pi = 0
    rows_c = np.arange(1, 1 + need_const_rows)
    first_arith = 1 + need_const_rows
    if ecdsa_gate_rows:
        first_arith = fill_ecdsa_gate_rows(b, first_arith, ecdsa_gate_rows, ecdsa_gate_subset)
    if extra_rows is not None:
        first_arith = extra_rows(b, first_arith)
    rows_a = np.arange(first_arith, n - num_noop_rows)
    b.set_rows(np.array([row_pi]), GATE_PUBLIC_INPUT, 0)
    b.set_rows(rows_c, GATE_CONSTANT, cfg.num_constants)
    b.set_rows(rows_a, GATE_ARITHMETIC, num_ops)
    na = len(rows_a)
    # ConstantGate rows: wire i = constant i.  Rows 0,1 carry the public-input hash.
    cvals = gl.rand(b.rng, (cfg.num_constants, need_const_rows))
    cvals[0, 0], cvals[1, 0], cvals[0, 1], cvals[1, 1] = pi_hash[0], pi_hash[1], pi_hash[2], pi_hash[3]
    b.gate_consts[:, rows_c] = cvals
    b.wires[:cfg.num_constants, rows_c] = cvals
    # PublicInputGate wires 0..3 = hash, copy-constrained to the constant cells
    b.wires[:4, row_pi] = pi_hash
    for k in range(4):
        b.connect_pairs(np.array([row_pi]), k, np.array([rows_c[k // 2]]), k % 2)
    # ArithmeticGate rows: per-row constants c0, c1; out_j = c0*m0*m1 + c1*addend; out_j -> addend_{j+1}
    c01 = gl.rand(b.rng, (2, na))
    b.gate_consts[0, rows_a], b.gate_consts[1, rows_a] = c01[0], c01[1]
    shared = gl.rand(b.rng, 1)[0]
    b.wires[1, rows_a] = shared                               # m1 of op 0: one big cycle over all rows
    b.connect_cycle(rows_a, np.full(na, 1))
    # remaining ConstantGate cells feed m0 of op 0 of the first arithmetic rows
    extra = [(r, k) for r in range(2, need_const_rows) for k in range(cfg.num_constants)]
    for t, (r, k) in enumerate(extra[:na]):
        b.wires[0, rows_a[t]] = cvals[k, r]
        b.connect_pairs(np.array([rows_c[r]]), k, np.array([rows_a[t]]), 0)
    for j in range(num_ops):
        m0, m1, ad = b.wires[4 * j, rows_a], b.wires[4 * j + 1, rows_a], b.wires[4 * j + 2, rows_a]
        out = gl.add(gl.mul(gl.mul(m0, m1), c01[0]), gl.mul(ad, c01[1]))
        b.wires[4 * j + 3, rows_a] = out
        if j + 1 < num_ops:
            b.wires[4 * (j + 1) + 2, rows_a] = out
            b.connect_pairs(rows_a, 4 * j + 3, rows_a, 4 * (j + 1) + 2)
    return b.build()


def ecdsa_shape_circuit(log_n, seed=3, rows_per_gate=2):
    """The headline stand-in: `standard_ecc_config`, every one of the 11 gate types the reference registers for
    its secp256k1 circuit [REF src/ecdsa/gadgets/ecdsa.rs:72-96] present (so the quotient kernel evaluates the
    same constraint set at every point, in three selector groups), ArithmeticGate rows filling the trace."""
    return arith_circuit(log_n, Config.standard_ecc_config(), seed=seed, ecdsa_gate_rows=rows_per_gate)


def _fill_interleave_rows(b, row, rows_per_gate=2):
    """Rows of the reference's three gates [REF src/u32/gates/*.rs] with satisfying witnesses."""
    cfg = b.cfg
    n_il = min(cfg.num_wires // 34, cfg.num_routed_wires // 2)
    n_ul = min(cfg.num_wires // 67, cfg.num_routed_wires // 3)

    def interleave(x):
        r = 0
        for i in range(32):
            r |= ((x >> i) & 1) << (2 * i)
        return r
    for _ in range(rows_per_gate):
        b.set_rows(np.array([row]), GATE_U32_INTERLEAVE, n_il)
        for op in range(n_il):
            x = int(b.rng.integers(0, 1 << 32))
            b.wires[2 * op, row], b.wires[2 * op + 1, row] = x, interleave(x)
            b.wires[2 * n_il + 32 * op:2 * n_il + 32 * op + 32, row] = [(x >> (31 - k)) & 1 for k in range(32)]
        row += 1
    for t in (GATE_UNINTERLEAVE_U32, GATE_UNINTERLEAVE_B32):
        for _ in range(rows_per_gate):
            b.set_rows(np.array([row]), t, n_ul)
            for op in range(n_ul):
                x, y = int(b.rng.integers(0, 1 << 31)), int(b.rng.integers(0, 1 << 31))
                v = interleave(x) + interleave(y)
                bits = [(v >> (63 - k)) & 1 for k in range(64)]
                ev = sum(bits[2 * j] << (31 - j) for j in range(32))
                od = sum(bits[2 * j + 1] << (31 - j) for j in range(32))
                if t == GATE_UNINTERLEAVE_B32:
                    ev, od = interleave(ev), interleave(od)
                b.wires[3 * op:3 * op + 3, row] = [v, ev, od]
                b.wires[3 * n_ul + 64 * op:3 * n_ul + 64 * op + 64, row] = bits
            row += 1
    return row


def keccak_shape_circuit(log_n, seed=4, rows_per_gate=2):
    """BASELINE configs 1 and 2 shape: `standard_recursion_config` (135 wires), the gate set SURVEY section 8 row Q
    lists for the u32 / Keccak circuits -- U32Arithmetic, U32AddMany, U32Subtraction, the reference's own
    U32Interleave / UninterleaveToU32 / UninterleaveToB32 [REF src/u32/interleaved_u32.rs:93-130], Constant,
    Arithmetic, PublicInput, Noop -- with ArithmeticGate rows filling the trace.  (The Keccak-f1600 wiring itself
    [REF src/hash/keccak256.rs:79-128] needs the Rust builder; the known-answer digests of that file pin the
    witness generator, not the prover.)"""
    return arith_circuit(log_n, Config.standard_recursion_config(), seed=seed, ecdsa_gate_rows=rows_per_gate,
                         ecdsa_gate_subset=(GATE_U32_ARITHMETIC, GATE_U32_ADD_MANY, GATE_U32_SUBTRACTION),
                         extra_rows=lambda b, row: _fill_interleave_rows(b, row, rows_per_gate))


def u32_circuit(log_n=6, config=None, seed=2):
    """Exercises the reference's own gates [REF src/u32/gates/interleave_u32.rs, uninterleave_to_u32.rs,
    uninterleave_to_b32.rs] next to arithmetic rows: x -> interleave(x), y -> interleave(y), then
    uninterleave(interleave(x) + interleave(y)) gives the AND (odds) and XOR (evens) bits -- the
    one-add XOR/AND trick of [REF src/u32/interleaved_u32.rs:145-179]."""
    cfg = config or Config.standard_recursion_config()
    b = Builder(cfg, log_n, seed)
    n = b.n
    n_il = min(cfg.num_wires // 34, cfg.num_routed_wires // 2)     # U32InterleaveGate::num_ops
    n_ul = min(cfg.num_wires // 67, cfg.num_routed_wires // 3)     # UninterleaveTo*Gate::num_ops
    num_ops = cfg.num_routed_wires // 4
    rows_il = np.arange(1, 1 + 4)
    rows_u32 = np.arange(5, 5 + 2)
    rows_b32 = np.arange(7, 7 + 2)
    rows_c = np.arange(9, 11)
    rows_a = np.arange(11, n - 2)
    b.set_rows(np.array([0]), GATE_PUBLIC_INPUT, 0)
    b.set_rows(rows_il, GATE_U32_INTERLEAVE, n_il)
    b.set_rows(rows_u32, GATE_UNINTERLEAVE_U32, n_ul)
    b.set_rows(rows_b32, GATE_UNINTERLEAVE_B32, n_ul)
    b.set_rows(rows_c, GATE_CONSTANT, cfg.num_constants)
    b.set_rows(rows_a, GATE_ARITHMETIC, num_ops)
    b.wires[:4, 0] = 0
    b.gate_consts[:, rows_c] = 0
    b.wires[:cfg.num_constants, rows_c] = 0
    for k in range(4):
        b.connect_pairs(np.array([0]), k, np.array([rows_c[k // 2]]), k % 2)

    def interleave(x):
        r = 0
        for i in range(32):
            r |= ((x >> i) & 1) << (2 * i)
        return r

    # interleave rows
    for r in rows_il:
        for op in range(n_il):
            x = int(b.rng.integers(0, 1 << 32))
            b.wires[2 * op, r] = x
            b.wires[2 * op + 1, r] = interleave(x)
            for k in range(32):     # big-endian bits
                b.wires[2 * n_il + 32 * op + k, r] = (x >> (31 - k)) & 1
    # uninterleave rows: input = interleave(x) + interleave(y) (< 2^64, in the field by construction of the test values)
    for rows, b32 in ((rows_u32, False), (rows_b32, True)):
        for r in rows:
            for op in range(n_ul):
                x, y = int(b.rng.integers(0, 1 << 31)), int(b.rng.integers(0, 1 << 31))
                v = interleave(x) + interleave(y)
                b.wires[3 * op, r] = v
                bits = [(v >> (63 - k)) & 1 for k in range(64)]
                ev = sum(bits[2 * j] << (31 - j) for j in range(32))
                od = sum(bits[2 * j + 1] << (31 - j) for j in range(32))
                if b32:
                    ev, od = interleave(ev), interleave(od)
                else:
                    assert od == (x ^ y) and ev == (x & y)
                b.wires[3 * op + 1, r], b.wires[3 * op + 2, r] = ev, od
                for k in range(64):
                    b.wires[3 * n_ul + 64 * op + k, r] = bits[k]
    # arithmetic rows as in arith_circuit (no cross-row wiring except the shared m1 cycle)
    na = len(rows_a)
    c01 = gl.rand(b.rng, (2, na))
    b.gate_consts[0, rows_a], b.gate_consts[1, rows_a] = c01[0], c01[1]
    b.wires[1, rows_a] = gl.rand(b.rng, 1)[0]
    b.connect_cycle(rows_a, np.full(na, 1))
    for j in range(num_ops):
        m0, m1, ad = b.wires[4 * j, rows_a], b.wires[4 * j + 1, rows_a], b.wires[4 * j + 2, rows_a]
        out = gl.add(gl.mul(gl.mul(m0, m1), c01[0]), gl.mul(ad, c01[1]))
        b.wires[4 * j + 3, rows_a] = out
        if j + 1 < num_ops:
            b.wires[4 * (j + 1) + 2, rows_a] = out
            b.connect_pairs(rows_a, 4 * j + 3, rows_a, 4 * (j + 1) + 2)
    return b.build()


def _fill_poseidon_row(b, row, inputs):
    """Sets the 135 wires of a PoseidonGate row (gates/poseidon.rs layout) for `inputs` with swap = 0."""
    from . import poseidon_py as pp
    out, f0, part, f1 = pp.permute_trace(inputs)
    w = b.wires
    for i in range(12):
        w[i, row] = inputs[i]
        w[12 + i, row] = out[i]
    w[24, row] = 0
    for i in range(4):
        w[25 + i, row] = 0
    for r in range(3):
        for i in range(12):
            w[29 + 12 * r + i, row] = f0[r][i]
    for r in range(22):
        w[65 + r, row] = part[r]
    for r in range(4):
        for i in range(12):
            w[87 + 12 * r + i, row] = f1[r][i]
    return out


def zkdsa_circuit(log_n=3, config=None, seed=5, private_key=None, message=None):
    """The reference's simple-signature circuit [REF src/zkdsa/circuits/mod.rs:24-43,
    src/zkdsa/gadgets/signature/mod.rs:49-62]: public_key = H(sk || sk), signature = H(sk || msg) with
    `poseidon_two_to_one` [REF src/poseidon/gadgets/mod.rs:7-22]; public inputs = message, public_key,
    signature (12 elements), whose in-circuit hash (two more permutations) feeds the PublicInputGate.
    Rows: PublicInputGate, 4 x PoseidonGate, ConstantGate (zero), NoopGate padding -- 2^3 rows like
    the real circuit; gate placement and wiring are this builder's, not plonky2's."""
    cfg = config or Config.standard_recursion_config()
    b = Builder(cfg, log_n, seed)
    rng = b.rng
    sk = [int(x) for x in (gl.rand(rng, 4) if private_key is None else private_key)]
    msg = [int(x) for x in (gl.rand(rng, 4) if message is None else message)]
    rows_p = [1, 2, 3, 4]
    row_c = 5
    b.set_rows(np.array([0]), GATE_PUBLIC_INPUT, 0)
    b.set_rows(np.array(rows_p), GATE_POSEIDON, 0)
    b.set_rows(np.array([row_c]), GATE_CONSTANT, cfg.num_constants)
    b.gate_consts[:, row_c] = 0
    b.wires[:cfg.num_constants, row_c] = 0
    pk = _fill_poseidon_row(b, 1, sk + sk + [0] * 4)[:4]
    sig = _fill_poseidon_row(b, 2, sk + msg + [0] * 4)[:4]
    pis = msg + pk + sig
    o3 = _fill_poseidon_row(b, 3, pis[:8] + [0] * 4)
    o4 = _fill_poseidon_row(b, 4, pis[8:12] + o3[4:12])
    b.public_inputs = np.array(pis, dtype=np.uint64)
    b.wires[:4, 0] = o4[:4]
    cyc = lambda cells: b.connect_cycle([r for r, _ in cells], [c for _, c in cells])
    for i in range(4):
        cyc([(1, i), (1, 4 + i), (2, i)])                 # private key
        cyc([(2, 4 + i), (3, i)])                          # message
        cyc([(1, 12 + i), (3, 4 + i)])                     # public key
        cyc([(2, 12 + i), (4, i)])                         # signature
        cyc([(4, 12 + i), (0, i)])                         # public-input hash -> PublicInputGate
    for i in range(4, 12):
        cyc([(3, 12 + i), (4, i)])                         # sponge state carried into the second absorb
    zeros = [(row_c, 0)] + [(r, 24) for r in rows_p] + [(r, 8 + i) for r in (1, 2, 3) for i in range(4)]
    cyc(zeros)                                             # constant zero: swap flags and capacity lanes
    return b.build()


def poseidon_chain_circuit(log_n, config=None, seed=6):
    """SMT-shaped stand-in [REF src/smt/gadgets/verify/verify_smt.rs:214-307: a chain of Poseidon hashes
    walking up a Merkle path]: rows 1..n-3 are PoseidonGate rows, each absorbing the previous digest."""
    cfg = config or Config.standard_recursion_config()
    b = Builder(cfg, log_n, seed)
    n = b.n
    rows_p = list(range(1, n - 2))
    row_c = n - 2
    b.set_rows(np.array([0]), GATE_PUBLIC_INPUT, 0)
    b.set_rows(np.array(rows_p), GATE_POSEIDON, 0)
    b.set_rows(np.array([row_c]), GATE_CONSTANT, cfg.num_constants)
    b.gate_consts[:, row_c] = 0
    b.wires[:cfg.num_constants, row_c] = 0
    b.wires[:4, 0] = 0
    cur = [int(x) for x in gl.rand(b.rng, 4)]
    zeros = [(row_c, 0), (0, 0), (0, 1), (0, 2), (0, 3)]
    prev = None
    for r in rows_p:
        sib = [int(x) for x in gl.rand(b.rng, 4)]
        out = _fill_poseidon_row(b, r, cur + sib + [0] * 4)
        if prev is not None:
            for i in range(4):
                b.connect_cycle([prev, r], [12 + i, i])
        zeros += [(r, 24)] + [(r, 8 + i) for i in range(4)]
        cur, prev = out[:4], r
    b.connect_cycle([r for r, _ in zeros], [c for _, c in zeros])
    return b.build()


def _fill_arith_rows(b, rows_a):
    """ArithmeticGate rows (20 ops wide at 80 routed wires): per-row constants c0, c1; out_j = c0 m0 m1 + c1 addend, each
    output copy-constrained into the next op's addend; m1 of op 0 is one cycle through all rows."""
    cfg = b.cfg
    num_ops = cfg.num_routed_wires // 4
    na = len(rows_a)
    if na == 0:
        return
    b.set_rows(rows_a, GATE_ARITHMETIC, num_ops)
    c01 = gl.rand(b.rng, (2, na))
    b.gate_consts[0, rows_a], b.gate_consts[1, rows_a] = c01[0], c01[1]
    b.wires[1, rows_a] = gl.rand(b.rng, 1)[0]
    b.connect_cycle(rows_a, np.full(na, 1))
    for j in range(num_ops):
        m0, m1, ad = b.wires[4 * j, rows_a], b.wires[4 * j + 1, rows_a], b.wires[4 * j + 2, rows_a]
        out = gl.add(gl.mul(gl.mul(m0, m1), c01[0]), gl.mul(ad, c01[1]))
        b.wires[4 * j + 3, rows_a] = out
        if j + 1 < num_ops:
            b.wires[4 * (j + 1) + 2, rows_a] = out
            b.connect_pairs(rows_a, 4 * j + 3, rows_a, 4 * (j + 1) + 2)


def smt_shape_circuit(log_n, config=None, seed=7, levels=16):
    """BASELINE config 4 stand-in: the gate mix of the sparse-Merkle-tree inclusion circuit
    [REF src/smt/gadgets/verify/verify_smt.rs:214-307, src/smt/gadgets/common.rs:87-112], `standard_recursion_config`.
    One inclusion proof of `levels` levels instantiates: 2 leaf hashes of 12 inputs (2 permutations each) + one
    two-to-one hash per level = levels + 4 PoseidonGate rows [common.rs:87-101,16-25]; `split_le(key[i], 64)` for the four
    key elements [verify_smt.rs:240-242] = 8 BaseSumGate<2> rows (63 limbs per gate at 80 routed wires: a 63-bit gate and a
    1-bit gate per element); the level state machine, conditional selects and equality checks = ArithmeticGate ops
    (about 30 per level, 20 ops per row).  The trace is filled with that proportion (levels + 4 : 8 : 1.5 levels),
    repeated as in a batch of inclusion proofs; ConstantGate / PublicInputGate / NoopGate as in every circuit.
    Rows: [PublicInput][Constant x2][Poseidon chain ...][BaseSum<2> ...][Arithmetic ...][Noop x2]."""
    cfg = config or Config.standard_recursion_config()
    b = Builder(cfg, log_n, seed)
    n = b.n
    if n < 16:
        raise ValueError("log_n too small")
    body = n - 5
    wp, wb, wa = levels + 4, 8, (3 * levels + 1) // 2
    n_p = max(1, body * wp // (wp + wb + wa))
    n_b = max(2, body * wb // (wp + wb + wa))
    rows_p = list(range(3, 3 + n_p))
    rows_b = list(range(3 + n_p, 3 + n_p + n_b))
    rows_a = np.arange(3 + n_p + n_b, n - 2)
    rows_c = np.array([1, 2])
    b.set_rows(np.array([0]), GATE_PUBLIC_INPUT, 0)
    b.set_rows(rows_c, GATE_CONSTANT, cfg.num_constants)
    b.set_rows(np.array(rows_p), GATE_POSEIDON, 0)
    b.gate_consts[:, rows_c] = 0
    b.wires[:cfg.num_constants, rows_c] = 0
    b.wires[:4, 0] = 0
    # Poseidon rows: chains of `levels` two-to-one hashes, each absorbing the previous digest and a sibling
    zeros = [(1, 0), (0, 0), (0, 1), (0, 2), (0, 3)]
    cur, prev = None, None
    for t, r in enumerate(rows_p):
        if t % (levels + 4) == 0:
            cur, prev = [int(x) for x in gl.rand(b.rng, 4)], None
        sib = [int(x) for x in gl.rand(b.rng, 4)]
        out = _fill_poseidon_row(b, r, cur + sib + [0] * 4)
        if prev is not None:
            for i in range(4):
                b.connect_cycle([prev, r], [12 + i, i])
        zeros += [(r, 24)] + [(r, 8 + i) for i in range(4)]
        cur, prev = out[:4], r
    b.connect_cycle([r for r, _ in zeros], [c for _, c in zeros])
    # BaseSumGate<2>, 63 limbs: alternately a 63-bit value and a single bit (the two gates of one split_le(x, 64))
    nl = min(cfg.num_routed_wires - 1, 63)
    for t, r in enumerate(rows_b):
        b.set_rows(np.array([r]), GATE_BASE_SUM, nl, 2)
        v = int(b.rng.integers(0, 1 << 62)) * 2 + int(b.rng.integers(0, 2)) if t % 2 == 0 else int(b.rng.integers(0, 2))
        b.wires[0, r] = v
        b.wires[1:1 + nl, r] = _digits(v, 1, nl)
    # the bit wires of a pair of split gates feed arithmetic rows in the real circuit; here: limb 0 of consecutive gates
    # with equal values are tied (a copy constraint between BaseSum rows)
    lim0 = {}
    for r in rows_b:
        lim0.setdefault(int(b.wires[1, r]), []).append(r)
    for rows in lim0.values():
        if len(rows) > 1:
            b.connect_cycle(rows, [1] * len(rows))
    _fill_arith_rows(b, rows_a)
    return b.build()
