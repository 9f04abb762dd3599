/*
 * oracle/gl_witness_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of the row-local `SimpleGenerator::run_once` bodies that the GPU kernel of
 * plonky2-lib_amd/csrc/witness.hip replaces.  One function per generator, written
 * against the Rust definitions (get_local_wire / set_wire on one row), not against the HIP code:
 *   U32InterleaveGenerator      [REF src/u32/gates/interleave_u32.rs:289-318]
 *   UninterleaveToU32Generator  [REF src/u32/gates/uninterleave_to_u32.rs:332-369]
 *   UninterleaveToB32Generator  [REF src/u32/gates/uninterleave_to_b32.rs:335-372]
 *   plonky2_u32 @552acaec (crate absent from /root/reference, restated from the published source):
 *     gates/arithmetic_u32.rs U32ArithmeticGenerator, add_many_u32.rs U32AddManyGenerator,
 *     subtraction_u32.rs U32SubtractionGenerator, range_check_u32.rs U32RangeCheckGenerator,
 *     comparison.rs ComparisonGenerator
 *   plonky2 0.1.4 (crate absent, restated): gates/base_sum.rs BaseSplitGenerator, arithmetic_base.rs
 *     ArithmeticBaseGenerator, random_access.rs RandomAccessGenerator, poseidon.rs PoseidonGenerator,
 *     constant.rs ConstantGenerator
 * Pinned by: the three reference generators are read from the reference; all of them must produce a witness
 * that the gate CONSTRAINTS (gl_prover_oracle.c, pinned as that file says) accept -- tests/test_oracle_witness.py
 * checks exactly that -- and must reproduce the witnesses that plonky2-lib_amd/synth.py builds independently in Python.
 */
#include "gl_circuit.h"
#include "poseidon_constants.h"
#include <string.h>

#define API __attribute__((visibility("default")))

typedef struct { u64 *w; const u64 *k; size_t n, row; int only_advice; u32 nr; } rowctx;
static u64 get_wire(const rowctx *r, u32 col) { return r->w[(size_t)col * r->n + r->row]; }
static void set_wire(const rowctx *r, u32 col, u64 v) {
    if (r->only_advice && col < r->nr) return;
    r->w[(size_t)col * r->n + r->row] = v;
}
static u64 gate_const(const rowctx *r, u32 nsel, u32 i) { return r->k[(size_t)(nsel + i) * r->n + r->row]; }

/* [REF src/u32/gates/interleave_u32.rs:289-318] */
static void gen_u32_interleave(const rowctx *r, u32 num_ops, u32 i) {
    const u32 num_bits = 32;
    u64 x = get_wire(r, 2 * i), x_interleaved = 0;
    for (u32 k = 0; k < num_bits; k++) {                       /* wires_ith_bit_decomposition(i).enumerate() */
        u64 bit = (x >> (num_bits - k - 1)) % 2;
        set_wire(r, 2 * num_ops + num_bits * i + k, bit);
        x_interleaved += bit * ((u64)1 << (2 * (num_bits - k - 1)));
    }
    set_wire(r, 2 * i + 1, x_interleaved);
}
/* [REF src/u32/gates/uninterleave_to_u32.rs:332-369] (b32 = 0), [REF src/u32/gates/uninterleave_to_b32.rs:335-372] (b32 = 1) */
static void gen_uninterleave(const rowctx *r, u32 num_ops, u32 i, int b32) {
    const u32 num_bits = 64, start_bits = num_ops * 3;
    u64 x_interleaved = get_wire(r, 3 * i), x_evens = 0, x_odds = 0;
    for (u32 j = 0; j < num_bits / 2; j++) {
        u32 shift = 2 * (num_bits / 2 - j - 1);
        u64 jth_even = (x_interleaved >> (shift + 1)) % 2, jth_odd = (x_interleaved >> shift) % 2;
        set_wire(r, 2 * j + start_bits + num_bits * i, jth_even);
        set_wire(r, 2 * j + 1 + start_bits + num_bits * i, jth_odd);
        u64 coeff = b32 ? (u64)1 << (2 * (num_bits / 2 - j - 1)) : (u64)1 << (num_bits / 2 - j - 1);
        x_evens += jth_even * coeff;
        x_odds += jth_odd * coeff;
    }
    set_wire(r, 3 * i + 1, x_evens);
    set_wire(r, 3 * i + 2, x_odds);
}
/* plonky2_u32 arithmetic_u32.rs: output = m0 * m1 + addend; high/low halves; inverse of (u32::MAX - high) or 0; 32 base-4 limbs */
static void gen_u32_arithmetic(const rowctx *r, u32 num_ops, u32 i) {
    u64 output = gl_add(gl_mul(get_wire(r, 6 * i), get_wire(r, 6 * i + 1)), get_wire(r, 6 * i + 2));
    u64 high = output >> 32, low = output & 0xFFFFFFFFull;
    set_wire(r, 6 * i + 3, low);
    set_wire(r, 6 * i + 4, high);
    u64 diff = 0xFFFFFFFFull - high;
    set_wire(r, 6 * i + 5, diff == 0 ? 0 : gl_inv(diff));
    u64 q = output;
    for (u32 j = 0; j < 32; j++) { set_wire(r, 6 * num_ops + 32 * i + j, q % 4); q /= 4; }
}
/* plonky2_u32 add_many_u32.rs: sum of addends + carry; result = low 32 bits, carry = the rest; 16 + 2 base-4 limbs */
static void gen_u32_add_many(const rowctx *r, u32 num_addends, u32 num_ops, u32 i) {
    u32 wd = num_addends + 3;
    u64 sum = 0;
    for (u32 j = 0; j < num_addends; j++) sum = gl_add(sum, get_wire(r, wd * i + j));
    sum = gl_add(sum, get_wire(r, wd * i + num_addends));
    u64 result = sum & 0xFFFFFFFFull, carry = sum >> 32;
    set_wire(r, wd * i + num_addends + 1, result);
    set_wire(r, wd * i + num_addends + 2, carry);
    u64 q = result;
    for (u32 j = 0; j < 16; j++) { set_wire(r, wd * num_ops + 18 * i + j, q % 4); q /= 4; }
    q = carry;
    for (u32 j = 0; j < 2; j++) { set_wire(r, wd * num_ops + 18 * i + 16 + j, q % 4); q /= 4; }
}
/* plonky2_u32 subtraction_u32.rs: result_initial = x - y - borrow; output_borrow = result_initial_u64 > 2^32 */
static void gen_u32_subtraction(const rowctx *r, u32 num_ops, u32 i) {
    u64 ri = gl_sub(gl_sub(get_wire(r, 5 * i), get_wire(r, 5 * i + 1)), get_wire(r, 5 * i + 2));
    u64 borrow = ri > ((u64)1 << 32) ? 1 : 0;
    u64 result = gl_add(ri, gl_mul((u64)1 << 32, borrow));
    set_wire(r, 5 * i + 3, result);
    set_wire(r, 5 * i + 4, borrow);
    u64 q = result;
    for (u32 j = 0; j < 16; j++) { set_wire(r, 5 * num_ops + 16 * i + j, q % 4); q /= 4; }
}
/* plonky2_u32 range_check_u32.rs */
static void gen_u32_range_check(const rowctx *r, u32 num_input_limbs, u32 i) {
    u64 q = get_wire(r, i);
    for (u32 j = 0; j < 16; j++) { set_wire(r, num_input_limbs + 16 * i + j, q % 4); q /= 4; }
}
/* plonky2_u32 comparison.rs */
static void gen_comparison(const rowctx *r, u32 num_bits, u32 num_chunks) {
    u32 chunk_bits = (num_bits + num_chunks - 1) / num_chunks;
    u64 first = get_wire(r, 0), second = get_wire(r, 1), base = (u64)1 << chunk_bits;
    u32 o_first = 4, o_second = 4 + num_chunks, o_dummy = 4 + 2 * num_chunks, o_eq = 4 + 3 * num_chunks, o_inter = 4 + 4 * num_chunks,
        o_bits = 4 + 5 * num_chunks;
    u64 msd = 0, qf = first, qs = second;
    for (u32 i = 0; i < num_chunks; i++) {
        u64 fc = qf % base, sc = qs % base;
        qf /= base; qs /= base;
        set_wire(r, o_first + i, fc);
        set_wire(r, o_second + i, sc);
        u64 eq = fc == sc ? 1 : 0;
        set_wire(r, o_dummy + i, eq ? 1 : gl_inv(gl_sub(sc, fc)));
        set_wire(r, o_eq + i, eq);
        u64 inter = eq ? msd : 0;                               /* chunks_equal * most_significant_diff_so_far */
        set_wire(r, o_inter + i, inter);
        msd = eq ? inter : gl_sub(sc, fc);                      /* intermediate + (1 - chunks_equal) * (second - first) */
    }
    set_wire(r, 3, msd);
    u64 two_n_plus_msd = gl_add(base, msd);
    for (u32 j = 0; j <= chunk_bits; j++) set_wire(r, o_bits + j, (two_n_plus_msd >> j) & 1);
    set_wire(r, 2, (two_n_plus_msd >> chunk_bits) & 1);
}
/* plonky2 gates/base_sum.rs BaseSplitGenerator */
static void gen_base_split(const rowctx *r, u32 num_limbs, u32 base) {
    u64 acc = get_wire(r, 0);
    for (u32 j = 0; j < num_limbs; j++) { set_wire(r, 1 + j, acc % base); acc /= base; }
}
/* plonky2 gates/arithmetic_base.rs */
static void gen_arithmetic(const rowctx *r, u32 nsel, u32 i) {
    u64 c0 = gate_const(r, nsel, 0), c1 = gate_const(r, nsel, 1);
    set_wire(r, 4 * i + 3, gl_add(gl_mul(gl_mul(get_wire(r, 4 * i), get_wire(r, 4 * i + 1)), c0), gl_mul(get_wire(r, 4 * i + 2), c1)));
}
/* plonky2 gates/random_access.rs */
static void gen_random_access(const rowctx *r, u32 bits, u32 copies, u32 nextra, u32 copy) {
    u32 vec_size = 1u << bits, o = (2 + vec_size) * copy, routed = (2 + vec_size) * copies + nextra;
    u64 access_index = get_wire(r, o);
    if (access_index >= vec_size) return;                       /* the Rust generator would panic on the list index */
    set_wire(r, o + 1, get_wire(r, o + 2 + (u32)access_index));
    for (u32 b = 0; b < bits; b++) set_wire(r, routed + bits * copy + b, (access_index >> b) & 1);
}
/* plonky2 gates/poseidon.rs PoseidonGenerator (naive round schedule: the S-box inputs it records are the same) */
static const u64 W_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static u64 w_sbox(u64 x) { u64 x2 = gl_sqr(x), x4 = gl_sqr(x2); return gl_mul(gl_mul(x, x2), x4); }
static void w_mds(u64 s[12]) {
    u64 o[12];
    for (int row = 0; row < 12; row++) {
        u128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (u128)s[(i + row) % 12] * W_CIRC[i];
        if (row == 0) acc += (u128)s[0] * 8;
        o[row] = gl_reduce128(acc);
    }
    memcpy(s, o, sizeof(o));
}
static void gen_poseidon(const rowctx *r) {
    u64 st[12], swap = get_wire(r, 24);
    for (u32 i = 0; i < 4; i++) {
        u64 lhs = get_wire(r, i), rhs = get_wire(r, i + 4);
        u64 delta = gl_mul(swap, gl_sub(rhs, lhs));
        set_wire(r, 25 + i, delta);
        st[i] = gl_add(lhs, delta);
        st[i + 4] = gl_sub(rhs, delta);
    }
    for (u32 i = 8; i < 12; i++) st[i] = get_wire(r, i);
    int rc = 0;
    for (int round = 0; round < 30; round++) {
        for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], (u64)GL_POSEIDON_RC[rc + i]);
        rc += 12;
        if (round < 4) {
            if (round != 0) for (u32 i = 0; i < 12; i++) set_wire(r, 29 + 12 * (round - 1) + i, st[i]);
            for (int i = 0; i < 12; i++) st[i] = w_sbox(st[i]);
        } else if (round < 26) {
            set_wire(r, 65 + (round - 4), st[0]);
            st[0] = w_sbox(st[0]);
        } else {
            for (u32 i = 0; i < 12; i++) set_wire(r, 87 + 12 * (round - 26) + i, st[i]);
            for (int i = 0; i < 12; i++) st[i] = w_sbox(st[i]);
        }
        w_mds(st);
    }
    for (u32 i = 0; i < 12; i++) set_wire(r, 12 + i, st[i]);
}

/* every row: find its gate through the selector polynomials, run that gate's generators once */
API void glo_witness_fill(const glo_circuit *c, u64 *wires, int only_advice) {
    size_t n = (size_t)1 << c->degree_bits;
#pragma omp parallel for schedule(static)
    for (size_t row = 0; row < n; row++) {
        u32 gi = 0xFFFFFFFFu;
        for (u32 s = 0; s < c->num_selectors; s++) {
            u64 v = c->constants[(size_t)s * n + row];
            if (c->num_selectors == 1 || v != 0xFFFFFFFFull) gi = (u32)v;
        }
        if (gi >= c->num_gates) continue;
        const glo_gate *g = &c->gates[gi];
        rowctx r = {wires, c->constants, n, row, only_advice, c->num_routed_wires};
        switch (g->type) {
        case GLO_GATE_CONSTANT: for (u32 i = 0; i < g->p0; i++) set_wire(&r, i, gate_const(&r, c->num_selectors, i)); break;
        case GLO_GATE_ARITHMETIC: for (u32 i = 0; i < g->p0; i++) gen_arithmetic(&r, c->num_selectors, i); break;
        case GLO_GATE_POSEIDON: gen_poseidon(&r); break;
        case GLO_GATE_U32_INTERLEAVE: for (u32 i = 0; i < g->p0; i++) gen_u32_interleave(&r, g->p0, i); break;
        case GLO_GATE_UNINTERLEAVE_U32: for (u32 i = 0; i < g->p0; i++) gen_uninterleave(&r, g->p0, i, 0); break;
        case GLO_GATE_UNINTERLEAVE_B32: for (u32 i = 0; i < g->p0; i++) gen_uninterleave(&r, g->p0, i, 1); break;
        case GLO_GATE_U32_ARITHMETIC: for (u32 i = 0; i < g->p0; i++) gen_u32_arithmetic(&r, g->p0, i); break;
        case GLO_GATE_U32_ADD_MANY: for (u32 i = 0; i < g->p1; i++) gen_u32_add_many(&r, g->p0, g->p1, i); break;
        case GLO_GATE_U32_SUBTRACTION: for (u32 i = 0; i < g->p0; i++) gen_u32_subtraction(&r, g->p0, i); break;
        case GLO_GATE_U32_RANGE_CHECK: for (u32 i = 0; i < g->p0; i++) gen_u32_range_check(&r, g->p0, i); break;
        case GLO_GATE_COMPARISON: gen_comparison(&r, g->p0, g->p1); break;
        case GLO_GATE_BASE_SUM: gen_base_split(&r, g->p0, g->p1); break;
        case GLO_GATE_RANDOM_ACCESS: {
            u32 copies = g->p1 & 0xFFFF, nextra = g->p1 >> 16, vs = 1u << g->p0;
            for (u32 cp = 0; cp < copies; cp++) gen_random_access(&r, g->p0, copies, nextra, cp);
            for (u32 e = 0; e < nextra; e++) set_wire(&r, (2 + vs) * copies + e, gate_const(&r, c->num_selectors, e));
            break;
        }
        default: break;
        }
    }
}
