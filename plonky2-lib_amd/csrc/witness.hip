// witness.hip -- the row-local half of witness generation on the GPU.
//
// plonky2's `generate_partial_witness` (iop/generator.rs) runs every gate's `SimpleGenerator`s as their
// dependencies become known.  Two kinds of work hide in it: the copy-constraint DATAFLOW between rows (serial,
// irregular: stays on the CPU with the reference's gadget code) and the ROW-LOCAL generators that derive a row's
// remaining wires from that row's own inputs -- bit and limb decompositions, inverses, S-box traces.  The second
// kind is most of the witness by volume (56 of 136 columns of the secp256k1 trace are base-4 limbs) and is
// embarrassingly row-parallel: one thread per trace row here, written straight into the HBM witness that
// glp_prove_device consumes, so those columns never cross PCIe.
//
// Generators restated (one `case` each below; wire layouts = the gate definitions in prover.hip `gate_terms`):
//   the reference's own     U32InterleaveGenerator      [REF src/u32/gates/interleave_u32.rs:289-318]
//                           UninterleaveToU32Generator  [REF src/u32/gates/uninterleave_to_u32.rs:332-369]
//                           UninterleaveToB32Generator  [REF src/u32/gates/uninterleave_to_b32.rs:335-372]
//   plonky2_u32 (crate absent, recalled)  U32ArithmeticGenerator, U32AddManyGenerator, U32SubtractionGenerator,
//                           U32RangeCheckGenerator, ComparisonGenerator
//   plonky2 (crate absent, recalled)      BaseSplitGenerator (gates/base_sum.rs), ArithmeticBaseGenerator,
//                           RandomAccessGenerator, PoseidonGenerator, ConstantGenerator (gates/constant.rs and the
//                           extra constants of RandomAccessGate)
// Every value written is a canonical field element; inputs are read as canonical u64 exactly as
// `to_canonical_u64()` hands them to the Rust generators.
#include "common.h"
#include "poseidon.h"
#include "prover_types.h"

namespace {

struct WArgs {
    u64 *wires;              // [num_wires][n]
    const u64 *consts;       // [num_constants][n] values on H (selectors first)
    const DevGate *gates;
    u32 lg, nsel, num_gates, only_advice, nr;
};

__device__ __forceinline__ u64 winv(u64 a) {          // a != 0
    u64 r = 1, b = a, e = P - 2;
    while (e) { if (e & 1) r = mul(r, b); b = sqr(b); e >>= 1; }
    return r;
}

// One thread = one trace row.  Rows of one gate type are contiguous in every circuit the builder emits, so a wave
// rarely sees more than one `case`.
__global__ __launch_bounds__(256) void k_witness_fill(WArgs a) {
    const size_t n = (size_t)1 << a.lg;
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    // the row's gate: the selector polynomial of its group holds the gate index, every other selector UNUSED
    u32 gi = 0xFFFFFFFFu;
    for (u32 s = 0; s < a.nsel; s++) {
        const u64 v = a.consts[(size_t)s * n + row];
        if (a.nsel == 1 || v != 0xFFFFFFFFull) gi = (u32)v;
    }
    if (gi >= a.num_gates) return;
    const DevGate g = a.gates[gi];
    u64 *W = a.wires + row;                                    // wire j -> W[j * n]
    const u64 *GC = a.consts + (size_t)a.nsel * n + row;       // gate constant i -> GC[i * n]
    const u32 nr = a.only_advice ? a.nr : 0xFFFFFFFFu;         // only_advice: columns < nr (routed) are never written
#define WR(col, val) do { const u32 _c = (col); if (!a.only_advice || _c >= nr) W[(size_t)_c * n] = (val); } while (0)
#define RD(col) W[(size_t)(col) * n]
    switch (g.type) {
    case GLP_GATE_CONSTANT:
        for (u32 i = 0; i < g.p0; i++) WR(i, GC[(size_t)i * n]);
        break;
    case GLP_GATE_ARITHMETIC: {
        const u64 c0 = GC[0], c1 = GC[n];
        for (u32 i = 0; i < g.p0; i++) WR(4 * i + 3, add(mul(mul(RD(4 * i), RD(4 * i + 1)), c0), mul(RD(4 * i + 2), c1)));
        break;
    }
    case GLP_GATE_POSEIDON: {
        // gates/poseidon.rs PoseidonGenerator: swap -> delta, then the round-by-round trace; the wires hold the S-box INPUTS
        u64 st[12];
        const u64 swap = RD(24);
        for (u32 i = 0; i < 4; i++) {
            const u64 lhs = RD(i), rhs = RD(i + 4);
            const u64 dl = mul(swap, sub(rhs, lhs));
            WR(25 + i, dl);
            st[i] = add(lhs, dl); st[i + 4] = sub(rhs, dl);
        }
        for (u32 i = 8; i < 12; i++) st[i] = RD(i);
        u32 rc = 0;
        for (u32 r = 0; r < 4; r++) {
            for (u32 i = 0; i < 12; i++) st[i] = add(st[i], pos::RC[rc + i]);
            rc += 12;
            if (r != 0) for (u32 i = 0; i < 12; i++) WR(29 + 12 * (r - 1) + i, st[i]);
            for (u32 i = 0; i < 12; i++) st[i] = pos::sbox7(st[i]);
            pos::mds_layer(st);
        }
        for (u32 r = 0; r < 22; r++) {
            for (u32 i = 0; i < 12; i++) st[i] = add(st[i], pos::RC[rc + i]);
            rc += 12;
            WR(65 + r, st[0]);
            st[0] = pos::sbox7(st[0]);
            pos::mds_layer(st);
        }
        for (u32 r = 0; r < 4; r++) {
            for (u32 i = 0; i < 12; i++) st[i] = add(st[i], pos::RC[rc + i]);
            rc += 12;
            for (u32 i = 0; i < 12; i++) WR(87 + 12 * r + i, st[i]);
            for (u32 i = 0; i < 12; i++) st[i] = pos::sbox7(st[i]);
            pos::mds_layer(st);
        }
        for (u32 i = 0; i < 12; i++) WR(12 + i, st[i]);
        break;
    }
    case GLP_GATE_U32_INTERLEAVE:
        // [REF src/u32/gates/interleave_u32.rs:289-318]: bit wire k = bit (31 - k) of x (big-endian), x_interleaved = sum bit 4^(31-k)
        for (u32 i = 0; i < g.p0; i++) {
            const u64 x = RD(2 * i);
            u64 xi = 0;
            for (u32 k = 0; k < 32; k++) {
                const u64 bit = (x >> (31 - k)) & 1;
                WR(2 * g.p0 + 32 * i + k, bit);
                xi += bit << (2 * (31 - k));
            }
            WR(2 * i + 1, canon(xi));
        }
        break;
    case GLP_GATE_UNINTERLEAVE_U32:
    case GLP_GATE_UNINTERLEAVE_B32:
        // [REF src/u32/gates/uninterleave_to_u32.rs:332-369, uninterleave_to_b32.rs:335-372]: 64 big-endian bits of
        // x_interleaved; even-position bits (shift + 1) -> x_evens, odd -> x_odds, weights 2^(31-j) (U32) or 4^(31-j) (B32)
        for (u32 i = 0; i < g.p0; i++) {
            const u64 x = RD(3 * i);
            u64 ev = 0, od = 0;
            for (u32 j = 0; j < 32; j++) {
                const u32 shift = 2 * (31 - j);
                const u64 be = (x >> (shift + 1)) & 1, bo = (x >> shift) & 1;
                WR(3 * g.p0 + 64 * i + 2 * j, be);
                WR(3 * g.p0 + 64 * i + 2 * j + 1, bo);
                const u32 cs = g.type == GLP_GATE_UNINTERLEAVE_U32 ? (31 - j) : 2 * (31 - j);
                ev += be << cs; od += bo << cs;
            }
            WR(3 * i + 1, canon(ev));
            WR(3 * i + 2, canon(od));
        }
        break;
    case GLP_GATE_U32_ARITHMETIC:
        // plonky2_u32 arithmetic_u32.rs U32ArithmeticGenerator: output = m0 m1 + addend in the field, as a canonical u64
        for (u32 i = 0; i < g.p0; i++) {
            const u64 out = add(mul(RD(6 * i), RD(6 * i + 1)), RD(6 * i + 2));
            const u64 lo = out & 0xFFFFFFFFull, hi = out >> 32;
            WR(6 * i + 3, lo);
            WR(6 * i + 4, hi);
            const u64 diff = 0xFFFFFFFFull - hi;
            WR(6 * i + 5, diff ? winv(diff) : 0);
            for (u32 j = 0; j < 32; j++) WR(6 * g.p0 + 32 * i + j, (out >> (2 * j)) & 3);
        }
        break;
    case GLP_GATE_U32_ADD_MANY: {
        // plonky2_u32 add_many_u32.rs U32AddManyGenerator: sum of the addends and the carry, split at bit 32
        const u32 na = g.p0, nops = g.p1, wd = na + 3;
        for (u32 i = 0; i < nops; i++) {
            u64 sum = RD(wd * i + na);
            for (u32 j = 0; j < na; j++) sum = add(sum, RD(wd * i + j));
            const u64 res = sum & 0xFFFFFFFFull, car = sum >> 32;
            WR(wd * i + na + 1, res);
            WR(wd * i + na + 2, car);
            for (u32 j = 0; j < 16; j++) WR(wd * nops + 18 * i + j, (res >> (2 * j)) & 3);
            for (u32 j = 0; j < 2; j++) WR(wd * nops + 18 * i + 16 + j, (car >> (2 * j)) & 3);
        }
        break;
    }
    case GLP_GATE_U32_SUBTRACTION:
        // plonky2_u32 subtraction_u32.rs U32SubtractionGenerator: x - y - borrow in the field; a wrapped result (> 2^32)
        // means a borrow, and 2^32 is added back
        for (u32 i = 0; i < g.p0; i++) {
            const u64 r0 = sub(sub(RD(5 * i), RD(5 * i + 1)), RD(5 * i + 2));
            const u64 bo = r0 > (1ull << 32) ? 1 : 0;
            const u64 res = add(r0, bo << 32);
            WR(5 * i + 3, res);
            WR(5 * i + 4, bo);
            for (u32 j = 0; j < 16; j++) WR(5 * g.p0 + 16 * i + j, (res >> (2 * j)) & 3);
        }
        break;
    case GLP_GATE_U32_RANGE_CHECK:
        // plonky2_u32 range_check_u32.rs U32RangeCheckGenerator: 16 base-4 limbs per input, little-endian
        for (u32 i = 0; i < g.p0; i++) {
            const u64 v = RD(i);
            for (u32 j = 0; j < 16; j++) WR(g.p0 + 16 * i + j, (v >> (2 * j)) & 3);
        }
        break;
    case GLP_GATE_COMPARISON: {
        // plonky2_u32 comparison.rs ComparisonGenerator (first <= second): chunks, equality dummies, chunk-equal flags,
        // intermediate values, most significant diff, its bits, result
        const u32 nb = g.p0, ncx = g.p1, cb = (nb + ncx - 1) / ncx;
        const u64 first = RD(0), second = RD(1);
        const u32 o_a = 4, o_b = o_a + ncx, o_ed = o_b + ncx, o_ce = o_ed + ncx, o_iv = o_ce + ncx, o_mb = o_iv + ncx;
        u64 msd = 0;
        for (u32 i = 0; i < ncx; i++) {
            const u64 ca = (first >> (cb * i)) & ((1ull << cb) - 1), cbv = (second >> (cb * i)) & ((1ull << cb) - 1);
            WR(o_a + i, ca);
            WR(o_b + i, cbv);
            const u64 diff = sub(cbv, ca);
            const u64 eq = diff == 0 ? 1 : 0;
            WR(o_ed + i, eq ? 1 : winv(diff));
            WR(o_ce + i, eq);
            const u64 inter = mul(eq, msd);
            WR(o_iv + i, inter);
            msd = add(inter, mul(sub(1, eq), diff));
        }
        WR(3, msd);
        const u64 top = add((u64)1 << cb, msd);
        for (u32 j = 0; j <= cb; j++) WR(o_mb + j, (top >> j) & 1);
        WR(2, (top >> cb) & 1);
        break;
    }
    case GLP_GATE_BASE_SUM: {
        // gates/base_sum.rs BaseSplitGenerator: little-endian digits of the sum in base B
        u64 v = RD(0);
        for (u32 j = 0; j < g.p0; j++) { WR(1 + j, v % g.p1); v /= g.p1; }
        break;
    }
    case GLP_GATE_RANDOM_ACCESS: {
        // gates/random_access.rs RandomAccessGenerator: index bits (little-endian) and the claimed element list[index];
        // the gate's extra constants are plain ConstantGenerators
        const u32 bits = g.p0, copies = g.p1 & 0xFFFF, nextra = g.p1 >> 16, vs = 1u << bits;
        const u32 routed = (2 + vs) * copies + nextra;
        for (u32 cpy = 0; cpy < copies; cpy++) {
            const u32 o = (2 + vs) * cpy;
            const u64 idx = RD(o);
            if (idx < vs) {                       // the Rust generator indexes the list: an out-of-range index panics there
                WR(o + 1, RD(o + 2 + (u32)idx));
                for (u32 b = 0; b < bits; b++) WR(routed + bits * cpy + b, (idx >> b) & 1);
            }
        }
        for (u32 e = 0; e < nextra; e++) WR((2 + vs) * copies + e, GC[(size_t)e * n]);
        break;
    }
    default: break;     // NoopGate; PublicInputGate (its wires are set from the public-input hash through copy constraints)
    }
#undef WR
#undef RD
}

}  // namespace

extern "C" int glp_witness_fill(glp_ctx *c, const glp_circuit *cc, uint64_t *dev_wires, int only_advice) {
    GLP_REQUIRE(c && cc && dev_wires, "null argument");
    GLP_REQUIRE(cc->ctx == c, "circuit belongs to another context");
    GLP_TRY(bind(c));
    WArgs a;
    a.wires = dev_wires; a.consts = cc->dev_consts; a.gates = cc->dev_gates;
    a.lg = cc->d.degree_bits; a.nsel = cc->d.num_selectors; a.num_gates = cc->d.num_gates;
    a.only_advice = only_advice ? 1u : 0u; a.nr = cc->d.num_routed_wires;
    const size_t n = (size_t)1 << a.lg;
    StageScope st(c, "witness_fill", 8.0 * n * cc->d.num_wires);
    hipLaunchKernelGGL(k_witness_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, a);
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}

// Which wire columns glp_witness_fill writes on rows of gate `gate_index` (1) and which it reads as inputs (2); 0 = untouched.
extern "C" int glp_witness_columns(const glp_circuit *cc, uint32_t gate_index, uint8_t *role_out /* [num_wires] */) {
    GLP_REQUIRE(cc && role_out, "null argument");
    GLP_REQUIRE(gate_index < cc->d.num_gates, "gate index out of range");
    const glp_gate &g = cc->gates[gate_index];
    const u32 nw = cc->d.num_wires;
    memset(role_out, 0, nw);
    auto out = [&](u32 c0, u32 cnt) { for (u32 i = 0; i < cnt && c0 + i < nw; i++) role_out[c0 + i] = 1; };
    auto in = [&](u32 c0, u32 cnt) { for (u32 i = 0; i < cnt && c0 + i < nw; i++) role_out[c0 + i] = 2; };
    switch (g.type) {
    case GLP_GATE_CONSTANT: out(0, g.p0); break;
    case GLP_GATE_ARITHMETIC: for (u32 i = 0; i < g.p0; i++) { in(4 * i, 3); out(4 * i + 3, 1); } break;
    case GLP_GATE_POSEIDON: in(0, 12); out(12, 12); in(24, 1); out(25, 110); break;
    case GLP_GATE_U32_INTERLEAVE: for (u32 i = 0; i < g.p0; i++) { in(2 * i, 1); out(2 * i + 1, 1); } out(2 * g.p0, 32 * g.p0); break;
    case GLP_GATE_UNINTERLEAVE_U32: case GLP_GATE_UNINTERLEAVE_B32:
        for (u32 i = 0; i < g.p0; i++) { in(3 * i, 1); out(3 * i + 1, 2); } out(3 * g.p0, 64 * g.p0); break;
    case GLP_GATE_U32_ARITHMETIC: for (u32 i = 0; i < g.p0; i++) { in(6 * i, 3); out(6 * i + 3, 3); } out(6 * g.p0, 32 * g.p0); break;
    case GLP_GATE_U32_ADD_MANY: {
        const u32 wd = g.p0 + 3;
        for (u32 i = 0; i < g.p1; i++) { in(wd * i, g.p0 + 1); out(wd * i + g.p0 + 1, 2); }
        out(wd * g.p1, 18 * g.p1);
        break;
    }
    case GLP_GATE_U32_SUBTRACTION: for (u32 i = 0; i < g.p0; i++) { in(5 * i, 3); out(5 * i + 3, 2); } out(5 * g.p0, 16 * g.p0); break;
    case GLP_GATE_U32_RANGE_CHECK: in(0, g.p0); out(g.p0, 16 * g.p0); break;
    case GLP_GATE_COMPARISON: { const u32 cb = (g.p0 + g.p1 - 1) / g.p1; in(0, 2); out(2, 2 + 5 * g.p1 + cb + 1); break; }
    case GLP_GATE_BASE_SUM: in(0, 1); out(1, g.p0); break;
    case GLP_GATE_RANDOM_ACCESS: {
        const u32 bits = g.p0, copies = g.p1 & 0xFFFF, nextra = g.p1 >> 16, vs = 1u << bits;
        for (u32 cpy = 0; cpy < copies; cpy++) { const u32 o = (2 + vs) * cpy; in(o, 1); out(o + 1, 1); in(o + 2, vs); }
        out((2 + vs) * copies, nextra);
        out((2 + vs) * copies + nextra, bits * copies);
        break;
    }
    default: break;
    }
    return GLP_OK;
}
