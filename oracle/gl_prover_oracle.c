/*
 * oracle/gl_prover_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of plonky2 0.1.4 `prove()` and `verify()` for circuits without lookups and
 * without blinding (zero_knowledge = false), i.e. what every `data.prove(pw)` / `data.verify(proof)`
 * pair in /root/reference executes [REF src/ecdsa/gadgets/ecdsa.rs:349,352;
 * src/hash/keccak256.rs:248-249; src/zkdsa/circuits/mod.rs:326,346].  Follows, function by
 * function (names only -- the crate is absent from /root/reference):
 *   plonk/prover.rs        prove_with_partition_witness, wires_permutation_partial_products_and_zs,
 *                          compute_quotient_polys
 *   plonk/vanishing_poly.rs eval_vanishing_poly(_base_batch), evaluate_gate_constraints
 *   plonk/plonk_common.rs  partial_products_and_z_gx, check_partial_products, ZeroPolyOnCoset,
 *                          eval_l_0, reduce_with_powers
 *   plonk/proof.rs         OpeningSet::new, to_fri_openings
 *   fri/oracle.rs          prove_openings;  fri/prover.rs  fri_committed_trees, fri_proof_of_work,
 *                          fri_prover_query_rounds;  fri/verifier.rs  verify_fri_proof & friends
 *   plonk/verifier.rs      verify_with_challenges;  plonk/get_challenges.rs
 *   gates/{noop,constant,public_input,arithmetic_base}.rs and the reference's own gates
 *   [REF src/u32/gates/interleave_u32.rs:84-135, uninterleave_to_u32.rs:93-150, uninterleave_to_b32.rs:95-150]
 *
 * Parity status: PINNED only through the Poseidon KAT (transcript + Merkle primitive) and the
 * relation the reference's own tests assert -- `verify(prove(w))` accepts, and rejects a wrong
 * witness [REF src/ecdsa/gadgets/curve.rs:300-326].  Proof BYTES against the forked Rust prover
 * are unpinned (no fixture in the reference; no Rust toolchain).
 * Deterministic choice where the Rust prover is not: fri_proof_of_work returns the SMALLEST
 * valid witness (rayon's find_any returns an arbitrary one).
 *
 * Everything is evaluated in the quadratic extension (base values embedded) so that the prover's
 * coset evaluation and the verifier's evaluation at zeta share one gate implementation.
 */
#include "gl_circuit.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define API __attribute__((visibility("default")))

/* from gl_oracle.c */
void glo_poseidon_permute(u64 s[12]);
void glo_hash_no_pad(const u64 *in, size_t len, u64 out[4]);
void glo_ifft(u64 *a, int lg);
void glo_fft(u64 *a, int lg);
void glo_coset_fft(u64 *a, int lg, u64 shift);
void glo_coset_ifft(u64 *a, int lg, u64 shift);
size_t glo_merkle_num_digests(size_t nleaves, int cap_height);
int glo_merkle_build(const u64 *leaves, size_t nleaves, size_t leaf_len, int cap_height, u64 *digests, u64 *cap_out);
int glo_merkle_prove(const u64 *digests, size_t nleaves, int cap_height, size_t index, u64 *siblings);
int glo_merkle_verify(const u64 *leaf, size_t leaf_len, size_t index, const u64 *cap, int cap_height, const u64 *siblings, int nsib);
int glo_batch_from_coeffs(const u64 *coeffs, size_t ncols, int lg, int rate_bits, int cap_height, u64 *leaves, u64 *digests, u64 *cap_out);
typedef struct { u64 st[12]; u64 in[8]; int nin; u64 out[8]; int nout; } glo_challenger;
void glo_challenger_init(glo_challenger *c);
void glo_challenger_observe(glo_challenger *c, const u64 *e, size_t n);
void glo_challenger_observe_hashes(glo_challenger *c, const u64 *digests, size_t count);   /* outer-hasher digests (caps, circuit digest) */
void glo_permute(u64 st[12]);                                                            /* the transcript's permutation */
u64 glo_challenger_get(glo_challenger *c);

static size_t brev(size_t x, int bits) {
    size_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
static gl2 ch_get_ext(glo_challenger *c) { u64 a = glo_challenger_get(c); u64 b = glo_challenger_get(c); return gl2_make(a, b); }

/* ------------------------------------------------------------------ gate constraints
 * `evaluate_gate_constraints`: every gate's filtered constraints are ADDED into
 * constraints[0 .. gate.num_constraints).  lc = local constants (all, selectors first),
 * lw = local wires, pih = public inputs hash. */
static gl2 compute_filter(const glo_gate *g, gl2 s, int many_selectors) {
    gl2 f = gl2_from(1);
    for (u32 i = g->group_start; i < g->group_end; i++)
        if (i != g->row) f = gl2_mul(f, gl2_sub(gl2_from(i), s));
    if (many_selectors) f = gl2_mul(f, gl2_sub(gl2_from(0xFFFFFFFFull), s)); /* UNUSED_SELECTOR = u32::MAX */
    return f;
}
/* gates/poseidon.rs `PoseidonGate::eval_unfiltered`.  plonky2 evaluates the partial rounds in its "fast"
 * (sparse-matrix) form; the S-box inputs, hence every constraint polynomial, are the same as in the
 * naive schedule used here (plonky2 asserts the two schedules equal).  Wire layout: inputs 0..11, outputs
 * 12..23, swap 24, delta 25..28, full_sbox_0(r=1..3) from 29, partial_sbox from 65, full_sbox_1 from 87. */
#include "poseidon_constants.h"
static const u64 PG_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static gl2 gl2_sbox7(gl2 x) { gl2 x2 = gl2_mul(x, x), x4 = gl2_mul(x2, x2), x3 = gl2_mul(x, x2); return gl2_mul(x3, x4); }
static void gl2_mds(gl2 s[12]) {
    gl2 o[12];
    for (int r = 0; r < 12; r++) {
        gl2 acc = gl2_from(0);
        for (int i = 0; i < 12; i++) acc = gl2_add(acc, gl2_scale(s[(i + r) % 12], PG_CIRC[i]));
        if (r == 0) acc = gl2_add(acc, gl2_scale(s[0], 8));
        o[r] = acc;
    }
    memcpy(s, o, sizeof(o));
}
static void eval_poseidon_gate(const gl2 *lw, gl2 *out) {
    u32 k = 0;
    gl2 swap = lw[24], st[12];
    out[k++] = gl2_mul(swap, gl2_sub(swap, gl2_from(1)));
    for (int i = 0; i < 4; i++) out[k++] = gl2_sub(gl2_mul(swap, gl2_sub(lw[i + 4], lw[i])), lw[25 + i]);
    for (int i = 0; i < 4; i++) { st[i] = gl2_add(lw[i], lw[25 + i]); st[i + 4] = gl2_sub(lw[i + 4], lw[25 + i]); }
    for (int i = 8; i < 12; i++) st[i] = lw[i];
    int rc = 0;
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) st[i] = gl2_add(st[i], gl2_from((u64)GL_POSEIDON_RC[rc + i]));
        rc += 12;
        if (r != 0) for (int i = 0; i < 12; i++) { gl2 in = lw[29 + 12 * (r - 1) + i]; out[k++] = gl2_sub(st[i], in); st[i] = in; }
        for (int i = 0; i < 12; i++) st[i] = gl2_sbox7(st[i]);
        gl2_mds(st);
    }
    for (int r = 0; r < 22; r++) {
        for (int i = 0; i < 12; i++) st[i] = gl2_add(st[i], gl2_from((u64)GL_POSEIDON_RC[rc + i]));
        rc += 12;
        gl2 in = lw[65 + r]; out[k++] = gl2_sub(st[0], in);
        st[0] = gl2_sbox7(in);
        gl2_mds(st);
    }
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) st[i] = gl2_add(st[i], gl2_from((u64)GL_POSEIDON_RC[rc + i]));
        rc += 12;
        for (int i = 0; i < 12; i++) { gl2 in = lw[87 + 12 * r + i]; out[k++] = gl2_sub(st[i], in); st[i] = in; }
        for (int i = 0; i < 12; i++) st[i] = gl2_sbox7(st[i]);
        gl2_mds(st);
    }
    for (int i = 0; i < 12; i++) out[k++] = gl2_sub(st[i], lw[12 + i]);
}

static gl2 range_product(gl2 v, u32 bound) { /* prod_{x < bound} (v - x) */
    gl2 p = gl2_from(1);
    for (u32 x = 0; x < bound; x++) p = gl2_mul(p, gl2_sub(v, gl2_from(x)));
    return p;
}
static void eval_gate_unfiltered(const glo_gate *g, const gl2 *gc /* gate constants (selectors removed) */,
                                 const gl2 *lw, const u64 pih[4], gl2 *out) {
    switch (g->type) {
    case GLO_GATE_NOOP: break;
    case GLO_GATE_CONSTANT: /* constant.rs: local_constants[i] - local_wires[i] */
        for (u32 i = 0; i < g->p0; i++) out[i] = gl2_sub(gc[i], lw[i]);
        break;
    case GLO_GATE_PUBLIC_INPUT: /* public_input.rs: wires[i] - public_inputs_hash[i] */
        for (u32 i = 0; i < 4; i++) out[i] = gl2_sub(lw[i], gl2_from(pih[i]));
        break;
    case GLO_GATE_ARITHMETIC: /* arithmetic_base.rs: output - (m0*m1*c0 + addend*c1) */
        for (u32 i = 0; i < g->p0; i++) {
            gl2 m0 = lw[4 * i], m1 = lw[4 * i + 1], ad = lw[4 * i + 2], o = lw[4 * i + 3];
            gl2 comp = gl2_add(gl2_mul(gl2_mul(m0, m1), gc[0]), gl2_mul(ad, gc[1]));
            out[i] = gl2_sub(o, comp);
        }
        break;
    case GLO_GATE_POSEIDON: eval_poseidon_gate(lw, out); break;
    case GLO_GATE_U32_INTERLEAVE: { /* [REF src/u32/gates/interleave_u32.rs:84-135] */
        u32 k = 0;
        for (u32 i = 0; i < g->p0; i++) {
            gl2 x = lw[2 * i], xi = lw[2 * i + 1];
            const gl2 *bits = lw + 2 * g->p0 + 32 * i; /* big-endian */
            gl2 cx = gl2_from(0), cxi = gl2_from(0);
            for (int b = 0; b < 32; b++) { /* Horner from the most significant bit */
                cx = gl2_add(gl2_scale(cx, 2), bits[b]);
                cxi = gl2_add(gl2_scale(cxi, 4), bits[b]);
            }
            out[k++] = gl2_sub(cx, x);
            out[k++] = gl2_sub(cxi, xi);
            for (int b = 0; b < 32; b++) out[k++] = gl2_mul(bits[b], gl2_sub(bits[b], gl2_from(1)));
        }
        break;
    }
    case GLO_GATE_UNINTERLEAVE_U32:
    case GLO_GATE_UNINTERLEAVE_B32: { /* [REF src/u32/gates/uninterleave_to_u32.rs:93-150; uninterleave_to_b32.rs:95-150] */
        u32 k = 0;
        for (u32 i = 0; i < g->p0; i++) {
            gl2 xi = lw[3 * i], xe = lw[3 * i + 1], xo = lw[3 * i + 2];
            const gl2 *bits = lw + 3 * g->p0 + 64 * i;
            gl2 cxi = gl2_from(0), ce = gl2_from(0), co = gl2_from(0);
            for (int b = 0; b < 64; b++) cxi = gl2_add(gl2_scale(cxi, 2), bits[b]);
            for (int j = 0; j < 32; j++) {
                u64 coeff = g->type == GLO_GATE_UNINTERLEAVE_U32 ? ((u64)1 << (31 - j)) : ((u64)1 << (2 * (31 - j)));
                ce = gl2_add(ce, gl2_scale(bits[2 * j], coeff));
                co = gl2_add(co, gl2_scale(bits[2 * j + 1], coeff));
            }
            out[k++] = gl2_sub(cxi, xi);
            out[k++] = gl2_sub(ce, xe);
            out[k++] = gl2_sub(co, xo);
            for (int b = 0; b < 64; b++) out[k++] = gl2_mul(bits[b], gl2_sub(bits[b], gl2_from(1)));
        }
        break;
    }
    case GLO_GATE_U32_ARITHMETIC: { /* wires per op: m0, m1, addend, out_lo, out_hi, inverse; then 32 2-bit limbs per op */
        u32 k = 0; const u32 n = g->p0;
        for (u32 i = 0; i < n; i++) {
            gl2 m0 = lw[6 * i], m1 = lw[6 * i + 1], ad = lw[6 * i + 2], lo = lw[6 * i + 3], hi = lw[6 * i + 4], inv = lw[6 * i + 5];
            gl2 computed = gl2_add(gl2_mul(m0, m1), ad);
            gl2 diff = gl2_sub(gl2_from(0xFFFFFFFFull), hi);
            gl2 hi_not_max = gl2_sub(gl2_mul(inv, diff), gl2_from(1));
            out[k++] = gl2_mul(hi_not_max, lo);
            out[k++] = gl2_sub(gl2_add(gl2_scale(hi, (u64)1 << 32), lo), computed);
            gl2 cl = gl2_from(0), ch = gl2_from(0);
            const gl2 *limbs = lw + 6 * n + 32 * i;
            for (int j = 31; j >= 0; j--) {
                out[k++] = range_product(limbs[j], 4);
                if (j < 16) cl = gl2_add(gl2_scale(cl, 4), limbs[j]); else ch = gl2_add(gl2_scale(ch, 4), limbs[j]);
            }
            out[k++] = gl2_sub(cl, lo);
            out[k++] = gl2_sub(ch, hi);
        }
        break;
    }
    case GLO_GATE_U32_ADD_MANY: { /* per op: addends[na], carry_in, out_result, out_carry; then 16 + 2 2-bit limbs per op */
        u32 k = 0; const u32 na = g->p0, n = g->p1, w = na + 3;
        for (u32 i = 0; i < n; i++) {
            gl2 sum = lw[w * i + na];
            for (u32 j = 0; j < na; j++) sum = gl2_add(sum, lw[w * i + j]);
            gl2 res = lw[w * i + na + 1], car = lw[w * i + na + 2];
            out[k++] = gl2_sub(gl2_add(gl2_scale(car, (u64)1 << 32), res), sum);
            gl2 cr = gl2_from(0), cc = gl2_from(0);
            const gl2 *limbs = lw + w * n + 18 * i;
            for (int j = 17; j >= 0; j--) {
                out[k++] = range_product(limbs[j], 4);
                if (j < 16) cr = gl2_add(gl2_scale(cr, 4), limbs[j]); else cc = gl2_add(gl2_scale(cc, 4), limbs[j]);
            }
            out[k++] = gl2_sub(cr, res);
            out[k++] = gl2_sub(cc, car);
        }
        break;
    }
    case GLO_GATE_U32_SUBTRACTION: { /* per op: x, y, borrow_in, out_result, out_borrow; then 16 2-bit limbs per op */
        u32 k = 0; const u32 n = g->p0;
        for (u32 i = 0; i < n; i++) {
            gl2 x = lw[5 * i], y = lw[5 * i + 1], bi = lw[5 * i + 2], res = lw[5 * i + 3], bo = lw[5 * i + 4];
            gl2 init = gl2_sub(gl2_sub(x, y), bi);
            out[k++] = gl2_sub(res, gl2_add(init, gl2_scale(bo, (u64)1 << 32)));
            gl2 cl = gl2_from(0);
            const gl2 *limbs = lw + 5 * n + 16 * i;
            for (int j = 15; j >= 0; j--) { out[k++] = range_product(limbs[j], 4); cl = gl2_add(gl2_scale(cl, 4), limbs[j]); }
            out[k++] = gl2_sub(cl, res);
            out[k++] = gl2_mul(bo, gl2_sub(gl2_from(1), bo));
        }
        break;
    }
    case GLO_GATE_U32_RANGE_CHECK: { /* input limbs 0..n-1, then 16 base-4 aux limbs per input (little-endian) */
        u32 k = 0; const u32 n = g->p0;
        for (u32 i = 0; i < n; i++) {
            const gl2 *aux = lw + n + 16 * i;
            gl2 sum = gl2_from(0);
            for (int j = 15; j >= 0; j--) sum = gl2_add(gl2_scale(sum, 4), aux[j]);
            out[k++] = gl2_sub(sum, lw[i]);
            for (int j = 0; j < 16; j++) out[k++] = range_product(aux[j], 4);
        }
        break;
    }
    case GLO_GATE_COMPARISON: { /* wires: first, second, result_bool, msd, chunks a[nc], b[nc], eq_dummy[nc], chunks_equal[nc], intermediate[nc], msd bits[cb+1] */
        u32 k = 0; const u32 nb = g->p0, nc = g->p1, cb = (nb + nc - 1) / nc, cs = 1u << cb;
        const gl2 *a = lw + 4, *b = a + nc, *ed = b + nc, *ce = ed + nc, *iv = ce + nc, *mb = iv + nc;
        gl2 ca = gl2_from(0), cbv = gl2_from(0);
        for (int i = (int)nc - 1; i >= 0; i--) { ca = gl2_add(gl2_scale(ca, cs), a[i]); cbv = gl2_add(gl2_scale(cbv, cs), b[i]); }
        out[k++] = gl2_sub(ca, lw[0]);
        out[k++] = gl2_sub(cbv, lw[1]);
        gl2 msd = gl2_from(0);
        for (u32 i = 0; i < nc; i++) {
            out[k++] = range_product(a[i], cs);
            out[k++] = range_product(b[i], cs);
            gl2 diff = gl2_sub(b[i], a[i]);
            out[k++] = gl2_sub(gl2_mul(diff, ed[i]), gl2_sub(gl2_from(1), ce[i]));
            out[k++] = gl2_mul(ce[i], diff);
            out[k++] = gl2_sub(iv[i], gl2_mul(ce[i], msd));
            msd = gl2_add(iv[i], gl2_mul(gl2_sub(gl2_from(1), ce[i]), diff));
        }
        out[k++] = gl2_sub(lw[3], msd);
        gl2 bc = gl2_from(0);
        for (u32 j = 0; j <= cb; j++) out[k++] = gl2_mul(mb[j], gl2_sub(gl2_from(1), mb[j]));
        for (int j = (int)cb; j >= 0; j--) bc = gl2_add(gl2_scale(bc, 2), mb[j]);
        out[k++] = gl2_sub(gl2_add(gl2_from(cs), lw[3]), bc);
        out[k++] = gl2_sub(lw[2], mb[cb]);
        break;
    }
    case GLO_GATE_BASE_SUM: { /* wire 0 = sum, limbs 1..num_limbs little-endian in base p1 */
        u32 k = 0; const u32 nl = g->p0, B = g->p1;
        gl2 sum = gl2_from(0);
        for (int j = (int)nl - 1; j >= 0; j--) sum = gl2_add(gl2_scale(sum, B), lw[1 + j]);
        out[k++] = gl2_sub(sum, lw[0]);
        for (u32 j = 0; j < nl; j++) out[k++] = range_product(lw[1 + j], B);
        break;
    }
    case GLO_GATE_RANDOM_ACCESS: { /* per copy: access_index, claimed, list[2^bits]; extra constants; then bits per copy */
        u32 k = 0; const u32 bits = g->p0, copies = g->p1 & 0xFFFF, nextra = g->p1 >> 16, vs = 1u << bits;
        const u32 routed = (2 + vs) * copies + nextra;
        for (u32 c = 0; c < copies; c++) {
            const gl2 *base = lw + (2 + vs) * c, *bw = lw + routed + bits * c;
            gl2 list[64];
            for (u32 j = 0; j < vs; j++) list[j] = base[2 + j];
            gl2 idx = gl2_from(0);
            for (u32 b = 0; b < bits; b++) out[k++] = gl2_mul(bw[b], gl2_sub(bw[b], gl2_from(1)));
            for (int b = (int)bits - 1; b >= 0; b--) idx = gl2_add(gl2_scale(idx, 2), bw[b]);
            out[k++] = gl2_sub(idx, base[0]);
            u32 len = vs;
            for (u32 b = 0; b < bits; b++) {
                for (u32 j = 0; j < len / 2; j++) list[j] = gl2_add(list[2 * j], gl2_mul(bw[b], gl2_sub(list[2 * j + 1], list[2 * j])));
                len /= 2;
            }
            out[k++] = gl2_sub(list[0], base[1]);
        }
        for (u32 e = 0; e < nextra; e++) out[k++] = gl2_sub(gc[e], lw[(2 + vs) * copies + e]);
        break;
    }
    default: break;
    }
}
static void eval_gate_constraints(const glo_circuit *c, const gl2 *lc, const gl2 *lw, const u64 pih[4], gl2 *constraints) {
    gl2 tmp[512];
    for (u32 i = 0; i < c->num_gate_constraints; i++) constraints[i] = gl2_from(0);
    for (u32 gi = 0; gi < c->num_gates; gi++) {
        const glo_gate *g = &c->gates[gi];
        gl2 filter = compute_filter(g, lc[g->selector_index], c->num_selectors > 1);
        for (u32 i = 0; i < g->num_constraints; i++) tmp[i] = gl2_from(0);
        eval_gate_unfiltered(g, lc + c->num_selectors, lw, pih, tmp);
        for (u32 i = 0; i < g->num_constraints; i++) constraints[i] = gl2_add(constraints[i], gl2_mul(filter, tmp[i]));
    }
}

/* `eval_vanishing_poly`: returns one combined value per challenge.
 *   x           evaluation point;  l0x = L_0(x);  zs / zs_next / pps / sig as named in plonky2 */
static void eval_vanishing(const glo_circuit *c, gl2 x, gl2 l0x, const gl2 *lc, const gl2 *lw, const gl2 *zs,
                           const gl2 *zs_next, const gl2 *pps, const gl2 *sig, const u64 *betas, const u64 *gammas,
                           const u64 *alphas, const u64 pih[4], gl2 *out) {
    const u32 nch = c->num_challenges, nr = c->num_routed_wires, qdf = c->quotient_degree_factor, npp = c->num_partial_products;
    const u32 nchunks = npp + 1;
    u32 nterms = nch + nch * nchunks + c->num_gate_constraints;
    gl2 *terms = (gl2 *)malloc(sizeof(gl2) * nterms);
    u32 t = 0;
    for (u32 i = 0; i < nch; i++) terms[t++] = gl2_mul(l0x, gl2_sub(zs[i], gl2_from(1)));  /* vanishing_z_1_terms */
    for (u32 i = 0; i < nch; i++) { /* check_partial_products */
        for (u32 ch = 0; ch < nchunks; ch++) {
            gl2 num = gl2_from(1), den = gl2_from(1);
            for (u32 j = ch * qdf; j < (ch + 1) * qdf && j < nr; j++) {
                gl2 sid = gl2_scale(x, c->k_is[j]);
                num = gl2_mul(num, gl2_add(gl2_add(lw[j], gl2_scale(sid, betas[i])), gl2_from(gammas[i])));
                den = gl2_mul(den, gl2_add(gl2_add(lw[j], gl2_scale(sig[j], betas[i])), gl2_from(gammas[i])));
            }
            gl2 prev = ch == 0 ? zs[i] : pps[i * npp + ch - 1];
            gl2 next = ch == nchunks - 1 ? zs_next[i] : pps[i * npp + ch];
            terms[t++] = gl2_sub(gl2_mul(prev, num), gl2_mul(next, den));
        }
    }
    eval_gate_constraints(c, lc, lw, pih, terms + t);
    for (u32 i = 0; i < nch; i++) { /* reduce_with_powers_multi: sum_k terms[k] * alpha^k */
        gl2 acc = gl2_from(0);
        for (u32 k = nterms; k-- > 0;) acc = gl2_add(gl2_scale(acc, alphas[i]), terms[k]);
        out[i] = acc;
    }
    free(terms);
}

/* ------------------------------------------------------------------ proof layout (flat u64 words)
 * Field order follows plonky2's `Buffer::write_proof`; sizes are implied by the circuit.
 *   wires_cap | zs_pp_cap | quotient_cap                                  each [2^ch][4]
 *   openings: constants[nc] sigmas[nr] wires[nw] zs[nch] zs_next[nch] pps[nch*npp] quotient[nch*qdf]   ext each
 *   commit_phase_merkle_caps[num_reductions][2^ch][4]
 *   query rounds[num_queries]: 4 x (leaf[ncols_k], path[depth0][4]) ; steps: evals[arity] ext, path[depth_i][4]
 *   final_poly[len] ext | pow_witness | public_inputs[npi]  */
typedef struct {
    size_t caps, openings, fri_caps, queries, final_poly, pow, pis, total;
    size_t nopen;              /* number of opened extension values */
    size_t query_stride;
    u32 oracle_cols[4];
    u32 depth0;                /* initial trees path length */
    u32 step_depth[16], step_leaves_lg[16];
    u32 final_len;
} layout_t;

static void make_layout(const glo_circuit *c, layout_t *L) {
    const u32 cap = 1u << c->cap_height, nch = c->num_challenges;
    memset(L, 0, sizeof(*L));
    L->oracle_cols[0] = c->num_constants + c->num_routed_wires;
    L->oracle_cols[1] = c->num_wires;
    L->oracle_cols[2] = nch * (1 + c->num_partial_products);
    L->oracle_cols[3] = nch * c->quotient_degree_factor;
    L->nopen = c->num_constants + c->num_routed_wires + c->num_wires + 2 * nch + nch * c->num_partial_products + nch * c->quotient_degree_factor;
    L->caps = 0;
    L->openings = 3 * cap * 4;
    L->fri_caps = L->openings + 2 * L->nopen;
    L->queries = L->fri_caps + (size_t)c->num_reductions * cap * 4;
    u32 lgN = c->degree_bits + c->rate_bits;
    L->depth0 = lgN - c->cap_height;
    size_t q = 0;
    for (int k = 0; k < 4; k++) q += L->oracle_cols[k] + 4 * (size_t)L->depth0;
    u32 lg = lgN;
    for (u32 i = 0; i < c->num_reductions; i++) {
        u32 ab = c->reduction_arity_bits[i];
        lg -= ab;
        L->step_leaves_lg[i] = lg;
        L->step_depth[i] = lg - c->cap_height;
        q += 2 * ((size_t)1 << ab) + 4 * (size_t)L->step_depth[i];
    }
    L->query_stride = q;
    L->final_len = 1u << (lg - c->rate_bits);
    L->final_poly = L->queries + q * c->num_query_rounds;
    L->pow = L->final_poly + 2 * (size_t)L->final_len;
    L->pis = L->pow + 1;
    L->total = L->pis + c->num_public_inputs;
}
API size_t glo_proof_words(const glo_circuit *c) { layout_t L; make_layout(c, &L); return L.total; }

/* ------------------------------------------------------------------ prove */
typedef struct { u64 *coeffs, *leaves, *digests, *cap; size_t ncols; } obatch;
static void obatch_free(obatch *b) { free(b->coeffs); free(b->leaves); free(b->digests); free(b->cap); }
static int obatch_from_coeffs(obatch *b, u64 *coeffs /* takes ownership */, size_t ncols, const glo_circuit *c) {
    size_t n = (size_t)1 << c->degree_bits, N = n << c->rate_bits;
    b->ncols = ncols; b->coeffs = coeffs;
    b->leaves = (u64 *)malloc(N * ncols * 8);
    b->digests = (u64 *)malloc(glo_merkle_num_digests(N, c->cap_height) * 32);
    b->cap = (u64 *)malloc(((size_t)32) << c->cap_height);
    return glo_batch_from_coeffs(coeffs, ncols, c->degree_bits, c->rate_bits, c->cap_height, b->leaves, b->digests, b->cap);
}
static int obatch_from_values(obatch *b, const u64 *values, size_t ncols, const glo_circuit *c) {
    size_t n = (size_t)1 << c->degree_bits;
    u64 *co = (u64 *)malloc(ncols * n * 8);
    memcpy(co, values, ncols * n * 8);
    for (size_t k = 0; k < ncols; k++) glo_ifft(co + k * n, c->degree_bits);
    return obatch_from_coeffs(b, co, ncols, c);
}
/* `PolynomialBatch::get_lde_values(index, step)` */
static const u64 *lde_row(const obatch *b, size_t index, size_t step, int lgN) { return b->leaves + brev(index * step, lgN) * b->ncols; }
static gl2 eval_poly_ext(const u64 *co, size_t n, gl2 z) { /* `to_extension().eval(z)` */
    gl2 acc = gl2_from(0);
    for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, z), gl2_from(co[i]));
    return acc;
}

/* The constants+sigmas batch is built once per circuit (`build()`); expose it so tests can pass the
 * same cap to the verifier. */
API int glo_constants_sigmas_cap(const glo_circuit *c, u64 *cap_out) {
    size_t n = (size_t)1 << c->degree_bits;
    size_t nc = c->num_constants, nr = c->num_routed_wires;
    u64 *v = (u64 *)malloc((nc + nr) * n * 8);
    memcpy(v, c->constants, nc * n * 8);
    memcpy(v + nc * n, c->sigmas, nr * n * 8);
    obatch b; int rc = obatch_from_values(&b, v, nc + nr, c);
    memcpy(cap_out, b.cap, ((size_t)32) << c->cap_height);
    obatch_free(&b); free(v);
    return rc;
}

/* Returns 0 on success; -5 if the quotient is not a polynomial of the expected degree (witness does
 * not satisfy the circuit: "Quotient has failed, the vanishing polynomial is not divisible by Z_H"). */
API int glo_prove(const glo_circuit *c, const u64 *wires /* [num_wires][n] */, const u64 *public_inputs, u64 *proof) {
    const int lg = c->degree_bits, rb = c->rate_bits, lgN = lg + rb;
    const size_t n = (size_t)1 << lg, N = n << rb;
    const u32 nch = c->num_challenges, nr = c->num_routed_wires, nw = c->num_wires, nc = c->num_constants;
    const u32 qdf = c->quotient_degree_factor, npp = c->num_partial_products, capn = 1u << c->cap_height;
    layout_t L; make_layout(c, &L);
    int rc = 0;
    memset(proof, 0, L.total * 8);

    /* constants_sigmas_commitment (prover_only data, from build()) */
    obatch cs;
    {
        u64 *v = (u64 *)malloc((size_t)(nc + nr) * n * 8);
        memcpy(v, c->constants, (size_t)nc * n * 8);
        memcpy(v + (size_t)nc * n, c->sigmas, (size_t)nr * n * 8);
        rc = obatch_from_values(&cs, v, nc + nr, c);
        free(v);
        if (rc) return rc;
    }
    u64 pih[4];
    glo_hash_no_pad(public_inputs, c->num_public_inputs, pih);
    if (c->num_public_inputs) memcpy(proof + L.pis, public_inputs, (size_t)c->num_public_inputs * 8);      /* (NULL, 0) is not a valid memcpy source */

    obatch wb; obatch_from_values(&wb, wires, nw, c);
    memcpy(proof + L.caps, wb.cap, capn * 32);

    glo_challenger ch; glo_challenger_init(&ch);
    glo_challenger_observe_hashes(&ch, c->circuit_digest, 1);
    glo_challenger_observe(&ch, pih, 4);
    glo_challenger_observe_hashes(&ch, wb.cap, capn);
    u64 betas[8], gammas[8], alphas[8];
    for (u32 i = 0; i < nch; i++) betas[i] = glo_challenger_get(&ch);
    for (u32 i = 0; i < nch; i++) gammas[i] = glo_challenger_get(&ch);

    /* wires_permutation_partial_products_and_zs, per challenge; column order of the batch:
     * [Z_0..Z_{nch-1}, pp_0_0..pp_0_{npp-1}, pp_1_0 ...] */
    const u32 nzp = nch * (1 + npp);
    u64 *zp = (u64 *)malloc((size_t)nzp * n * 8);
    {
        u64 w = gl_root_of_unity(lg);
        for (u32 i = 0; i < nch; i++) {
            /* quotient_chunk_products per row (parallel over rows, as rayon does), then the running product */
            u64 *qc_ = (u64 *)malloc((size_t)(npp + 1) * n * 8);
#pragma omp parallel for schedule(static)
            for (size_t row = 0; row < n; row++) {
                u64 x = gl_pow(w, row);
                for (u32 chunk = 0; chunk <= npp; chunk++) {
                    u64 num = 1, den = 1;
                    for (u32 j = chunk * qdf; j < (chunk + 1) * qdf && j < nr; j++) {
                        u64 wv = wires[(size_t)j * n + row];
                        num = gl_mul(num, gl_add(gl_add(wv, gl_mul(betas[i], gl_mul(c->k_is[j], x))), gammas[i]));
                        den = gl_mul(den, gl_add(gl_add(wv, gl_mul(betas[i], c->sigmas[(size_t)j * n + row])), gammas[i]));
                    }
                    qc_[(size_t)chunk * n + row] = gl_mul(num, gl_inv(den));
                }
            }
            u64 z = 1;
            for (size_t row = 0; row < n; row++) {
                u64 acc = z;
                zp[(size_t)i * n + row] = z; /* Z(x) */
                for (u32 chunk = 0; chunk <= npp; chunk++) {
                    acc = gl_mul(acc, qc_[(size_t)chunk * n + row]);
                    if (chunk < npp) zp[((size_t)nch + (size_t)i * npp + chunk) * n + row] = acc;
                }
                z = acc; /* Z(g x) */
            }
            free(qc_);
        }
    }
    obatch zb; obatch_from_values(&zb, zp, nzp, c);
    free(zp);
    memcpy(proof + L.caps + capn * 4, zb.cap, capn * 32);
    glo_challenger_observe_hashes(&ch, zb.cap, capn);
    for (u32 i = 0; i < nch; i++) alphas[i] = glo_challenger_get(&ch);

    /* compute_quotient_polys: evaluate on the coset 7*<w_{n*qdf}> (qdf = 2^quotient_degree_bits <= 2^rate_bits) */
    int qdb = 0; while ((1u << qdb) < qdf) qdb++;
    const size_t M = n << qdb, step = (size_t)1 << (rb - qdb), next_step = (size_t)1 << qdb;
    u64 *qv = (u64 *)malloc((size_t)nch * M * 8);
    {
        u64 wM = gl_root_of_unity(lg + qdb);
        u64 gn = gl_pow(GL_GEN, n);              /* ZeroPolyOnCoset: Z_H(g w^i) = g^n * w_rate^i - 1 */
        u64 wr = gl_root_of_unity(qdb);
        /* rayon's par_chunks over the points in the Rust prover; OpenMP over points here */
#pragma omp parallel
        {
            gl2 *lc = (gl2 *)malloc(sizeof(gl2) * (nc + nr + nw + 4 * nzp + 16));
            gl2 *sg = lc + nc, *lw = sg + nr, *zs = lw + nw, *zn = zs + nch, *pp = zn + nch;
#pragma omp for schedule(static)
            for (size_t i = 0; i < M; i++) {
                u64 sx = gl_mul(GL_GEN, gl_pow(wM, i));
                const u64 *rc_ = lde_row(&cs, i, step, lgN);
                const u64 *rw = lde_row(&wb, i, step, lgN);
                const u64 *rz = lde_row(&zb, i, step, lgN);
                const u64 *rzn = lde_row(&zb, (i + next_step) % M, step, lgN);
                for (u32 k = 0; k < nc; k++) lc[k] = gl2_from(rc_[k]);
                for (u32 k = 0; k < nr; k++) sg[k] = gl2_from(rc_[nc + k]);
                for (u32 k = 0; k < nw; k++) lw[k] = gl2_from(rw[k]);
                for (u32 k = 0; k < nch; k++) { zs[k] = gl2_from(rz[k]); zn[k] = gl2_from(rzn[k]); }
                for (u32 k = 0; k < nch * npp; k++) pp[k] = gl2_from(rz[nch + k]);
                u64 zh = gl_sub(gl_mul(gn, gl_pow(wr, i % next_step)), 1);
                /* eval_l_0(i, x) = Z_H(x) / (n (x - 1)) */
                u64 l0 = gl_mul(zh, gl_inv(gl_mul((u64)n % GL_P, gl_sub(sx, 1))));
                gl2 res[8];
                eval_vanishing(c, gl2_from(sx), gl2_from(l0), lc, lw, zs, zn, pp, sg, betas, gammas, alphas, pih, res);
                u64 zhi = gl_inv(zh);
                for (u32 k = 0; k < nch; k++) qv[(size_t)k * M + i] = gl_mul(res[k].a[0], zhi);
            }
            free(lc);
        }
    }
    /* coset_ifft, trim_to_len(quotient_degree = qdf * n), chunks(n) */
    u64 *qc = (u64 *)malloc((size_t)nch * qdf * n * 8);
    for (u32 k = 0; k < nch; k++) {
        glo_coset_ifft(qv + (size_t)k * M, lg + qdb, GL_GEN);
        for (size_t i = (size_t)qdf * n; i < M; i++) if (qv[(size_t)k * M + i]) rc = -5;
        memcpy(qc + (size_t)k * qdf * n, qv + (size_t)k * M, (size_t)qdf * n * 8);
    }
    free(qv);
    /* a degree check that also works when M == qdf*n: re-evaluate is not needed; the polynomial
     * identity is checked by the verifier.  An unsatisfied witness makes t(X) not a polynomial of
     * degree < qdf*n on the coset, detected below by the opening check in tests. */
    obatch qb; obatch_from_coeffs(&qb, qc, (size_t)nch * qdf, c);
    memcpy(proof + L.caps + 2 * capn * 4, qb.cap, capn * 32);
    glo_challenger_observe_hashes(&ch, qb.cap, capn);

    gl2 zeta = ch_get_ext(&ch);
    gl2 g = gl2_from(gl_root_of_unity(lg));
    gl2 zeta_next = gl2_mul(g, zeta);

    /* OpeningSet::new ; order of `to_fri_openings` zeta batch = constants, sigmas, wires, zs, pps, quotient */
    u64 *op = proof + L.openings;
    {
        /* (coefficient pointer, point) per opened value, in `to_fri_openings` order; evaluated in parallel */
        size_t cnt = L.nopen, o = 0;
        const u64 **src = (const u64 **)malloc(cnt * sizeof(*src));
        gl2 *pt = (gl2 *)malloc(cnt * sizeof(gl2));
        #define PUT(ptr, z) do { src[o] = (ptr); pt[o] = (z); o++; } while (0)
        for (u32 k = 0; k < nc + nr; k++) PUT(cs.coeffs + (size_t)k * n, zeta);
        for (u32 k = 0; k < nw; k++) PUT(wb.coeffs + (size_t)k * n, zeta);
        for (u32 k = 0; k < nch; k++) PUT(zb.coeffs + (size_t)k * n, zeta);
        for (u32 k = 0; k < nch; k++) PUT(zb.coeffs + (size_t)k * n, zeta_next);
        for (u32 k = 0; k < nch * npp; k++) PUT(zb.coeffs + (size_t)(nch + k) * n, zeta);
        for (u32 k = 0; k < nch * qdf; k++) PUT(qb.coeffs + (size_t)k * n, zeta);
        #undef PUT
#pragma omp parallel for schedule(dynamic)
        for (size_t k = 0; k < cnt; k++) { gl2 e = eval_poly_ext(src[k], n, pt[k]); op[2 * k] = e.a[0]; op[2 * k + 1] = e.a[1]; }
        free(src); free(pt);
    }
    /* observe_openings: batch zeta = [constants, sigmas, wires, zs, pps, quotient]; batch zeta_next = [zs_next] */
    {
        const u64 *p_cs = op, *p_w = op + 2 * (nc + nr), *p_zs = p_w + 2 * nw, *p_zn = p_zs + 2 * nch;
        const u64 *p_pp = p_zn + 2 * nch, *p_q = p_pp + 2 * nch * npp;
        glo_challenger_observe(&ch, p_cs, 2 * (nc + nr));
        glo_challenger_observe(&ch, p_w, 2 * nw);
        glo_challenger_observe(&ch, p_zs, 2 * nch);
        glo_challenger_observe(&ch, p_pp, 2 * nch * npp);
        glo_challenger_observe(&ch, p_q, 2 * nch * qdf);
        glo_challenger_observe(&ch, p_zn, 2 * nch);
    }

    /* prove_openings */
    gl2 alpha = ch_get_ext(&ch);
    gl2 *fp = (gl2 *)calloc(N, sizeof(gl2)); /* final_poly coefficients, zero padded to N (= lde) */
    {
        const obatch *ob[4] = {&cs, &wb, &zb, &qb};
        gl2 *comp = (gl2 *)malloc(n * sizeof(gl2));
        for (int batch = 0; batch < 2; batch++) {
            gl2 point = batch == 0 ? zeta : zeta_next;
            for (size_t i = 0; i < n; i++) comp[i] = gl2_from(0);
            gl2 ap = gl2_from(1);
            u64 count = 0;
            if (batch == 0) {
                for (int k = 0; k < 4; k++)
                    for (size_t col = 0; col < ob[k]->ncols; col++) {
                        const u64 *co = ob[k]->coeffs + col * n;
                        for (size_t i = 0; i < n; i++) comp[i] = gl2_add(comp[i], gl2_scale(ap, co[i]));
                        ap = gl2_mul(ap, alpha); count++;
                    }
            } else {
                for (size_t col = 0; col < nch; col++) {
                    const u64 *co = zb.coeffs + col * n;
                    for (size_t i = 0; i < n; i++) comp[i] = gl2_add(comp[i], gl2_scale(ap, co[i]));
                    ap = gl2_mul(ap, alpha); count++;
                }
            }
            /* divide_by_linear(point): b_{k-1} = b_k * z + c_k scan from the top; drop the remainder */
            gl2 acc = gl2_from(0);
            gl2 *quot = (gl2 *)malloc(n * sizeof(gl2));
            for (size_t i = n; i-- > 0;) {
                acc = gl2_add(gl2_mul(acc, point), comp[i]);
                if (i > 0) quot[i - 1] = acc;
            }
            quot[n - 1] = gl2_from(0); /* pad back to a power of two */
            /* alpha.shift_poly(final_poly): *= alpha^count */
            gl2 sh = gl2_pow(alpha, count);
            for (size_t i = 0; i < n; i++) fp[i] = gl2_add(gl2_mul(fp[i], sh), quot[i]);
            free(quot);
        }
        free(comp);
    }
    /* lde + coset_fft over the extension = two base-field transforms */
    u64 *va = (u64 *)malloc(N * 8), *vb = (u64 *)malloc(N * 8);
    gl2 *coeffs = fp;
    size_t len = N;
    int lglen = lgN;
    u64 shift = GL_GEN;
    for (size_t i = 0; i < N; i++) { va[i] = coeffs[i].a[0]; vb[i] = coeffs[i].a[1]; }
    glo_coset_fft(va, lglen, shift); glo_coset_fft(vb, lglen, shift);

    /* fri_committed_trees */
    u64 *tree_dig[16]; u64 *tree_leaves[16];
    for (u32 r = 0; r < c->num_reductions; r++) {
        const u32 ab = c->reduction_arity_bits[r], arity = 1u << ab;
        const size_t nl = len >> ab;
        u64 *leaves = (u64 *)malloc(len * 2 * 8);
        for (size_t j = 0; j < len; j++) { size_t s = brev(j, lglen); leaves[2 * j] = va[s]; leaves[2 * j + 1] = vb[s]; }
        u64 *dig = (u64 *)malloc(glo_merkle_num_digests(nl, c->cap_height) * 32);
        u64 cap[64 * 4];
        glo_merkle_build(leaves, nl, 2 * arity, c->cap_height, dig, cap);
        memcpy(proof + L.fri_caps + (size_t)r * capn * 4, cap, capn * 32);
        glo_challenger_observe_hashes(&ch, cap, capn);
        tree_dig[r] = dig; tree_leaves[r] = leaves;
        gl2 beta = ch_get_ext(&ch);
        /* coeffs <- chunks(arity).map(reduce_with_powers(chunk, beta)) */
        for (size_t k = 0; k < nl; k++) {
            gl2 acc = gl2_from(0);
            for (u32 t = arity; t-- > 0;) acc = gl2_add(gl2_mul(acc, beta), coeffs[k * arity + t]);
            coeffs[k] = acc;
        }
        len = nl; lglen -= ab;
        shift = gl_pow(shift, arity);
        for (size_t i = 0; i < len; i++) { va[i] = coeffs[i].a[0]; vb[i] = coeffs[i].a[1]; }
        glo_coset_fft(va, lglen, shift); glo_coset_fft(vb, lglen, shift);
    }
    /* final poly: truncate the zero tail (len >> rate_bits), observe */
    for (u32 i = 0; i < L.final_len; i++) { proof[L.final_poly + 2 * i] = coeffs[i].a[0]; proof[L.final_poly + 2 * i + 1] = coeffs[i].a[1]; }
    glo_challenger_observe(&ch, proof + L.final_poly, 2 * (size_t)L.final_len);

    /* fri_proof_of_work: smallest witness whose response has >= pow_bits leading zeros */
    {
        u64 st[12]; memcpy(st, ch.st, 96);
        for (int i = 0; i < ch.nin; i++) st[i] = ch.in[i];
        int pos = ch.nin;
        u64 cand = 0;
        for (;; cand++) {
            u64 t[12]; memcpy(t, st, 96);
            t[pos] = cand;
            glo_permute(t);
            if (c->proof_of_work_bits == 0 || (t[7] >> (64 - c->proof_of_work_bits)) == 0) break;
        }
        proof[L.pow] = cand;
        glo_challenger_observe(&ch, &cand, 1);
        u64 resp = glo_challenger_get(&ch);
        if (c->proof_of_work_bits && (resp >> (64 - c->proof_of_work_bits)) != 0) rc = -6;
    }
    /* fri_prover_query_rounds */
    {
        const obatch *ob[4] = {&cs, &wb, &zb, &qb};
        for (u32 q = 0; q < c->num_query_rounds; q++) {
            size_t x_index = (size_t)(glo_challenger_get(&ch) % (u64)N);
            u64 *w = proof + L.queries + (size_t)q * L.query_stride;
            for (int k = 0; k < 4; k++) {
                memcpy(w, ob[k]->leaves + x_index * ob[k]->ncols, ob[k]->ncols * 8); w += ob[k]->ncols;
                glo_merkle_prove(ob[k]->digests, N, c->cap_height, x_index, w); w += 4 * (size_t)L.depth0;
            }
            size_t nl = N;
            for (u32 r = 0; r < c->num_reductions; r++) {
                const u32 ab = c->reduction_arity_bits[r], arity = 1u << ab;
                nl >>= ab;
                size_t li = x_index >> ab;
                memcpy(w, tree_leaves[r] + li * 2 * arity, 2 * arity * 8); w += 2 * arity;
                glo_merkle_prove(tree_dig[r], nl, c->cap_height, li, w); w += 4 * (size_t)L.step_depth[r];
                x_index = li;
            }
        }
    }
    for (u32 r = 0; r < c->num_reductions; r++) { free(tree_dig[r]); free(tree_leaves[r]); }
    free(va); free(vb); free(fp);
    obatch_free(&cs); obatch_free(&wb); obatch_free(&zb); obatch_free(&qb);
    return rc;
}

/* ------------------------------------------------------------------ verify (plonk/verifier.rs, fri/verifier.rs)
 * Returns 0 if the proof verifies, otherwise a positive code naming the failed check. */
static gl2 rd2(const u64 *p) { return gl2_make(p[0], p[1]); }

API int glo_verify(const glo_circuit *c, const u64 *constants_sigmas_cap, const u64 *proof) {
    const int lg = c->degree_bits, rb = c->rate_bits, lgN = lg + rb;
    const size_t n = (size_t)1 << lg, N = n << rb;
    const u32 nch = c->num_challenges, nr = c->num_routed_wires, nw = c->num_wires, nc = c->num_constants;
    const u32 qdf = c->quotient_degree_factor, npp = c->num_partial_products, capn = 1u << c->cap_height;
    layout_t L; make_layout(c, &L);
    for (size_t i = 0; i < L.total; i++) if (proof[i] >= GL_P) return 1; /* non-canonical element */

    u64 pih[4];
    glo_hash_no_pad(proof + L.pis, c->num_public_inputs, pih);
    glo_challenger ch; glo_challenger_init(&ch);
    glo_challenger_observe_hashes(&ch, c->circuit_digest, 1);
    glo_challenger_observe(&ch, pih, 4);
    glo_challenger_observe_hashes(&ch, proof + L.caps, capn);
    u64 betas[8], gammas[8], alphas[8];
    for (u32 i = 0; i < nch; i++) betas[i] = glo_challenger_get(&ch);
    for (u32 i = 0; i < nch; i++) gammas[i] = glo_challenger_get(&ch);
    glo_challenger_observe_hashes(&ch, proof + L.caps + capn * 4, capn);
    for (u32 i = 0; i < nch; i++) alphas[i] = glo_challenger_get(&ch);
    glo_challenger_observe_hashes(&ch, proof + L.caps + 2 * capn * 4, capn);
    gl2 zeta = ch_get_ext(&ch);

    const u64 *op = proof + L.openings;
    const u64 *p_cs = op, *p_w = op + 2 * (nc + nr), *p_zs = p_w + 2 * nw, *p_zn = p_zs + 2 * nch;
    const u64 *p_pp = p_zn + 2 * nch, *p_q = p_pp + 2 * nch * npp;
    glo_challenger_observe(&ch, p_cs, 2 * (nc + nr));
    glo_challenger_observe(&ch, p_w, 2 * nw);
    glo_challenger_observe(&ch, p_zs, 2 * nch);
    glo_challenger_observe(&ch, p_pp, 2 * nch * npp);
    glo_challenger_observe(&ch, p_q, 2 * nch * qdf);
    glo_challenger_observe(&ch, p_zn, 2 * nch);
    /* fri_challenges */
    gl2 fri_alpha = ch_get_ext(&ch);
    gl2 fri_betas[16];
    for (u32 r = 0; r < c->num_reductions; r++) {
        glo_challenger_observe_hashes(&ch, proof + L.fri_caps + (size_t)r * capn * 4, capn);
        fri_betas[r] = ch_get_ext(&ch);
    }
    glo_challenger_observe(&ch, proof + L.final_poly, 2 * (size_t)L.final_len);
    glo_challenger_observe(&ch, proof + L.pow, 1);
    u64 pow_resp = glo_challenger_get(&ch);
    if (c->proof_of_work_bits && (pow_resp >> (64 - c->proof_of_work_bits)) != 0) return 2;

    /* vanishing(zeta) == Z_H(zeta) * reduce_with_powers(quotient chunks, zeta^n) */
    {
        gl2 *lc = (gl2 *)malloc(sizeof(gl2) * (nc + nr + nw + 4 * nch * (1 + npp) + 16));
        gl2 *sg = lc + nc, *lw = sg + nr, *zs = lw + nw, *zn = zs + nch, *pp = zn + nch;
        for (u32 k = 0; k < nc; k++) lc[k] = rd2(p_cs + 2 * k);
        for (u32 k = 0; k < nr; k++) sg[k] = rd2(p_cs + 2 * (nc + k));
        for (u32 k = 0; k < nw; k++) lw[k] = rd2(p_w + 2 * k);
        for (u32 k = 0; k < nch; k++) { zs[k] = rd2(p_zs + 2 * k); zn[k] = rd2(p_zn + 2 * k); }
        for (u32 k = 0; k < nch * npp; k++) pp[k] = rd2(p_pp + 2 * k);
        gl2 zpow = zeta;
        for (int i = 0; i < lg; i++) zpow = gl2_mul(zpow, zpow);
        gl2 zh = gl2_sub(zpow, gl2_from(1));
        gl2 l0 = gl2_mul(zh, gl2_inv(gl2_scale(gl2_sub(zeta, gl2_from(1)), (u64)n % GL_P)));
        gl2 van[8];
        eval_vanishing(c, zeta, l0, lc, lw, zs, zn, pp, sg, betas, gammas, alphas, pih, van);
        free(lc);
        for (u32 i = 0; i < nch; i++) {
            gl2 acc = gl2_from(0);
            for (u32 k = qdf; k-- > 0;) acc = gl2_add(gl2_mul(acc, zpow), rd2(p_q + 2 * (i * qdf + k)));
            if (!gl2_eq(van[i], gl2_mul(zh, acc))) return 3;
        }
    }
    /* verify_fri_proof */
    gl2 g = gl2_from(gl_root_of_unity(lg));
    gl2 zeta_next = gl2_mul(g, zeta);
    /* PrecomputedReducedOpenings: reduce(batch values) = sum_j alpha^j v_j */
    gl2 red0 = gl2_from(0), red1 = gl2_from(0);
    {
        /* zeta batch order: constants, sigmas, wires, zs, pps, quotient */
        size_t tot = 0;
        const u64 *parts[5] = {p_cs, p_w, p_zs, p_pp, p_q};
        size_t lens[5] = {nc + nr, nw, nch, (size_t)nch * npp, (size_t)nch * qdf};
        gl2 ap = gl2_from(1);
        for (int k = 0; k < 5; k++)
            for (size_t j = 0; j < lens[k]; j++) { red0 = gl2_add(red0, gl2_mul(ap, rd2(parts[k] + 2 * j))); ap = gl2_mul(ap, fri_alpha); tot++; }
        ap = gl2_from(1);
        for (size_t j = 0; j < nch; j++) { red1 = gl2_add(red1, gl2_mul(ap, rd2(p_zn + 2 * j))); ap = gl2_mul(ap, fri_alpha); }
        (void)tot;
    }
    const u64 *caps4[4] = {constants_sigmas_cap, proof + L.caps, proof + L.caps + capn * 4, proof + L.caps + 2 * capn * 4};
    u64 wN = gl_root_of_unity(lgN);
    for (u32 q = 0; q < c->num_query_rounds; q++) {
        size_t x_index = (size_t)(glo_challenger_get(&ch) % (u64)N);
        const u64 *w = proof + L.queries + (size_t)q * L.query_stride;
        const u64 *evals[4];
        for (int k = 0; k < 4; k++) {
            evals[k] = w;
            if (glo_merkle_verify(w, L.oracle_cols[k], x_index, caps4[k], c->cap_height, w + L.oracle_cols[k], (int)L.depth0)) return 4;
            w += L.oracle_cols[k] + 4 * (size_t)L.depth0;
        }
        u64 sx = gl_mul(GL_GEN, gl_pow(wN, brev(x_index, lgN)));
        /* fri_combine_initial */
        gl2 sum = gl2_from(0);
        {
            /* batch 0: all polys in oracle order; batch 1: first nch polys of oracle 2 */
            gl2 r0 = gl2_from(0), ap = gl2_from(1);
            u64 cnt0 = 0;
            for (int k = 0; k < 4; k++)
                for (u32 j = 0; j < L.oracle_cols[k]; j++) { r0 = gl2_add(r0, gl2_scale(ap, evals[k][j])); ap = gl2_mul(ap, fri_alpha); cnt0++; }
            gl2 r1 = gl2_from(0); ap = gl2_from(1);
            for (u32 j = 0; j < nch; j++) { r1 = gl2_add(r1, gl2_scale(ap, evals[2][j])); ap = gl2_mul(ap, fri_alpha); }
            gl2 d0 = gl2_sub(gl2_from(sx), zeta), d1 = gl2_sub(gl2_from(sx), zeta_next);
            /* sum = alpha.shift(sum) uses the count of the batch just reduced */
            sum = gl2_mul(sum, gl2_pow(fri_alpha, cnt0));
            sum = gl2_add(sum, gl2_mul(gl2_sub(r0, red0), gl2_inv(d0)));
            sum = gl2_mul(sum, gl2_pow(fri_alpha, nch));
            sum = gl2_add(sum, gl2_mul(gl2_sub(r1, red1), gl2_inv(d1)));
        }
        gl2 old_eval = sum;
        u64 subgroup_x = sx;
        for (u32 r = 0; r < c->num_reductions; r++) {
            const u32 ab = c->reduction_arity_bits[r], arity = 1u << ab;
            const u64 *ev = w;
            const u64 *path = w + 2 * arity;
            size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
            if (!gl2_eq(rd2(ev + 2 * within), old_eval)) return 5;
            /* compute_evaluation: interpolate {(x g^i, P(x g^i))} and evaluate at beta */
            {
                u64 gA = gl_root_of_unity(ab);
                size_t rev_within = brev(within, ab);
                u64 coset_start = gl_mul(subgroup_x, gl_pow(gA, arity - rev_within));
                gl2 pts[64], vals[64];
                u64 y = 1;
                for (u32 i = 0; i < arity; i++) {
                    pts[i] = gl2_from(gl_mul(coset_start, y));
                    vals[i] = rd2(ev + 2 * brev(i, ab)); /* reverse_index_bits(evals) */
                    y = gl_mul(y, gA);
                }
                /* Lagrange interpolation at beta */
                gl2 beta = fri_betas[r], acc = gl2_from(0);
                for (u32 i = 0; i < arity; i++) {
                    gl2 num = gl2_from(1), den = gl2_from(1);
                    for (u32 j = 0; j < arity; j++) if (j != i) {
                        num = gl2_mul(num, gl2_sub(beta, pts[j]));
                        den = gl2_mul(den, gl2_sub(pts[i], pts[j]));
                    }
                    acc = gl2_add(acc, gl2_mul(vals[i], gl2_mul(num, gl2_inv(den))));
                }
                old_eval = acc;
            }
            if (glo_merkle_verify(ev, 2 * arity, coset_index, proof + L.fri_caps + (size_t)r * capn * 4, c->cap_height, path, (int)L.step_depth[r])) return 6;
            for (u32 i = 0; i < ab; i++) subgroup_x = gl_sqr(subgroup_x);
            x_index = coset_index;
            w += 2 * arity + 4 * (size_t)L.step_depth[r];
        }
        /* final_poly.eval(subgroup_x) == old_eval */
        gl2 acc = gl2_from(0);
        for (u32 i = L.final_len; i-- > 0;) acc = gl2_add(gl2_scale(acc, subgroup_x), rd2(proof + L.final_poly + 2 * i));
        if (!gl2_eq(acc, old_eval)) return 7;
    }
    return 0;
}
