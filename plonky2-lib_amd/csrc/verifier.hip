// verifier.hip -- `CircuitData::verify(proof)` for the proofs this library produces [REF src/ecdsa/gadgets/ecdsa.rs:352,
// src/zkdsa/circuits/mod.rs:346: every reference test ends in `data.verify(proof)`].
//
// Restates plonky2 0.1.4 `plonk/verifier.rs::verify_with_challenges`, `plonk/vanishing_poly.rs::eval_vanishing_poly`,
// `gates/*::eval_unfiltered` (extension-field form), `fri/verifier.rs::{verify_fri_proof, fri_combine_initial,
// compute_evaluation}` and `hash/merkle_proofs.rs::verify_merkle_proof_to_cap`.
//
// Verification is a few thousand Poseidon permutations and one constraint evaluation at a single point: it runs on the
// host, like the transcript, and touches no device memory (the circuit handle supplies the gate table, the coset
// shifts, the digest and the constants/sigmas cap).  It shares no code with the CPU checker the tests use: the two
// verifiers and the two provers are cross-checked against each other in tests/test_gpu_prove.py.
#include "prover_types.h"

namespace {

struct E {          // F_{p^2} element with operators (host only)
    ext2 v;
    E() : v(e_from(0)) {}
    E(ext2 x) : v(x) {}
    explicit E(u64 x) : v(e_from(x % P)) {}
};
inline E operator+(E a, E b) { return E(e_add(a.v, b.v)); }
inline E operator-(E a, E b) { return E(e_sub(a.v, b.v)); }
inline E operator*(E a, E b) { return E(e_mul(a.v, b.v)); }
inline E operator*(E a, u64 s) { return E(e_scale(a.v, s % P)); }
inline bool operator==(E a, E b) { return e_eq(a.v, b.v); }
inline E rd(const u64 *p) { return E(e_make(p[0], p[1])); }
inline E inv(E a) { return E(e_inv(a.v)); }
inline E epow(E a, u64 e) { return E(e_pow(a.v, e)); }
const E ONE = E((u64)1), ZERO = E((u64)0);

E range_product(E v, u32 bound) {
    E p = ONE;
    for (u32 x = 0; x < bound; x++) p = p * (v - E((u64)x));
    return p;
}
E sbox7(E x) { const E x2 = x * x, x4 = x2 * x2, x3 = x * x2; return x3 * x4; }
void mds(E s[12]) {
    static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    E o[12];
    for (int r = 0; r < 12; r++) {
        E acc = ZERO;
        for (int i = 0; i < 12; i++) acc = acc + s[(i + r) % 12] * C[i];
        if (r == 0) acc = acc + s[0] * (u64)8;
        o[r] = acc;
    }
    for (int r = 0; r < 12; r++) s[r] = o[r];
}

// unfiltered constraints of one gate at one point: w = local wires, gc = the gate's constants (selectors removed)
void gate_constraints(const glp_gate &g, const E *gc, const E *w, const u64 pih[4], std::vector<E> &out) {
    out.assign(g.num_constraints, ZERO);
    u32 k = 0;
    auto emit = [&](E v) { out.at(k++) = v; };
    switch (g.type) {
    case GLP_GATE_CONSTANT:
        for (u32 i = 0; i < g.p0; i++) emit(gc[i] - w[i]);
        break;
    case GLP_GATE_PUBLIC_INPUT:
        for (u32 i = 0; i < 4; i++) emit(w[i] - E(pih[i]));
        break;
    case GLP_GATE_ARITHMETIC:
        for (u32 i = 0; i < g.p0; i++) emit(w[4 * i + 3] - (w[4 * i] * w[4 * i + 1] * gc[0] + w[4 * i + 2] * gc[1]));
        break;
    case GLP_GATE_POSEIDON: {
        // wires: inputs 0..11, outputs 12..23, swap 24, delta 25..28, full_sbox_0 (rounds 1..3) from 29, partial_sbox from
        // 65, full_sbox_1 from 87; naive round schedule (same S-box inputs as plonky2's sparse form)
        const E swap = w[24];
        E st[12];
        emit(swap * (swap - ONE));
        for (int i = 0; i < 4; i++) emit(swap * (w[i + 4] - w[i]) - w[25 + i]);
        for (int i = 0; i < 4; i++) { st[i] = w[i] + w[25 + i]; st[i + 4] = w[i + 4] - w[25 + i]; }
        for (int i = 8; i < 12; i++) st[i] = w[i];
        int rc = 0;
        for (int r = 0; r < 4; r++) {
            for (int i = 0; i < 12; i++) st[i] = st[i] + E(pos::RC[rc + i]);
            rc += 12;
            if (r != 0) for (int i = 0; i < 12; i++) { const E in = w[29 + 12 * (r - 1) + i]; emit(st[i] - in); st[i] = in; }
            for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
            mds(st);
        }
        for (int r = 0; r < 22; r++) {
            for (int i = 0; i < 12; i++) st[i] = st[i] + E(pos::RC[rc + i]);
            rc += 12;
            const E in = w[65 + r];
            emit(st[0] - in);
            st[0] = sbox7(in);
            mds(st);
        }
        for (int r = 0; r < 4; r++) {
            for (int i = 0; i < 12; i++) st[i] = st[i] + E(pos::RC[rc + i]);
            rc += 12;
            for (int i = 0; i < 12; i++) { const E in = w[87 + 12 * r + i]; emit(st[i] - in); st[i] = in; }
            for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
            mds(st);
        }
        for (int i = 0; i < 12; i++) emit(st[i] - w[12 + i]);
        break;
    }
    case GLP_GATE_U32_INTERLEAVE:       // [REF src/u32/gates/interleave_u32.rs:84-135]
        for (u32 i = 0; i < g.p0; i++) {
            const E *bits = w + 2 * g.p0 + 32 * i;      // big-endian
            E cx = ZERO, cxi = ZERO;
            for (int b = 0; b < 32; b++) { cx = cx * (u64)2 + bits[b]; cxi = cxi * (u64)4 + bits[b]; }
            emit(cx - w[2 * i]);
            emit(cxi - w[2 * i + 1]);
            for (int b = 0; b < 32; b++) emit(bits[b] * (bits[b] - ONE));
        }
        break;
    case GLP_GATE_UNINTERLEAVE_U32:     // [REF src/u32/gates/uninterleave_to_u32.rs:93-150]
    case GLP_GATE_UNINTERLEAVE_B32:     // [REF src/u32/gates/uninterleave_to_b32.rs:95-150]
        for (u32 i = 0; i < g.p0; i++) {
            const E *bits = w + 3 * g.p0 + 64 * i;
            E cxi = ZERO, ce = ZERO, co = ZERO;
            for (int b = 0; b < 64; b++) cxi = cxi * (u64)2 + bits[b];
            for (int j = 0; j < 32; j++) {
                const u64 coeff = g.type == GLP_GATE_UNINTERLEAVE_U32 ? ((u64)1 << (31 - j)) : ((u64)1 << (2 * (31 - j)));
                ce = ce + bits[2 * j] * coeff;
                co = co + bits[2 * j + 1] * coeff;
            }
            emit(cxi - w[3 * i]);
            emit(ce - w[3 * i + 1]);
            emit(co - w[3 * i + 2]);
            for (int b = 0; b < 64; b++) emit(bits[b] * (bits[b] - ONE));
        }
        break;
    case GLP_GATE_U32_ARITHMETIC: {     // per op: m0, m1, addend, out_lo, out_hi, inverse; then 32 base-4 limbs per op
        const u32 n = g.p0;
        for (u32 i = 0; i < n; i++) {
            const E m0 = w[6 * i], m1 = w[6 * i + 1], ad = w[6 * i + 2], lo = w[6 * i + 3], hi = w[6 * i + 4], iv = w[6 * i + 5];
            emit((iv * (E((u64)0xFFFFFFFFull) - hi) - ONE) * lo);
            emit(hi * ((u64)1 << 32) + lo - (m0 * m1 + ad));
            E cl = ZERO, ch = ZERO;
            const E *limbs = w + 6 * n + 32 * i;
            for (int j = 31; j >= 0; j--) {
                emit(range_product(limbs[j], 4));
                if (j < 16) cl = cl * (u64)4 + limbs[j]; else ch = ch * (u64)4 + limbs[j];
            }
            emit(cl - lo);
            emit(ch - hi);
        }
        break;
    }
    case GLP_GATE_U32_ADD_MANY: {       // per op: addends, carry_in, result, carry_out; then 16 + 2 base-4 limbs per op
        const u32 na = g.p0, n = g.p1, wd = na + 3;
        for (u32 i = 0; i < n; i++) {
            E sum = w[wd * i + na];
            for (u32 j = 0; j < na; j++) sum = sum + w[wd * i + j];
            const E res = w[wd * i + na + 1], car = w[wd * i + na + 2];
            emit(car * ((u64)1 << 32) + res - sum);
            E cr = ZERO, cc = ZERO;
            const E *limbs = w + wd * n + 18 * i;
            for (int j = 17; j >= 0; j--) {
                emit(range_product(limbs[j], 4));
                if (j < 16) cr = cr * (u64)4 + limbs[j]; else cc = cc * (u64)4 + limbs[j];
            }
            emit(cr - res);
            emit(cc - car);
        }
        break;
    }
    case GLP_GATE_U32_SUBTRACTION: {    // per op: x, y, borrow_in, result, borrow_out; then 16 base-4 limbs per op
        const u32 n = g.p0;
        for (u32 i = 0; i < n; i++) {
            const E x = w[5 * i], y = w[5 * i + 1], bi = w[5 * i + 2], res = w[5 * i + 3], bo = w[5 * i + 4];
            emit(res - (x - y - bi + bo * ((u64)1 << 32)));
            E cl = ZERO;
            const E *limbs = w + 5 * n + 16 * i;
            for (int j = 15; j >= 0; j--) { emit(range_product(limbs[j], 4)); cl = cl * (u64)4 + limbs[j]; }
            emit(cl - res);
            emit(bo * (ONE - bo));
        }
        break;
    }
    case GLP_GATE_U32_RANGE_CHECK: {    // inputs 0..n-1, then 16 base-4 limbs per input (little-endian)
        const u32 n = g.p0;
        for (u32 i = 0; i < n; i++) {
            const E *aux = w + n + 16 * i;
            E sum = ZERO;
            for (int j = 15; j >= 0; j--) sum = sum * (u64)4 + aux[j];
            emit(sum - w[i]);
            for (int j = 0; j < 16; j++) emit(range_product(aux[j], 4));
        }
        break;
    }
    case GLP_GATE_COMPARISON: {
        const u32 nb = g.p0, ncx = g.p1, cb = (nb + ncx - 1) / ncx, cs = 1u << cb;
        const E *a = w + 4, *b = a + ncx, *ed = b + ncx, *ce = ed + ncx, *iv = ce + ncx, *mb = iv + ncx;
        E ca = ZERO, cbv = ZERO;
        for (int i = (int)ncx - 1; i >= 0; i--) { ca = ca * (u64)cs + a[i]; cbv = cbv * (u64)cs + b[i]; }
        emit(ca - w[0]);
        emit(cbv - w[1]);
        E msd = ZERO;
        for (u32 i = 0; i < ncx; i++) {
            emit(range_product(a[i], cs));
            emit(range_product(b[i], cs));
            const E diff = b[i] - a[i];
            emit(diff * ed[i] - (ONE - ce[i]));
            emit(ce[i] * diff);
            emit(iv[i] - ce[i] * msd);
            msd = iv[i] + (ONE - ce[i]) * diff;
        }
        emit(w[3] - msd);
        E bc = ZERO;
        for (u32 j = 0; j <= cb; j++) emit(mb[j] * (ONE - mb[j]));
        for (int j = (int)cb; j >= 0; j--) bc = bc * (u64)2 + mb[j];
        emit(E((u64)cs) + w[3] - bc);
        emit(w[2] - mb[cb]);
        break;
    }
    case GLP_GATE_BASE_SUM: {
        const u32 nl = g.p0, B = g.p1;
        E sum = ZERO;
        for (int j = (int)nl - 1; j >= 0; j--) sum = sum * (u64)B + w[1 + j];
        emit(sum - w[0]);
        for (u32 j = 0; j < nl; j++) emit(range_product(w[1 + j], B));
        break;
    }
    case GLP_GATE_RANDOM_ACCESS: {
        const u32 bits = g.p0, copies = g.p1 & 0xFFFF, nextra = g.p1 >> 16, vs = 1u << bits;
        const u32 routed = (2 + vs) * copies + nextra;
        for (u32 c = 0; c < copies; c++) {
            const E *base = w + (2 + vs) * c, *bw = w + routed + bits * c;
            std::vector<E> list(base + 2, base + 2 + vs);
            E idx = ZERO;
            for (u32 b = 0; b < bits; b++) emit(bw[b] * (bw[b] - ONE));
            for (int b = (int)bits - 1; b >= 0; b--) idx = idx * (u64)2 + bw[b];
            emit(idx - base[0]);
            u32 len = vs;
            for (u32 b = 0; b < bits; b++) {
                for (u32 j = 0; j < len / 2; j++) list[j] = list[2 * j] + bw[b] * (list[2 * j + 1] - list[2 * j]);
                len /= 2;
            }
            emit(list[0] - base[1]);
        }
        for (u32 e = 0; e < nextra; e++) emit(gc[e] - w[(2 + vs) * copies + e]);
        break;
    }
    default: break;    // NoopGate
    }
}

size_t brev(size_t x, int bits) {
    size_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

// hash/merkle_proofs.rs verify_merkle_proof_to_cap
bool merkle_ok(int hasher, const u64 *leaf, size_t ncols, size_t index, const u64 *cap, const u64 *path, u32 depth) {
    u64 cur[4] = {0, 0, 0, 0};
    const bool kec25 = hasher == GLP_HASH_KECCAK25;
    if (kec25) kec::host_hash_or_noop(leaf, ncols, cur);
    else if (ncols <= 4) for (size_t i = 0; i < ncols; i++) cur[i] = leaf[i];       // hash_or_noop
    else host_hash_no_pad(leaf, ncols, cur);
    for (u32 d = 0; d < depth; d++) {
        u64 nxt[4];
        if (kec25) { if (index & 1) kec::two_to_one(path + 4 * d, cur, nxt); else kec::two_to_one(cur, path + 4 * d, nxt); }
        else if (index & 1) pos::two_to_one(path + 4 * d, cur, nxt); else pos::two_to_one(cur, path + 4 * d, nxt);
        memcpy(cur, nxt, 32);
        index >>= 1;
    }
    return memcmp(cur, cap + 4 * index, 32) == 0;
}

int verify_impl(const glp_circuit *cc, const u64 *proof) {
    const glp_circuit_desc &d = cc->d;
    const Layout &L = cc->L;
    const int lg = (int)d.degree_bits, rb = (int)d.rate_bits, lgN = lg + rb;
    const size_t n = (size_t)1 << lg, N = n << rb;
    const u32 nch = d.num_challenges, nr = d.num_routed_wires, nw = d.num_wires, nc = d.num_constants;
    const u32 qdf = d.quotient_degree_factor, npp = d.num_partial_products, capn = 1u << d.cap_height;
    const u32 nchunks = npp + 1;
#define FAIL(...) return set_error(GLP_ERR_PROVE, __VA_ARGS__)
    {   // field elements (and Poseidon digests) must be canonical; a KeccakHash<25> digest is 25 bytes in a 4-word slot
        const bool kec25 = d.hasher == GLP_HASH_KECCAK25;
        const size_t dig_lo[2] = {0, L.fri_caps}, dig_hi[2] = {L.openings, L.queries};        // cap regions
        auto in_caps = [&](size_t i) { return (i >= dig_lo[0] && i < dig_hi[0]) || (i >= dig_lo[1] && i < dig_hi[1]); };
        std::vector<unsigned char> is_dig(kec25 ? L.total : 0, 0);
        if (kec25) {
            for (size_t i = 0; i < L.queries; i++) is_dig[i] = in_caps(i);
            for (u32 q = 0; q < d.num_query_rounds; q++) {
                size_t o = L.queries + (size_t)q * L.query_stride;
                for (int k = 0; k < 4; k++) { o += L.oracle_cols[k]; for (size_t j = 0; j < 4 * (size_t)L.depth0; j++) is_dig[o + j] = 1; o += 4 * (size_t)L.depth0; }
                for (u32 r = 0; r < d.num_reductions; r++) {
                    o += (size_t)2 << d.reduction_arity_bits[r];
                    for (size_t j = 0; j < 4 * (size_t)L.step_depth[r]; j++) is_dig[o + j] = 1;
                    o += 4 * (size_t)L.step_depth[r];
                }
            }
        }
        size_t run = 0;
        for (size_t i = 0; i < L.total; i++) {
            if (kec25 && is_dig[i]) { if ((run & 3) == 3 && proof[i] > 0xFF) FAIL("digest at word %zu is longer than 25 bytes", i - 3); run++; continue; }
            run = 0;
            if (proof[i] >= P) FAIL("proof word %zu is not a canonical field element", i);
        }
    }

    // ---- transcript (plonk/get_challenges.rs)
    u64 pih[4];
    host_hash_no_pad(proof + L.pis, d.num_public_inputs, pih);
    Challenger ch((int)d.hasher);
    ch.observe_hashes(cc->digest, 1);
    ch.observe(pih, 4);
    ch.observe_hashes(proof + L.caps, capn);
    u64 betas[MAXCH], gammas[MAXCH], alphas[MAXCH];
    for (u32 i = 0; i < nch; i++) betas[i] = ch.get();
    for (u32 i = 0; i < nch; i++) gammas[i] = ch.get();
    ch.observe_hashes(proof + L.caps + capn * 4, capn);
    for (u32 i = 0; i < nch; i++) alphas[i] = ch.get();
    ch.observe_hashes(proof + L.caps + 2 * capn * 4, capn);
    const E zeta(ch.get_ext());
    const u64 *op = proof + L.openings;
    const u64 *p_cs = op, *p_w = op + 2 * (nc + nr), *p_zs = p_w + 2 * nw, *p_zn = p_zs + 2 * nch;
    const u64 *p_pp = p_zn + 2 * nch, *p_q = p_pp + 2 * nch * npp;
    ch.observe(p_cs, 2 * (nc + nr)); ch.observe(p_w, 2 * nw); ch.observe(p_zs, 2 * nch);
    ch.observe(p_pp, 2 * (size_t)nch * npp); ch.observe(p_q, 2 * (size_t)nch * qdf); ch.observe(p_zn, 2 * nch);
    const E fri_alpha(ch.get_ext());
    std::vector<E> fri_betas(d.num_reductions);
    for (u32 r = 0; r < d.num_reductions; r++) {
        ch.observe_hashes(proof + L.fri_caps + (size_t)r * capn * 4, capn);
        fri_betas[r] = E(ch.get_ext());
    }
    ch.observe(proof + L.final_poly, 2 * (size_t)L.final_len);
    ch.observe(proof + L.pow, 1);
    const u64 pow_resp = ch.get();
    if (d.proof_of_work_bits && (pow_resp >> (64 - d.proof_of_work_bits)) != 0) FAIL("Invalid proof of work witness.");

    // ---- vanishing(zeta) == Z_H(zeta) * reduce_with_powers(quotient chunks, zeta^n)     (plonk/verifier.rs)
    {
        std::vector<E> lc(nc), sg(nr), lw(nw), zs(nch), zn(nch), pp((size_t)nch * npp);
        for (u32 k = 0; k < nc; k++) lc[k] = rd(p_cs + 2 * k);
        for (u32 k = 0; k < nr; k++) sg[k] = rd(p_cs + 2 * (nc + k));
        for (u32 k = 0; k < nw; k++) lw[k] = rd(p_w + 2 * k);
        for (u32 k = 0; k < nch; k++) { zs[k] = rd(p_zs + 2 * k); zn[k] = rd(p_zn + 2 * k); }
        for (u32 k = 0; k < nch * npp; k++) pp[k] = rd(p_pp + 2 * k);
        E zpow = zeta;
        for (int i = 0; i < lg; i++) zpow = zpow * zpow;
        const E zh = zpow - ONE;
        if (zeta == ONE) FAIL("zeta = 1");
        const E l0 = zh * inv((zeta - ONE) * ((u64)n % P));
        std::vector<E> terms;
        terms.reserve(nch + nch * nchunks + d.num_gate_constraints);
        for (u32 i = 0; i < nch; i++) terms.push_back(l0 * (zs[i] - ONE));
        for (u32 i = 0; i < nch; i++)
            for (u32 c = 0; c < nchunks; c++) {
                E num = ONE, den = ONE;
                for (u32 j = c * qdf; j < (c + 1) * qdf && j < nr; j++) {
                    num = num * (lw[j] + zeta * cc->k_is[j] * betas[i] + E(gammas[i]));
                    den = den * (lw[j] + sg[j] * betas[i] + E(gammas[i]));
                }
                const E prev = c == 0 ? zs[i] : pp[i * npp + c - 1];
                const E next = c == nchunks - 1 ? zn[i] : pp[i * npp + c];
                terms.push_back(prev * num - next * den);
            }
        std::vector<E> gate_terms(d.num_gate_constraints, ZERO), tmp;
        for (u32 gi = 0; gi < d.num_gates; gi++) {
            const glp_gate &g = cc->gates[gi];
            const E s = lc[g.selector_index];
            E filter = ONE;
            for (u32 i = g.group_start; i < g.group_end; i++)
                if (i != g.row) filter = filter * (E((u64)i) - s);
            if (d.num_selectors > 1) filter = filter * (E((u64)0xFFFFFFFFull) - s);       // UNUSED_SELECTOR
            gate_constraints(g, lc.data() + d.num_selectors, lw.data(), pih, tmp);
            for (u32 i = 0; i < g.num_constraints; i++) gate_terms[i] = gate_terms[i] + filter * tmp[i];
        }
        terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
        for (u32 i = 0; i < nch; i++) {
            E van = ZERO;
            for (size_t k = terms.size(); k-- > 0;) van = van * alphas[i] + terms[k];
            E t = ZERO;
            for (u32 k = qdf; k-- > 0;) t = t * zpow + rd(p_q + 2 * (i * qdf + k));
            if (!(van == zh * t)) FAIL("Mismatch between evaluation and opening of quotient polynomial (challenge %u)", i);
        }
    }

    // ---- FRI (fri/verifier.rs)
    const E zeta_next = zeta * root_of_unity(lg);
    E red0 = ZERO, red1 = ZERO;            // PrecomputedReducedOpenings
    {
        const u64 *parts[5] = {p_cs, p_w, p_zs, p_pp, p_q};
        const size_t lens[5] = {(size_t)nc + nr, nw, nch, (size_t)nch * npp, (size_t)nch * qdf};
        E ap = ONE;
        for (int k = 0; k < 5; k++)
            for (size_t j = 0; j < lens[k]; j++) { red0 = red0 + ap * rd(parts[k] + 2 * j); ap = ap * fri_alpha; }
        ap = ONE;
        for (size_t j = 0; j < nch; j++) { red1 = red1 + ap * rd(p_zn + 2 * j); ap = ap * fri_alpha; }
    }
    const u64 *caps4[4] = {cc->cs_cap.data(), proof + L.caps, proof + L.caps + capn * 4, proof + L.caps + 2 * capn * 4};
    size_t cols_total = 0;
    for (int k = 0; k < 4; k++) cols_total += L.oracle_cols[k];
    const E shift0 = epow(fri_alpha, cols_total), shift1 = epow(fri_alpha, nch);
    const u64 wN = root_of_unity(lgN);
    for (u32 q = 0; q < d.num_query_rounds; q++) {
        size_t x_index = (size_t)(ch.get() % (u64)N);
        const u64 *w = proof + L.queries + (size_t)q * L.query_stride;
        const u64 *evals[4];
        for (int k = 0; k < 4; k++) {
            evals[k] = w;
            if (!merkle_ok((int)d.hasher, w, L.oracle_cols[k], x_index, caps4[k], w + L.oracle_cols[k], L.depth0))
                FAIL("Invalid Merkle proof (query %u, initial tree %d)", q, k);
            w += L.oracle_cols[k] + 4 * (size_t)L.depth0;
        }
        u64 subgroup_x = mul(GEN, pow(wN, (u64)brev(x_index, lgN)));
        E old_eval;
        {   // fri_combine_initial: batch 0 = every polynomial at zeta, batch 1 = the Z's at g zeta
            E r0 = ZERO, ap = ONE;
            for (int k = 0; k < 4; k++)
                for (u32 j = 0; j < L.oracle_cols[k]; j++) { r0 = r0 + ap * evals[k][j]; ap = ap * fri_alpha; }
            E r1 = ZERO;
            ap = ONE;
            for (u32 j = 0; j < nch; j++) { r1 = r1 + ap * evals[2][j]; ap = ap * fri_alpha; }
            const E sx((u64)subgroup_x);
            if (sx == zeta || sx == zeta_next) FAIL("query point equals an opening point");
            E sum = ZERO;
            sum = sum * shift0 + (r0 - red0) * inv(sx - zeta);
            sum = sum * shift1 + (r1 - red1) * inv(sx - zeta_next);
            old_eval = sum;
        }
        for (u32 r = 0; r < d.num_reductions; r++) {
            const u32 ab = d.reduction_arity_bits[r], arity = 1u << ab;
            const u64 *ev = w, *path = w + 2 * arity;
            const size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
            if (!(rd(ev + 2 * within) == old_eval)) FAIL("FRI consistency check failed (query %u, reduction %u)", q, r);
            {   // compute_evaluation: interpolate the coset and evaluate at beta
                const u64 gA = root_of_unity((int)ab);
                const u64 coset_start = mul(subgroup_x, pow(gA, (u64)(arity - brev(within, (int)ab))));
                std::vector<E> pts(arity), vals(arity);
                u64 y = 1;
                for (u32 i = 0; i < arity; i++) {
                    pts[i] = E(mul(coset_start, y));
                    vals[i] = rd(ev + 2 * brev(i, (int)ab));
                    y = mul(y, gA);
                }
                E acc = ZERO;
                for (u32 i = 0; i < arity; i++) {
                    E num = ONE, den = ONE;
                    for (u32 j = 0; j < arity; j++)
                        if (j != i) { num = num * (fri_betas[r] - pts[j]); den = den * (pts[i] - pts[j]); }
                    acc = acc + vals[i] * num * inv(den);
                }
                old_eval = acc;
            }
            if (!merkle_ok((int)d.hasher, ev, 2 * (size_t)arity, coset_index, proof + L.fri_caps + (size_t)r * capn * 4, path, L.step_depth[r]))
                FAIL("Invalid Merkle proof (query %u, reduction %u)", q, r);
            for (u32 i = 0; i < ab; i++) subgroup_x = sqr(subgroup_x);
            x_index = coset_index;
            w += 2 * (size_t)arity + 4 * (size_t)L.step_depth[r];
        }
        E acc = ZERO;       // final_poly.eval(subgroup_x)
        for (u32 i = L.final_len; i-- > 0;) acc = acc * (u64)subgroup_x + rd(proof + L.final_poly + 2 * (size_t)i);
        if (!(acc == old_eval)) FAIL("Final polynomial evaluation is invalid (query %u)", q);
    }
#undef FAIL
    return GLP_OK;
}
}  // namespace

extern "C" int glp_verify(const glp_circuit *cc, const uint64_t *proof_words) {
    GLP_REQUIRE(cc && proof_words, "null argument");
    return verify_impl(cc, proof_words);
}
// Same, with the caller's buffer length stated: a truncated or over-long buffer is an argument error, never a read past it.
extern "C" int glp_verify_n(const glp_circuit *cc, const uint64_t *proof_words, size_t num_words) {
    GLP_REQUIRE(cc && proof_words, "null argument");
    GLP_REQUIRE(num_words == cc->L.total, "proof has %zu words, a proof of this circuit has %zu", num_words, (size_t)cc->L.total);
    return verify_impl(cc, proof_words);
}
