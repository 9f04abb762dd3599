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
#include <chrono>
#include <string>
#include <thread>
#include "merkle.h"
#include "prover_types.h"
#include "transcript_dev.h"

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
void mds(E s[12]) {     // circulant + diagonal, entries < 64: both coordinates accumulate unreduced in 128 bits, one reduction per row and coordinate
    static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    E o[12];
    for (int r = 0; r < 12; r++) {
        unsigned __int128 a = 0, b = 0;
        for (int i = 0; i < 12; i++) {
            const ext2 &x = s[(i + r) % 12].v;
            a += (unsigned __int128)x.a * C[i];
            b += (unsigned __int128)x.b * C[i];
        }
        if (r == 0) { a += (unsigned __int128)s[0].v.a * 8; b += (unsigned __int128)s[0].v.b * 8; }
        o[r] = E(e_make(reduce128((u64)a, (u64)(a >> 64)), reduce128((u64)b, (u64)(b >> 64))));
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

// What the transcript yields for the FRI half of the verification (fri/verifier.rs): challenges, the reduced openings and the
// query indices.  verify_front fills it (canonical-form check, plonk/get_challenges.rs, proof of work, the vanishing-polynomial
// identity at zeta); the query rounds then run on the host (verify_fri_host: glp_verify) or on the device (glp_verify_batch).
struct VChal {
    E zeta, zeta_next, fri_alpha, red0, red1, shift1;
    std::vector<E> fri_betas;
    std::vector<u64> x_index;
};
#define FAIL(...) return set_error(GLP_ERR_PROVE, __VA_ARGS__)
// field elements (and Poseidon digests) must be canonical; a KeccakHash<25> digest is 25 bytes in a 4-word slot
int verify_canonical(const glp_circuit *cc, const u64 *proof) {
    const glp_circuit_desc &d = cc->d;
    const Layout &L = cc->L;
    {
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
    return GLP_OK;
}
// vanishing(zeta) == Z_H(zeta) * reduce_with_powers(quotient chunks, zeta^n)     (plonk/verifier.rs), given the challenges
int verify_vanishing(const glp_circuit *cc, const u64 *proof, const u64 *betas, const u64 *gammas, const u64 *alphas, const E &zeta, const u64 pih[4]);

int verify_vanishing(const glp_circuit *cc, const u64 *proof, const u64 *betas, const u64 *gammas, const u64 *alphas, const E &zeta, const u64 pih[4]) {
    const glp_circuit_desc &d = cc->d;
    const Layout &L = cc->L;
    const int lg = (int)d.degree_bits;
    const size_t n = (size_t)1 << lg;
    const u32 nch = d.num_challenges, nr = d.num_routed_wires, nw = d.num_wires, nc = d.num_constants;
    const u32 qdf = d.quotient_degree_factor, npp = d.num_partial_products, nchunks = npp + 1;
    const u64 *op = proof + L.openings;
    const u64 *p_cs = op, *p_w = op + 2 * (nc + nr), *p_zs = p_w + 2 * nw, *p_zn = p_zs + 2 * nch;
    const u64 *p_pp = p_zn + 2 * nch, *p_q = p_pp + 2 * nch * npp;
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
    return GLP_OK;
}

int verify_front(const glp_circuit *cc, const u64 *proof, VChal &vc) {
    const glp_circuit_desc &d = cc->d;
    const Layout &L = cc->L;
    const int lg = (int)d.degree_bits, rb = (int)d.rate_bits, lgN = lg + rb;
    const size_t n = (size_t)1 << lg, N = n << rb;
    const u32 nch = d.num_challenges, nr = d.num_routed_wires, nw = d.num_wires, nc = d.num_constants;
    const u32 qdf = d.quotient_degree_factor, npp = d.num_partial_products, capn = 1u << d.cap_height;
    (void)n;
    {
        const int rc0 = verify_canonical(cc, proof);
        if (rc0 != GLP_OK) return rc0;
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
    {
        const int rcv = verify_vanishing(cc, proof, betas, gammas, alphas, zeta, pih);
        if (rcv != GLP_OK) return rcv;
    }

    // ---- FRI (fri/verifier.rs): reduced openings and query indices
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
    vc.zeta = zeta; vc.zeta_next = zeta_next; vc.fri_alpha = fri_alpha; vc.red0 = red0; vc.red1 = red1;
    vc.shift1 = epow(fri_alpha, nch);
    vc.fri_betas = fri_betas;
    vc.x_index.resize(d.num_query_rounds);
    for (u32 q = 0; q < d.num_query_rounds; q++) vc.x_index[q] = ch.get() % (u64)N;
    (void)lgN;
#undef FAIL
    return GLP_OK;
}

// messages of the query rounds, shared by the host and the device path (code: see k_verify_queries)
int query_failure(u32 q, u32 code, u32 nred) {
    if (code >= 1 && code <= 4) return set_error(GLP_ERR_PROVE, "Invalid Merkle proof (query %u, initial tree %d)", q, (int)code - 1);
    if (code == 5) return set_error(GLP_ERR_PROVE, "query point equals an opening point");
    if (code >= 6 && code < 6 + 2 * nred) {
        const u32 r = (code - 6) / 2;
        if ((code - 6) % 2 == 0) return set_error(GLP_ERR_PROVE, "FRI consistency check failed (query %u, reduction %u)", q, r);
        return set_error(GLP_ERR_PROVE, "Invalid Merkle proof (query %u, reduction %u)", q, r);
    }
    return set_error(GLP_ERR_PROVE, "Final polynomial evaluation is invalid (query %u)", q);
}

int verify_fri_host(const glp_circuit *cc, const u64 *proof, const VChal &vc) {
    const glp_circuit_desc &d = cc->d;
    const Layout &L = cc->L;
    const int lg = (int)d.degree_bits, rb = (int)d.rate_bits, lgN = lg + rb;
    const u32 nch = d.num_challenges, capn = 1u << d.cap_height, nred = d.num_reductions;
    const E zeta = vc.zeta, zeta_next = vc.zeta_next, fri_alpha = vc.fri_alpha, red0 = vc.red0, red1 = vc.red1, shift1 = vc.shift1;
    const std::vector<E> &fri_betas = vc.fri_betas;
#define FAIL(CODE) return query_failure(q, (CODE), nred)
    const u64 *caps4[4] = {cc->cs_cap.data(), proof + L.caps, proof + L.caps + capn * 4, proof + L.caps + 2 * capn * 4};
    size_t cols_total = 0;
    for (int k = 0; k < 4; k++) cols_total += L.oracle_cols[k];
    const E shift0 = epow(fri_alpha, cols_total);
    const u64 wN = root_of_unity(lgN);
    for (u32 q = 0; q < d.num_query_rounds; q++) {
        size_t x_index = (size_t)vc.x_index[q];
        const u64 *w = proof + L.queries + (size_t)q * L.query_stride;
        const u64 *evals[4];
        for (int k = 0; k < 4; k++) {
            evals[k] = w;
            if (!merkle_ok((int)d.hasher, w, L.oracle_cols[k], x_index, caps4[k], w + L.oracle_cols[k], L.depth0))
                FAIL(1 + (u32)k);
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
            if (sx == zeta || sx == zeta_next) FAIL(5);
            E sum = ZERO;
            sum = sum * shift0 + (r0 - red0) * inv(sx - zeta);
            sum = sum * shift1 + (r1 - red1) * inv(sx - zeta_next);
            old_eval = sum;
        }
        for (u32 r = 0; r < d.num_reductions; r++) {
            const u32 ab = d.reduction_arity_bits[r], arity = 1u << ab;
            const u64 *ev = w, *path = w + 2 * arity;
            const size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
            if (!(rd(ev + 2 * within) == old_eval)) FAIL(6 + 2 * r);
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
                FAIL(7 + 2 * r);
            for (u32 i = 0; i < ab; i++) subgroup_x = sqr(subgroup_x);
            x_index = coset_index;
            w += 2 * (size_t)arity + 4 * (size_t)L.step_depth[r];
        }
        E acc = ZERO;       // final_poly.eval(subgroup_x)
        for (u32 i = L.final_len; i-- > 0;) acc = acc * (u64)subgroup_x + rd(proof + L.final_poly + 2 * (size_t)i);
        if (!(acc == old_eval)) FAIL(6 + 2 * nred);
    }
#undef FAIL
    return GLP_OK;
}
int verify_impl(const glp_circuit *cc, const u64 *proof) {
    VChal vc;
    GLP_TRY(verify_front(cc, proof, vc));
    return verify_fri_host(cc, proof, vc);
}
}  // namespace

// ------------------------------------------------------------------------------------------ query rounds on the device
// glp_verify_batch (a GPU verifier for batch self-checking; every reference driver proves and then
// verifies [REF src/zkdsa/circuits/mod.rs:341-347, src/ecdsa/gadgets/ecdsa.rs:349-352]).  Per proof the transcript, the proof of
// work and the one constraint evaluation at zeta stay on host threads (verify_front: a few dozen permutations); the query
// rounds -- K x num_query_rounds independent checks, each a few dozen to a few hundred permutations (Merkle paths of the four
// initial oracles and of every FRI layer), `fri_combine_initial`, the arity-2^k interpolations and the final polynomial -- are
// one launch: a 16-lane group per (proof, query), hashes in the 12-lane cooperative form (poseidon.h permute_coop: ~5x lower
// latency per hash than one state per lane, which is what matters for a chain of dependent hashes), the alpha-combination and
// the Lagrange terms spread over the lanes of the group.
namespace {
constexpr u32 VC_ALPHA = 0, VC_ZETA = 2, VC_ZETA_NEXT = 4, VC_RED0 = 6, VC_RED1 = 8, VC_SHIFT1 = 10, VC_BETAS = 12, VC_XIDX = 44;
struct VQArgs {
    const u64 *proofs;      // [K][total]
    const u64 *vchal;       // [K][vstride]: fri_alpha, zeta, zeta_next, red0, red1, alpha^nch, betas[16] (ext each), x_index[nq]
    const u64 *cs_cap;      // [2^cap_height][4]
    u32 *status;            // [K][nq]: 0 = accepted, else the code query_failure() turns into the verifier's message
    size_t total, caps, fri_caps, queries, query_stride, final_poly;
    u32 vstride, nq, K, nch, lgN, cap_height, depth0, nred, final_len;
    u32 oracle_cols[4], ab[16], step_depth[16];
    u64 wN, gA[16];         // root_of_unity(lgN), root_of_unity(ab[r])
};
using pos::shfl64;
using pos::shfl_xor64;
__device__ __forceinline__ ext2 group_sum(ext2 v) {         // sum over the 16 lanes of a group, result on every lane
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v = e_add(v, e_make(shfl_xor64(v.a, m), shfl_xor64(v.b, m)));
    return v;
}
__device__ __forceinline__ u32 group_or(u32 v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v |= (u32)__shfl_xor((int)v, m, 64);
    return v;
}
__device__ __forceinline__ u64 dvpow(u64 b, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = mul(r, b); b = sqr(b); e >>= 1; }
    return r;
}
// verify_merkle_proof_to_cap on one 16-lane group: true (on every lane of the group) if the leaf does NOT hash to the cap entry.
// Every lane of the wave must call it (shuffles); len, depth are the same for all groups of a launch.
template <int HASHER>
__device__ __forceinline__ bool merkle_bad(const u64 *leaf, u32 len, size_t index, const u64 *path, u32 depth, const u64 *cap, int l, int gb) {
    u32 bad = 0;
    if (HASHER == GLP_HASH_KECCAK25) {
        if (l == 0) {                                     // Keccak is 64-bit logic at full rate: one lane walks the path
            u64 cur[4] = {0, 0, 0, 0};
            if (8 * len <= 25) { for (u32 i = 0; i < len; i++) cur[i] = leaf[i]; }
            else {
                kec::Sponge sp;
                kec::sponge_init(sp);
                for (u32 i = 0; i < len; i++) kec::sponge_absorb(sp, leaf[i]);
                kec::sponge_finish(sp);
                kec::sponge_digest25(sp, cur);
            }
            for (u32 dd = 0; dd < depth; dd++) {
                u64 nxt[4];
                if (index & 1) kec::two_to_one(path + 4 * dd, cur, nxt); else kec::two_to_one(cur, path + 4 * dd, nxt);
                cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2]; cur[3] = nxt[3];
                index >>= 1;
            }
            for (int i = 0; i < 4; i++) bad |= cur[i] != cap[4 * index + i];
        }
    } else {
        u64 x = 0;
        if (len <= 4) x = (u32)l < len ? leaf[l] : 0;     // hash_or_noop: copied, zero padded
        else
            for (u32 c0 = 0; c0 < len; c0 += 8) {         // sponge, overwrite mode: lanes past the chunk keep their state
                if (l < 8 && c0 + (u32)l < len) x = leaf[c0 + l];
                x = pos::permute_coop(x, l, gb);
            }
        for (u32 dd = 0; dd < depth; dd++) {              // two_to_one(left, right) = permute(left || right || 0000)[0..4]
            const u64 sib = path[4 * dd + (l & 3)];
            const u64 cur = shfl64(x, gb + (l & 3));
            const bool right = index & 1;
            u64 nx = 0;
            if (l < 4) nx = right ? sib : cur; else if (l < 8) nx = right ? cur : sib;
            x = pos::permute_coop(nx, l, gb);
            index >>= 1;
        }
        if (l < 4) bad = x != cap[4 * index + l];
    }
    return group_or(bad) != 0;
}
__device__ __forceinline__ ext2 rd2(const u64 *p) { return e_make(p[0], p[1]); }

template <int HASHER>
__global__ __launch_bounds__(256) void k_verify_queries(VQArgs a) {
    const int tid = threadIdx.x, l = tid & 15, lane = tid & 63, gb = lane & ~15;
    const size_t grp0 = (size_t)blockIdx.x * 16 + (tid >> 4), ngrp = (size_t)a.K * a.nq;
    const bool live = grp0 < ngrp;
    const size_t grp = live ? grp0 : 0;                   // idle groups redo group 0 (the shuffles need every lane) and write nothing
    const u32 k = (u32)(grp / a.nq), q = (u32)(grp % a.nq);
    const u64 *proof = a.proofs + (size_t)k * a.total, *vc = a.vchal + (size_t)k * a.vstride;
    const u64 *w = proof + a.queries + (size_t)q * a.query_stride;
    size_t x_index = (size_t)vc[VC_XIDX + q];
    const ext2 alpha = rd2(vc + VC_ALPHA), zeta = rd2(vc + VC_ZETA), zeta_next = rd2(vc + VC_ZETA_NEXT);
    const u32 capn4 = 4u << a.cap_height;
    u32 code = 0;
#define VQ_FAIL(C) do { if (code == 0) code = (C); } while (0)
    // ---- initial trees; sum_j alpha^j evals_j over the four oracles in order (batch 0) and the Z columns (batch 1)
    ext2 a16 = alpha;
#pragma unroll
    for (int i = 0; i < 4; i++) a16 = e_sqr(a16);
    const ext2 al = e_pow(alpha, (u64)l);
    ext2 r0 = e_from(0), r1 = e_from(0), off = e_from(1);
    for (int t = 0; t < 4; t++) {
        const u32 ncols = a.oracle_cols[t];
        const u64 *cap = t == 0 ? a.cs_cap : proof + a.caps + (size_t)(t - 1) * capn4;
        if (merkle_bad<HASHER>(w, ncols, x_index, w + ncols, a.depth0, cap, l, gb)) VQ_FAIL(1 + (u32)t);
        ext2 part = e_from(0);                            // lane l: sum_i evals[l + 16 i] (alpha^16)^i, Horner from the top
        if ((u32)l < ncols)
            for (int j = (int)(((ncols - 1 - (u32)l) >> 4) << 4) + l; j >= 0; j -= 16) part = e_add(e_mul(part, a16), e_from(w[j]));
        r0 = e_add(r0, e_mul(off, group_sum(e_mul(part, al))));
        off = e_mul(off, e_pow(alpha, (u64)ncols));
        if (t == 2) { ext2 ap = e_from(1); for (u32 j = 0; j < a.nch; j++) { r1 = e_add(r1, e_scale(ap, w[j])); ap = e_mul(ap, alpha); } }
        w += ncols + 4 * (size_t)a.depth0;
    }
    u64 subgroup_x = mul(GEN, dvpow(a.wN, (u64)(__brevll((unsigned long long)x_index) >> (64 - a.lgN))));
    ext2 old_eval;
    {
        const ext2 d0 = e_sub(e_from(subgroup_x), zeta), d1 = e_sub(e_from(subgroup_x), zeta_next);
        if ((d0.a == 0 && d0.b == 0) || (d1.a == 0 && d1.b == 0)) VQ_FAIL(5);
        ext2 sum = e_mul(e_sub(r0, rd2(vc + VC_RED0)), e_inv(d0));
        sum = e_add(e_mul(sum, rd2(vc + VC_SHIFT1)), e_mul(e_sub(r1, rd2(vc + VC_RED1)), e_inv(d1)));
        old_eval = sum;
    }
    // ---- reductions: consistency with the previous layer, interpolation of the coset at beta, Merkle path of the layer
    for (u32 r = 0; r < a.nred; r++) {
        const u32 ab = a.ab[r], arity = 1u << ab;
        const u64 *ev = w, *path = w + 2 * (size_t)arity;
        const size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
        if (!e_eq(rd2(ev + 2 * within), old_eval)) VQ_FAIL(6 + 2 * r);
        {   // compute_evaluation: sum_i vals_i prod_{j != i} (beta - p_j) / (p_i - p_j), p_j = coset_start gA^j; lane l takes i = l, l + 16
            const u64 gA = a.gA[r];
            const u32 rw = (u32)(__brev((unsigned)within) >> (32 - ab));
            const u64 coset_start = mul(subgroup_x, dvpow(gA, (u64)(arity - rw)));
            const ext2 beta = rd2(vc + VC_BETAS + 2 * r);
            ext2 acc = e_from(0);
            for (u32 i = (u32)l; i < arity; i += 16) {
                const u64 pi = mul(coset_start, dvpow(gA, (u64)i));
                ext2 num = e_from(1);
                u64 den = 1, pj = coset_start;
                for (u32 j = 0; j < arity; j++) {
                    if (j != i) { num = e_mul(num, e_sub(beta, e_from(pj))); den = mul(den, sub(pi, pj)); }
                    pj = mul(pj, gA);
                }
                const u32 bi = (u32)(__brev((unsigned)i) >> (32 - ab));                   // reverse_index_bits(evals)
                acc = e_add(acc, e_scale(e_mul(rd2(ev + 2 * bi), num), glf::inv(den)));
            }
            old_eval = group_sum(acc);
        }
        const u64 *cap = proof + a.fri_caps + (size_t)r * capn4;
        if (merkle_bad<HASHER>(ev, 2 * arity, coset_index, path, a.step_depth[r], cap, l, gb)) VQ_FAIL(7 + 2 * r);
        for (u32 i = 0; i < ab; i++) subgroup_x = sqr(subgroup_x);
        x_index = coset_index;
        w += 2 * (size_t)arity + 4 * (size_t)a.step_depth[r];
    }
    {   // final_poly.eval(subgroup_x)
        ext2 acc = e_from(0);
        for (u32 i = a.final_len; i-- > 0;) acc = e_add(e_scale(acc, subgroup_x), rd2(proof + a.final_poly + 2 * (size_t)i));
        if (!e_eq(acc, old_eval)) VQ_FAIL(6 + 2 * a.nred);
    }
#undef VQ_FAIL
    if (live && l == 0) a.status[grp] = code;
}
}  // namespace

#include "host_pool.h"
extern "C" int glp_verify_batch(glp_ctx *c, const glp_circuit *cc, uint32_t K, const uint64_t *proofs, int32_t *status_out, char *reasons_out) {
    GLP_REQUIRE(c && cc && proofs && status_out, "null argument");
    GLP_REQUIRE(cc->ctx == c, "circuit belongs to another context");
    GLP_REQUIRE(K >= 1 && K <= 65536, "glp_verify_batch: batch of %u proofs outside 1..65536", K);
    GLP_TRY(bind(c));
    const glp_circuit_desc &d = cc->d;
    const Layout &L = cc->L;
    const u32 nq = d.num_query_rounds, nred = d.num_reductions;
    const u32 vstride = (VC_XIDX + nq + 3) & ~3u;
    std::vector<VChal> vcs(K);
    std::vector<int> rc(K, GLP_OK);
    std::vector<std::string> why(K);
    std::vector<u64> hv((size_t)K * vstride, 0);
    // the proofs go up (pageable host memory: the copy blocks its caller) on a thread of their own while the host half runs
    struct Scratch { glp_ctx *c; std::vector<void *> p; ~Scratch() { (void)hipStreamSynchronize(c->stream); for (void *q : p) c->release(q); } } sc{c, {}};
    auto get = [&](void **p, size_t bytes) -> int { int r = c->alloc(p, bytes); if (r == GLP_OK) sc.p.push_back(*p); return r; };
    u64 *dev_proofs = nullptr, *dev_vc = nullptr;
    u32 *dev_status = nullptr;
    GLP_TRY(get((void **)&dev_proofs, (size_t)K * L.total * 8));
    GLP_TRY(get((void **)&dev_vc, hv.size() * 8));
    GLP_TRY(get((void **)&dev_status, (size_t)K * nq * 4));
    const bool trace = getenv("GLP_BATCH_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    hipError_t up_err = hipSuccess;
    std::thread uploader([&] {
        up_err = hipSetDevice(c->device);
        if (up_err == hipSuccess) up_err = hipMemcpyAsync(dev_proofs, proofs, (size_t)K * L.total * 8, hipMemcpyHostToDevice, c->stream);
    });
    // PoseidonGoldilocksConfig: the K transcripts run on the device, as the batch prover's do (transcript_dev.h; the uploaded proofs are
    // the image its kernels read).  The host keeps the canonical-form scan (beside the upload) and the vanishing-polynomial identity
    // (beside the query rounds).  KeccakGoldilocksConfig, or GLP_VERIFY_HOST_TRANSCRIPT=1: everything of verify_front on host threads.
    const bool dev_tr = d.hasher == GLP_HASH_POSEIDON && getenv("GLP_VERIFY_HOST_TRANSCRIPT") == nullptr;
    const u32 nch = d.num_challenges, capn = 1u << d.cap_height;
    u64 *dev_chal = nullptr, *dev_qpp = nullptr, *dev_apow = nullptr, *dev_zetas = nullptr;
    u32 *dev_err = nullptr;
    std::vector<u64> h_back;
    const u64 *h_chal = nullptr, *h_qpp = nullptr, *h_apow = nullptr, *h_zetas = nullptr;
    const u32 *h_err = nullptr;
    hipEvent_t ev_chal = nullptr;
    struct EvGuard { hipEvent_t &e; ~EvGuard() { if (e) (void)hipEventDestroy(e); } } evg{ev_chal};
    double t_front = 0, t_up = 0;
    if (!dev_tr) {
        // host half, one proof per task on the context's pool: canonical form, transcript, proof of work, vanishing polynomial at zeta
        ctx_host_pool(c).run(K, [&](size_t k) {
            rc[k] = verify_front(cc, proofs + k * L.total, vcs[k]);
            if (rc[k] != GLP_OK) { why[k] = g_last_error; return; }
            const VChal &v = vcs[k];
            u64 *o = &hv[k * vstride];
            auto put = [&](u32 at, const E &e) { o[at] = e.v.a; o[at + 1] = e.v.b; };
            put(VC_ALPHA, v.fri_alpha); put(VC_ZETA, v.zeta); put(VC_ZETA_NEXT, v.zeta_next); put(VC_RED0, v.red0); put(VC_RED1, v.red1); put(VC_SHIFT1, v.shift1);
            for (u32 r = 0; r < nred; r++) put(VC_BETAS + 2 * r, v.fri_betas[r]);
            for (u32 q = 0; q < nq; q++) o[VC_XIDX + q] = v.x_index[q];
        });
        // device half: every query round of every proof in one launch.  Proofs the host half already rejected still ride along
        // (their slots hold zero challenges and every index is in range); their device status is ignored.
        t_front = since();
        uploader.join();
        t_up = since();
        if (up_err != hipSuccess) return set_error(GLP_ERR_HIP, "upload of the proofs: %s", hipGetErrorString(up_err));
        GLP_HIP(hipMemcpyAsync(dev_vc, hv.data(), hv.size() * 8, hipMemcpyHostToDevice, c->stream));
    } else {
        ctx_host_pool(c).run(K, [&](size_t k) {
            rc[k] = verify_canonical(cc, proofs + k * L.total);
            if (rc[k] != GLP_OK) why[k] = g_last_error;
        });
        t_front = since();
        uploader.join();
        t_up = since();
        if (up_err != hipSuccess) return set_error(GLP_ERR_HIP, "upload of the proofs: %s", hipGetErrorString(up_err));
        const u32 total_cols = L.oracle_cols[0] + L.oracle_cols[1] + L.oracle_cols[2] + L.oracle_cols[3];
        u64 *dch, *dev_apl, *dev_fap, *dev_fpp, *dev_betas, *dev_idx;
        GLP_TRY(get((void **)&dch, (size_t)K * DCH_WORDS * 8));
        // betas | gammas, public-input hash, alpha powers, zetas, error bits: one block, one copy back
        const size_t w_chal = (size_t)K * 2 * MAXCH, w_qpp = (size_t)K * 3 * MAXCH, w_apow = (size_t)K * nch * 2, w_zetas = (size_t)K * 4, w_err = (K + 1) / 2;
        u64 *dev_back;
        GLP_TRY(get((void **)&dev_back, (w_chal + w_qpp + w_apow + w_zetas + w_err) * 8));
        dev_chal = dev_back; dev_qpp = dev_chal + w_chal; dev_apow = dev_qpp + w_qpp; dev_zetas = dev_apow + w_apow; dev_err = (u32 *)(dev_zetas + w_zetas);
        GLP_TRY(get((void **)&dev_apl, (size_t)K * nch * 2 * APL_WORDS * 8));
        GLP_TRY(get((void **)&dev_fap, (size_t)K * 2 * total_cols * 8));
        GLP_TRY(get((void **)&dev_fpp, (size_t)K * 10 * 8));
        GLP_TRY(get((void **)&dev_betas, (size_t)std::max<u32>(nred, 1) * K * 2 * 8));
        GLP_TRY(get((void **)&dev_idx, (size_t)K * nq * 8));
        GLP_HIP(hipMemsetAsync(dev_back, 0, (w_chal + w_qpp + w_apow + w_zetas + w_err) * 8, c->stream));
        TrGeo g;
        g.dch = dch; g.image = dev_proofs; g.total = L.total; g.K = K; g.capn = capn; g.nch = nch;
        const dim3 tg((K + 15) / 16), tb(256);
        const size_t cap4 = (size_t)capn * 4;
        // a cap is "copied" from the image onto itself: source = image + offset with the proof stride, destination offset the same
        hipLaunchKernelGGL(k_tr_begin, tg, tb, 0, c->stream, g, cc->digest[0], cc->digest[1], cc->digest[2], cc->digest[3], L.pis, d.num_public_inputs,
                           dev_proofs + L.caps, L.total, L.caps, dev_chal, dev_qpp);
        hipLaunchKernelGGL(k_tr_alphas, tg, tb, 0, c->stream, g, dev_proofs + L.caps + cap4, L.total, L.caps + cap4, 2u, dev_apow, dev_apl);
        hipLaunchKernelGGL(k_tr_zeta, tg, tb, 0, c->stream, g, dev_proofs + L.caps + 2 * cap4, L.total, L.caps + 2 * cap4, d.degree_bits,
                           root_of_unity((int)d.degree_bits), dev_zetas, dev_err);
        OpenGeo og;
        memset(&og, 0, sizeof(og));
        for (int b = 0; b < 4; b++) og.cols[b] = L.oracle_cols[b];
        og.openings_off = L.openings; og.nch = nch; og.npp = d.num_partial_products; og.nopen = (u32)L.nopen;
        hipLaunchKernelGGL(k_tr_fri_alpha, tg, tb, 0, c->stream, g, og, dev_zetas, dev_fap, dev_fpp);
        for (u32 r = 0; r < nred; r++)
            hipLaunchKernelGGL(k_tr_beta, tg, tb, 0, c->stream, g, dev_proofs + L.fri_caps + r * cap4, L.total, L.fri_caps + r * cap4, dev_betas + (size_t)r * K * 2);
        hipLaunchKernelGGL(k_trv_final, tg, tb, 0, c->stream, g, L.final_poly, 2u * L.final_len);
        hipLaunchKernelGGL(k_trv_queries, tg, tb, 0, c->stream, g, d.proof_of_work_bits, L.pow, nq, (u64)1 << (d.degree_bits + d.rate_bits), dev_idx, dev_err);
        hipLaunchKernelGGL(k_trv_pack, dim3((K + 255) / 256), dim3(256), 0, c->stream, K, vstride, dev_fap, total_cols, dev_fpp, dev_betas, nred, dev_idx, nq, dev_vc);
        GLP_HIP(hipGetLastError());
        // what the host's half of the check needs comes back while the query rounds run
        h_back.resize(w_chal + w_qpp + w_apow + w_zetas + w_err);
        GLP_HIP(hipMemcpyAsync(h_back.data(), dev_back, h_back.size() * 8, hipMemcpyDeviceToHost, c->stream));
        GLP_HIP(hipEventCreateWithFlags(&ev_chal, hipEventDisableTiming));
        GLP_HIP(hipEventRecord(ev_chal, c->stream));
        h_chal = h_back.data(); h_qpp = h_chal + w_chal; h_apow = h_qpp + w_qpp; h_zetas = h_apow + w_apow; h_err = (const u32 *)(h_zetas + w_zetas);
    }
    VQArgs a;
    memset(&a, 0, sizeof(a));
    a.proofs = dev_proofs; a.vchal = dev_vc; a.status = dev_status;
    const size_t N = (size_t)1 << (d.degree_bits + d.rate_bits);
    a.cs_cap = cc->cs->digests + 4 * merkle_cap_offset(N, (int)d.cap_height);
    a.total = L.total; a.caps = L.caps; a.fri_caps = L.fri_caps; a.queries = L.queries; a.query_stride = L.query_stride; a.final_poly = L.final_poly;
    a.vstride = vstride; a.nq = nq; a.K = K; a.nch = d.num_challenges; a.lgN = d.degree_bits + d.rate_bits; a.cap_height = d.cap_height;
    a.depth0 = L.depth0; a.nred = nred; a.final_len = L.final_len;
    for (int t = 0; t < 4; t++) a.oracle_cols[t] = L.oracle_cols[t];
    for (u32 r = 0; r < nred; r++) { a.ab[r] = d.reduction_arity_bits[r]; a.step_depth[r] = L.step_depth[r]; a.gA[r] = root_of_unity((int)d.reduction_arity_bits[r]); }
    a.wN = root_of_unity((int)a.lgN);
    {
        StageScope st(c, "verify_queries", 8.0 * K * L.total);
        const unsigned nblocks = (unsigned)(((size_t)K * nq + 15) / 16);
        if (d.hasher == GLP_HASH_KECCAK25) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_verify_queries<GLP_HASH_KECCAK25>), dim3(nblocks), dim3(256), 0, c->stream, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_verify_queries<GLP_HASH_POSEIDON>), dim3(nblocks), dim3(256), 0, c->stream, a);
        GLP_HIP(hipGetLastError());
    }
    std::vector<u32> hs((size_t)K * nq);
    double t_chal = 0, t_van = 0;
    if (dev_tr) {
        GLP_HIP(hipEventSynchronize(ev_chal));
        t_chal = since();
        // same order of checks as verify_front: canonical form (above), proof of work, then the identity at zeta
        ctx_host_pool(c).run(K, [&](size_t k) {
            if (rc[k] != GLP_OK) return;
            if (h_err[k] & 8u) { rc[k] = set_error(GLP_ERR_PROVE, "Invalid proof of work witness."); why[k] = g_last_error; return; }
            u64 alphas[MAXCH] = {0, 0, 0, 0};
            for (u32 i = 0; i < nch; i++) alphas[i] = h_apow[(k * nch + i) * 2 + 1];
            const E zeta(e_make(h_zetas[4 * k], h_zetas[4 * k + 1]));
            rc[k] = verify_vanishing(cc, proofs + k * L.total, &h_chal[k * 2 * MAXCH], &h_chal[k * 2 * MAXCH + MAXCH], alphas, zeta, &h_qpp[k * 3 * MAXCH + 2 * MAXCH]);
            if (rc[k] != GLP_OK) why[k] = g_last_error;
        });
        t_van = since();
    }
    // (a copy into pageable memory blocks its caller until the stream gets there: enqueued only now, the identity at zeta ran beside the query rounds)
    GLP_HIP(hipMemcpyAsync(hs.data(), dev_status, hs.size() * 4, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    if (trace) {
        if (dev_tr) fprintf(stderr, "[glp_verify_batch K=%u dev] canonical scan %.3f ms | upload done at %.3f | challenges back at %.3f | identity at zeta done at %.3f | query rounds done at %.3f ms\n",
                            K, t_front, t_up, t_chal, t_van, since());
        else fprintf(stderr, "[glp_verify_batch K=%u] host half %.3f ms | upload done at %.3f | query rounds done at %.3f ms\n", K, t_front, t_up, since());
    }
    for (u32 k = 0; k < K; k++) {
        if (rc[k] == GLP_OK)
            for (u32 q = 0; q < nq && rc[k] == GLP_OK; q++)
                if (hs[(size_t)k * nq + q]) { rc[k] = query_failure(q, hs[(size_t)k * nq + q], nred); why[k] = g_last_error; }
        status_out[k] = rc[k];
        if (reasons_out) {
            char *o = reasons_out + (size_t)k * GLP_REASON_LEN;
            memset(o, 0, GLP_REASON_LEN);
            if (rc[k] != GLP_OK) strncpy(o, why[k].c_str(), GLP_REASON_LEN - 1);
        }
    }
    return GLP_OK;
}

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
