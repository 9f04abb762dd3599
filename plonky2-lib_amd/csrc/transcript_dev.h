// transcript_dev.h -- plonky2's duplex `Challenger` (iop/challenger.rs) on the device, one 16-lane group per proof, and the kernels that
// run the Fiat-Shamir steps of K proofs of one circuit in lock step.  Two users, each with its own copy (anonymous namespace):
//   prover.hip  (prover_batch_dev.inc): glp_prove_batch assembles the K proofs in a device image and the k_tr_* kernels observe what the
//               previous device stage left there;
//   verifier.hip (glp_verify_batch):    the uploaded proofs ARE that image -- the same kernels (caps "copied" onto themselves) plus
//               k_trv_final / k_trv_queries, which read the final polynomial and the proof-of-work witness from the image instead of
//               from the prover's buffers, and k_trv_pack, which lays the challenges out for k_verify_queries.
// [REF: reached from `data.prove(pw)` / `data.verify(proof)`, src/zkdsa/circuits/mod.rs:326-347]
#pragma once
#include "poseidon.h"
#include "prover_types.h"

namespace {
__device__ __forceinline__ u64 tr_pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = mul(r, b); b = sqr(b); e >>= 1; }
    return r;
}
constexpr u32 DCH_WORDS = 32;        // per proof: st[12] | in[8] | (unused 8) | nin | nout | pad
// duplex sponge of one proof on the 16 lanes of a group: lane l < 12 holds state element l, lane l < 8 input-buffer slot l
struct DSponge { u64 st, inb; u32 nin, nout; };
__device__ __forceinline__ void ds_load(DSponge &s, const u64 *p, int l) {
    s.st = l < 12 ? p[l] : 0; s.inb = l < 8 ? p[12 + l] : 0; s.nin = (u32)p[28]; s.nout = (u32)p[29];
}
__device__ __forceinline__ void ds_store(const DSponge &s, u64 *p, int l, bool live) {
    if (!live) return;
    if (l < 12) p[l] = s.st;
    if (l < 8) p[12 + l] = s.inb;
    if (l == 0) { p[28] = s.nin; p[29] = s.nout; }
}
__device__ __noinline__ void ds_duplex(DSponge &s, int l, int gb) {
    if ((u32)l < s.nin) s.st = s.inb;
    s.nin = 0;
    s.st = pos::permute_coop(s.st, l, gb);
    s.nout = 8;
}
// observe n elements fetch(0) .. fetch(n - 1) (the same sequence on every group of the launch: control flow is wave-uniform)
template <class F>
__device__ __forceinline__ void ds_observe_f(DSponge &s, u32 n, int l, int gb, F fetch) {
    u32 i = 0;
    while (i < n) {
        const u32 take = min(8u - s.nin, n - i);
        if ((u32)l >= s.nin && (u32)l < s.nin + take) s.inb = fetch(i + (u32)l - s.nin);
        s.nin += take; i += take; s.nout = 0;
        if (s.nin == 8) ds_duplex(s, l, gb);
    }
}
__device__ __forceinline__ void ds_observe(DSponge &s, const u64 *p, u32 n, int l, int gb) {
    ds_observe_f(s, n, l, gb, [p](u32 i) { return p[i]; });
}
__device__ __forceinline__ void ds_observe1(DSponge &s, u64 v, int l, int gb) {      // one element held by every lane
    if ((u32)l == s.nin) s.inb = v;
    s.nin++; s.nout = 0;
    if (s.nin == 8) ds_duplex(s, l, gb);
}
__device__ __forceinline__ u64 ds_get(DSponge &s, int l, int gb) {
    if (s.nin > 0 || s.nout == 0) ds_duplex(s, l, gb);
    s.nout--;
    return pos::shfl64(s.st, gb + (int)s.nout);
}
__device__ __forceinline__ ext2 ds_get_ext(DSponge &s, int l, int gb) { const u64 a = ds_get(s, l, gb); const u64 b = ds_get(s, l, gb); return e_make(a, b); }
__device__ __forceinline__ ext2 dgroup_sum(ext2 v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v = e_add(v, e_make(pos::shfl_xor64(v.a, m), pos::shfl_xor64(v.b, m)));
    return v;
}

struct TrGeo {                          // what every transcript kernel needs
    u64 *dch;                           // [K][DCH_WORDS]
    u64 *image;                         // [K][total]: the proofs being assembled
    size_t total;
    u32 K, capn, nch;
};
#define TR_PROLOGUE                                                                           \
    const int tid = threadIdx.x, l = tid & 15, gb = (tid & 63) & ~15;                         \
    const u32 k0 = blockIdx.x * 16 + (tid >> 4);                                              \
    const bool live = k0 < g.K;                                                               \
    const u32 k = live ? k0 : 0;                                                              \
    u64 *dch = g.dch + (size_t)k * DCH_WORDS, *img = g.image + (size_t)k * g.total;           \
    (void)img
// copy a cap (capn digests) from a digest buffer into the image and observe it
__device__ __forceinline__ void tr_cap(DSponge &s, const u64 *cap, u64 *dst, u32 capn, bool live, int l, int gb) {
    if (live) for (u32 i = (u32)l; i < 4 * capn; i += 16) dst[i] = cap[i];
    ds_observe(s, cap, 4 * capn, l, gb);
}

// T1: fresh transcript; public-input hash (InnerHasher = Poseidon sponge over the public inputs, already in the image);
// observe circuit digest, that hash, the wires cap; betas, gammas -> chal[k][2 MAXCH], qpp[k][3 MAXCH] (betas, gammas, pih)
__global__ __launch_bounds__(256) void k_tr_begin(TrGeo g, u64 d0, u64 d1, u64 d2, u64 d3, size_t pis_off, u32 npi, const u64 *cap_b, size_t cap_stride,
                                                  size_t caps_off, u64 *chal, u64 *qpp) {
    TR_PROLOGUE;
    u64 x = 0;                                            // hash_n_to_hash_no_pad(public_inputs)
    for (u32 c0 = 0; c0 < npi; c0 += 8) {
        if (l < 8 && c0 + (u32)l < npi) x = img[pis_off + c0 + l];
        x = pos::permute_coop(x, l, gb);
    }
    const u64 pih0 = pos::shfl64(x, gb), pih1 = pos::shfl64(x, gb + 1), pih2 = pos::shfl64(x, gb + 2), pih3 = pos::shfl64(x, gb + 3);
    DSponge s;
    s.st = 0; s.inb = 0; s.nin = 0; s.nout = 0;
    ds_observe1(s, d0, l, gb); ds_observe1(s, d1, l, gb); ds_observe1(s, d2, l, gb); ds_observe1(s, d3, l, gb);
    ds_observe1(s, pih0, l, gb); ds_observe1(s, pih1, l, gb); ds_observe1(s, pih2, l, gb); ds_observe1(s, pih3, l, gb);
    tr_cap(s, cap_b + (size_t)k * cap_stride, img + caps_off, g.capn, live, l, gb);
    u64 *ck = chal + (size_t)k * 2 * MAXCH, *qk = qpp + (size_t)k * 3 * MAXCH;
    for (u32 i = 0; i < g.nch; i++) { const u64 b = ds_get(s, l, gb); if (live && l == 0) { ck[i] = b; qk[i] = b; } }
    for (u32 i = 0; i < g.nch; i++) { const u64 b = ds_get(s, l, gb); if (live && l == 0) { ck[MAXCH + i] = b; qk[MAXCH + i] = b; } }
    if (live && l == 0) { qk[2 * MAXCH] = pih0; qk[2 * MAXCH + 1] = pih1; qk[2 * MAXCH + 2] = pih2; qk[2 * MAXCH + 3] = pih3; }
    ds_store(s, dch, l, live);
}
// T2: observe the Z / partial-products cap; alphas; their powers alpha^t, t < nterms, whole (apow[k][i][t]) and as the 22-bit limb
// words of AccHL (apl, APL_WORDS per power), lane l taking t = l, l + 16, ...
__global__ __launch_bounds__(256) void k_tr_alphas(TrGeo g, const u64 *cap_b, size_t cap_stride, size_t caps_off, u32 nterms, u64 *apow, u64 *apl) {
    TR_PROLOGUE;
    DSponge s;
    ds_load(s, dch, l);
    tr_cap(s, cap_b + (size_t)k * cap_stride, img + caps_off, g.capn, live, l, gb);
    for (u32 i = 0; i < g.nch; i++) {
        const u64 alpha = ds_get(s, l, gb);
        u64 a16 = alpha;
#pragma unroll
        for (int j = 0; j < 4; j++) a16 = sqr(a16);
        u64 x = tr_pow(alpha, (u64)l);
        for (u32 t = (u32)l; t < nterms; t += 16) {
            const size_t e = ((size_t)k * g.nch + i) * nterms + t;
            if (live) {
                apow[e] = x;
                const u64 mp = mul(x, 1ull << 32);
                u64 *o = apl + APL_WORDS * e;
                o[0] = (x & 0x3FFFFFull) | (((x >> 22) & 0x3FFFFFull) << 32); o[1] = x >> 44;
                o[2] = (mp & 0x3FFFFFull) | (((mp >> 22) & 0x3FFFFFull) << 32); o[3] = mp >> 44;
            }
            x = mul(x, a16);
        }
    }
    ds_store(s, dch, l, live);
}
// T3: observe the quotient cap; zeta; zetas[k] = (zeta, g_n zeta); err[k] |= 1 if zeta lies in the subgroup
__global__ __launch_bounds__(256) void k_tr_zeta(TrGeo g, const u64 *cap_b, size_t cap_stride, size_t caps_off, u32 lg, u64 wn, u64 *zetas, u32 *err) {
    TR_PROLOGUE;
    DSponge s;
    ds_load(s, dch, l);
    tr_cap(s, cap_b + (size_t)k * cap_stride, img + caps_off, g.capn, live, l, gb);
    const ext2 zeta = ds_get_ext(s, l, gb);
    ext2 zp = zeta;
    for (u32 i = 0; i < lg; i++) zp = e_sqr(zp);
    const ext2 zn = e_scale(zeta, wn);
    if (live && l == 0) {
        u64 *z = zetas + 4 * (size_t)k;
        z[0] = zeta.a; z[1] = zeta.b; z[2] = zn.a; z[3] = zn.b;
        if (zp.a == 1 && zp.b == 0) err[k] |= 1u;
    }
    ds_store(s, dch, l, live);
}
// openings: fold the nob partial sums of every column into the image, in proof order (constants, sigmas, wires, zs, zs_next,
// partial products, quotient chunks).  One thread per (proof, opening).
struct OpenGeo { size_t poff[6], openings_off; u32 cols[4], nob, nch, npp, nopen; };
__global__ __launch_bounds__(256) void k_open_reduce(TrGeo g, OpenGeo og, const u64 *partial) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (size_t)g.K * og.nopen) return;
    const u32 k = (u32)(id / og.nopen);
    u32 o = (u32)(id % og.nopen), b, col;
    const u32 nzs = og.nch, npps = og.nch * og.npp;
    if (o < og.cols[0]) { b = 0; col = o; }
    else if ((o -= og.cols[0]) < og.cols[1]) { b = 1; col = o; }
    else if ((o -= og.cols[1]) < nzs) { b = 2; col = o; }
    else if ((o -= nzs) < nzs) { b = 4; col = o; }                         // Z at g zeta: the fifth group of partial sums
    else if ((o -= nzs) < npps) { b = 2; col = nzs + o; }
    else { o -= npps; b = 3; col = o; }
    const u64 *p = partial + (size_t)k * og.poff[5] + og.poff[b] + 2 * (size_t)col * og.nob;
    u64 a = 0, bb = 0;
    for (u32 j = 0; j < og.nob; j++) { a = add(a, p[2 * j]); bb = add(bb, p[2 * j + 1]); }
    u64 *dst = g.image + (size_t)k * g.total + og.openings_off + 2 * (id % og.nopen);
    dst[0] = a; dst[1] = bb;
}
// T4: observe the openings (transcript order: constants/sigmas, wires, zs, partial products, quotient, zs_next); FRI alpha;
// fap[k][j] = alpha^j over the columns of the four oracles in ORACLE order (what k_final_values multiplies the LDE rows by);
// fpp[k] = red0, red1, zeta, zeta_next, alpha^nch
__global__ __launch_bounds__(256) void k_tr_fri_alpha(TrGeo g, OpenGeo og, const u64 *zetas, u64 *fap, u64 *fpp) {
    TR_PROLOGUE;
    DSponge s;
    ds_load(s, dch, l);
    const u64 *op = img + og.openings_off;
    const u32 nzs = og.nch, npps = og.nch * og.npp;
    const u64 *p_cs = op, *p_w = p_cs + 2 * og.cols[0], *p_zs = p_w + 2 * og.cols[1], *p_zn = p_zs + 2 * nzs, *p_pp = p_zn + 2 * nzs, *p_q = p_pp + 2 * npps;
    ds_observe(s, p_cs, 2 * og.cols[0], l, gb); ds_observe(s, p_w, 2 * og.cols[1], l, gb); ds_observe(s, p_zs, 2 * nzs, l, gb);
    ds_observe(s, p_pp, 2 * npps, l, gb); ds_observe(s, p_q, 2 * og.cols[3], l, gb); ds_observe(s, p_zn, 2 * nzs, l, gb);
    const ext2 alpha = ds_get_ext(s, l, gb);
    ext2 a16 = alpha;
#pragma unroll
    for (int j = 0; j < 4; j++) a16 = e_sqr(a16);
    const u32 total_cols = og.cols[0] + og.cols[1] + og.cols[2] + og.cols[3];
    ext2 x = e_pow(alpha, (u64)l), r0 = e_from(0);
    u64 *ap = fap + (size_t)k * 2 * total_cols;
    for (u32 j = (u32)l; j < total_cols; j += 16) {
        // oracle-order column j -> its opening: cs | wires | zb (zs then partial products; zs_next sits between them in the proof) | quotient
        const u64 *src;
        u32 c = j;
        if (c < og.cols[0]) src = p_cs + 2 * c;
        else if ((c -= og.cols[0]) < og.cols[1]) src = p_w + 2 * c;
        else if ((c -= og.cols[1]) < og.cols[2]) src = c < nzs ? p_zs + 2 * c : p_pp + 2 * (c - nzs);
        else src = p_q + 2 * (c - og.cols[2]);
        if (live) { ap[2 * j] = x.a; ap[2 * j + 1] = x.b; }
        r0 = e_add(r0, e_mul(x, e_make(src[0], src[1])));
        x = e_mul(x, a16);
    }
    r0 = dgroup_sum(r0);
    ext2 r1 = e_from(0), y = e_from(1);
    for (u32 j = 0; j < nzs; j++) { r1 = e_add(r1, e_mul(y, e_make(p_zn[2 * j], p_zn[2 * j + 1]))); y = e_mul(y, alpha); }
    if (live && l == 0) {
        u64 *v = fpp + (size_t)k * 10;
        const u64 *z = zetas + 4 * (size_t)k;
        v[0] = r0.a; v[1] = r0.b; v[2] = r1.a; v[3] = r1.b; v[4] = z[0]; v[5] = z[1]; v[6] = z[2]; v[7] = z[3]; v[8] = y.a; v[9] = y.b;     // y = alpha^nch
    }
    ds_store(s, dch, l, live);
}
// T5 (per reduction): observe the layer's cap; beta -> betas[k]
__global__ __launch_bounds__(256) void k_tr_beta(TrGeo g, const u64 *cap_b, size_t cap_stride, size_t caps_off, u64 *betas) {
    TR_PROLOGUE;
    DSponge s;
    ds_load(s, dch, l);
    tr_cap(s, cap_b + (size_t)k * cap_stride, img + caps_off, g.capn, live, l, gb);
    const ext2 beta = ds_get_ext(s, l, gb);
    if (live && l == 0) { betas[2 * (size_t)k] = beta.a; betas[2 * (size_t)k + 1] = beta.b; }
    ds_store(s, dch, l, live);
}
// T6: final polynomial (bit-reversed coefficient slots [2][fl]) -> image in natural order; observe it; export the sponge for the
// proof-of-work search: pst[k] = state with the pending inputs written over it, ppos[k] = number of pending inputs
__global__ __launch_bounds__(256) void k_tr_final(TrGeo g, const u64 *cur, u32 lgf, size_t final_off, u64 *pst, u32 *ppos, u32 *err) {
    TR_PROLOGUE;
    DSponge s;
    ds_load(s, dch, l);
    const u32 fl = 1u << lgf;
    const u64 *h = cur + (size_t)k * 2 * fl;
    u64 *pf = img + final_off;
    if (live) for (u32 p = (u32)l; p < fl; p += 16) { const u32 kk = bitrev32(p, (int)lgf); pf[2 * kk] = h[p]; pf[2 * kk + 1] = h[fl + p]; }
    // observed from the source (natural coefficient kk sits in slot bitrev(kk)), not read back from the image just written
    ds_observe_f(s, 2 * fl, l, gb, [h, fl, lgf](u32 e) { return h[(e & 1 ? fl : 0) + bitrev32(e >> 1, (int)lgf)]; });
    if (live) {
        if (l < 12) pst[(size_t)k * 12 + l] = ((u32)l < s.nin) ? s.inb : s.st;
        if (l == 0) { ppos[k] = s.nin; if (s.nin >= 8) err[k] |= 2u; }
    }
    ds_store(s, dch, l, live);
}
// T7: observe the proof-of-work witness, check the response, draw the query indices
__global__ __launch_bounds__(256) void k_tr_queries(TrGeo g, const u64 *best, u32 pow_bits, size_t pow_off, u32 nq, u64 Nmask_plus1, u64 *idx, u32 *err) {
    TR_PROLOGUE;
    DSponge s;
    ds_load(s, dch, l);
    const u64 w = best[k];
    if (live && l == 0) { img[pow_off] = w; if (w == ~0ull) err[k] |= 4u; }
    ds_observe1(s, w, l, gb);
    const u64 resp = ds_get(s, l, gb);
    if (live && l == 0 && pow_bits && (resp >> (64 - pow_bits)) != 0) err[k] |= 8u;
    for (u32 q = 0; q < nq; q++) {
        const u64 x = ds_get(s, l, gb) % Nmask_plus1;
        if (live && l == 0) idx[(size_t)k * nq + q] = x;
    }
    ds_store(s, dch, l, live);
}
#undef TR_PROLOGUE

// ---- the verifier's variants (the image is a batch of finished proofs) ------------------------------------------------------------
#define TR_PROLOGUE                                                                           \
    const int tid = threadIdx.x, l = tid & 15, gb = (tid & 63) & ~15;                         \
    const u32 k0 = blockIdx.x * 16 + (tid >> 4);                                              \
    const bool live = k0 < g.K;                                                               \
    const u32 k = live ? k0 : 0;                                                              \
    u64 *dch = g.dch + (size_t)k * DCH_WORDS, *img = g.image + (size_t)k * g.total;           \
    (void)img
// T6': observe the final polynomial where the proof holds it (natural order, (a, b) pairs)
__global__ __launch_bounds__(256) void k_trv_final(TrGeo g, size_t final_off, u32 nwords) {
    TR_PROLOGUE;
    DSponge s;
    ds_load(s, dch, l);
    ds_observe(s, img + final_off, nwords, l, gb);
    ds_store(s, dch, l, live);
}
// T7': observe the proof's proof-of-work witness, check the response (err bit 8), draw the query indices
__global__ __launch_bounds__(256) void k_trv_queries(TrGeo g, u32 pow_bits, size_t pow_off, u32 nq, u64 N, u64 *idx, u32 *err) {
    TR_PROLOGUE;
    DSponge s;
    ds_load(s, dch, l);
    ds_observe1(s, img[pow_off], l, gb);
    const u64 resp = ds_get(s, l, gb);
    if (live && l == 0 && pow_bits && (resp >> (64 - pow_bits)) != 0) err[k] |= 8u;
    for (u32 q = 0; q < nq; q++) {
        const u64 x = ds_get(s, l, gb) % N;
        if (live && l == 0) idx[(size_t)k * nq + q] = x;
    }
    ds_store(s, dch, l, live);
}
#undef TR_PROLOGUE
// what k_verify_queries reads per proof: fri_alpha, zeta, zeta_next, red0, red1, alpha^nch, betas[r] (ext each) from word 12, x_index[q] from word 44
__global__ __launch_bounds__(256) void k_trv_pack(u32 K, u32 vstride, const u64 *fap, u32 total_cols, const u64 *fpp, const u64 *betas, u32 nred,
                                                  const u64 *idx, u32 nq, u64 *vc) {
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    u64 *o = vc + (size_t)k * vstride;
    const u64 *v = fpp + (size_t)k * 10, *ap = fap + (size_t)k * 2 * total_cols;
    o[0] = ap[2]; o[1] = ap[3];                           // alpha^1 (every circuit has more than one opened column)
    o[2] = v[4]; o[3] = v[5]; o[4] = v[6]; o[5] = v[7];   // zeta, zeta_next
    o[6] = v[0]; o[7] = v[1]; o[8] = v[2]; o[9] = v[3];   // red0, red1
    o[10] = v[8]; o[11] = v[9];                           // alpha^nch
    for (u32 r = 0; r < nred; r++) { o[12 + 2 * r] = betas[((size_t)r * K + k) * 2]; o[13 + 2 * r] = betas[((size_t)r * K + k) * 2 + 1]; }
    for (u32 q = 0; q < nq; q++) o[44 + q] = idx[(size_t)k * nq + q];
}
}  // namespace
