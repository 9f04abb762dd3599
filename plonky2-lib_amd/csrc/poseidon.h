// poseidon.h -- plonky2's Poseidon permutation over Goldilocks (width 12, x^7, 4+22+4 rounds,
// MDS = circulant(17,15,41,16,2,28,13,13,39,18,34,20) + diag(8,0,...)) for gfx950 and the host
// transcript.  One state per lane; the 12x12 MDS has 6-bit entries, so each output row is
// accumulated unreduced from the 32-bit halves of the inputs and reduced once.
//
// Replaces plonky2 `hash/poseidon.rs` + `poseidon_goldilocks.rs` on the prove() path
// [REF src/ecdsa/gadgets/ecdsa.rs:349]; same primitive the reference calls natively at
// [REF src/zkdsa/account.rs:165, src/smt/goldilocks_poseidon/mod.rs:165].
#pragma once
#include "glf.h"
#include "poseidon_rc.inc"

namespace pos {
using namespace glf;

#if defined(__HIP_DEVICE_COMPILE__)
static __device__ const u64 RC[360] = { GLP_POSEIDON_RC_LIST };
#else
static const u64 RC[360] = { GLP_POSEIDON_RC_LIST };
#endif

GLF_HD u64 sbox7(u64 x) {
    u64 x2 = sqr(x), x4 = sqr(x2), x3 = mul(x, x2);
    return mul(x3, x4);
}

// out[r] = sum_i s[(i+r)%12]*CIRC[i] + s[r]*DIAG[r]   (plonky2 `mds_row_shf`)
GLF_HD void mds_layer(u64 s[12]) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u64 al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (u64)lo[(i + r) % 12] * C[i];
            ah += (u64)hi[(i + r) % 12] * C[i];
        }
        if (r == 0) { al += (u64)lo[0] * 8; ah += (u64)hi[0] * 8; }
        // value = al + ah * 2^32, both < 2^42
        u64 l = al + (ah << 32);
        u32 h = (u32)(ah >> 32) + (l < al ? 1u : 0u);
        s[r] = reduce96(l, h);
    }
}

// Reference schedule (host transcript; also the definition the device path is tested against).
GLF_HD void permute_ref(u64 s[12]) {
    int rc = 0;
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox7(add(s[i], RC[rc + i]));
        rc += 12;
        mds_layer(s);
    }
    for (int r = 0; r < 22; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = add(s[i], RC[rc + i]);
        rc += 12;
        s[0] = sbox7(s[0]);
        mds_layer(s);
    }
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox7(add(s[i], RC[rc + i]));
        rc += 12;
        mds_layer(s);
    }
}

GLF_HD u32 mds_entry(int r, int i) {     // M[r][i] of the circulant-plus-diagonal MDS matrix
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    return C[(i - r + 12) % 12] + (r == 0 && i == 0 ? 8u : 0u);
}
// ---- partial rounds, K at a time -------------------------------------------------------------------
// A partial round is x <- M (E x + e0 sbox(x0)) + c  (E zeroes coordinate 0, c = the next round's constants):
// linear except for ONE S-box, so K of them compose into
//     z_j = G_j y + sum_{i=1..j} B_{j-i} sigma_i + Kc_j,    sigma_1 = sbox(x0),  sigma_{j+1} = sbox(z_j[0]),
//     A_1 = M, A_j = M E A_{j-1};  G_j = A_j E (y = the entering state);  B_0 = M[:,0], B_t = M E B_{t-1};
//     Kc_1 = c_1, Kc_j = M E Kc_{j-1} + c_j
// and only z_K is needed in full: rows 0 of G_1..G_{K-1} (24 multiply-adds each) plus ONE 12-row product with
// G_K instead of K of them.  The entries stay small integers (A_4 < 2^29, row weight < 2^31.8), so the same
// unreduced 32-bit-limb accumulation + single fold as the plain MDS layer applies: 438 v_mad_u64_u32 per 4
// rounds instead of 1152.  (plonky2's own "fast partial round" form reaches fewer multiplications but with
// full-width field constants, which on gfx950 costs more issue slots than this: profiles/r01_ubench_variants.txt.)
// MERGED: the block also swallows the MDS layer of the full round in front of it (y = that round's S-box outputs,
// x = M y + c_0, G_j = A_j E M, Kc_0 = c_0): one more 24-term row instead of a 12-row layer; K <= 3 there.
// Same permutation: tests/test_gpu_commit.py checks it against the oracle's naive schedule.
namespace pblk {
constexpr u64 PRIME = 0xFFFFFFFF00000001ULL;
constexpr u32 CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
constexpr u64 RC_CE[360] = { GLP_POSEIDON_RC_LIST };
constexpr u64 m_at(int r, int c) { return CIRC[(c - r + 12) % 12] + ((r == 0 && c == 0) ? 8u : 0u); }
template <int K, bool MERGED> struct Tab {
    u32 g0[K][12] = {};      // g0[j-1] = row 0 of G_j           (j = 1..K-1 used)
    u32 gK[12][12] = {};     // G_K
    u32 bt[K][12] = {};      // B_t, t = 0..K-1
    u64 max_weight = 0;      // largest total coefficient weight of one output (overflow bound)
};
template <int K, bool MERGED> constexpr Tab<K, MERGED> make_tab() {
    Tab<K, MERGED> t{};
    u64 A[12][12] = {}, N[12][12] = {}, G[12][12] = {};
    for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) A[r][c] = m_at(r, c);
    for (int j = 1; j <= K; j++) {
        if (j > 1) {
            for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) {
                u64 acc = 0;
                for (int k = 1; k < 12; k++) acc += m_at(r, k) * A[k][c];
                N[r][c] = acc;
            }
            for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) A[r][c] = N[r][c];
        }
        for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) {       // G_j = A_j E (M)
            u64 acc = 0;
            if (MERGED) { for (int k = 1; k < 12; k++) acc += A[r][k] * m_at(k, c); }
            else acc = c == 0 ? 0 : A[r][c];
            G[r][c] = acc;
        }
        if (j < K) for (int c = 0; c < 12; c++) t.g0[j - 1][c] = (u32)G[0][c];
    }
    for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) t.gK[r][c] = (u32)G[r][c];
    for (int r = 0; r < 12; r++) t.bt[0][r] = (u32)m_at(r, 0);
    for (int s = 1; s < K; s++)
        for (int r = 0; r < 12; r++) {
            u64 acc = 0;
            for (int k = 1; k < 12; k++) acc += m_at(r, k) * t.bt[s - 1][k];
            t.bt[s][r] = (u32)acc;
        }
    for (int r = 0; r < 12; r++) {
        u64 w = 0;
        for (int c = 0; c < 12; c++) { w += G[r][c]; if (G[r][c] >> 32) w = ~0ULL >> 1; }
        for (int i = 1; i <= K; i++) w += t.bt[K - i][r];
        if (w > t.max_weight) t.max_weight = w;
    }
    return t;
}
// per-block affine offsets (they depend on the block's round constants): kpre = c_0[0] (MERGED only),
// k0[j-1] = Kc_j[0] for j < K, kK = Kc_K
template <int K> struct BlkConst { u64 kpre; u64 k0[K]; u64 kK[12]; };
// the block does partial rounds first_round .. first_round+K-1; MERGED: plus the linear layer in front of them
template <int K, bool MERGED> constexpr BlkConst<K> make_blk(int first_round) {
    BlkConst<K> b{};
    u64 kc[12] = {}, nx[12] = {};
    for (int i = 0; i < 12; i++) kc[i] = RC_CE[12 * (first_round + (MERGED ? 0 : 1)) + i];
    b.kpre = MERGED ? kc[0] : 0;
    for (int j = MERGED ? 1 : 2; j <= K; j++) {
        for (int r = 0; r < 12; r++) {
            unsigned __int128 acc = RC_CE[12 * (first_round + j) + r];
            for (int k = 1; k < 12; k++) acc += (unsigned __int128)m_at(r, k) * kc[k];
            nx[r] = (u64)(acc % PRIME);
        }
        for (int r = 0; r < 12; r++) kc[r] = nx[r];
        if (j < K) b.k0[j - 1] = kc[0];
    }
    if (!MERGED && K > 1) b.k0[0] = RC_CE[12 * (first_round + 1)];
    b.k0[K - 1] = 0;
    for (int r = 0; r < 12; r++) b.kK[r] = kc[r];
    return b;
}
}  // namespace pblk

#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 schedule ---------------------------------------------------------------------------
// Same permutation, restructured for the VALU (the kernel is integer-issue bound, not HBM bound):
//  * values between layers are ANY u64 congruent to the field element (no canonicalisation until
//    the end), so no layer contains a compare-and-subtract;
//  * the next round's constant layer is folded into the MDS accumulators' initial value, so the
//    only stand-alone modular additions are the 12 of round 0;
//  * reductions are written on 32-bit limbs with carry builtins (v_add_co/v_addc chains) instead
//    of 64-bit compares + selects.
// Measured (profiles/r01_ubench_int_issue.txt): v_mad_u64_u32 and v_lshl_add_u64 issue at half
// rate on gfx950, so one MDS row (24 multiply-adds by 6-bit constants + one 96-bit fold) is the
// dominant cost.
static __device__ const u64 RC_ZERO[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ u64 sbox7_nc(u64 x) {
#ifdef GLP_SBOX_PLAIN_MUL
    const u64 x2 = mul_nc(x, x), x4 = mul_nc(x2, x2), x3 = mul_nc(x, x2);
    return mul_nc(x3, x4);
#else
    const u64 x2 = mul_nc_cc(x, x), x4 = mul_nc_cc(x2, x2), x3 = mul_nc_cc(x, x2);
    return mul_nc_cc(x3, x4);
#endif
}
// 12 S-boxes in groups of SBOX_GROUP: the scheduling barrier keeps hipcc from interleaving all 12 chains (which
// costs ~50 VGPRs and a wave of occupancy); within a group the chains still overlap
#ifndef GLP_MDS_GROUP
#define GLP_MDS_GROUP 4
#endif
#ifndef GLP_SBOX_GROUP
#define GLP_SBOX_GROUP 4
#endif
__device__ __forceinline__ void sbox_layer_nc(u64 s[12]) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
        s[i] = sbox7_nc(s[i]);
        if (i % GLP_SBOX_GROUP == GLP_SBOX_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
}
// s <- MDS * s + rc   (rc = the NEXT round's constants), inputs and outputs non-canonical
__device__ __forceinline__ void mds_add_nc(u64 s[12], const u64 *rc) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
    // 16 and 2 held in SGPRs the compiler cannot see through: otherwise it turns those terms into
    // v_lshl_add_u64 on a (limb, 0) register pair it has to build with two moves (3.75 issue slots against 1.75)
    u32 c16 = 16, c2 = 2;
    asm volatile("" : "+s"(c16), "+s"(c2));
#pragma unroll
    for (int r = 0; r < 12; r++) {
        // low halves first; the high-half chain then starts from the low chain's overflow, so the 96-bit row value
        // (ah : low 32 bits of al) needs no carry combine
        u64 al = (u64)(u32)rc[r] + (u64)lo[r] * (r == 0 ? C[0] + 8 : C[0]);
        asm("" : "+v"(al));            // keep the constant as the first multiply-add's addend (no separate 64-bit add)
#pragma unroll
        for (int i = 1; i < 12; i++) al += (u64)lo[(i + r) % 12] * (C[i] == 16 ? c16 : C[i] == 2 ? c2 : C[i]);
        u64 ah = (al >> 32) + (rc[r] >> 32);
#pragma unroll
        for (int i = 0; i < 12; i++) ah += (u64)hi[(i + r) % 12] * (i == 0 && r == 0 ? C[0] + 8 : C[i] == 16 ? c16 : C[i] == 2 ? c2 : C[i]);
#ifdef GLP_MDS_FOLD_C
        s[r] = fold96_c((ah << 32) | (u32)al, (u32)(ah >> 32));
#else
        s[r] = fold96_nc((ah << 32) | (u32)al, (u32)(ah >> 32));
#endif
        if (r % GLP_MDS_GROUP == GLP_MDS_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// rounds 3(linear layer)+4..6 merged, 7..22 as four blocks of 4, 23..25 as one block of 3
static __device__ const pblk::BlkConst<3> PBM[1] = {pblk::make_blk<3, true>(4)};
static __device__ const pblk::BlkConst<4> PB4[4] = {pblk::make_blk<4, false>(7), pblk::make_blk<4, false>(11),
                                                    pblk::make_blk<4, false>(15), pblk::make_blk<4, false>(19)};
static __device__ const pblk::BlkConst<3> PB3[1] = {pblk::make_blk<3, false>(23)};

// unreduced row: sum_c y_c * g[c] + sum_{i<ns} sigma_i * bcoef(i) + k, folded to a non-canonical u64
#define GLP_PB_ROW(OUT, GROW, KCONST, NS, BCOEF)                                                    \
    do {                                                                                            \
        const u64 _k = (KCONST);                                                                    \
        u64 al = (u32)_k;                                                                           \
        _Pragma("unroll") for (int c = 0; c < 12; c++) if ((GROW)[c] != 0) al += (u64)lo[c] * (GROW)[c];   \
        _Pragma("unroll") for (int i = 0; i < (NS); i++) al += (u64)slo[i] * (BCOEF);               \
        u64 ah = (al >> 32) + (_k >> 32);                                                           \
        _Pragma("unroll") for (int c = 0; c < 12; c++) if ((GROW)[c] != 0) ah += (u64)hi[c] * (GROW)[c];   \
        _Pragma("unroll") for (int i = 0; i < (NS); i++) ah += (u64)shi[i] * (BCOEF);               \
        (OUT) = fold96_nc((ah << 32) | (u32)al, (u32)(ah >> 32));                                   \
    } while (0)

// !MERGED: s = state entering partial round t with that round's constants already added.  MERGED: s = S-box outputs
// of the full round before partial round t.  On return: the state entering round t+K with ITS constants added.
// Non-canonical in and out.
// hook(j, z): z = the state element 0 entering the S-box of the block's j-th round (j = 0..K-1); what it returns goes through the S-box
// (the permutation: z itself; the PoseidonGate constraint evaluation: the S-box-input wire, after emitting z - wire).
template <int K, bool MERGED, class Hook>
__device__ __forceinline__ void partial_block_nc(u64 s[12], const pblk::BlkConst<K> &kc, Hook hook) {
    constexpr pblk::Tab<K, MERGED> T = pblk::make_tab<K, MERGED>();
    // al <= (2^32-1) W + 2^32 and ah <= 2^32 + 2^32 + (2^32-1) W must fit 64 bits, value < 2^96
    static_assert(T.max_weight + 2 < (1ULL << 32), "unreduced accumulation would overflow");
    u32 lo[12], hi[12], slo[K], shi[K];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
    u64 x0 = s[0];
    if constexpr (MERGED) {
        constexpr u32 MROW0[12] = {25, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};   // M[0,:]
        GLP_PB_ROW(x0, MROW0, kc.kpre, 0, 0u);
    }
    u64 sig = sbox7_nc(hook(0, x0));
    slo[0] = (u32)sig; shi[0] = (u32)(sig >> 32);
#pragma unroll
    for (int j = 1; j < K; j++) {           // z_j[0] -> sigma_{j+1}
        u64 z;
        GLP_PB_ROW(z, T.g0[j - 1], kc.k0[j - 1], j, T.bt[j - 1 - i][0]);
        sig = sbox7_nc(hook(j, z));
        slo[j] = (u32)sig; shi[j] = (u32)(sig >> 32);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {          // z_K
        GLP_PB_ROW(s[r], T.gK[r], kc.kK[r], K, T.bt[K - 1 - i][r]);
        if (r % GLP_MDS_GROUP == GLP_MDS_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int K, bool MERGED>
__device__ __forceinline__ void partial_block_nc(u64 s[12], const pblk::BlkConst<K> &kc) {
    partial_block_nc<K, MERGED>(s, kc, [](int, u64 z) { return z; });
}

// constants added by the full rounds' linear layers (= the NEXT round's constants): rounds 0..2 add those of rounds
// 1..3, rounds 26..28 those of 27..29, round 29 adds none; round 3's linear layer is inside the merged block
struct FullNext { u64 k[8][12]; };
constexpr FullNext make_full_next() {
    FullNext f{};
    for (int r = 0; r < 3; r++) for (int i = 0; i < 12; i++) { f.k[r][i] = pblk::RC_CE[12 * (r + 1) + i]; f.k[4 + r][i] = pblk::RC_CE[12 * (27 + r) + i]; }
    return f;
}
static __device__ const FullNext RCN = make_full_next();

// One copy of the S-box layer and of the MDS layer serves both halves (the two full-round phases are the same loop):
// the whole permutation is ~30 KB of code instead of ~62 KB, inside the 64 KB instruction cache a CU pair shares.
__device__ __forceinline__ void permute(u64 s[12]) {    // canonical in, canonical out
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = add(s[i], RC[i]);
#ifdef GLP_POSEIDON_PLAIN_PARTIAL
    int rc = 12;
    for (int r = 0; r < 4; r++) { sbox_layer_nc(s); mds_add_nc(s, RC + rc); rc += 12; }
    for (int r = 0; r < 22; r++) { s[0] = sbox7_nc(s[0]); mds_add_nc(s, RC + rc); rc += 12; }
    for (int r = 0; r < 3; r++) { sbox_layer_nc(s); mds_add_nc(s, RC + rc); rc += 12; }
    sbox_layer_nc(s);
    mds_add_nc(s, RC_ZERO);
#else
#ifdef GLP_POSEIDON_FLAT       // counting build for profiles/isa_slots.py: the same rounds with no loop at all
#define GLP_FULL_ROUND(i) sbox_layer_nc(s); mds_add_nc(s, RCN.k[i]);
    GLP_FULL_ROUND(0) GLP_FULL_ROUND(1) GLP_FULL_ROUND(2)
    sbox_layer_nc(s);
    partial_block_nc<3, true>(s, PBM[0]);
    partial_block_nc<4, false>(s, PB4[0]); partial_block_nc<4, false>(s, PB4[1]);
    partial_block_nc<4, false>(s, PB4[2]); partial_block_nc<4, false>(s, PB4[3]);
    partial_block_nc<3, false>(s, PB3[0]);
    GLP_FULL_ROUND(4) GLP_FULL_ROUND(5) GLP_FULL_ROUND(6) GLP_FULL_ROUND(7)
#undef GLP_FULL_ROUND
#else
#pragma nounroll
    for (int half = 0; half < 2; half++) {
#pragma nounroll
        for (int r = 0; r < 4; r++) {
            sbox_layer_nc(s);
            if (half == 0 && r == 3) break;                 // round 3's linear layer is part of the merged block
            mds_add_nc(s, RCN.k[4 * half + r]);
        }
        if (half == 0) {
            partial_block_nc<3, true>(s, PBM[0]);           // linear layer of round 3 + rounds 4..6
#pragma nounroll
            for (int b = 0; b < 4; b++) partial_block_nc<4, false>(s, PB4[b]);    // rounds 7..22
            partial_block_nc<3, false>(s, PB3[0]);          // rounds 23..25
        }
    }
#endif
#endif
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = canon(s[i]);
}
// ---- proof-of-work form -------------------------------------------------------------------------------
// The grinding loop permutes states that differ in ONE input element (the candidate) and looks at ONE output element.  Round 0
// therefore collapses to one S-box and twelve multiply-adds per candidate: with c_i = sbox(x_i + rc0_i) for the eleven fixed
// inputs, the state entering round 1's S-boxes is  K_r + M[r][pos] sbox(cand + rc0_pos),  K_r = rc1_r + sum_{i != pos} M[r][i] c_i
// (pow_round0_consts, once per proof); and the last linear layer is needed for row 7 only.  ~10 % fewer issue slots per candidate.
// st: the sponge state with the pending inputs written over it (canonical), pos: the slot the candidate goes to; out[r] = K_r (canonical)
__device__ __forceinline__ void pow_round0_consts(const u64 st[12], u32 pos, u64 out[12]) {
    u64 c[12];
    for (int i = 0; i < 12; i++) c[i] = (u32)i == pos ? 0 : canon(sbox7_nc(add(st[i], RC[i])));
    for (int r = 0; r < 12; r++) {
        u64 acc = RC[12 + r];
        for (int i = 0; i < 12; i++) if ((u32)i != pos) acc = add(acc, mul(c[i], (u64)mds_entry(r, i)));
        out[r] = acc;
    }
}
// s = the state entering the S-box layer of round 1 (any u64 congruent to it); returns output element 7, canonical.
// give_up() is asked at four points on the way (after rounds 3, 14, 25 and 27): true = the caller has no use for the result any more
// (a smaller witness is known), the function returns ~0 at once.
template <class GiveUp>
__device__ __forceinline__ u64 permute_tail7(u64 s[12], GiveUp give_up) {
#pragma nounroll
    for (int half = 0; half < 2; half++) {
#pragma nounroll
        for (int r = half == 0 ? 1 : 0; r < 4; r++) {
            sbox_layer_nc(s);
            if (r == 3) break;                              // half 0: round 3's linear layer is part of the merged block; half 1: row 7 only, below
            mds_add_nc(s, RCN.k[4 * half + r]);
            if (half == 1 && r == 1 && give_up()) return ~0ull;
        }
        if (half == 0) {
            if (give_up()) return ~0ull;
            partial_block_nc<3, true>(s, PBM[0]);
#pragma nounroll
            for (int b = 0; b < 4; b++) {
                partial_block_nc<4, false>(s, PB4[b]);
                if (b == 1 && give_up()) return ~0ull;
            }
            partial_block_nc<3, false>(s, PB3[0]);
            if (give_up()) return ~0ull;
        }
    }
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u64 al = 0, ah = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { al += (u64)(u32)s[(i + 7) % 12] * C[i]; ah += (u64)(u32)(s[(i + 7) % 12] >> 32) * C[i]; }
    ah += al >> 32;
    return canon(fold96_nc((ah << 32) | (u32)al, (u32)(ah >> 32)));
}
__device__ __forceinline__ u64 permute_tail7(u64 s[12]) { return permute_tail7(s, [] { return false; }); }
// ---- one permutation on 12 lanes --------------------------------------------------------------------
// Latency form for the small levels near the top of a Merkle tree, the small FRI layers and the Fiat-Shamir transcripts of a batch, where
// there are far fewer hashes than lanes: lane l (0..11 of a 16-lane group = one DPP row) owns state element l and S-boxes run 12-wide.
// The linear layer is row l of the matrix against the whole state: element j reaches every lane of the row by a DPP row broadcast
// (v_mov_b64 row_newbcast:j, no LDS) and is multiplied by the lane's own coefficient M[l][j], held in twelve registers.
// (The first form rotated the data instead -- 24 ds_bpermute per round with their LDS latency: ~16 us per permutation against ~11.)
// A wave running alone pays ~4 clocks per VALU instruction and 8 per multiply-add, so what counts is the instruction count of one lane:
// ~120 per round here against ~470 with one state per lane.  Lower throughput (partial rounds idle 11 lanes), so the big levels keep the
// one-state-per-lane kernel.  x: canonical in, canonical out; all 64 lanes must call it.
static __device__ const u32 MDS_C24[24] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20, 17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
template <int J> __device__ __forceinline__ u64 row_bcast(u64 v) {     // every lane of a 16-lane row receives lane J's v (v_mov_b64_dpp)
    return (u64)__builtin_amdgcn_update_dpp((long long)0, (long long)v, 0x150 + J, 0xf, 0xf, true);
}
template <int J> __device__ __forceinline__ void coop_row_terms(u64 &al, u64 &ah, u64 x, const u32 (&cf)[12]) {
    if constexpr (J < 12) {
        const u64 b = row_bcast<J>(x);
        al += (u64)(u32)b * cf[J];
        ah += (u64)(u32)(b >> 32) * cf[J];
        coop_row_terms<J + 1>(al, ah, x, cf);
    }
}
__device__ __forceinline__ u64 permute_coop(u64 x, int l /* lane in group, 0..15 */, int /* group_base: first lane of the group in the wave */) {
    const int ll = l < 12 ? l : 0;
    u32 cf[12];                                                 // cf[j] = M[l][j] = C[(j - l) mod 12] (+ 8 at [0][0])
#pragma unroll
    for (int j = 0; j < 12; j++) cf[j] = MDS_C24[12 + j - ll];
    if (l == 0) cf[0] += 8;
    x = add(x, RC[ll]);
    for (int r = 0; r < 30; r++) {
        const bool full = r < 4 || r >= 26;
        const u64 sb = sbox7_nc(x);
        if (full || l == 0) x = sb;
        const u64 rcn = r < 29 ? RC[12 * (r + 1) + ll] : 0;
        u64 al = (u32)rcn, ah = rcn >> 32;
        coop_row_terms<0>(al, ah, x, cf);
        ah += al >> 32;
        x = fold96_nc((ah << 32) | (u32)al, (u32)(ah >> 32));
    }
    return canon(x);
}
// ---- one permutation on 4 lanes -----------------------------------------------------------------------
// Between the two forms above: lane q of a quad (4 adjacent lanes) owns state elements 3q, 3q+1, 3q+2.  The circulant row of
// element 3q+a reads x[(3q + a + i) % 12] = slot (a + i) % 3 of the lane ((a + i) / 3) places further round the quad: the nine foreign
// words arrive by DPP quad rotations (v_mov_b32 quad_perm, full rate, no LDS), the coefficients stay compile-time constants.
// Per lane a full round is 3 S-boxes + 72 multiply-adds, a partial round 1 + 72 -- about 7.6 k instructions per permutation on
// 4 lanes against 14 k on one and 5.5 k on sixteen: for trees of 2^12..2^16 leaves, where one state per lane leaves most SIMDs
// empty and the 12-lane form has more lanes than the chip has slots.  x[s]: canonical in, canonical out; whole quads must call it.
template <int D> __device__ __forceinline__ u32 quad_rot(u32 v) {      // lane q receives v of lane (q + D) % 4 of its quad
    constexpr int CTRL = (D & 3) | (((D + 1) & 3) << 2) | (((D + 2) & 3) << 4) | (((D + 3) & 3) << 6);
    return (u32)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ void permute_quad(u64 x[3], int q /* lane in quad, 0..3 */) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const u64 *rc = RC + 3 * q;
    const u32 diag = q == 0 ? 8u : 0u;                          // M[0][0] = C[0] + 8
#pragma unroll
    for (int s = 0; s < 3; s++) x[s] = add(x[s], rc[s]);
#pragma nounroll
    for (int r = 0; r < 30; r++) {
        const bool full = r < 4 || r >= 26;
        const u64 sb = sbox7_nc(x[0]);
        if (full) { x[0] = sb; x[1] = sbox7_nc(x[1]); x[2] = sbox7_nc(x[2]); }
        else if (q == 0) x[0] = sb;
        u32 lo[4][3], hi[4][3];
#pragma unroll
        for (int s = 0; s < 3; s++) {
            lo[0][s] = (u32)x[s]; hi[0][s] = (u32)(x[s] >> 32);
            lo[1][s] = quad_rot<1>(lo[0][s]); hi[1][s] = quad_rot<1>(hi[0][s]);
            lo[2][s] = quad_rot<2>(lo[0][s]); hi[2][s] = quad_rot<2>(hi[0][s]);
            lo[3][s] = quad_rot<3>(lo[0][s]); hi[3][s] = quad_rot<3>(hi[0][s]);
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const u64 rcn = r < 29 ? rc[12 * (r + 1) + a] : 0;
            u64 al = (u32)rcn;
#pragma unroll
            for (int i = 0; i < 12; i++) al += (u64)lo[((a + i) / 3) & 3][(a + i) % 3] * C[i];
            if (a == 0) al += (u64)lo[0][0] * diag;
            u64 ah = (al >> 32) + (rcn >> 32);
#pragma unroll
            for (int i = 0; i < 12; i++) ah += (u64)hi[((a + i) / 3) & 3][(a + i) % 3] * C[i];
            if (a == 0) ah += (u64)hi[0][0] * diag;
            x[a] = fold96_nc((ah << 32) | (u32)al, (u32)(ah >> 32));
        }
    }
#pragma unroll
    for (int s = 0; s < 3; s++) x[s] = canon(x[s]);
}
// 64-bit wavefront shuffles (two ds_bpermute each) for the cooperative forms
__device__ __forceinline__ u64 shfl64(u64 v, int src) {
    const u32 lo = (u32)__shfl((int)(u32)v, src, 64), hi = (u32)__shfl((int)(u32)(v >> 32), src, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_xor64(u64 v, int m) {
    const u32 lo = (u32)__shfl_xor((int)(u32)v, m, 64), hi = (u32)__shfl_xor((int)(u32)(v >> 32), m, 64);
    return ((u64)hi << 32) | lo;
}
#else
// ---- host schedule (the Fiat-Shamir transcripts of glp_prove / glp_verify and the verifiers' Merkle paths) -----------------------------
// Same permutation as permute_ref with the 22 partial rounds in the blocks of namespace pblk (3 merged with round 3's linear layer, 4 x 4, 3):
// one 12-row product per block instead of one per round.  What bounds a block on a CPU core is its chain of S-boxes (each four dependent
// 64 x 64 multiplications and reductions) with a 24-term row between two of them: 3.5 -> 2.3 us per permutation on the build machine.
// Full rounds keep the reference layers (hipcc's host pass vectorises mds_layer's unrolled rows; branch-free reductions measured slower).
namespace host {
inline u64 sbox(u64 x) { return sbox7(x); }              // any u64 in, canonical out (glf::mul: rarely-taken branches predict well on a CPU)
// (al + 2^32 ah + k) mod p as some u64: al, ah unreduced sums of 32-bit-half products with total weight < 2^32
inline u64 fold(u64 al, u64 ah, u64 k) {
    al += (u32)k;
    const u64 h = ah + (al >> 32) + (k >> 32);
    const u64 l = (h << 32) | (u32)al, t1 = (h >> 32) * EPS;
    u64 t2 = l + t1;
    if (t2 < t1) t2 += EPS;
    return t2;
}
// s <- MDS s + rc   (rc canonical: the next round's constants, or zeros); canonical out
inline void mds_add(u64 s[12], const u64 *rc) {
    mds_layer(s);
    for (int i = 0; i < 12; i++) s[i] = add(s[i], rc[i]);
}
// K partial rounds as one block (the tables of namespace pblk, shared with the gfx950 schedule): s = the state entering partial round t with
// that round's constants added (MERGED: the S-box outputs of the full round before it); on return the state entering round t + K with its
// constants.  Every row is an unreduced sum of 32-bit-half products (weights < 2^32) folded once.
template <int K, bool MERGED>
inline void partial_block(u64 s[12], const pblk::BlkConst<K> &kc) {
    constexpr pblk::Tab<K, MERGED> T = pblk::make_tab<K, MERGED>();
    static_assert(T.max_weight + 2 < (1ULL << 32), "unreduced accumulation would overflow");
    u32 lo[12], hi[12], slo[K], shi[K];
    for (int i = 0; i < 12; i++) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
    u64 x0 = s[0];
    if (MERGED) {
        constexpr u32 MROW0[12] = {25, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
        u64 al = 0, ah = 0;
        _Pragma("unroll") for (int c = 0; c < 12; c++) { al += (u64)lo[c] * MROW0[c]; ah += (u64)hi[c] * MROW0[c]; }
        x0 = fold(al, ah, kc.kpre);
    }
    u64 sig = sbox(x0);
    slo[0] = (u32)sig; shi[0] = (u32)(sig >> 32);
    _Pragma("unroll") for (int j = 1; j < K; j++) {
        u64 al = 0, ah = 0;
        _Pragma("unroll") for (int c = 0; c < 12; c++) { al += (u64)lo[c] * T.g0[j - 1][c]; ah += (u64)hi[c] * T.g0[j - 1][c]; }
        _Pragma("unroll") for (int i = 0; i < j; i++) { al += (u64)slo[i] * T.bt[j - 1 - i][0]; ah += (u64)shi[i] * T.bt[j - 1 - i][0]; }
        sig = sbox(fold(al, ah, kc.k0[j - 1]));
        slo[j] = (u32)sig; shi[j] = (u32)(sig >> 32);
    }
    _Pragma("unroll") for (int r = 0; r < 12; r++) {
        u64 al = 0, ah = 0;
        _Pragma("unroll") for (int c = 0; c < 12; c++) { al += (u64)lo[c] * T.gK[r][c]; ah += (u64)hi[c] * T.gK[r][c]; }
        _Pragma("unroll") for (int i = 0; i < K; i++) { al += (u64)slo[i] * T.bt[K - 1 - i][r]; ah += (u64)shi[i] * T.bt[K - 1 - i][r]; }
        s[r] = fold(al, ah, kc.kK[r]);
    }
}
static constexpr pblk::BlkConst<3> HPBM = pblk::make_blk<3, true>(4);
static constexpr pblk::BlkConst<4> HPB4[4] = {pblk::make_blk<4, false>(7), pblk::make_blk<4, false>(11), pblk::make_blk<4, false>(15), pblk::make_blk<4, false>(19)};
static constexpr pblk::BlkConst<3> HPB3 = pblk::make_blk<3, false>(23);
}  // namespace host
inline void permute(u64 s[12]) {                         // canonical in, canonical out
    static const u64 ZERO12[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 12; i++) s[i] = add(s[i], RC[i]);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) s[i] = host::sbox(s[i]);
        if (r < 3) host::mds_add(s, RC + 12 * (r + 1));      // round 3's linear layer is inside the merged block
    }
    host::partial_block<3, true>(s, host::HPBM);             // rounds 4..6
    for (int b = 0; b < 4; b++) host::partial_block<4, false>(s, host::HPB4[b]);     // rounds 7..22
    host::partial_block<3, false>(s, host::HPB3);            // rounds 23..25
    for (int r = 26; r < 30; r++) {
        for (int i = 0; i < 12; i++) s[i] = host::sbox(s[i]);
        host::mds_add(s, r < 29 ? RC + 12 * (r + 1) : ZERO12);
    }
    for (int i = 0; i < 12; i++) s[i] = canon(canon(s[i]));
}
inline u64 permute_coop(u64 x, int, int) { return x; }   // device-only; declared for the host parsing pass
inline void permute_quad(u64 *, int) {}
// device-only below; declared for the host parsing pass
GLF_HD u64 sbox7_nc(u64 x) { return x; }
GLF_HD void pow_round0_consts(const u64 *, u32, u64 *) {}
GLF_HD u64 permute_tail7(u64 *) { return 0; }
template <class GiveUp> GLF_HD u64 permute_tail7(u64 *, GiveUp) { return 0; }
GLF_HD u64 shfl64(u64 v, int) { return v; }
GLF_HD u64 shfl_xor64(u64 v, int) { return v; }
#endif

// hashing.rs `compress` (= Hasher::two_to_one): perm(l || r || 0000)[0..4]
GLF_HD void two_to_one(const u64 l[4], const u64 r[4], u64 out[4]) {
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 4; i++) { s[i] = l[i]; s[4 + i] = r[i]; s[8 + i] = 0; }
    permute(s);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = s[i];
}

}  // namespace pos
