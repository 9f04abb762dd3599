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
    const u64 x2 = mul_nc(x, x), x4 = mul_nc(x2, x2), x3 = mul_nc(x, x2);
    return mul_nc(x3, x4);
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
#pragma unroll
    for (int r = 0; r < 12; r++) {
        // low halves first; the high-half chain then starts from the low chain's overflow, so the 96-bit row value
        // (ah : low 32 bits of al) needs no carry combine
        u64 al = (u32)rc[r];
#pragma unroll
        for (int i = 0; i < 12; i++) al += (u64)lo[(i + r) % 12] * C[i];
        if (r == 0) al += (u64)lo[0] * 8;
        u64 ah = (al >> 32) + (rc[r] >> 32);
#pragma unroll
        for (int i = 0; i < 12; i++) ah += (u64)hi[(i + r) % 12] * C[i];
        if (r == 0) ah += (u64)hi[0] * 8;
#ifdef GLP_MDS_FOLD_C
        s[r] = fold96_c((ah << 32) | (u32)al, (u32)(ah >> 32));
#else
        s[r] = fold96_nc((ah << 32) | (u32)al, (u32)(ah >> 32));
#endif
        if (r % GLP_MDS_GROUP == GLP_MDS_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void permute(u64 s[12]) {    // canonical in, canonical out
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = add(s[i], RC[i]);
    int rc = 12;
    for (int r = 0; r < 4; r++) {
        sbox_layer_nc(s);
        mds_add_nc(s, RC + rc);
        rc += 12;
    }
    for (int r = 0; r < 22; r++) {
        s[0] = sbox7_nc(s[0]);
        mds_add_nc(s, RC + rc);
        rc += 12;
    }
    for (int r = 0; r < 3; r++) {
        sbox_layer_nc(s);
        mds_add_nc(s, RC + rc);
        rc += 12;
    }
    sbox_layer_nc(s);
    mds_add_nc(s, RC_ZERO);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = canon(s[i]);
}
// ---- one permutation on 12 lanes --------------------------------------------------------------------
// Latency form for the small levels near the top of a Merkle tree and the small FRI layers, where there are far
// fewer hashes than lanes: lane l (0..11 of a 16-lane group) owns state element l, S-boxes run 12-wide, and the
// circulant MDS row of lane l gathers x[(i + l) % 12] from its neighbours with wavefront shuffles (ds_bpermute).
// About 5x lower latency per hash than one-state-per-lane; lower throughput (partial rounds idle 11 lanes), so the
// big levels keep the one-state-per-lane kernel.  x: canonical in, canonical out; all 64 lanes must call it.
__device__ __forceinline__ u64 permute_coop(u64 x, int l /* lane in group, 0..15 */, int group_base /* first lane of the group in the wave */) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const int ll = l < 12 ? l : 0;
    x = add(x, RC[ll]);
    for (int r = 0; r < 30; r++) {
        const bool full = r < 4 || r >= 26;
        const u64 sb = sbox7_nc(x);
        if (full || l == 0) x = sb;
        const u64 rcn = r < 29 ? RC[12 * (r + 1) + ll] : 0;
        const u32 xlo = (u32)x, xhi = (u32)(x >> 32);
        u64 al = (u32)rcn;
        u32 hi_g[12];
#pragma unroll
        for (int i = 0; i < 12; i++) {
            int src = ll + i; src = src >= 12 ? src - 12 : src;
            const u32 lo_i = (u32)__shfl((int)xlo, group_base + src, 64);
            hi_g[i] = (u32)__shfl((int)xhi, group_base + src, 64);
            al += (u64)lo_i * C[i];
        }
        if (l == 0) al += (u64)xlo * 8;
        u64 ah = (al >> 32) + (rcn >> 32);
#pragma unroll
        for (int i = 0; i < 12; i++) ah += (u64)hi_g[i] * C[i];
        if (l == 0) ah += (u64)xhi * 8;
        x = fold96_nc((ah << 32) | (u32)al, (u32)(ah >> 32));
    }
    return canon(x);
}
#else
inline void permute(u64 s[12]) { permute_ref(s); }
inline u64 permute_coop(u64 x, int, int) { return x; }   // device-only; declared for the host parsing pass
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
