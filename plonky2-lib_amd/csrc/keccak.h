// keccak.h -- Keccak-256 and plonky2's `KeccakHash<25>` / `KeccakPermutation` built on it, for gfx950 device code and the
// host-side transcript: the hasher of `KeccakGoldilocksConfig`, which one driver of the reference selects
// [REF src/hash/keccak256.rs:281 `type C = KeccakGoldilocksConfig`] (every other driver uses PoseidonGoldilocksConfig).
//
// Replaces plonky2 0.1.4 `hash/keccak.rs` (crate absent from /root/reference; restated from the published source):
//   KeccakHash<25>::hash_no_pad(elements) = first 25 bytes of keccak256(8 little-endian bytes per element)
//   hash_or_noop: <= 3 elements are copied (zero-padded to 25 bytes), more are hashed
//   two_to_one(l, r) = first 25 bytes of keccak256(l || r)
//   BytesHash<25>::to_vec: 7-byte little-endian chunks as field elements (how a digest enters the transcript)
//   KeccakPermutation: 12 elements -> keccak256 -> keccak256 of that -> ..., the hash chain's 8-byte words below p, first 12
// A 25-byte digest is kept as 4 u64 words (little-endian; the top 7 bytes of word 3 are zero), the same slot size as a
// Poseidon HashOut, so digest arrays, caps, paths and gathers are hasher-independent.  Field elements ARE 64-bit Keccak lanes
// (8 little-endian bytes each): absorbing a row of the LDE matrix is 17 lane XORs per 136-byte rate block.
// The Keccak-256 primitive is pinned by the reference's (input, digest) pairs [REF src/hash/keccak256.rs:196-212,256-277]
// (tests/test_oracle_keccak.py for the checker, tests/test_gpu_keccak.py for this file through the C ABI).
#pragma once
#include "glf.h"

namespace kec {
using namespace glf;

constexpr int RATE_LANES = 17;                  // 136-byte rate of Keccak-256
constexpr u64 DIGEST_TOP_MASK = 0xFFull;        // word 3 of a 25-byte digest keeps one byte

GLF_HD u64 rotl(u64 x, int n) { return (x << n) | (x >> (64 - n)); }

// Keccak-f[1600], lanes a[x + 5 y].  One round written out with the rho offsets and the pi permutation as literals.
GLF_HD void f1600(u64 (&a)[25]) {
    constexpr u64 RC[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
        0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
        0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
        0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        // theta
        const u64 c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20], c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21];
        const u64 c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22], c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23];
        const u64 c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];
        const u64 d0 = c4 ^ rotl(c1, 1), d1 = c0 ^ rotl(c2, 1), d2 = c1 ^ rotl(c3, 1), d3 = c2 ^ rotl(c4, 1), d4 = c3 ^ rotl(c0, 1);
        // rho + pi: b[y + 5 ((2x + 3y) mod 5)] = rotl(a[x + 5y] ^ d[x], r[x][y])
        u64 b[25];
        b[0] = a[0] ^ d0;
        b[10] = rotl(a[1] ^ d1, 1);   b[20] = rotl(a[2] ^ d2, 62);  b[5] = rotl(a[3] ^ d3, 28);   b[15] = rotl(a[4] ^ d4, 27);
        b[16] = rotl(a[5] ^ d0, 36);  b[1] = rotl(a[6] ^ d1, 44);   b[11] = rotl(a[7] ^ d2, 6);   b[21] = rotl(a[8] ^ d3, 55);
        b[6] = rotl(a[9] ^ d4, 20);   b[7] = rotl(a[10] ^ d0, 3);   b[17] = rotl(a[11] ^ d1, 10); b[2] = rotl(a[12] ^ d2, 43);
        b[12] = rotl(a[13] ^ d3, 25); b[22] = rotl(a[14] ^ d4, 39); b[23] = rotl(a[15] ^ d0, 41); b[8] = rotl(a[16] ^ d1, 45);
        b[18] = rotl(a[17] ^ d2, 15); b[3] = rotl(a[18] ^ d3, 21);  b[13] = rotl(a[19] ^ d4, 8);  b[14] = rotl(a[20] ^ d0, 18);
        b[24] = rotl(a[21] ^ d1, 2);  b[9] = rotl(a[22] ^ d2, 61);  b[19] = rotl(a[23] ^ d3, 56); b[4] = rotl(a[24] ^ d4, 14);
        // chi
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            a[y] = b[y] ^ (~b[y + 1] & b[y + 2]);
            a[y + 1] = b[y + 1] ^ (~b[y + 2] & b[y + 3]);
            a[y + 2] = b[y + 2] ^ (~b[y + 3] & b[y + 4]);
            a[y + 3] = b[y + 3] ^ (~b[y + 4] & b[y]);
            a[y + 4] = b[y + 4] ^ (~b[y] & b[y + 1]);
        }
        a[0] ^= RC[round];      // iota
    }
}

// Sponge over 64-bit lanes (= field elements).  `fill` counts the lanes absorbed into the current rate block.
struct Sponge {
    u64 a[25];
    int fill;
};
GLF_HD void sponge_init(Sponge &s) {
#pragma unroll
    for (int i = 0; i < 25; i++) s.a[i] = 0;
    s.fill = 0;
}
// XOR one lane into position `fill` of the rate (written as a select chain: no dynamic register indexing on the device)
GLF_HD void sponge_absorb(Sponge &s, u64 lane) {
#pragma unroll
    for (int i = 0; i < RATE_LANES; i++) if (i == s.fill) s.a[i] ^= lane;
    if (++s.fill == RATE_LANES) { f1600(s.a); s.fill = 0; }
}
// pad10*1 of the original Keccak (0x01 ... 0x80) after a whole number of lanes plus `extra_bytes` (< 8) bytes `extra` of a partial lane
GLF_HD void sponge_finish(Sponge &s, u64 extra = 0, int extra_bytes = 0) {
    const u64 tail = extra | ((u64)0x01 << (8 * extra_bytes));
#pragma unroll
    for (int i = 0; i < RATE_LANES; i++) if (i == s.fill) s.a[i] ^= tail;
    s.a[RATE_LANES - 1] ^= 0x8000000000000000ull;
    f1600(s.a);
}
// first 25 bytes of the squeezed hash as a 4-word digest
GLF_HD void sponge_digest25(const Sponge &s, u64 out[4]) {
    out[0] = s.a[0]; out[1] = s.a[1]; out[2] = s.a[2]; out[3] = s.a[3] & DIGEST_TOP_MASK;
}

// KeccakHash<25>::two_to_one on 4-word digests: keccak256(l[0..25) || r[0..25)), 50 bytes = 6 lanes + 2 bytes
GLF_HD void two_to_one(const u64 l[4], const u64 r[4], u64 out[4]) {
    Sponge s;
    sponge_init(s);
    s.a[0] = l[0]; s.a[1] = l[1]; s.a[2] = l[2];
    s.a[3] = (l[3] & 0xFF) | (r[0] << 8);
    s.a[4] = (r[0] >> 56) | (r[1] << 8);
    s.a[5] = (r[1] >> 56) | (r[2] << 8);
    s.fill = 6;
    sponge_finish(s, (r[2] >> 56) | ((r[3] & 0xFF) << 8), 2);
    sponge_digest25(s, out);
}

// BytesHash<25>::to_vec: bytes 0..6, 7..13, 14..20, 21..24 as little-endian integers
GLF_HD void digest_to_elements(const u64 h[4], u64 out[4]) {
    constexpr u64 M56 = 0x00FFFFFFFFFFFFFFull;
    out[0] = h[0] & M56;
    out[1] = ((h[0] >> 56) | (h[1] << 8)) & M56;
    out[2] = ((h[1] >> 48) | (h[2] << 16)) & M56;
    out[3] = ((h[2] >> 40) | ((h[3] & 0xFF) << 24)) & 0xFFFFFFFFull;
}

// KeccakPermutation::permute: st <- first 12 words below p of keccak256(st bytes), keccak256(that), ...
GLF_HD void permute(u64 (&st)[12]) {
    u64 outv[12];
    int got = 0;
    Sponge s;
    sponge_init(s);
#pragma unroll
    for (int i = 0; i < 12; i++) s.a[i] = st[i];
    s.fill = 12;
    sponge_finish(s);
    for (;;) {
        const u64 h0 = s.a[0], h1 = s.a[1], h2 = s.a[2], h3 = s.a[3];
        const u64 w[4] = {h0, h1, h2, h3};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (w[k] < P && got < 12) {
#pragma unroll
                for (int j = 0; j < 12; j++) if (j == got) outv[j] = w[k];
                got++;
            }
        }
        if (got >= 12) break;
        sponge_init(s);
        s.a[0] = h0; s.a[1] = h1; s.a[2] = h2; s.a[3] = h3;
        s.fill = 4;
        sponge_finish(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = outv[i];
}

// host-side helpers for the transcript and the verifier (short inputs)
inline void host_hash_no_pad(const u64 *in, size_t len, u64 out[4]) {
    Sponge s;
    sponge_init(s);
    for (size_t i = 0; i < len; i++) sponge_absorb(s, in[i]);
    sponge_finish(s);
    sponge_digest25(s, out);
}
inline void host_hash_or_noop(const u64 *in, size_t len, u64 out[4]) {
    if (8 * len <= 25) { for (int i = 0; i < 4; i++) out[i] = (size_t)i < len ? in[i] : 0; return; }
    host_hash_no_pad(in, len, out);
}

}  // namespace kec
