/*
 * oracle/gl_keccak.h -- TEST INFRASTRUCTURE, not product code.
 * Keccak-256 (original Keccak padding 0x01 .. 0x80, rate 136) and the plonky2 0.1.4 `KeccakHash<25>` / `KeccakPermutation`
 * conventions built on it (plonky2/src/hash/keccak.rs; crate absent from /root/reference, restated from the published source):
 *   hash_no_pad(elements)  = first 25 bytes of keccak256(8 little-endian bytes per element)
 *   hash_or_noop(elements) = the elements' bytes zero-padded to 25 if they fit (<= 3 elements), else hash_no_pad
 *   two_to_one(l, r)       = first 25 bytes of keccak256(l || r)  (50 bytes)
 *   BytesHash<25>::to_vec  = 7-byte little-endian chunks as field elements (4 of them: 7 + 7 + 7 + 4 bytes)
 *   KeccakPermutation      = state (12 elements, 96 bytes) -> keccak256 -> keccak256 of that -> ...; the 8-byte little-endian words
 *                            of the hash chain that are < p, first 12 of them (rejection sampling)
 * Pinned by: the Keccak-256 (input, digest) pairs the reference checks natively against the `sha3` crate
 * [REF src/hash/keccak256.rs:196-212,256-277] fix the primitive (tests/test_oracle_keccak.py).  The conventions above (truncation to
 * 25 bytes, 7-byte chunking, hash-chain permutation) are RECALLED: parity unpinned, like the rest of the proof format.
 * A 25-byte digest travels as 4 u64 words, little-endian, the top 7 bytes of the last word zero.
 */
#ifndef GL_KECCAK_H
#define GL_KECCAK_H
#include "gl_field.h"
void glo_keccak_f1600(u64 st[25]);
void glo_keccak256(const unsigned char *msg, size_t len, unsigned char out[32]);
void glo_keccak_hash_no_pad(const u64 *in, size_t len, u64 out[4]);
void glo_keccak_hash_or_noop(const u64 *in, size_t len, u64 out[4]);
void glo_keccak_two_to_one(const u64 l[4], const u64 r[4], u64 out[4]);
void glo_keccak_hash_to_elements(const u64 h[4], u64 out[4]);
void glo_keccak_permute(u64 st[12]);
#endif
