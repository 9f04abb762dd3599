/*
 * oracle/gl_proof_bytes.c -- TEST INFRASTRUCTURE, not product code.
 * `ProofWithPublicInputs::to_bytes()` of plonky2 0.1.4 (util/serialization.rs, crate absent from /root/reference; the
 * reference round-trips circuits through the same `Buffer` machinery at [REF src/ecdsa/gadgets/ecdsa.rs:298-316] and builds
 * its gate / generator serializers on it [REF src/ecdsa/serialization.rs:7-46]).  RECALLED, UNPINNED: the reference holds no
 * proof bytes, so this restatement and the product's glp_proof_to_bytes can only be checked against each other -- which is the
 * point of keeping them structurally independent: the product walks a flat list of (offset, count, kind) pieces
 * (csrc/prover.hip walk_proof), this file follows the Rust writer's call tree, one function per `Write` method:
 *
 *   write_proof_with_public_inputs = write_proof ; write_field_vec(public_inputs)            (no length prefix: `read_field_vec(len)`)
 *   write_proof           = write_merkle_cap x3 (wires, plonk_zs_partial_products, quotient_polys) ; write_opening_set ; write_fri_proof
 *   write_merkle_cap      = write_hash per cap entry (count implied by cap_height)
 *   write_hash            = H::Hash::to_bytes(): HashOut = 4 x 8 little-endian bytes; BytesHash<25> = its 25 bytes
 *   write_opening_set     = write_field_ext_vec of constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products,
 *                           quotient_polys (lookup_zs / next_lookup_zs are empty without lookup tables: zero bytes)
 *   write_fri_proof       = commit_phase_merkle_caps ; write_fri_query_round per query ; write_field_ext_vec(final_poly.coeffs) ;
 *                           write_field(pow_witness)
 *   write_fri_query_round = write_fri_initial_proof ; per reduction: write_field_ext_vec(evals) ; write_merkle_proof
 *   write_fri_initial_proof = per oracle (constants_sigmas, wires, zs_partial_products, quotient): write_field_vec(evals) ; write_merkle_proof
 *   write_merkle_proof    = write_u8(siblings.len()) ; write_hash per sibling
 *   write_field           = to_canonical_u64().to_le_bytes() ; write_field_ext = its D = 2 coefficients
 *
 * Input: the proof as flat words in the order gl_prover_oracle.c documents (which is this same order with every digest in a
 * 4-word slot).
 */
#include <string.h>
#include "gl_circuit.h"

#define API __attribute__((visibility("default")))
int glo_get_hasher(void);

typedef struct { const u64 *w; unsigned char *out; size_t cap, len; int digest_bytes; } wr_t;

static void put(wr_t *b, const void *p, size_t n) {
    if (b->out && b->len + n <= b->cap) memcpy(b->out + b->len, p, n);
    b->len += n;                                      /* keeps counting past cap: the caller sees the size it needed */
}
static void write_u8(wr_t *b, unsigned v) { unsigned char c = (unsigned char)v; put(b, &c, 1); }
static void write_field(wr_t *b) {
    unsigned char le[8];
    u64 v = *b->w++;
    for (int i = 0; i < 8; i++) le[i] = (unsigned char)(v >> (8 * i));
    put(b, le, 8);
}
static void write_field_vec(wr_t *b, size_t n) { for (size_t i = 0; i < n; i++) write_field(b); }
static void write_field_ext_vec(wr_t *b, size_t n) { for (size_t i = 0; i < n; i++) { write_field(b); write_field(b); } }
static void write_hash(wr_t *b) {                     /* one digest = 4 words in; 32 (HashOut) or 25 (BytesHash<25>) bytes out */
    unsigned char le[32];
    for (int k = 0; k < 4; k++) for (int i = 0; i < 8; i++) le[8 * k + i] = (unsigned char)(b->w[k] >> (8 * i));
    b->w += 4;
    put(b, le, (size_t)b->digest_bytes);
}
static void write_merkle_cap(wr_t *b, u32 cap_height) { for (u32 i = 0; i < (1u << cap_height); i++) write_hash(b); }
static void write_merkle_proof(wr_t *b, u32 siblings) {
    write_u8(b, siblings);
    for (u32 i = 0; i < siblings; i++) write_hash(b);
}
static void write_opening_set(wr_t *b, const glo_circuit *c) {
    const u32 nch = c->num_challenges;
    write_field_ext_vec(b, c->num_constants);                         /* constants */
    write_field_ext_vec(b, c->num_routed_wires);                      /* plonk_sigmas */
    write_field_ext_vec(b, c->num_wires);                             /* wires */
    write_field_ext_vec(b, nch);                                      /* plonk_zs */
    write_field_ext_vec(b, nch);                                      /* plonk_zs_next */
    write_field_ext_vec(b, (size_t)nch * c->num_partial_products);    /* partial_products */
    write_field_ext_vec(b, (size_t)nch * c->quotient_degree_factor);  /* quotient_polys */
}
static void write_fri_initial_proof(wr_t *b, const glo_circuit *c) {
    const u32 nch = c->num_challenges, depth = c->degree_bits + c->rate_bits - c->cap_height;
    const size_t cols[4] = {(size_t)c->num_constants + c->num_routed_wires, c->num_wires, (size_t)nch * (1 + c->num_partial_products),
                            (size_t)nch * c->quotient_degree_factor};
    for (int k = 0; k < 4; k++) { write_field_vec(b, cols[k]); write_merkle_proof(b, depth); }
}
static void write_fri_query_round(wr_t *b, const glo_circuit *c) {
    write_fri_initial_proof(b, c);
    u32 lg = c->degree_bits + c->rate_bits;
    for (u32 r = 0; r < c->num_reductions; r++) {                     /* FriQueryStep: evals of one coset, path in the layer's tree */
        const u32 ab = c->reduction_arity_bits[r];
        lg -= ab;
        write_field_ext_vec(b, (size_t)1 << ab);
        write_merkle_proof(b, lg - c->cap_height);
    }
}
static void write_fri_proof(wr_t *b, const glo_circuit *c) {
    u32 sum_ab = 0;
    for (u32 r = 0; r < c->num_reductions; r++) { write_merkle_cap(b, c->cap_height); sum_ab += c->reduction_arity_bits[r]; }
    for (u32 q = 0; q < c->num_query_rounds; q++) write_fri_query_round(b, c);
    write_field_ext_vec(b, (size_t)1 << (c->degree_bits - sum_ab));   /* final_poly.coeffs */
    write_field(b);                                                   /* pow_witness */
}
static void write_proof(wr_t *b, const glo_circuit *c) {
    write_merkle_cap(b, c->cap_height);                               /* wires_cap */
    write_merkle_cap(b, c->cap_height);                               /* plonk_zs_partial_products_cap */
    write_merkle_cap(b, c->cap_height);                               /* quotient_polys_cap */
    write_opening_set(b, c);
    write_fri_proof(b, c);
}

/* Returns the number of bytes of the serialized proof; writes them if out != NULL and cap is large enough.  The digest width
 * follows glo_set_hasher (0: PoseidonHash, 32 bytes; 1: KeccakHash<25>, 25 bytes). */
API size_t glo_proof_to_bytes(const glo_circuit *c, const u64 *proof_words, unsigned char *out, size_t cap) {
    wr_t b = {proof_words, out, cap, 0, glo_get_hasher() == 1 ? 25 : 32};
    write_proof(&b, c);
    write_field_vec(&b, c->num_public_inputs);
    return b.len;
}
