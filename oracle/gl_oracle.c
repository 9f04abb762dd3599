/*
 * oracle/gl_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C CPU restatement of the commitment half of the plonky2 prover hot path that
 * /root/reference reaches through `data.prove(pw)` [REF src/ecdsa/gadgets/ecdsa.rs:349]:
 *   Poseidon permutation / sponge / two_to_one   (plonky2 0.1.4 hash/poseidon.rs, hashing.rs)
 *   fft / ifft / coset_fft / lde                  (field/src/fft.rs, polynomial/mod.rs)
 *   MerkleTree::new / prove                       (hash/merkle_tree.rs)
 *   PolynomialBatch::from_values / from_coeffs    (fri/oracle.rs)
 *   Challenger                                    (iop/challenger.rs)
 * The plonky2 crate is a path-patched, un-vendored dependency [REF Cargo.toml:10-12,32-34;
 * Cargo.lock:952-978] and is absent from /root/reference, so the published upstream algorithm is
 * restated.  Pinned by the reference's own golden vectors:
 *   - PoseidonHash::two_to_one(0,0)  [REF src/zkdsa/circuits/mod.rs:85-101,143,149]
 *   - hash_pad padding relation      [REF src/smt/goldilocks_poseidon/mod.rs:170-180 vs
 *                                          src/smt/gadgets/common.rs:87-101]
 * Proof-byte parity against the forked Rust prover is otherwise UNPINNED (no fixture exists in
 * the reference, no Rust toolchain here) -- see DESIGN.md "Oracle".
 *
 * The implementation here is deliberately the textbook one (bit-reverse + iterative radix-2,
 * naive 30-round Poseidon, row-major leaves) and shares no code with the HIP library.
 */
#include "gl_field.h"
#include "gl_keccak.h"
#include "poseidon_constants.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ helpers for python */
API void glo_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
API int glo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
API u64 glo_add(u64 a, u64 b) { return gl_add(a, b); }
API u64 glo_sub(u64 a, u64 b) { return gl_sub(a, b); }
API u64 glo_mul(u64 a, u64 b) { return gl_mul(a, b); }
API u64 glo_pow(u64 a, u64 e) { return gl_pow(a, e); }
API u64 glo_inv(u64 a) { return gl_inv(a); }
API u64 glo_root_of_unity(int n_log) { return gl_root_of_unity(n_log); }
API void glo_vec_mul(const u64 *a, const u64 *b, u64 *o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = gl_mul(a[i], b[i]); }
API void glo_vec_add(const u64 *a, const u64 *b, u64 *o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = gl_add(a[i], b[i]); }
API void glo_vec_sub(const u64 *a, const u64 *b, u64 *o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = gl_sub(a[i], b[i]); }
API void glo_vec_scale(const u64 *a, u64 s, u64 *o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = gl_mul(a[i], s); }
API void glo_vec_inv(const u64 *a, u64 *o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = gl_inv(a[i]); }
API void glo_ext_mul(const u64 *x, const u64 *y, u64 *o) {
    gl2 r = gl2_mul(gl2_make(x[0], x[1]), gl2_make(y[0], y[1])); o[0] = r.a[0]; o[1] = r.a[1];
}
API void glo_ext_inv(const u64 *x, u64 *o) { gl2 r = gl2_inv(gl2_make(x[0], x[1])); o[0] = r.a[0]; o[1] = r.a[1]; }

/* ------------------------------------------------------------------ Poseidon (width 12)
 * hash/poseidon.rs: N_ROUNDS = 4 + 22 + 4, S-box x^7, MDS = circulant(CIRC) + diag(DIAG);
 * `poseidon_naive` schedule (constant layer on all lanes every round, S-box on lane 0 only in
 * partial rounds, full MDS every round) -- plonky2 asserts it equal to its fast schedule. */
static const u64 MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const u64 MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static inline u64 sbox7(u64 x) {
    u64 x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
}
static inline void mds_layer(u64 s[12]) {
    u64 o[12], d[24];
    memcpy(d, s, 96); memcpy(d + 12, s, 96);      /* d[i + r] == s[(i + r) % 12] */
    for (int r = 0; r < 12; r++) { /* mds_row_shf */
        u128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (u128)d[i + r] * MDS_CIRC[i];
        acc += (u128)s[r] * MDS_DIAG[r];
        o[r] = gl_reduce128(acc);
    }
    memcpy(s, o, sizeof(o));
}
API void glo_poseidon_permute(u64 s[12]) {
    int rc = 0;
    for (int r = 0; r < 30; r++) {
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], (u64)GL_POSEIDON_RC[rc + i]);
        rc += 12;
        if (r < 4 || r >= 26) { for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]); }
        else s[0] = sbox7(s[0]);
        mds_layer(s);
    }
}
/* hashing.rs `hash_n_to_m_no_pad` with m = 4: overwrite-mode sponge, rate 8, no padding. */
API void glo_hash_no_pad(const u64 *in, size_t len, u64 out[4]) {
    u64 st[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t c = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < c; i++) st[i] = in[off + i];
        glo_poseidon_permute(st);
    }
    if (len == 0) { /* zero chunks absorbed: squeeze straight from the zero state */ }
    memcpy(out, st, 4 * sizeof(u64));
}
/* Hasher::hash_or_noop: <=4 elements are copied (zero padded), not hashed. */
/* ---- GenericConfig::Hasher: 0 = PoseidonHash (PoseidonGoldilocksConfig), 1 = KeccakHash<25> (KeccakGoldilocksConfig, the one
 * driver [REF src/hash/keccak256.rs:281]; conventions in gl_keccak.h).  A process-wide switch: this is test infrastructure, the
 * Python side sets it around each call (oracle.py).  The public-input hash is the INNER hasher, Poseidon in both configs. */
static int g_hasher = 0;
API void glo_set_hasher(int h) { g_hasher = h; }
API int glo_get_hasher(void) { return g_hasher; }
API void glo_hash_or_noop(const u64 *in, size_t len, u64 out[4]) {
    if (g_hasher == 1) { glo_keccak_hash_or_noop(in, len, out); return; }
    if (len <= 4) { for (int i = 0; i < 4; i++) out[i] = (size_t)i < len ? in[i] : 0; }
    else glo_hash_no_pad(in, len, out);
}
/* C::Hasher::hash_no_pad (circuit digest) */
API void glo_outer_hash_no_pad(const u64 *in, size_t len, u64 out[4]) {
    if (g_hasher == 1) glo_keccak_hash_no_pad(in, len, out); else glo_hash_no_pad(in, len, out);
}
/* the sponge permutation of the transcript: H::Permutation */
API void glo_permute(u64 st[12]) {
    if (g_hasher == 1) glo_keccak_permute(st); else glo_poseidon_permute(st);
}
/* hashing.rs `compress`: perm(left || right || 0000)[0..4] */
API void glo_two_to_one(const u64 l[4], const u64 r[4], u64 out[4]) {
    if (g_hasher == 1) { glo_keccak_two_to_one(l, r, out); return; }
    u64 st[12] = {0};
    memcpy(st, l, 32); memcpy(st + 4, r, 32);
    glo_poseidon_permute(st);
    memcpy(out, st, 32);
}
/* Hasher::hash_pad: append 1, zero-fill until len+1 = 0 mod 12, append 1, hash_no_pad. */
API void glo_hash_pad(const u64 *in, size_t len, u64 out[4]) {
    size_t cap = len + 14;
    u64 *b = (u64 *)calloc(cap, sizeof(u64));
    memcpy(b, in, len * sizeof(u64));
    size_t l = len;
    b[l++] = 1;
    while ((l + 1) % 12 != 0) b[l++] = 0;
    b[l++] = 1;
    glo_hash_no_pad(b, l, out);
    free(b);
}

/* ------------------------------------------------------------------ FFT
 * fft.rs semantics: fft(coeffs)[i] = poly(w^i), w = primitive_root_of_unity(log n), natural
 * order in and out.  Implementation: bit-reverse, then iterative radix-2 DIT. */
static inline size_t bitrev(size_t x, int bits) {
    size_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
static void fft_inplace(u64 *a, int lg) {
    size_t n = (size_t)1 << lg;
    for (size_t i = 0; i < n; i++) { size_t j = bitrev(i, lg); if (i < j) { u64 t = a[i]; a[i] = a[j]; a[j] = t; } }
    for (int s = 1; s <= lg; s++) {
        size_t m = (size_t)1 << s, h = m >> 1;
        u64 wm = gl_root_of_unity(s);
        for (size_t k = 0; k < n; k += m) {
            u64 w = 1;
            for (size_t j = 0; j < h; j++) {
                u64 t = gl_mul(w, a[k + j + h]), u = a[k + j];
                a[k + j] = gl_add(u, t);
                a[k + j + h] = gl_sub(u, t);
                w = gl_mul(w, wm);
            }
        }
    }
}
API void glo_fft(u64 *a, int lg) { fft_inplace(a, lg); }
/* `ifft_with_options`: forward FFT, then reverse all but the first and divide by n. */
API void glo_ifft(u64 *a, int lg) {
    size_t n = (size_t)1 << lg;
    fft_inplace(a, lg);
    u64 ninv = gl_inv((u64)n % GL_P);
    a[0] = gl_mul(a[0], ninv);
    if (n > 1) a[n / 2] = gl_mul(a[n / 2], ninv);
    for (size_t i = 1; i < n / 2; i++) {
        size_t j = n - i;
        u64 ci = gl_mul(a[j], ninv), cj = gl_mul(a[i], ninv);
        a[i] = ci; a[j] = cj;
    }
}
/* `coset_fft(shift)`: coefficient i scaled by shift^i, then fft. */
API void glo_coset_fft(u64 *a, int lg, u64 shift) {
    size_t n = (size_t)1 << lg; u64 p = 1;
    for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], p); p = gl_mul(p, shift); }
    fft_inplace(a, lg);
}
/* `coset_ifft(shift)`: ifft, then coefficient i scaled by shift^-i. */
API void glo_coset_ifft(u64 *a, int lg, u64 shift) {
    size_t n = (size_t)1 << lg; u64 si = gl_inv(shift), p = 1;
    glo_ifft(a, lg);
    for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], p); p = gl_mul(p, si); }
}
/* `PolynomialCoeffs::lde(rate_bits)` + `coset_fft(F::coset_shift() = 7)`; out has n << rate_bits. */
API void glo_lde(const u64 *coeffs, int lg, int rate_bits, u64 shift, u64 *out) {
    size_t n = (size_t)1 << lg, N = n << rate_bits;
    memcpy(out, coeffs, n * sizeof(u64));
    memset(out + n, 0, (N - n) * sizeof(u64));
    glo_coset_fft(out, lg + rate_bits, shift);
}

/* ------------------------------------------------------------------ Merkle tree
 * merkle_tree.rs: leaf digest = hash_or_noop(leaf); parent = two_to_one(left,right); the cap is
 * the level with 2^cap_height nodes.  Digest storage here: level 0 (leaf digests) first, then
 * level 1, ... up to and including the cap level; `glo_merkle_num_digests` gives the total. */
API size_t glo_merkle_num_digests(size_t nleaves, int cap_height) {
    size_t t = 0, w = nleaves, cap = (size_t)1 << cap_height;
    for (;;) { t += w; if (w == cap) break; w >>= 1; }
    return t;
}
API int glo_merkle_build(const u64 *leaves, size_t nleaves, size_t leaf_len, int cap_height, u64 *digests, u64 *cap_out) {
    size_t cap = (size_t)1 << cap_height;
    if (cap > nleaves) return -1;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < nleaves; i++) glo_hash_or_noop(leaves + i * leaf_len, leaf_len, digests + 4 * i);
    u64 *lvl = digests; size_t w = nleaves;
    while (w > cap) {
        u64 *nxt = lvl + 4 * w;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < w / 2; i++) glo_two_to_one(lvl + 8 * i, lvl + 8 * i + 4, nxt + 4 * i);
        lvl = nxt; w >>= 1;
    }
    memcpy(cap_out, lvl, cap * 32);
    return 0;
}
/* MerkleTree::prove: siblings bottom-up, stopping below the cap level. Returns the sibling count. */
API int glo_merkle_prove(const u64 *digests, size_t nleaves, int cap_height, size_t index, u64 *siblings) {
    size_t cap = (size_t)1 << cap_height, w = nleaves; const u64 *lvl = digests; int k = 0;
    while (w > cap) {
        memcpy(siblings + 4 * k, lvl + 4 * (index ^ 1), 32);
        k++; lvl += 4 * w; w >>= 1; index >>= 1;
    }
    return k;
}
/* merkle_proofs.rs `verify_merkle_proof_to_cap`. Returns 0 if the path recomputes the cap entry. */
API int glo_merkle_verify(const u64 *leaf, size_t leaf_len, size_t index, const u64 *cap, int cap_height,
                          const u64 *siblings, int nsib) {
    (void)cap_height;
    u64 cur[4], nx[4];
    glo_hash_or_noop(leaf, leaf_len, cur);
    for (int k = 0; k < nsib; k++) {
        if (index & 1) glo_two_to_one(siblings + 4 * k, cur, nx); else glo_two_to_one(cur, siblings + 4 * k, nx);
        memcpy(cur, nx, 32); index >>= 1;
    }
    return memcmp(cur, cap + 4 * index, 32) == 0 ? 0 : 1;
}

/* ------------------------------------------------------------------ PolynomialBatch
 * fri/oracle.rs `from_coeffs`: lde every polynomial (rate_bits, shift 7), transpose, bit-reverse
 * the row order, Merkle-commit the rows. No blinding (zero_knowledge = false in every config the
 * reference uses [REF src/ecdsa/gadgets/ecdsa.rs:476,483]).
 *   coeffs  [ncols][n]   in, natural order
 *   leaves  [N][ncols]   out, leaf j = row bitrev_N(j) of the LDE matrix (N = n << rate_bits)
 *   digests, cap         out (layout: glo_merkle_build) */
API int glo_batch_from_coeffs(const u64 *coeffs, size_t ncols, int lg, int rate_bits, int cap_height,
                              u64 *leaves, u64 *digests, u64 *cap_out) {
    size_t n = (size_t)1 << lg, N = n << rate_bits; int LG = lg + rate_bits;
    int err = 0;
#pragma omp parallel
    {
        u64 *tmp = (u64 *)malloc(N * sizeof(u64));
#pragma omp for schedule(dynamic)
        for (size_t c = 0; c < ncols; c++) {
            glo_lde(coeffs + c * n, lg, rate_bits, GL_GEN, tmp);
            for (size_t j = 0; j < N; j++) leaves[j * ncols + c] = tmp[bitrev(j, LG)];
        }
        free(tmp);
    }
    err = glo_merkle_build(leaves, N, ncols, cap_height, digests, cap_out);
    return err;
}
/* `from_values`: ifft every column first; coeffs_out [ncols][n] receives the coefficients. */
API int glo_batch_from_values(const u64 *values, size_t ncols, int lg, int rate_bits, int cap_height,
                              u64 *coeffs_out, u64 *leaves, u64 *digests, u64 *cap_out) {
    size_t n = (size_t)1 << lg;
    memcpy(coeffs_out, values, ncols * n * sizeof(u64));
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < ncols; c++) glo_ifft(coeffs_out + c * n, lg);
    return glo_batch_from_coeffs(coeffs_out, ncols, lg, rate_bits, cap_height, leaves, digests, cap_out);
}

/* ------------------------------------------------------------------ Challenger (iop/challenger.rs)
 * Duplex sponge in overwrite mode; challenges are popped from the END of the squeezed rate. */
typedef struct { u64 st[12]; u64 in[8]; int nin; u64 out[8]; int nout; } glo_challenger;
API size_t glo_challenger_size(void) { return sizeof(glo_challenger); }
API void glo_challenger_init(glo_challenger *c) { memset(c, 0, sizeof(*c)); }
static void ch_duplex(glo_challenger *c) {
    for (int i = 0; i < c->nin; i++) c->st[i] = c->in[i];
    c->nin = 0;
    glo_permute(c->st);
    memcpy(c->out, c->st, 64); c->nout = 8;
}
API void glo_challenger_observe(glo_challenger *c, const u64 *e, size_t n) {
    for (size_t i = 0; i < n; i++) {
        c->nout = 0;
        c->in[c->nin++] = e[i];
        if (c->nin == 8) ch_duplex(c);
    }
}
/* observe_hash / observe_cap for hashes of the OUTER hasher: a Poseidon HashOut is its 4 elements, a BytesHash<25> its 7-byte chunks */
API void glo_challenger_observe_hashes(glo_challenger *c, const u64 *digests, size_t count) {
    for (size_t i = 0; i < count; i++) {
        if (g_hasher == 1) { u64 e[4]; glo_keccak_hash_to_elements(digests + 4 * i, e); glo_challenger_observe(c, e, 4); }
        else glo_challenger_observe(c, digests + 4 * i, 4);
    }
}
API u64 glo_challenger_get(glo_challenger *c) {
    if (c->nin > 0 || c->nout == 0) ch_duplex(c);
    return c->out[--c->nout];
}
API void glo_challenger_get_n(glo_challenger *c, u64 *o, size_t n) { for (size_t i = 0; i < n; i++) o[i] = glo_challenger_get(c); }
