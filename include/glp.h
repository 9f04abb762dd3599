/*
 * glp.h -- C ABI of libglprover.so: an MI355X (gfx950) backend for the Goldilocks-field
 * `CircuitData::prove()` path of the plonky2 fork that Orbiter-Finance/Plonky2-lib drives
 * (every `data.prove(pw)` call site, e.g. [REF src/ecdsa/gadgets/ecdsa.rs:349],
 * [REF src/hash/keccak256.rs:248], [REF src/zkdsa/circuits/mod.rs:326]).
 *
 * The prover's arithmetic lives in the `plonky2` crate, a path-patched dependency that is NOT
 * part of /root/reference [REF Cargo.toml:10-12,32-34]; the entry points below are what an
 * `extern "C"` block inside that crate would bind, at the same seams where the fork links its
 * CUDA NTT library (sppark, [REF Cargo.lock:955-977,1411-1416]).  Each function names the
 * plonky2 0.1.4 item it replaces.  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions: every function returns GLP_OK (0) or a negative error code and records a message
 * retrievable with glp_last_error() (thread-local).  Pointers are caller-owned HOST memory
 * unless the parameter name starts with `dev_`.  Field elements are canonical uint64_t
 * (< 2^64 - 2^32 + 1); extension elements are two consecutive uint64_t (a + b*X, X^2 = 7).
 * A glp_ctx is bound to one GPU and one HIP stream; calls on one ctx must be serialised by the
 * caller; distinct ctxs are independent (this is what shards a batch of proofs over 8 GPUs).
 * There is no CPU fallback: without a usable gfx950 device glp_ctx_create fails.
 */
#ifndef GLP_H
#define GLP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define GLP_API __attribute__((visibility("default")))
#else
#define GLP_API
#endif

#define GLP_OK 0
#define GLP_ERR_ARG (-1)         /* bad argument (null pointer, size out of range, ...) */
#define GLP_ERR_HIP (-2)         /* a HIP runtime call failed; message has the HIP error string */
#define GLP_ERR_UNSUPPORTED (-3) /* valid request outside what this build implements */
#define GLP_ERR_NOGPU (-4)       /* no gfx950 device visible */
#define GLP_ERR_PROVE (-5)       /* the witness does not satisfy the circuit (quotient not a polynomial) */

typedef struct glp_ctx glp_ctx;
typedef struct glp_batch glp_batch; /* plonky2 `PolynomialBatch`: coefficients + LDE + Merkle tree, device resident */

GLP_API const char *glp_last_error(void);
GLP_API const char *glp_version(void);
GLP_API int glp_device_count(void);

GLP_API int glp_ctx_create(int device_id, glp_ctx **out);
GLP_API void glp_ctx_destroy(glp_ctx *ctx);
GLP_API int glp_ctx_synchronize(glp_ctx *ctx);
/* The HIP stream (hipStream_t) every kernel of this ctx is launched on. */
GLP_API void *glp_ctx_stream(glp_ctx *ctx);

/* ---- per-stage device timing (hipEvent pairs on the ctx stream) -------------------------------
 * With profiling on, each kernel stage of the following calls is bracketed by events.
 * glp_ctx_stage_count / glp_ctx_stage_get read them back after synchronising. */
GLP_API int glp_ctx_set_profiling(glp_ctx *ctx, int on);
GLP_API int glp_ctx_stage_reset(glp_ctx *ctx);
GLP_API int glp_ctx_stage_count(glp_ctx *ctx);
GLP_API int glp_ctx_stage_get(glp_ctx *ctx, int index, const char **name, float *ms, double *algorithmic_bytes);

/* ---- primitives (used by the parity tests and by a fine-grained FFI binding) ------------------ */
/* hash/poseidon.rs `Poseidon::poseidon`: permute `count` states of 12 elements on the GPU. */
GLP_API int glp_poseidon_permute(glp_ctx *ctx, uint64_t *states, size_t count);
/* field/src/fft.rs `fft` / `ifft` on `ncols` independent columns of n = 2^log_n elements,
 * natural order in and out (values[i] = p(w^i), w = primitive_root_of_unity(log_n)). */
GLP_API int glp_fft(glp_ctx *ctx, uint64_t *cols, uint32_t ncols, uint32_t log_n);
GLP_API int glp_ifft(glp_ctx *ctx, uint64_t *cols, uint32_t ncols, uint32_t log_n);
/* polynomial/mod.rs `lde(rate_bits)` + `coset_fft(shift)`: in [ncols][n] coefficients,
 * out [ncols][n << rate_bits] values, out[c][i] = p_c(shift * W^i), natural order. */
GLP_API int glp_lde(glp_ctx *ctx, const uint64_t *coeffs, uint32_t ncols, uint32_t log_n, uint32_t rate_bits,
            uint64_t shift, uint64_t *out);

/* Synthetic-trace helper for benchmarks: dev_out[i] = SplitMix64(seed, i) folded into [0, p)
 * (z >= p ? z - p : z).  Lets bench.py create HBM-resident inputs without a PCIe copy. */
GLP_API int glp_fill_random_device(glp_ctx *ctx, uint64_t *dev_out, size_t count, uint64_t seed);

/* ---- PolynomialBatch (fri/oracle.rs) ----------------------------------------------------------
 * glp_batch_from_values  = PolynomialBatch::from_values  (ifft, lde x 2^rate_bits on the coset 7*H,
 *                          transpose + bit-reverse, MerkleTree::new(leaves, cap_height))
 * glp_batch_from_coeffs  = PolynomialBatch::from_coeffs
 * Input layout: column-major [ncols][n], n = 2^log_n, natural order.  No blinding
 * (zero_knowledge = false in every CircuitConfig the reference uses).
 * The *_device variants take a device pointer on the ctx's GPU (input already resident in HBM). */
GLP_API int glp_batch_from_values(glp_ctx *ctx, const uint64_t *values, uint32_t ncols, uint32_t log_n,
                          uint32_t rate_bits, uint32_t cap_height, glp_batch **out);
GLP_API int glp_batch_from_values_device(glp_ctx *ctx, const uint64_t *dev_values, uint32_t ncols, uint32_t log_n,
                                 uint32_t rate_bits, uint32_t cap_height, glp_batch **out);
GLP_API int glp_batch_from_coeffs(glp_ctx *ctx, const uint64_t *coeffs, uint32_t ncols, uint32_t log_n,
                          uint32_t rate_bits, uint32_t cap_height, glp_batch **out);
GLP_API int glp_batch_from_coeffs_device(glp_ctx *ctx, const uint64_t *dev_coeffs, uint32_t ncols, uint32_t log_n,
                                 uint32_t rate_bits, uint32_t cap_height, glp_batch **out);
GLP_API void glp_batch_free(glp_batch *b);
GLP_API int glp_batch_info(const glp_batch *b, uint32_t *ncols, uint32_t *log_n, uint32_t *rate_bits, uint32_t *cap_height);
/* merkle_tree.cap: [2^cap_height][4] */
GLP_API int glp_batch_cap(const glp_batch *b, uint64_t *cap_out);
/* `polynomials[col].coeffs`, natural order, for cols [col_begin, col_begin + ncols) -> [ncols][n] */
GLP_API int glp_batch_coeffs(const glp_batch *b, uint32_t col_begin, uint32_t ncols, uint64_t *out);
/* merkle_tree.leaves[leaf_index] (= `MerkleTree::get`): the ncols LDE values of row bitrev(leaf_index) */
GLP_API int glp_batch_leaf(const glp_batch *b, uint64_t leaf_index, uint64_t *out);
/* `MerkleTree::prove(leaf_index)`: (log_n + rate_bits - cap_height) sibling digests, bottom-up, [k][4] */
GLP_API int glp_batch_merkle_proof(const glp_batch *b, uint64_t leaf_index, uint64_t *siblings_out);
/* every digest bottom-up: level 0 (leaf digests, index = leaf index) ... cap level; count =
 * sum_{w = N, N/2, .., 2^cap_height} w.  For tests. */
GLP_API size_t glp_batch_num_digests(const glp_batch *b);
GLP_API int glp_batch_digests(const glp_batch *b, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif /* GLP_H */
