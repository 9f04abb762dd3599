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
#define GLP_ERR_PROVE (-5)       /* prove: the transcript cannot continue (PoW search exhausted, zeta in the subgroup); verify: proof rejected */

#define GLP_HASH_POSEIDON 0      /* PoseidonGoldilocksConfig */
#define GLP_HASH_KECCAK25 1      /* KeccakGoldilocksConfig: KeccakHash<25> [REF src/hash/keccak256.rs:281] */

typedef struct glp_ctx glp_ctx;
typedef struct glp_batch glp_batch; /* plonky2 `PolynomialBatch`: coefficients + LDE + Merkle tree, device resident */

GLP_API const char *glp_last_error(void);
GLP_API const char *glp_version(void);
GLP_API int glp_device_count(void);

/* Environment variables read when a context is made (tuning switches for the A/B rows under profiles/; the defaults ship):
 *   GLP_NTT_2PASS_LG      = 20..22 (default 22): largest log2(n) transformed in two passes; above it a third pass over 2^20-point blocks
 *   GLP_NTT_STRIDED32_LW  = 3 | 4  (default 4):  log2 columns of the 512- / 1024-row strided tile (64- or 128-byte row segments)
 *   GLP_NTT_STRIDED32_TL  = 1..64  (default 8):  tiles one block of that kernel walks, software-pipelined (1 = one tile per block)
 *   GLP_MERKLE_COOP_MAX   (default 4096), GLP_MERKLE_QUAD_MAX (default 32768): Poseidon leaf hashing spreads one sponge over 12 of 16 lanes up to
 *                         the first many leaves per launch, over a quad of lanes up to the second, and keeps one sponge per lane above it
 *                         (FRI layers alike; Merkle levels: 12 of 16 lanes up to 8192 parents, a quad up to the second threshold)
 *   GLP_HOST_THREADS      (read on the first glp_prove_batch of a context): host threads for the transcripts of a batch */
GLP_API int glp_ctx_create(int device_id, glp_ctx **out);
GLP_API void glp_ctx_destroy(glp_ctx *ctx);
GLP_API int glp_ctx_synchronize(glp_ctx *ctx);
/* The HIP stream (hipStream_t) every kernel of this ctx is launched on. */
GLP_API void *glp_ctx_stream(glp_ctx *ctx);

/* Device buffers for the *_device entry points, for callers that do not link a HIP runtime themselves (a Rust / Go host
 * that keeps a witness resident in HBM across proofs).  Memory comes from and returns to the context's pool. */
GLP_API int glp_dev_alloc(glp_ctx *ctx, size_t bytes, void **dev_out);
GLP_API int glp_dev_free(glp_ctx *ctx, void *dev);
GLP_API int glp_dev_upload(glp_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);     /* synchronous */
GLP_API int glp_dev_download(glp_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);   /* synchronous */

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
/* The hash of the commitment: plonky2's `GenericConfig::Hasher`.  GLP_HASH_POSEIDON = PoseidonGoldilocksConfig (every driver of the
 * reference but one); GLP_HASH_KECCAK25 = KeccakGoldilocksConfig, `KeccakHash<25>` [REF src/hash/keccak256.rs:281]: Keccak-256
 * truncated to 25 bytes; such a digest occupies the same 4-word slot as a Poseidon HashOut (little-endian bytes, top 7 bytes of the
 * last word zero), so caps, paths and digest arrays keep their shapes.  The *_h entry points are the ones above with the hash named. */
GLP_API int glp_batch_from_values_h(glp_ctx *ctx, const uint64_t *values, uint32_t ncols, uint32_t log_n, uint32_t rate_bits,
                                    uint32_t cap_height, uint32_t hasher, glp_batch **out);
GLP_API int glp_batch_from_coeffs_h(glp_ctx *ctx, const uint64_t *coeffs, uint32_t ncols, uint32_t log_n, uint32_t rate_bits,
                                    uint32_t cap_height, uint32_t hasher, glp_batch **out);
/* Keccak-256 of `count` messages of `len` bytes each (msgs [count][len], digests_out [count][32]) on the GPU: the primitive under
 * GLP_HASH_KECCAK25, exposed so that it can be checked against the reference's (input, digest) pairs
 * [REF src/hash/keccak256.rs:196-212,256-277]. */
GLP_API int glp_keccak256(glp_ctx *ctx, const uint8_t *msgs, size_t count, size_t len, uint8_t *digests_out);
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

/* ---- circuits and whole proofs ------------------------------------------------------------------
 * glp_circuit_desc carries the parts of plonky2's CommonCircuitData / ProverOnlyCircuitData /
 * VerifierOnlyCircuitData that `prove` reads (plonk/circuit_data.rs), i.e. what
 * `builder.build::<C>()` returns at [REF src/ecdsa/gadgets/ecdsa.rs:298].  Scope of this build:
 * no lookup tables, zero_knowledge = false, quotient_degree_factor a power of two <= 2^rate_bits,
 * D = 2, Poseidon hashing (PoseidonGoldilocksConfig, the `type C` of every reference driver but
 * one [REF src/hash/keccak256.rs:281]).  Gate types a circuit may contain: see GLP_GATE_*. */
enum {
    GLP_GATE_NOOP = 0,             /* plonky2 gates/noop.rs */
    GLP_GATE_CONSTANT = 1,         /* gates/constant.rs, p0 = num_consts */
    GLP_GATE_PUBLIC_INPUT = 2,     /* gates/public_input.rs */
    GLP_GATE_ARITHMETIC = 3,       /* gates/arithmetic_base.rs, p0 = num_ops */
    GLP_GATE_POSEIDON = 4,         /* gates/poseidon.rs */
    GLP_GATE_U32_INTERLEAVE = 5,   /* [REF src/u32/gates/interleave_u32.rs:33-82,84-135], p0 = num_ops */
    GLP_GATE_UNINTERLEAVE_U32 = 6, /* [REF src/u32/gates/uninterleave_to_u32.rs:30-91,93-150], p0 = num_ops */
    GLP_GATE_UNINTERLEAVE_B32 = 7, /* [REF src/u32/gates/uninterleave_to_b32.rs:95-150], p0 = num_ops */
    /* the gate set of the secp256k1 circuit [REF src/ecdsa/gadgets/ecdsa.rs:72-96]; sources absent from the
     * reference (plonky2, plonky2_u32 @552acaec [REF Cargo.lock:1001-1009]), restated from the published crates */
    GLP_GATE_U32_ARITHMETIC = 8,   /* plonky2_u32 gates/arithmetic_u32.rs, p0 = num_ops */
    GLP_GATE_U32_ADD_MANY = 9,     /* plonky2_u32 gates/add_many_u32.rs, p0 = num_addends, p1 = num_ops */
    GLP_GATE_U32_SUBTRACTION = 10, /* plonky2_u32 gates/subtraction_u32.rs, p0 = num_ops */
    GLP_GATE_U32_RANGE_CHECK = 11, /* plonky2_u32 gates/range_check_u32.rs, p0 = num_input_limbs */
    GLP_GATE_COMPARISON = 12,      /* plonky2_u32 gates/comparison.rs, p0 = num_bits, p1 = num_chunks */
    GLP_GATE_BASE_SUM = 13,        /* plonky2 gates/base_sum.rs, p0 = num_limbs, p1 = base */
    GLP_GATE_RANDOM_ACCESS = 14    /* plonky2 gates/random_access.rs, p0 = bits, p1 = num_copies | num_extra_constants << 16 */
};

typedef struct {
    uint32_t type;
    uint32_t selector_index;           /* SelectorsInfo.selector_indices[gate] */
    uint32_t group_start, group_end;   /* SelectorsInfo.groups[selector_index] */
    uint32_t row;                      /* index of the gate in CommonCircuitData.gates */
    uint32_t num_constraints;
    uint32_t p0, p1;                   /* gate parameters (see GLP_GATE_*) */
} glp_gate;

typedef struct {
    uint32_t degree_bits;
    uint32_t num_wires, num_routed_wires;
    uint32_t num_constants;            /* all constant polynomials: selectors first, then gate constants */
    uint32_t num_selectors;
    uint32_t num_challenges;
    uint32_t quotient_degree_factor;
    uint32_t num_partial_products;
    uint32_t num_gate_constraints;
    uint32_t rate_bits, cap_height, proof_of_work_bits, num_query_rounds;
    uint32_t num_reductions;
    uint32_t reduction_arity_bits[16];
    uint32_t num_gates;
    uint32_t num_public_inputs;
    const glp_gate *gates;
    const uint64_t *k_is;              /* [num_routed_wires] */
    uint64_t circuit_digest[4];        /* verifier_only.circuit_digest; all-zero = let the library derive it */
    const uint64_t *constants;         /* [num_constants][n]    values on H (constant_vecs) */
    const uint64_t *sigmas;            /* [num_routed_wires][n] values on H (sigma_vecs) */
    uint32_t hasher;                   /* GLP_HASH_POSEIDON (0) or GLP_HASH_KECCAK25: `GenericConfig::Hasher` of the proof (Merkle trees,
                                          transcript, proof of work).  The public-input hash is the InnerHasher, Poseidon in both.
                                          With GLP_HASH_KECCAK25 every digest in caps, paths and circuit_digest is a 25-byte value in
                                          a 4-word slot (see GLP_HASH_KECCAK25 above) */
} glp_circuit_desc;

typedef struct glp_circuit glp_circuit;

/* Uploads the description and does the prover-side part of `build()`: commits constants ++ sigmas
 * (`constants_sigmas_commitment`) and, if desc->circuit_digest is all zero, derives the digest as
 * hash_no_pad(cap.flatten() ++ hash_pad([]) ++ [degree_bits]) (plonk/circuit_builder.rs). */
GLP_API int glp_circuit_create(glp_ctx *ctx, const glp_circuit_desc *desc, glp_circuit **out);
GLP_API void glp_circuit_free(glp_circuit *circuit);
GLP_API int glp_circuit_digest(const glp_circuit *circuit, uint64_t digest_out[4]);
GLP_API int glp_circuit_constants_sigmas_cap(const glp_circuit *circuit, uint64_t *cap_out);
/* Number of uint64_t words of a proof of this circuit (layout below). */
GLP_API size_t glp_proof_words(const glp_circuit *circuit);

/* ---- circuit hand-off file --------------------------------------------------------------
 * The reference persists circuits with `CircuitData::to_bytes(&gate_serializer, &generator_serializer)` and reloads them
 * with `from_bytes` [REF src/ecdsa/gadgets/ecdsa.rs:298-316; serializer tables REF src/ecdsa/gadgets/ecdsa.rs:68-135,
 * src/ecdsa/serialization.rs:7-46].  That byte layout belongs to the absent plonky2 crate and carries the generator list,
 * which the GPU prover does not need.  The file below is this library's own flat dump of glp_circuit_desc (+ optionally one
 * witness and its public inputs): versioned header, little-endian sections, FNV-1a checksum; the exact layout is at the top
 * of plonky2-lib_amd/csrc/circuit_file.hip, the Rust writer that fills it from `data.common` / `data.prover_only` is in
 * INTEGRATION.md.  Host-only calls (no GPU needed); glp_circuit_file_open maps the file (GB-sized circuits are not copied)
 * and the descriptor's pointers stay valid until glp_circuit_file_close. */
typedef struct glp_circuit_file glp_circuit_file;
GLP_API int glp_circuit_file_write(const char *path, const glp_circuit_desc *desc, const uint64_t *wires /* [num_wires][n] or NULL */,
                                   const uint64_t *public_inputs /* [num_public_inputs], with wires */);
GLP_API int glp_circuit_file_open(const char *path, int verify_checksum, glp_circuit_file **out);
GLP_API void glp_circuit_file_close(glp_circuit_file *file);
GLP_API const glp_circuit_desc *glp_circuit_file_desc(const glp_circuit_file *file);       /* feed to glp_circuit_create */
GLP_API const uint64_t *glp_circuit_file_wires(const glp_circuit_file *file);              /* NULL if the file has no witness */
GLP_API const uint64_t *glp_circuit_file_public_inputs(const glp_circuit_file *file);

/* `CircuitData::prove` after witness generation (plonk/prover.rs `prove_with_partition_witness`,
 * steps "compute wires commitment" .. "compute opening proofs").
 *   wires          [num_wires][n] full witness (`witness.wire_values`), host or (…_device) HBM resident
 *   public_inputs  [num_public_inputs]
 *   proof_out      glp_proof_words() words, field order of plonky2's `Buffer::write_proof`:
 *     wires_cap | plonk_zs_partial_products_cap | quotient_polys_cap           each [2^cap_height][4]
 *     openings: constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys   (ext each)
 *     commit_phase_merkle_caps [num_reductions][2^cap_height][4]
 *     query_round_proofs [num_query_rounds]: 4 x (leaf values, merkle path) then per reduction (evals, merkle path)
 *     final_poly (ext coefficients) | pow_witness | public_inputs
 * The FRI proof-of-work witness is the SMALLEST valid one (the Rust prover's rayon `find_any`
 * returns an arbitrary valid one; every other word of the proof is a deterministic function of
 * the inputs and of that witness). */
GLP_API int glp_prove(glp_ctx *ctx, const glp_circuit *circuit, const uint64_t *wires, const uint64_t *public_inputs,
                      uint64_t *proof_out);
GLP_API int glp_prove_device(glp_ctx *ctx, const glp_circuit *circuit, const uint64_t *dev_wires,
                             const uint64_t *public_inputs, uint64_t *proof_out);

/* ---- witnesses that start in host memory, pipelined -------------------------------------------------------------------
 * `data.prove(pw)` starts from a witness the CPU has just generated [REF src/ecdsa/gadgets/ecdsa.rs:332-349: set the targets, then
 * prove].  glp_prove(host wires) uploads it inside the call; a caller that proves witness after witness hides the upload behind
 * the previous proof instead:
 *     glp_witness_stage(ctx, circuit, wires[i+1], flags, &w_next);        returns at once: the copy runs on the ctx's copy stream
 *     glp_prove_staged(ctx, circuit, w_cur, public_inputs, proof_out);    the compute stream waits for w_cur's copy, then proves
 *     glp_witness_free(w_cur);
 * For the copy to overlap anything the host buffer must be page-locked: glp_host_alloc hands out such memory (have witness
 * generation write into it); pageable memory works and is staged by the HIP runtime, synchronously.  Either way host_wires must stay
 * valid and unchanged until glp_prove_staged has returned for that witness (or glp_witness_free has).
 * GLP_WITNESS_ROUTED_ONLY: host_wires holds only the routed columns [num_routed_wires][n] (= the first columns of the full
 * [num_wires][n] layout, so the full array may be passed as well); the advice columns are zero-filled on the device and derived
 * there by glp_witness_fill(only_advice) before the proof starts -- for the secp256k1 trace 56 of 136 columns (41 %) never cross
 * PCIe.  The proof is word for word the proof of the complete witness provided every advice wire of the witness is either
 * written by a row-local generator or zero (true for witnesses of plonky2's generators; tests/test_gpu_witness.py). */
typedef struct glp_witness glp_witness;
#define GLP_WITNESS_ROUTED_ONLY 1u
GLP_API int glp_host_alloc(glp_ctx *ctx, size_t bytes, void **host_out);      /* page-locked host memory */
GLP_API int glp_host_free(glp_ctx *ctx, void *host);
GLP_API int glp_witness_stage(glp_ctx *ctx, const glp_circuit *circuit, const uint64_t *host_wires, uint32_t flags, glp_witness **out);
GLP_API int glp_prove_staged(glp_ctx *ctx, const glp_circuit *circuit, glp_witness *witness, const uint64_t *public_inputs,
                             uint64_t *proof_out);
GLP_API void glp_witness_free(glp_witness *witness);

/* Many independent proofs of ONE circuit in lock step (BASELINE config 5: a batch of zkdsa simple-signature proofs, the unit
 * [REF src/zkdsa/circuits/mod.rs:24-43,322-339] proves one at a time).  Small circuits are bound by launch and host round-trip
 * latency when proved one by one; here every device stage is one launch over all num_proofs proofs and every host round trip
 * carries all their caps / openings, while the num_proofs Fiat-Shamir transcripts run on host threads (environment variable GLP_HOST_THREADS; default: the
 * cores this process may use -- affinity mask capped by the cgroup CPU quota -- up to 32; every context has its own pool).  proofs_out[k] is word for word what glp_prove returns for witness k.
 *   wires          [num_proofs][num_wires][n], host memory, or (wires_on_device != 0) an HBM pointer on the ctx's GPU
 *   public_inputs  [num_proofs][num_public_inputs]
 *   proofs_out     [num_proofs][glp_proof_words(circuit)]
 * Scope: num_challenges = 2 (every CircuitConfig the reference uses). */
GLP_API int glp_prove_batch(glp_ctx *ctx, const glp_circuit *circuit, uint32_t num_proofs, const uint64_t *wires, int wires_on_device,
                            const uint64_t *public_inputs, uint64_t *proofs_out);

/* ---- witness generation, the row-local half ------------------------------------------
 * plonky2 `iop/generator.rs::generate_partial_witness` interleaves two kinds of work: the copy-constraint dataflow
 * between rows (stays with the reference's CPU gadget code) and the row-local `SimpleGenerator`s that derive the rest
 * of a row from that row's own inputs.  glp_witness_fill applies every row-local generator of the gate library once,
 * one GPU thread per trace row, in place on an HBM-resident witness [num_wires][n]:
 *   the reference's own generators  U32InterleaveGenerator [REF src/u32/gates/interleave_u32.rs:289-318],
 *     UninterleaveToU32Generator [REF src/u32/gates/uninterleave_to_u32.rs:332-369], UninterleaveToB32Generator
 *     [REF src/u32/gates/uninterleave_to_b32.rs:335-372];
 *   plonky2_u32 (recalled) U32Arithmetic / U32AddMany / U32Subtraction / U32RangeCheck / Comparison generators;
 *   plonky2 (recalled)     BaseSplitGenerator, ArithmeticBaseGenerator, RandomAccessGenerator, PoseidonGenerator,
 *                          ConstantGenerator.
 * Each generator reads its dependencies (glp_witness_columns role 2) from the row and writes its outputs (role 1):
 * bits, base-4 limbs, inverses, S-box traces, u32 results.  only_advice != 0: only non-routed columns
 * (index >= num_routed_wires) are written, for a caller whose CPU pass already resolved every routed wire -- then the
 * GPU derives the 56 limb columns of the secp256k1 trace (41 % of the witness) and they never cross PCIe.
 * The row's gate is read from the circuit's selector polynomials.  Asynchronous on the ctx stream. */
GLP_API int glp_witness_fill(glp_ctx *ctx, const glp_circuit *circuit, uint64_t *dev_wires, int only_advice);
/* role_out[col] for rows of gate `gate_index`: 1 = written by glp_witness_fill, 2 = read as a generator input, 0 = untouched */
GLP_API int glp_witness_columns(const glp_circuit *circuit, uint32_t gate_index, uint8_t *role_out /* [num_wires] */);

/* ---- the same proof, stepped by the caller's own transcript -------------------------------------------------
 * `prove_with_partition_witness` is Fiat-Shamir glue around six device stages.  A Rust integration that keeps
 * plonky2's own `Challenger` (so that the transcript is the fork's by construction) calls the stages one by one:
 * each call returns exactly what the Rust prover observes next, and takes exactly the challenges it draws next
 * [UPSTREAM plonky2 plonk/prover.rs: `challenger.observe_cap(..)` / `get_n_challenges` / `get_extension_challenge`;
 *  fri/prover.rs: `fri_committed_trees`, `fri_proof_of_work`, `fri_prover_query_rounds`; reached from
 *  REF src/ecdsa/gadgets/ecdsa.rs:349].  Order (enforced; GLP_ERR_ARG otherwise):
 *
 *   glp_session_begin              wires -> iNTT, LDE, Merkle           out: wires cap, hash of the public inputs
 *   glp_session_partial_products   in: betas, gammas [num_challenges]   out: plonk_zs_partial_products cap
 *   glp_session_quotient           in: alphas [num_challenges]          out: quotient_polys cap
 *   glp_session_open               in: zeta (ext)                       out: OpeningSet, 2 words per opening, proof order
 *   glp_session_fri_combine        in: FRI alpha (ext)
 *   num_reductions x { glp_session_fri_commit (out: layer cap) ; glp_session_fri_fold (in: beta, ext) }
 *   glp_session_fri_final_poly     out: final polynomial coefficients (ext)
 *   glp_pow_search                 (stateless) in: the challenger's sponge state + pending inputs   out: witness
 *   glp_session_queries            in: pow witness, query indices x_index in [0, 2^(degree_bits+rate_bits))
 *   glp_session_proof              out: the assembled proof words (same layout as glp_prove)
 *   glp_session_end
 *
 * glp_prove() is this sequence driven by the library's built-in transcript; tests/test_gpu_prove.py drives it with the
 * oracle's Challenger and requires the identical proof. */
typedef struct glp_session glp_session;
/* wires: [num_wires][n]; wires_on_device != 0: an HBM pointer that must stay valid until glp_session_end */
GLP_API int glp_session_begin(glp_ctx *ctx, const glp_circuit *circuit, const uint64_t *wires, int wires_on_device,
                              const uint64_t *public_inputs, glp_session **out, uint64_t *wires_cap_out /* [2^cap_height][4] */,
                              uint64_t public_inputs_hash_out[4]);
GLP_API int glp_session_partial_products(glp_session *s, const uint64_t *betas, const uint64_t *gammas, uint64_t *zs_cap_out);
GLP_API int glp_session_quotient(glp_session *s, const uint64_t *alphas, uint64_t *quotient_cap_out);
/* openings_out: 2 * glp_num_openings(circuit) words: constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next,
 * partial_products, quotient_polys (the order `OpeningSet::to_fri_openings` / the proof uses) */
GLP_API size_t glp_num_openings(const glp_circuit *circuit);
GLP_API int glp_session_open(glp_session *s, const uint64_t zeta[2], uint64_t *openings_out);
GLP_API int glp_session_fri_combine(glp_session *s, const uint64_t alpha[2]);
GLP_API int glp_session_fri_commit(glp_session *s, uint64_t *cap_out);
GLP_API int glp_session_fri_fold(glp_session *s, const uint64_t beta[2]);
GLP_API size_t glp_final_poly_len(const glp_circuit *circuit);          /* ext coefficients */
GLP_API int glp_session_fri_final_poly(glp_session *s, uint64_t *coeffs_out /* [2 * glp_final_poly_len] */);
/* smallest w such that a duplex sponge in `sponge_state` with `num_pending` (< 8) buffered inputs, after also
 * observing w, squeezes an element with `bits` leading zero bits (fri_proof_of_work) */
GLP_API int glp_pow_search(glp_ctx *ctx, const uint64_t sponge_state[12], const uint64_t *pending_inputs, uint32_t num_pending,
                           uint32_t bits, uint64_t *witness_out);
/* the same for a transcript whose permutation is that of `hasher` (GLP_HASH_KECCAK25: KeccakPermutation) */
GLP_API int glp_pow_search_h(glp_ctx *ctx, uint32_t hasher, const uint64_t sponge_state[12], const uint64_t *pending_inputs,
                             uint32_t num_pending, uint32_t bits, uint64_t *witness_out);
GLP_API int glp_session_queries(glp_session *s, uint64_t pow_witness, const uint64_t *indices, uint32_t num_indices);
GLP_API int glp_session_proof(glp_session *s, uint64_t *proof_out /* glp_proof_words(circuit) */);
GLP_API void glp_session_end(glp_session *s);

/* `CircuitData::verify(proof)` [REF src/ecdsa/gadgets/ecdsa.rs:352, src/zkdsa/circuits/mod.rs:346]: checks a proof in
 * the word layout above against the circuit (gate table, coset shifts, digest, constants/sigmas cap): transcript,
 * proof of work, vanishing polynomial at zeta against the quotient openings, every FRI query (Merkle paths, initial
 * combination, arity-2^k consistency, final polynomial).  Host computation (a few thousand Poseidon permutations), no
 * device work.  GLP_OK = accepted; GLP_ERR_PROVE with the reason in glp_last_error() = rejected. */
GLP_API int glp_verify(const glp_circuit *circuit, const uint64_t *proof_words);
/* The same with the buffer length stated (num_words must equal glp_proof_words(circuit)): for callers that hold a proof of
 * unknown provenance; a wrong length is GLP_ERR_ARG, not a read past the buffer. */
GLP_API int glp_verify_n(const glp_circuit *circuit, const uint64_t *proof_words, size_t num_words);

/* `data.verify(proof)` for K proofs of one circuit, the query rounds on the GPU (a verifier for batch
 * self-checking behind glp_prove_batch; the reference proves and then verifies every proof [REF src/zkdsa/circuits/mod.rs:341-347,
 * src/ecdsa/gadgets/ecdsa.rs:349-352]).  Per proof, host threads run the transcript, the proof-of-work check and the
 * vanishing-polynomial identity at zeta; every FRI query round of every proof (Merkle paths of the four initial oracles and of each
 * commit-phase layer, `fri_combine_initial`, the arity-2^k consistency checks, the final polynomial) is one device launch over
 * K x num_query_rounds 16-lane groups.  Accepts and rejects exactly what glp_verify does, with the same reason.
 *   proofs       [K][glp_proof_words(circuit)], host memory
 *   status_out   [K]: GLP_OK = accepted, GLP_ERR_PROVE = rejected
 *   reasons_out  NULL, or [K][GLP_REASON_LEN] chars: the rejection reason of proof k (empty string if accepted)
 * Under PoseidonGoldilocksConfig the K transcripts run on the device too (the batch prover's kernels over the uploaded proofs); the host keeps
 * the canonical-form scan and the identity at zeta.  Environment variable GLP_VERIFY_HOST_TRANSCRIPT=1 (read per call): transcripts on host
 * threads, as under KeccakGoldilocksConfig.
 * Returns GLP_OK when the batch was checked (whatever the verdicts), an error code for bad arguments / HIP failures. */
#define GLP_REASON_LEN 160
GLP_API int glp_verify_batch(glp_ctx *ctx, const glp_circuit *circuit, uint32_t num_proofs, const uint64_t *proofs, int32_t *status_out,
                             char *reasons_out);

/* plonky2 `ProofWithPublicInputs::to_bytes()` (util/serialization.rs `Buffer::write_proof_with_public_inputs`):
 * every field element as 8 little-endian bytes in the word order above, plus the one-byte sibling
 * count that `write_merkle_proof` puts in front of every Merkle path.  This is the wire format the
 * reference round-trips at [REF src/ecdsa/gadgets/ecdsa.rs:298-316] for circuits and that a Rust
 * `ProofWithPublicInputs::from_bytes(bytes, &data.common)` consumes.  (Recalled format: not pinned by
 * a fixture, see DESIGN.md.) */
GLP_API size_t glp_proof_bytes_len(const glp_circuit *circuit);
GLP_API int glp_proof_to_bytes(const glp_circuit *circuit, const uint64_t *proof_words, uint8_t *bytes_out, size_t bytes_len);
GLP_API int glp_proof_from_bytes(const glp_circuit *circuit, const uint8_t *bytes, size_t bytes_len, uint64_t *proof_words_out);

#ifdef __cplusplus
}
#endif
#endif /* GLP_H */
