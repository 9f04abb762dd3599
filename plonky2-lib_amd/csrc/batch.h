// batch.h -- device-resident PolynomialBatch (plonky2 `fri/oracle.rs`).
#pragma once
#include "common.h"

struct glp_batch {
    glp_ctx *ctx = nullptr;
    u32 ncols = 0;
    int lg = 0, rate_bits = 0, cap_height = 0;
    u64 *coeffs = nullptr;    // [ncols][n]     bit-reversed coefficient order
    u64 *lde = nullptr;       // [ncols][R][n]  coset-major LDE values
    u64 *digests = nullptr;   // [ndigests][4]  level 0 (leaf j at slot j) ... cap level
    size_t ndigests = 0;
    int hasher = GLP_HASH_POSEIDON;   // GenericConfig::Hasher of the tree
    u32 K = 1;                // many-proofs batches (glp_prove_batch): K independent oracles of identical shape, arrays [K][...]
};

namespace glp {
enum BatchInput { BATCH_VALUES = 0, BATCH_COEFFS_NATURAL = 1, BATCH_COEFFS_BITREV = 2 };
int batch_build(glp_ctx *c, const u64 *dev_in, int input_kind, u32 ncols, int lg, int rate_bits, int cap_height,
                glp_batch **out, const u64 *host_src = nullptr, u32 K = 1, int hasher = GLP_HASH_POSEIDON);
void batch_destroy(glp_batch *b);
}  // namespace glp
