// merkle.h -- Poseidon Merkle tree over the rows of a coset-major LDE matrix.
#pragma once
#include "common.h"

namespace glp {

// number of digests stored bottom-up (level 0 = leaf digests ... cap level inclusive)
size_t merkle_num_digests(size_t nleaves, int cap_height);

// MerkleTree::new over leaves taken from a coset-major LDE matrix [ncols][R][n] (see ntt.h):
// leaf j (plonky2 order) = the ncols values at point index bitrev_N(j).  Digest of leaf j is
// written to digests[j]; upper levels follow.  dev_cap (may be null) receives the cap level.
int merkle_from_lde(glp_ctx *c, const u64 *dev_lde, u32 ncols, int lg, int rate_bits, int cap_height, u64 *dev_digests, u32 K = 1,
                    size_t lde_stride = 0, size_t dig_stride = 0, int hasher = GLP_HASH_POSEIDON);
// MerkleTree::new over row-major leaves [nleaves][leaf_len] (FRI commit-phase trees)
int merkle_from_rows(glp_ctx *c, const u64 *dev_rows, size_t nleaves, u32 leaf_len, int cap_height, u64 *dev_digests);
// the 2-to-1 levels above an already hashed leaf level (digests[0..nleaves) filled)
int merkle_levels(glp_ctx *c, u64 *dev_digests, size_t nleaves, int cap_height, u32 K = 1, size_t dig_stride = 0, int hasher = GLP_HASH_POSEIDON);
// offset (in digests) of the cap level inside the digest buffer
size_t merkle_cap_offset(size_t nleaves, int cap_height);
// gather `count` leaves / proofs
int merkle_gather_lde_rows(glp_ctx *c, const u64 *dev_lde, u32 ncols, int lg, int rate_bits, const u64 *dev_leaf_idx,
                           u32 count, u64 *dev_out /*[count][ncols]*/, size_t out_stride = 0 /* 0: ncols */, u32 K = 1,
                           size_t lde_bstride = 0, size_t out_bstride = 0);
int merkle_gather_paths(glp_ctx *c, const u64 *dev_digests, size_t nleaves, int cap_height, const u64 *dev_leaf_idx,
                        u32 count, u64 *dev_out /*[count][depth][4]*/, size_t out_stride = 0 /* 0: 4 depth */,
                        u32 idx_shift = 0 /* leaf = idx >> idx_shift */, u32 K = 1, size_t dig_bstride = 0, size_t out_bstride = 0);
int poseidon_permute_states(glp_ctx *c, u64 *dev_states, size_t count);

}  // namespace glp
