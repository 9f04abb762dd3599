// merkle.hip -- Poseidon sponge leaf hashing and 2-to-1 tree levels for gfx950.
//
// Replaces plonky2 `hash/merkle_tree.rs` (MerkleTree::new / prove / get) with
// `PoseidonHash::{hash_or_noop, two_to_one}` (`hash/poseidon.rs`, `hashing.rs`) as called from
// `PolynomialBatch::from_coeffs` and `fri_committed_trees` on the prove() path
// [REF src/ecdsa/gadgets/ecdsa.rs:349].  The same two primitives are what the reference calls
// natively at [REF src/smt/goldilocks_poseidon/mod.rs:165,170-180].
//
// One sponge state per lane.  The LDE matrix is column-major with rows in coset-major order, so
// lane `pos` reads lde[c][pos] for every column c: one fully coalesced 512-byte request per
// column per wave.  No transpose pass exists; the bit-reversed leaf order of plonky2 is produced
// by scattering the 32-byte digest to slot bitrev(pos).
#include "merkle.h"
#include "keccak.h"
#include "poseidon.h"

namespace glp {
using namespace glf;

constexpr size_t MERKLE_COOP_MAX_PARENTS = 8192;   // at or below this many hashes per launch: 12 lanes per hash
// Leaf hashing has three forms (crossovers measured by profiles/leaf_form_probe.py, 135 columns: 12-lane 0.27 ms / quad 0.34 / one per lane 0.65 at
// 2^12 leaves; 0.61 / 0.36 / 0.65 at 2^14; 2.09 / 0.83 / 0.66 at 2^16): the thresholds are glp_ctx::merkle_coop_max / merkle_quad_max.

size_t merkle_num_digests(size_t nleaves, int cap_height) {
    size_t t = 0, w = nleaves, cap = (size_t)1 << cap_height;
    for (;;) { t += w; if (w <= cap) break; w >>= 1; }
    return t;
}
size_t merkle_cap_offset(size_t nleaves, int cap_height) {
    return merkle_num_digests(nleaves, cap_height) - ((size_t)1 << cap_height);
}

__device__ __forceinline__ void store_digest(u64 *dst, const u64 s[12]) {
    ulonglong2 a, b;
    a.x = s[0]; a.y = s[1]; b.x = s[2]; b.y = s[3];
    reinterpret_cast<ulonglong2 *>(dst)[0] = a;
    reinterpret_cast<ulonglong2 *>(dst)[1] = b;
}

// hash_or_noop of one LDE row per lane.  grid.x * 256 >= N
// blockIdx.y = proof of a many-proofs batch (glp_prove_batch): LDE matrix and digest array of proof k start
// k * lde_stride / k * dig_stride words further on; a single proof launches with gridDim.y = 1.
__global__ __launch_bounds__(256, 4) void k_leaf_hash_lde(const u64 *__restrict__ lde, u64 *__restrict__ digests,
                                                       u32 ncols, int lg, int rate_bits, size_t lde_stride, size_t dig_stride) {
    lde += (size_t)blockIdx.y * lde_stride; digests += (size_t)blockIdx.y * dig_stride;
    const size_t N = (size_t)1 << (lg + rate_bits);
    const size_t pos = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= N) return;
    const u32 r = (u32)(pos >> lg), q = (u32)(pos & (((size_t)1 << lg) - 1));
    const size_t leaf = ((size_t)bitrev32(r, rate_bits) << lg) | bitrev32(q, lg);
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    const u64 *p = lde + pos;
    if (ncols <= 4) {
        for (u32 c = 0; c < ncols; c++) s[c] = p[(size_t)c * N];
    } else {
        // the next eight columns are requested before the current permutation starts (the sponge overwrites s[0..8)
        // with them afterwards), so their HBM latency hides under ~25k VALU instructions
        u64 nx[8];
#pragma unroll
        for (int i = 0; i < 8; i++) nx[i] = i < (int)ncols ? p[(size_t)i * N] : 0;
        for (u32 c = 0; c < ncols; c += 8) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (c + i < ncols) s[i] = nx[i];
            if (c + 8 < ncols) {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (c + 8 + i < ncols) nx[i] = p[(size_t)(c + 8 + i) * N];
            }
            pos::permute(s);
        }
    }
    store_digest(digests + 4 * leaf, s);
}

// Latency form of the leaf hash for small trees (N <= glp_ctx::merkle_coop_max): one leaf per 16-lane group, the
// sponge state spread over 12 lanes (pos::permute_coop).  17 sequential permutations per leaf take ~0.2 ms here
// instead of ~0.95 ms with one state per lane.
__global__ __launch_bounds__(256) void k_leaf_hash_lde_coop(const u64 *__restrict__ lde, u64 *__restrict__ digests,
                                                            u32 ncols, int lg, int rate_bits, size_t lde_stride, size_t dig_stride) {
    lde += (size_t)blockIdx.y * lde_stride; digests += (size_t)blockIdx.y * dig_stride;
    const size_t N = (size_t)1 << (lg + rate_bits);
    const int tid = threadIdx.x, l = tid & 15, lane = tid & 63, group_base = lane & ~15;
    const size_t pos = (size_t)blockIdx.x * 16 + (tid >> 4);
    const bool live = pos < N;
    const size_t p = live ? pos : 0;
    const u32 r = (u32)(p >> lg), q = (u32)(p & (((size_t)1 << lg) - 1));
    const size_t leaf = ((size_t)bitrev32(r, rate_bits) << lg) | bitrev32(q, lg);
    u64 x = 0;
    if (ncols <= 4) {
        if (live && l < 4) digests[4 * leaf + l] = (u32)l < ncols ? lde[(size_t)l * N + p] : 0;
        return;
    }
    for (u32 c = 0; c < ncols; c += 8) {
        if (l < 8 && c + l < ncols) x = lde[(size_t)(c + l) * N + p];
        x = pos::permute_coop(x, l, group_base);
    }
    if (live && l < 4) digests[4 * leaf + l] = x;
}

// The form between the two (merkle_coop_max < N <= merkle_quad_max): one leaf per quad of lanes, three state elements per
// lane (pos::permute_quad); 64 leaves per workgroup.  Lane q takes elements 3q..3q+2 of each block of eight columns.
__global__ __launch_bounds__(256) void k_leaf_hash_lde_quad(const u64 *__restrict__ lde, u64 *__restrict__ digests,
                                                            u32 ncols, int lg, int rate_bits, size_t lde_stride, size_t dig_stride) {
    lde += (size_t)blockIdx.y * lde_stride; digests += (size_t)blockIdx.y * dig_stride;
    const size_t N = (size_t)1 << (lg + rate_bits);
    const int tid = threadIdx.x, q = tid & 3;
    const size_t pos = (size_t)blockIdx.x * 64 + (tid >> 2);
    const bool live = pos < N;
    const size_t p = live ? pos : 0;
    const u32 r = (u32)(p >> lg), qq = (u32)(p & (((size_t)1 << lg) - 1));
    const size_t leaf = ((size_t)bitrev32(r, rate_bits) << lg) | bitrev32(qq, lg);
    if (ncols <= 4) {
        if (live) digests[4 * leaf + q] = (u32)q < ncols ? lde[(size_t)q * N + p] : 0;
        return;
    }
    u64 x[3] = {0, 0, 0};
    for (u32 c = 0; c < ncols; c += 8) {
#pragma unroll
        for (int s = 0; s < 3; s++) {
            const u32 e = 3 * q + s;
            if (e < 8 && c + e < ncols) x[s] = lde[(size_t)(c + e) * N + p];
        }
        pos::permute_quad(x, q);
    }
    if (live) {
        if (q == 0) { digests[4 * leaf] = x[0]; digests[4 * leaf + 1] = x[1]; digests[4 * leaf + 2] = x[2]; }
        if (q == 1) digests[4 * leaf + 3] = x[0];
    }
}

// hash_or_noop of row-major leaves [nleaves][leaf_len]
__global__ __launch_bounds__(256, 4) void k_leaf_hash_rows(const u64 *__restrict__ rows, u64 *__restrict__ digests,
                                                        size_t nleaves, u32 leaf_len) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= nleaves) return;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    const u64 *p = rows + j * leaf_len;
    if (leaf_len <= 4) {
        for (u32 c = 0; c < leaf_len; c++) s[c] = p[c];
    } else {
        u32 c = 0;
        for (; c + 8 <= leaf_len; c += 8) {
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = p[c + i];
            pos::permute(s);
        }
        if (c < leaf_len) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (c + i < leaf_len) s[i] = p[c + i];
            pos::permute(s);
        }
    }
    store_digest(digests + 4 * j, s);
}

// one tree level: out[i] = two_to_one(in[2i], in[2i+1])
__global__ __launch_bounds__(256) void k_merkle_level(const u64 *__restrict__ in, u64 *__restrict__ out, size_t m, size_t dig_stride) {
    in += (size_t)blockIdx.y * dig_stride; out += (size_t)blockIdx.y * dig_stride;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(in + 8 * i);
    ulonglong2 a = src[0], b = src[1], c2 = src[2], d = src[3];
    u64 s[12] = {a.x, a.y, b.x, b.y, c2.x, c2.y, d.x, d.y, 0, 0, 0, 0};
    pos::permute(s);
    store_digest(out + 4 * i, s);
}

// Small levels: one parent hash per 16-lane group (12 lanes active), 16 hashes per 256-thread workgroup.
__global__ __launch_bounds__(256) void k_merkle_level_coop(const u64 *__restrict__ in, u64 *__restrict__ out, size_t m, size_t dig_stride) {
    in += (size_t)blockIdx.y * dig_stride; out += (size_t)blockIdx.y * dig_stride;
    const int tid = threadIdx.x, l = tid & 15, lane = tid & 63, group_base = lane & ~15;
    const size_t i = (size_t)blockIdx.x * 16 + (tid >> 4);
    const bool live = i < m;
    u64 x = (live && l < 8) ? in[8 * i + l] : 0;
    x = pos::permute_coop(x, l, group_base);
    if (live && l < 4) out[4 * i + l] = x;
}

// Levels between the two (merkle_quad_max >= parents > MERKLE_COOP_MAX_PARENTS): one parent per quad of lanes (pos::permute_quad)
__global__ __launch_bounds__(256) void k_merkle_level_quad(const u64 *__restrict__ in, u64 *__restrict__ out, size_t m, size_t dig_stride) {
    in += (size_t)blockIdx.y * dig_stride; out += (size_t)blockIdx.y * dig_stride;
    const int tid = threadIdx.x, q = tid & 3;
    const size_t i0 = (size_t)blockIdx.x * 64 + (tid >> 2);
    const bool live = i0 < m;
    const size_t i = live ? i0 : 0;
    u64 x[3];
#pragma unroll
    for (int s = 0; s < 3; s++) x[s] = 3 * q + s < 8 ? in[8 * i + 3 * q + s] : 0;
    pos::permute_quad(x, q);
    if (live) {
        if (q == 0) { out[4 * i] = x[0]; out[4 * i + 1] = x[1]; out[4 * i + 2] = x[2]; }
        if (q == 1) out[4 * i + 3] = x[0];
    }
}

__global__ void k_permute_states(u64 *states, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = states[12 * i + k];
    pos::permute(s);
#pragma unroll
    for (int k = 0; k < 12; k++) states[12 * i + k] = s[k];
}

// ---- KeccakHash<25> (KeccakGoldilocksConfig): the same tree with Keccak-256 leaves and nodes (keccak.h) ----------------------
// hash_or_noop of one LDE row per lane: <= 3 columns are copied, otherwise the row's elements ARE the sponge's 64-bit lanes, 17 per
// rate block.  Bitwise work only (full-rate VALU): this tree is HBM / latency bound, not issue bound like the Poseidon one.
__global__ __launch_bounds__(256) void k_leaf_hash_lde_keccak(const u64 *__restrict__ lde, u64 *__restrict__ digests, u32 ncols, int lg,
                                                              int rate_bits, size_t lde_stride, size_t dig_stride) {
    lde += (size_t)blockIdx.y * lde_stride; digests += (size_t)blockIdx.y * dig_stride;
    const size_t N = (size_t)1 << (lg + rate_bits);
    const size_t pos = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= N) return;
    const u32 r = (u32)(pos >> lg), q = (u32)(pos & (((size_t)1 << lg) - 1));
    const size_t leaf = ((size_t)bitrev32(r, rate_bits) << lg) | bitrev32(q, lg);
    const u64 *p = lde + pos;
    u64 d[4] = {0, 0, 0, 0};
    if (ncols <= 3) {
        for (u32 c = 0; c < ncols; c++) d[c] = p[(size_t)c * N];
    } else {
        kec::Sponge s;
        kec::sponge_init(s);
        u32 c = 0;
        for (; c + kec::RATE_LANES <= ncols; c += kec::RATE_LANES) {
#pragma unroll
            for (int i = 0; i < kec::RATE_LANES; i++) s.a[i] ^= p[(size_t)(c + i) * N];
            kec::f1600(s.a);
        }
#pragma unroll
        for (int i = 0; i < kec::RATE_LANES; i++) if (c + i < ncols) s.a[i] ^= p[(size_t)(c + i) * N];
        s.fill = (int)(ncols - c);
        kec::sponge_finish(s);
        kec::sponge_digest25(s, d);
    }
    store_digest(digests + 4 * leaf, d);
}
__global__ __launch_bounds__(256) void k_merkle_level_keccak(const u64 *__restrict__ in, u64 *__restrict__ out, size_t m, size_t dig_stride) {
    in += (size_t)blockIdx.y * dig_stride; out += (size_t)blockIdx.y * dig_stride;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(in + 8 * i);
    const ulonglong2 a = src[0], b = src[1], c2 = src[2], e = src[3];
    const u64 l[4] = {a.x, a.y, b.x, b.y}, r[4] = {c2.x, c2.y, e.x, e.y};
    u64 d[4];
    kec::two_to_one(l, r, d);
    store_digest(out + 4 * i, d);
}

static int build_levels(glp_ctx *c, u64 *dev_digests, size_t nleaves, int cap_height, u32 K = 1, size_t dig_stride = 0, int hasher = 0) {
    size_t w = nleaves, cap = (size_t)1 << cap_height;
    u64 *lvl = dev_digests;
    while (w > cap) {
        u64 *nxt = lvl + 4 * w;
        size_t m = w >> 1;
        if (hasher == GLP_HASH_KECCAK25)
            hipLaunchKernelGGL(k_merkle_level_keccak, dim3((unsigned)((m + 255) / 256), K), dim3(256), 0, c->stream, lvl, nxt, m, dig_stride);
        else if (m * K <= MERKLE_COOP_MAX_PARENTS)   // few hashes: 12 lanes per hash for latency
            hipLaunchKernelGGL(k_merkle_level_coop, dim3((unsigned)((m + 15) / 16), K), dim3(256), 0, c->stream, lvl, nxt, m, dig_stride);
        else if (m * K <= c->merkle_quad_max)
            hipLaunchKernelGGL(k_merkle_level_quad, dim3((unsigned)((m + 63) / 64), K), dim3(256), 0, c->stream, lvl, nxt, m, dig_stride);
        else
            hipLaunchKernelGGL(k_merkle_level, dim3((unsigned)((m + 255) / 256), K), dim3(256), 0, c->stream, lvl, nxt, m, dig_stride);
        GLP_HIP(hipGetLastError());
        lvl = nxt; w = m;
    }
    return GLP_OK;
}

int merkle_levels(glp_ctx *c, u64 *dev_digests, size_t nleaves, int cap_height, u32 K, size_t dig_stride, int hasher) {
    return build_levels(c, dev_digests, nleaves, cap_height, K, dig_stride, hasher);
}

// K > 1: K trees over K LDE matrices (stride lde_stride words) into K digest arrays (stride dig_stride words)
int merkle_from_lde(glp_ctx *c, const u64 *dev_lde, u32 ncols, int lg, int rate_bits, int cap_height, u64 *dev_digests, u32 K,
                    size_t lde_stride, size_t dig_stride, int hasher) {
    const size_t N = (size_t)1 << (lg + rate_bits);
    if (cap_height < 0 || ((size_t)1 << cap_height) > N)
        return set_error(GLP_ERR_ARG, "cap_height=%d should be at most log2(leaves)=%d", cap_height, lg + rate_bits);
    GLP_REQUIRE(K >= 1 && K <= 65535, "batch of %u trees outside 1..65535", K);
    {
        StageScope st(c, "merkle_leaves", (double)N * K * (8.0 * ncols + 32.0));
        if (hasher == GLP_HASH_KECCAK25)
            hipLaunchKernelGGL(k_leaf_hash_lde_keccak, dim3((unsigned)((N + 255) / 256), K), dim3(256), 0, c->stream, dev_lde, dev_digests,
                               ncols, lg, rate_bits, lde_stride, dig_stride);
        else if (N * K <= c->merkle_coop_max)
            hipLaunchKernelGGL(k_leaf_hash_lde_coop, dim3((unsigned)((N + 15) / 16), K), dim3(256), 0, c->stream, dev_lde, dev_digests,
                               ncols, lg, rate_bits, lde_stride, dig_stride);
        else if (N * K <= c->merkle_quad_max)
            hipLaunchKernelGGL(k_leaf_hash_lde_quad, dim3((unsigned)((N + 63) / 64), K), dim3(256), 0, c->stream, dev_lde, dev_digests,
                               ncols, lg, rate_bits, lde_stride, dig_stride);
        else
            hipLaunchKernelGGL(k_leaf_hash_lde, dim3((unsigned)((N + 255) / 256), K), dim3(256), 0, c->stream, dev_lde, dev_digests,
                               ncols, lg, rate_bits, lde_stride, dig_stride);
        GLP_HIP(hipGetLastError());
    }
    StageScope st(c, "merkle_levels", (double)N * K * 32.0 * 1.5);
    return build_levels(c, dev_digests, N, cap_height, K, dig_stride, hasher);
}

int merkle_from_rows(glp_ctx *c, const u64 *dev_rows, size_t nleaves, u32 leaf_len, int cap_height, u64 *dev_digests) {
    if (cap_height < 0 || ((size_t)1 << cap_height) > nleaves)
        return set_error(GLP_ERR_ARG, "cap_height=%d should be at most log2(leaves)", cap_height);
    hipLaunchKernelGGL(k_leaf_hash_rows, dim3((unsigned)((nleaves + 255) / 256)), dim3(256), 0, c->stream, dev_rows,
                       dev_digests, nleaves, leaf_len);
    GLP_HIP(hipGetLastError());
    return build_levels(c, dev_digests, nleaves, cap_height);
}

// out[k * out_stride + col]: a record stride lets the prover gather straight into the proof's query layout
// blockIdx.y = proof of a batch: lde / leaf_idx / out advance by lde_bstride / count / out_bstride per proof
__global__ void k_gather_lde_rows(const u64 *__restrict__ lde, u32 ncols, int lg, int rate_bits,
                                  const u64 *__restrict__ leaf_idx, u32 count, u64 *__restrict__ out, size_t out_stride,
                                  size_t lde_bstride, size_t out_bstride) {
    lde += (size_t)blockIdx.y * lde_bstride; leaf_idx += (size_t)blockIdx.y * count; out += (size_t)blockIdx.y * out_bstride;
    const size_t N = (size_t)1 << (lg + rate_bits);
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)count * ncols) return;
    const u32 k = (u32)(t / ncols), col = (u32)(t % ncols);
    const u64 j = leaf_idx[k];
    // leaf j <-> point index i = bitrev_N(j) = q*R + r  <-> slot (r, q)
    const u32 rtop = (u32)(j >> lg), jl = (u32)(j & (((u64)1 << lg) - 1));
    const size_t pos = ((size_t)bitrev32(rtop, rate_bits) << lg) | bitrev32(jl, lg);
    out[(size_t)k * out_stride + col] = lde[(size_t)col * N + pos];
}

// leaf index = leaf_idx[k] >> idx_shift (FRI layer r looks at x_index >> sum of the arities so far)
__global__ void k_gather_paths(const u64 *__restrict__ digests, size_t nleaves, int depth,
                               const u64 *__restrict__ leaf_idx, u32 idx_shift, u32 count, u64 *__restrict__ out, size_t out_stride,
                               size_t dig_bstride, size_t out_bstride) {
    digests += (size_t)blockIdx.y * dig_bstride; leaf_idx += (size_t)blockIdx.y * count; out += (size_t)blockIdx.y * out_bstride;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)count * depth * 4) return;
    const u32 e = (u32)(t & 3);
    const u32 lvl = (u32)((t >> 2) % depth), k = (u32)((t >> 2) / depth);
    size_t off = 0, w = nleaves;
    for (u32 l = 0; l < lvl; l++) { off += w; w >>= 1; }
    const size_t idx = ((leaf_idx[k] >> idx_shift) >> lvl) ^ 1;
    out[(size_t)k * out_stride + 4 * lvl + e] = digests[4 * (off + idx) + e];
}

int merkle_gather_lde_rows(glp_ctx *c, const u64 *dev_lde, u32 ncols, int lg, int rate_bits, const u64 *dev_leaf_idx,
                           u32 count, u64 *dev_out, size_t out_stride, u32 K, size_t lde_bstride, size_t out_bstride) {
    if (out_stride == 0) out_stride = ncols;
    const size_t total = (size_t)count * ncols;
    if (!total) return GLP_OK;
    hipLaunchKernelGGL(k_gather_lde_rows, dim3((unsigned)((total + 255) / 256), K), dim3(256), 0, c->stream, dev_lde, ncols, lg,
                       rate_bits, dev_leaf_idx, count, dev_out, out_stride, lde_bstride, out_bstride);
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}

int merkle_gather_paths(glp_ctx *c, const u64 *dev_digests, size_t nleaves, int cap_height, const u64 *dev_leaf_idx,
                        u32 count, u64 *dev_out, size_t out_stride, u32 idx_shift, u32 K, size_t dig_bstride, size_t out_bstride) {
    int depth = 0;
    for (size_t w = nleaves; w > ((size_t)1 << cap_height); w >>= 1) depth++;
    if (out_stride == 0) out_stride = (size_t)depth * 4;
    const size_t total = (size_t)count * depth * 4;
    if (!total) return GLP_OK;
    hipLaunchKernelGGL(k_gather_paths, dim3((unsigned)((total + 255) / 256), K), dim3(256), 0, c->stream, dev_digests, nleaves,
                       depth, dev_leaf_idx, idx_shift, count, dev_out, out_stride, dig_bstride, out_bstride);
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}

int poseidon_permute_states(glp_ctx *c, u64 *dev_states, size_t count) {
    if (!count) return GLP_OK;
    hipLaunchKernelGGL(k_permute_states, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, dev_states, count);
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}

}  // namespace glp
