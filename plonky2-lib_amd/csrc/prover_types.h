// prover_types.h -- host-side types shared by the prover (prover.hip) and the verifier (verifier.hip):
// the circuit handle, the flat proof layout and the Fiat-Shamir transcript.
#pragma once
#include <algorithm>
#include <string.h>
#include <vector>
#include "batch.h"
#include "common.h"
#include "keccak.h"
#include "poseidon.h"

using namespace glp;
using namespace glf;

constexpr int MAXCH = 4;      // num_challenges supported
constexpr int MAXR = 16;      // 2^rate_bits supported
constexpr u32 APL_WORDS = 4;  // table words per alpha power in the quotient kernels' limb form (prover.hip AccHL): m0 | m1 << 32, m2, m0' | m1' << 32, m2'

struct DevGate { u32 type, selector_index, group_start, group_end, row, num_constraints, p0, p1; };

struct Layout {
    size_t caps, openings, fri_caps, queries, final_poly, pow, pis, total, nopen, query_stride;
    u32 oracle_cols[4], depth0, step_depth[16], final_len;
};

struct glp_circuit {
    glp_ctx *ctx = nullptr;
    glp_circuit_desc d;            // scalars + host copies below
    std::vector<glp_gate> gates;
    std::vector<u64> k_is;
    u64 digest[4];
    DevGate *dev_gates = nullptr;
    u64 *dev_k_is = nullptr;
    u32 k_ratio = 0;               // g if k_is[j] = g^j for all j with g < 2^32 (then the quotient kernel chains by g), else 0
    u64 *dev_sigmas = nullptr;     // [nr][n] values on H (natural order), for the partial products
    u64 *dev_consts = nullptr;     // [nc][n] values on H (selectors first): which gate sits on a row (witness.hip)
    // quotient launch plan (built once in glp_circuit_create): which gates share a launch
    u64 *dev_limb_desc = nullptr;  // [group][num_wires][LIMB_SLOTS] column programs of k_quotient_limbs
    // limb gates in groups of up to LIMB_SLOTS = 5 (one set of accumulators in registers per group; the kernel walks the groups in turn):
    // limb_count = gates over all groups, group g holds gates limb_gi[5 g .. 5 g + limb_gcount[g])
    u32 limb_count = 0, limb_groups = 0, limb_gcount[4] = {0, 0, 0, 0}, limb_gi[20] = {0}, limb_jlo[4] = {0, 0, 0, 0}, limb_jhi[4] = {0, 0, 0, 0};
    u32 limb_extra_count = 0, limb_extra_gi[4] = {0, 0, 0, 0};
    u32 arith_gi = 0, arith_ops = 0;   // ArithmeticGate evaluated inside the permutation loop (0 ops: none)
    u32 light_count = 0, light_gi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<u32> single_gates; // gates that keep a launch of their own
    glp_batch *cs = nullptr;       // constants_sigmas_commitment
    std::vector<u64> cs_cap;
    Layout L;
};

// ------------------------------------------------------------------------------------------ transcript
struct Challenger {   // iop/challenger.rs, overwrite-mode duplex sponge; challenges pop from the END of the rate
    u64 st[12]; u64 in[8]; int nin = 0; u64 out[8]; int nout = 0;
    int hasher = GLP_HASH_POSEIDON;      // H::Permutation: Poseidon, or KeccakPermutation (hash chain, keccak.h)
    explicit Challenger(int hasher_ = GLP_HASH_POSEIDON) : hasher(hasher_) { memset(st, 0, sizeof(st)); }
    void duplex() {
        for (int i = 0; i < nin; i++) st[i] = in[i];
        nin = 0;
        if (hasher == GLP_HASH_KECCAK25) kec::permute(st); else pos::permute(st);
        memcpy(out, st, 64); nout = 8;
    }
    // observe_hash / observe_cap for digests of the proof's hasher: a HashOut is its 4 elements, a BytesHash<25> its four 7-byte chunks
    void observe_hashes(const u64 *digests, size_t count) {
        for (size_t i = 0; i < count; i++) {
            if (hasher == GLP_HASH_KECCAK25) { u64 e[4]; kec::digest_to_elements(digests + 4 * i, e); observe(e, 4); }
            else observe(digests + 4 * i, 4);
        }
    }
    void observe(const u64 *e, size_t n) {
        for (size_t i = 0; i < n; i++) { nout = 0; in[nin++] = e[i]; if (nin == 8) duplex(); }
    }
    u64 get() { if (nin > 0 || nout == 0) duplex(); return out[--nout]; }
    ext2 get_ext() { u64 a = get(); u64 b = get(); return e_make(a, b); }
};
inline void host_hash_no_pad(const u64 *in, size_t len, u64 out[4]) {
    u64 st[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t c = std::min<size_t>(8, len - off);
        for (size_t i = 0; i < c; i++) st[i] = in[off + i];
        pos::permute(st);
    }
    memcpy(out, st, 32);
}
