// ntt.h -- batched Goldilocks NTT / iNTT / low-degree extension over column-major device arrays.
//
// Device layouts (chosen for coalescing on gfx950, not plonky2's host layouts):
//   trace values   [ncols][n]            natural order (row i = w^i)
//   coefficients   [ncols][n]            BIT-REVERSED order: slot p holds the coefficient of X^bitrev(p)
//   LDE values     [ncols][R][n]         "coset-major": slot (r, q) holds p(shift * W^(q*R + r)),
//                                        R = 2^rate_bits, W = primitive_root_of_unity(log_n + rate_bits)
// An inverse transform is decimation-in-frequency (natural -> bit-reversed), a forward one
// decimation-in-time (bit-reversed -> natural), so no permutation pass ever touches HBM.
// A size-n transform is two LDS-staged passes, n = A * B: a strided pass over A rows (tile A x 16
// columns, 128-byte row segments) and a contiguous pass over B <= 4096 elements.  A <= 256 fits the
// 32 KB tiles of k_strided16; A = 512 / 1024 (n = 2^21 / 2^22) use the 160 KB LDS of gfx950
// (k_strided32: radix-32 x radix-16/32 register transforms, tiles of 64-128 KB).
#pragma once
#include "common.h"

namespace glp {

constexpr int NTT_2PASS_LG = 22;    // two-pass limit: B <= 2^12 (contiguous), A <= 2^10 (strided; 2^9 and 2^10 in k_strided32).
                                    // glp_ctx::two_pass_lg (GLP_NTT_2PASS_LG in the environment: 20..22) lowers it per context
constexpr int NTT_INNER_LG = 20;    // above the two-pass limit: per 2^20 block the two-pass transform, then an outer strided pass
constexpr int NTT_MAX_LG = 24;      //   over A' = n / 2^20 <= 16 blocks (three passes)
constexpr int NTT_LGB_MAX = 12;
constexpr int NTT_LGA_MAX = 8;      // rows of the generic radix-2 strided tile (static LDS); the radix-32 kernel goes to 2^10
constexpr int NTT_STRIDED_W = 16;   // columns per strided tile (16 x 8 B = one 128-B line per row)

struct NttPlan {
    int lg, lgA, lgB;
    u64 *tw_B = nullptr, *itw_B = nullptr;   // w_B^j / w_B^-j, j < B/2
    u64 *tw_A = nullptr, *itw_A = nullptr;   // w_A^j / w_A^-j, j < A/2
    u64 *tw4096 = nullptr, *itw4096 = nullptr;  // w_4096^(+-j), j < 4096: inter-step twiddles of the radix-16 kernels
    u64 w_n, w_n_inv, n_inv;
    // three-pass plans only (lg above the context's two-pass limit): n = A' * 2^20
    int lgAo = 0;
    const NttPlan *inner = nullptr;          // the 2^20 plan
    u64 *tw_Ao = nullptr, *itw_Ao = nullptr; // w_A'^(+-j), j < A'/2
    u64 *it0 = nullptr, *it1 = nullptr;      // inverse outer twiddle: (w_n^-k1o)^q = it1[pbo][q >> 10] * it0[pbo][q & 1023], it1 carries 1/A'
    NttPlan() = default;
    NttPlan(const NttPlan &) = delete;
    NttPlan &operator=(const NttPlan &) = delete;
    ~NttPlan() {                             // tables belong to the plan: an error path that drops a half-built plan frees them
        for (u64 *t : {tw_B, itw_B, tw_A, itw_A, tw4096, itw4096, tw_Ao, itw_Ao, it0, it1}) if (t) (void)hipFree(t);
    }
};

struct LdePlan {
    const NttPlan *ntt;
    int rate_bits;
    u64 shift;
    u64 *pre = nullptr;      // [R][B]: (s_r^A)^bitrev_B(pl),  s_r = shift * W^r
    u64 *s_r = nullptr;      // [R]
    // three-pass plans only
    const LdePlan *inner = nullptr;          // (2^20, rate_bits, shift^A')
    u64 *t0 = nullptr, *t1 = nullptr;        // outer twiddle s_r^k1o (w_n^k1o)^q = t1[r][pbo][q >> 10] * t0[pbo][q & 1023]
    u64 last_use = 0;                        // LRU stamp (glp_ctx::lde_clock); plans other LdePlans point at are pinned
    int pins = 0;
    LdePlan() = default;
    LdePlan(const LdePlan &) = delete;
    LdePlan &operator=(const LdePlan &) = delete;
    ~LdePlan() { for (u64 *t : {pre, s_r, t0, t1}) if (t) (void)hipFree(t); }
};
constexpr size_t LDE_PLAN_CACHE_MAX = 48;    // distinct (log_n, rate_bits, shift) tables kept per context

int get_ntt_plan(glp_ctx *c, int lg, NttPlan **out);
int get_lde_plan(glp_ctx *c, int lg, int rate_bits, u64 shift, LdePlan **out);
void free_plans(glp_ctx *c);

// values [ncols][n] natural  ->  coefficients [ncols][n] bit-reversed   (PolynomialValues::ifft)
int intt_values_to_coeffs(glp_ctx *c, const u64 *dev_values, u64 *dev_coeffs, u32 ncols, int lg);
// coefficients bit-reversed -> values natural (PolynomialCoeffs::fft), in place allowed (out == in)
int ntt_coeffs_to_values(glp_ctx *c, const u64 *dev_coeffs, u64 *dev_values, u32 ncols, int lg);
// coefficients [ncols][n] bit-reversed -> LDE [ncols][R][n] coset-major   (lde + coset_fft(shift))
int lde_coeffs(glp_ctx *c, const u64 *dev_coeffs, u64 *dev_lde, u32 ncols, int lg, int rate_bits, u64 shift);
// out[c][bitrev(p)] = in[c][p]  (layout conversion for accessors; not on the prove path)
int bitrev_copy(glp_ctx *c, const u64 *dev_in, u64 *dev_out, u32 ncols, int lg);
// coset-major [ncols][R][n] -> natural [ncols][N]  (accessor only)
int lde_to_natural(glp_ctx *c, const u64 *dev_lde, u64 *dev_out, u32 ncols, int lg, int rate_bits);

}  // namespace glp
