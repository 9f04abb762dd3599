// ntt.hip -- LDS-staged two-pass Goldilocks NTT kernels for gfx950 (see ntt.h for layouts).
//
// Replaces plonky2 `field/src/fft.rs` (fft_dispatch / ifft_with_options) and
// `polynomial/mod.rs` (lde, coset_fft_with_options) as called from
// `PolynomialBatch::from_values / from_coeffs` on the prove() path
// [REF src/ecdsa/gadgets/ecdsa.rs:349].  No MFMA: 64-bit modular integer butterflies.
#include <algorithm>
#include <utility>
#include "ntt.h"

namespace glp {
using namespace glf;

// ------------------------------------------------------------------------------------------
// host-side plan construction (tables are a few KB; built once per (log_n, rate_bits, shift))
// ------------------------------------------------------------------------------------------
static int upload(glp_ctx *c, const std::vector<u64> &h, u64 **dev) {
    size_t bytes = (h.size() ? h.size() : 1) * sizeof(u64);
    GLP_HIP(hipMalloc((void **)dev, bytes));
    if (h.size()) GLP_HIP(hipMemcpyAsync(*dev, h.data(), h.size() * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

int get_ntt_plan(glp_ctx *c, int lg, NttPlan **out) {
    if (lg < 0 || lg > NTT_MAX_LG)
        return set_error(GLP_ERR_UNSUPPORTED, "log_n=%d outside the supported range 0..%d", lg, NTT_MAX_LG);
    auto it = c->ntt_plans.find(lg);
    if (it != c->ntt_plans.end()) { *out = it->second; return GLP_OK; }
    std::unique_ptr<NttPlan> p(new NttPlan());
    p->lg = lg;
    if (lg > c->two_pass_lg) {
        NttPlan *in;
        GLP_TRY(get_ntt_plan(c, NTT_INNER_LG, &in));
        p->inner = in;
        p->lgAo = lg - NTT_INNER_LG;
        p->lgB = in->lgB; p->lgA = in->lgA;
        p->w_n = root_of_unity(lg); p->w_n_inv = inv(p->w_n); p->n_inv = inv((u64)1 << lg);
        const size_t Ao = (size_t)1 << p->lgAo;
        std::vector<u64> f(Ao / 2), b(Ao / 2), t0(Ao * 1024), t1(Ao * 1024);
        const u64 wA = root_of_unity(p->lgAo), wAi = inv(wA), ainv = inv((u64)Ao);
        u64 x = 1, y = 1;
        for (size_t j = 0; j < Ao / 2; j++) { f[j] = x; b[j] = y; x = mul(x, wA); y = mul(y, wAi); }
        for (size_t pbo = 0; pbo < Ao; pbo++) {
            const u64 base = pow(p->w_n_inv, (u64)bitrev32((u32)pbo, p->lgAo)), base1024 = pow(base, 1024);
            u64 a0 = 1, a1 = ainv;
            for (size_t j = 0; j < 1024; j++) { t0[pbo * 1024 + j] = a0; t1[pbo * 1024 + j] = a1; a0 = mul(a0, base); a1 = mul(a1, base1024); }
        }
        GLP_TRY(upload(c, f, &p->tw_Ao));
        GLP_TRY(upload(c, b, &p->itw_Ao));
        GLP_TRY(upload(c, t0, &p->it0));
        GLP_TRY(upload(c, t1, &p->it1));
        *out = p.get();
        c->ntt_plans[lg] = p.release();
        return GLP_OK;
    }
    p->lgB = lg < NTT_LGB_MAX ? lg : NTT_LGB_MAX;
    p->lgA = lg - p->lgB;
    p->w_n = root_of_unity(lg);
    p->w_n_inv = inv(p->w_n);
    p->n_inv = inv((u64)1 << lg);
    auto table = [](int lgs, bool inverse) {
        size_t half = lgs > 0 ? ((size_t)1 << (lgs - 1)) : 0;
        std::vector<u64> t(half);
        u64 w = root_of_unity(lgs);
        if (inverse) w = inv(w);
        u64 x = 1;
        for (size_t j = 0; j < half; j++) { t[j] = x; x = mul(x, w); }
        return t;
    };
    GLP_TRY(upload(c, table(p->lgB, false), &p->tw_B));
    GLP_TRY(upload(c, table(p->lgB, true), &p->itw_B));
    GLP_TRY(upload(c, table(p->lgA, false), &p->tw_A));
    GLP_TRY(upload(c, table(p->lgA, true), &p->itw_A));
    {
        std::vector<u64> f(4096), b(4096);
        const u64 w = root_of_unity(12), wi = inv(w);
        u64 x = 1, y = 1;
        for (int j = 0; j < 4096; j++) { f[j] = x; b[j] = y; x = mul(x, w); y = mul(y, wi); }
        GLP_TRY(upload(c, f, &p->tw4096));
        GLP_TRY(upload(c, b, &p->itw4096));
        // the radix-16 butterflies hard-code w_16 = 2^156 = -2^60 and w_16^-1 = 2^36
        if (glf::pow((u64)2, (u64)156) != root_of_unity(4) || glf::pow((u64)2, (u64)36) != inv(root_of_unity(4)))
            return set_error(GLP_ERR_ARG, "internal: 16th root of unity is not the expected power of two");
    }
    *out = p.get();
    c->ntt_plans[lg] = p.release();
    return GLP_OK;
}

int get_lde_plan(glp_ctx *c, int lg, int rate_bits, u64 shift, LdePlan **out) {
    if (rate_bits < 0 || rate_bits > 4) return set_error(GLP_ERR_UNSUPPORTED, "rate_bits=%d outside 0..4", rate_bits);
    auto key = std::make_pair(std::make_pair(lg, rate_bits), shift);
    auto it = c->lde_plans.find(key);
    if (it != c->lde_plans.end()) { it->second->last_use = ++c->lde_clock; *out = it->second; return GLP_OK; }
    // The cache is keyed by the caller-supplied shift (glp_lde, FRI layers): bound it.  Evict the least recently used
    // plan that no other plan points at; nothing is in flight that could read it only if the stream is idle, so wait.
    while (c->lde_plans.size() >= LDE_PLAN_CACHE_MAX) {
        auto victim = c->lde_plans.end();
        for (auto jt = c->lde_plans.begin(); jt != c->lde_plans.end(); ++jt)
            if (jt->second->pins == 0 && (victim == c->lde_plans.end() || jt->second->last_use < victim->second->last_use)) victim = jt;
        if (victim == c->lde_plans.end()) break;
        GLP_HIP(hipStreamSynchronize(c->stream));
        if (victim->second->inner) const_cast<LdePlan *>(victim->second->inner)->pins--;
        delete victim->second;
        c->lde_plans.erase(victim);
    }
    NttPlan *np;
    GLP_TRY(get_ntt_plan(c, lg, &np));
    std::unique_ptr<LdePlan> p(new LdePlan());
    p->ntt = np; p->rate_bits = rate_bits; p->shift = shift;
    const int R = 1 << rate_bits;
    if (np->lgAo > 0) {
        const size_t Ao = (size_t)1 << np->lgAo;
        LdePlan *in;
        GLP_TRY(get_lde_plan(c, NTT_INNER_LG, rate_bits, pow(shift, (u64)Ao), &in));
        p->inner = in;
        in->pins++;
        struct Unpin { LdePlan *q; ~Unpin() { if (q) q->pins--; } } unpin{in};     // undone if this plan is not completed
        const u64 Wbig = root_of_unity(lg + rate_bits);
        std::vector<u64> t0(Ao * 1024), t1((size_t)R * Ao * 1024);
        for (size_t pbo = 0; pbo < Ao; pbo++) {
            const u64 k1o = bitrev32((u32)pbo, np->lgAo);
            const u64 base = pow(np->w_n, k1o), base1024 = pow(base, 1024);
            u64 a0 = 1;
            for (size_t j = 0; j < 1024; j++) { t0[pbo * 1024 + j] = a0; a0 = mul(a0, base); }
            for (int r = 0; r < R; r++) {
                u64 a1 = pow(mul(shift, pow(Wbig, (u64)r)), k1o);      // s_r^k1o
                for (size_t j = 0; j < 1024; j++) { t1[((size_t)r * Ao + pbo) * 1024 + j] = a1; a1 = mul(a1, base1024); }
            }
        }
        GLP_TRY(upload(c, t0, &p->t0));
        GLP_TRY(upload(c, t1, &p->t1));
        unpin.q = nullptr;
        p->last_use = ++c->lde_clock;
        *out = p.get();
        c->lde_plans[key] = p.release();
        return GLP_OK;
    }
    const size_t B = (size_t)1 << np->lgB, A = (size_t)1 << np->lgA;
    const u64 Wbig = root_of_unity(lg + rate_bits);
    std::vector<u64> s(R), pre((size_t)R * B);
    for (int r = 0; r < R; r++) {
        s[r] = mul(shift, pow(Wbig, (u64)r));
        u64 sA = pow(s[r], (u64)A);
        // pre[r][pl] = sA^bitrev_B(pl): walk k2 = 0..B-1 in natural order, scatter to bitrev slot
        u64 x = 1;
        for (size_t k2 = 0; k2 < B; k2++) {
            pre[(size_t)r * B + bitrev32((u32)k2, np->lgB)] = x;
            x = mul(x, sA);
        }
    }
    GLP_TRY(upload(c, s, &p->s_r));
    GLP_TRY(upload(c, pre, &p->pre));
    p->last_use = ++c->lde_clock;
    *out = p.get();
    c->lde_plans[key] = p.release();
    return GLP_OK;
}

void free_plans(glp_ctx *c) {
    for (auto &kv : c->lde_plans) delete kv.second;       // the plans' destructors free their tables
    c->lde_plans.clear();
    for (auto &kv : c->ntt_plans) delete kv.second;
    c->ntt_plans.clear();
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
constexpr int TPB = 256;
constexpr int MAXR_LDE = 16;
constexpr int EPT = (1 << NTT_LGB_MAX) / TPB;   // 16 elements per thread in the contiguous tile

__device__ __forceinline__ u64 dev_pow(u64 b, u32 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = mul(r, b); b = sqr(b); e >>= 1; }
    return r;
}

// Radix-2 decimation-in-time stages over a tile of 2^lgT elements in LDS (bit-reversed in,
// natural out).  tw[j] = w_T^j, j < T/2.
__device__ __forceinline__ void dit_stages(u64 *tile, const u64 *tw, int lgT, int tid) {
    const int halfT = 1 << (lgT > 0 ? lgT - 1 : 0);
    for (int s = 0; s < lgT; s++) {
        const int half = 1 << s;
        for (int bf = tid; bf < halfT; bf += TPB) {
            const int lo = bf & (half - 1);
            const int j = ((bf >> s) << (s + 1)) | lo;
            const u64 w = tw[lo << (lgT - 1 - s)];
            const u64 u = tile[j];
            const u64 v = mul(tile[j + half], w);
            tile[j] = add(u, v);
            tile[j + half] = sub(u, v);
        }
        __syncthreads();
    }
}
// Radix-2 decimation-in-frequency stages (natural in, bit-reversed out).
__device__ __forceinline__ void dif_stages(u64 *tile, const u64 *tw, int lgT, int tid) {
    const int halfT = 1 << (lgT > 0 ? lgT - 1 : 0);
    for (int s = lgT - 1; s >= 0; s--) {
        const int half = 1 << s;
        for (int bf = tid; bf < halfT; bf += TPB) {
            const int lo = bf & (half - 1);
            const int j = ((bf >> s) << (s + 1)) | lo;
            const u64 w = tw[lo << (lgT - 1 - s)];
            const u64 u = tile[j];
            const u64 v = tile[j + half];
            tile[j] = add(u, v);
            tile[j + half] = mul(sub(u, v), w);
        }
        __syncthreads();
    }
}

// Forward contiguous pass of the (coset) NTT, all R cosets from one read of the coefficients.
//   in : coeffs [ncols][n] bit-reversed
//   out: [ncols][R][n]; block pb of plane r receives  s_r^k1 * w_n^(q2*k1) * DFT_B(c * pre_r)[q2]
// grid = (n / B, ncols); dynamic LDS = (B + B/2 + 64 + 64 + 16) * 8 bytes
__global__ __launch_bounds__(TPB) void k_lde_contig(const u64 *__restrict__ coeffs, u64 *__restrict__ out,
                                                    const u64 *__restrict__ tw_B, const u64 *__restrict__ pre,
                                                    const u64 *__restrict__ s_r, u64 w_n, int lg, int lgA, int lgB,
                                                    int R) {
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    const int B = 1 << lgB, tid = threadIdx.x;
    u64 *tile = smem;
    u64 *tw = tile + B;
    u64 *T0 = tw + (B >> 1) + 1;
    u64 *T1 = T0 + 64;
    u64 *sk = T1 + 64;
    const u32 pb = blockIdx.x, col = blockIdx.y;
    const size_t n = (size_t)1 << lg;

    for (int j = tid; j < (B >> 1); j += TPB) tw[j] = tw_B[j];
    if (lgA > 0) {
        const u32 k1 = bitrev32(pb, lgA);
        if (tid < 128) {
            const u64 base = dev_pow(w_n, k1);
            if (tid < 64) T0[tid] = dev_pow(base, tid);
            else T1[tid - 64] = dev_pow(base, (u32)(tid - 64) << 6);
        } else if (tid < 128 + R) {
            sk[tid - 128] = dev_pow(s_r[tid - 128], k1);
        }
    }
    u64 c[EPT];
    const u64 *src = coeffs + (size_t)col * n + (size_t)pb * B;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int pl = tid + TPB * e;
        c[e] = pl < B ? src[pl] : 0;
    }
    __syncthreads();
    u64 post[EPT];
    if (lgA > 0) {
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int q2 = tid + TPB * e;
            post[e] = q2 < B ? mul(T1[q2 >> 6], T0[q2 & 63]) : 0;
        }
    }
    for (int r = 0; r < R; r++) {
        const u64 *pr = pre + (size_t)r * B;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int pl = tid + TPB * e;
            if (pl < B) tile[pl] = mul(c[e], pr[pl]);
        }
        __syncthreads();
        dit_stages(tile, tw, lgB, tid);
        u64 *dst = out + ((size_t)col * R + r) * n + (size_t)pb * B;
        if (lgA > 0) {
            const u64 skr = sk[r];
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int q2 = tid + TPB * e;
                if (q2 < B) dst[q2] = mul(tile[q2], mul(post[e], skr));
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int q2 = tid + TPB * e;
                if (q2 < B) dst[q2] = tile[q2];
            }
        }
        __syncthreads();
    }
}

// Strided pass: size-A transforms down the rows of an A x B matrix (row stride B), tile = A rows
// x 16 columns.  DIT (rows bit-reversed in -> natural out) for the forward transform, DIF
// (natural in -> bit-reversed out) for the inverse.  grid = (B / 16, planes)
template <bool DIF>
__global__ __launch_bounds__(TPB) void k_strided(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                 const u64 *__restrict__ tw_A, int lg, int lgA, int lgB) {
    __shared__ __attribute__((aligned(16))) u64 tile[(1 << NTT_LGA_MAX) * NTT_STRIDED_W];
    __shared__ u64 tw[1 << (NTT_LGA_MAX - 1)];
    const int A = 1 << lgA, tid = threadIdx.x;
    const size_t n = (size_t)1 << lg, B = (size_t)1 << lgB;
    const size_t base = (size_t)blockIdx.y * n + (size_t)blockIdx.x * NTT_STRIDED_W;
    const int w = tid & (NTT_STRIDED_W - 1), r0 = tid >> 4;
    for (int j = tid; j < (A >> 1); j += TPB) tw[j] = tw_A[j];
    for (int row = r0; row < A; row += TPB / NTT_STRIDED_W) tile[row * NTT_STRIDED_W + w] = in[base + (size_t)row * B + w];
    __syncthreads();
    const int nbf = (A >> 1) * NTT_STRIDED_W;
    if (DIF) {
        for (int s = lgA - 1; s >= 0; s--) {
            const int half = 1 << s;
            for (int idx = tid; idx < nbf; idx += TPB) {
                const int bf = idx >> 4, lo = bf & (half - 1);
                const int j = (((bf >> s) << (s + 1)) | lo) * NTT_STRIDED_W + w;
                const u64 t = tw[lo << (lgA - 1 - s)];
                const u64 u = tile[j], v = tile[j + half * NTT_STRIDED_W];
                tile[j] = add(u, v);
                tile[j + half * NTT_STRIDED_W] = mul(sub(u, v), t);
            }
            __syncthreads();
        }
    } else {
        for (int s = 0; s < lgA; s++) {
            const int half = 1 << s;
            for (int idx = tid; idx < nbf; idx += TPB) {
                const int bf = idx >> 4, lo = bf & (half - 1);
                const int j = (((bf >> s) << (s + 1)) | lo) * NTT_STRIDED_W + w;
                const u64 t = tw[lo << (lgA - 1 - s)];
                const u64 u = tile[j], v = mul(tile[j + half * NTT_STRIDED_W], t);
                tile[j] = add(u, v);
                tile[j + half * NTT_STRIDED_W] = sub(u, v);
            }
            __syncthreads();
        }
    }
    for (int row = r0; row < A; row += TPB / NTT_STRIDED_W) out[base + (size_t)row * B + w] = tile[row * NTT_STRIDED_W + w];
}

// Inverse contiguous pass: block pb (k1 = bitrev_A(pb)) of every column:
//   x[i2] = in[i2] * (w_n^-k1)^i2 * n^-1 ;  DIF over i2 ;  store in natural tile order (= bit-reversed k2)
// grid = (n / B, ncols)
__global__ __launch_bounds__(TPB) void k_intt_contig(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                     const u64 *__restrict__ itw_B, u64 w_n_inv, u64 n_inv, int lg,
                                                     int lgA, int lgB) {
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    const int B = 1 << lgB, tid = threadIdx.x;
    u64 *tile = smem;
    u64 *tw = tile + B;
    u64 *T0 = tw + (B >> 1) + 1;
    u64 *T1 = T0 + 64;
    const u32 pb = blockIdx.x, col = blockIdx.y;
    const size_t n = (size_t)1 << lg;
    for (int j = tid; j < (B >> 1); j += TPB) tw[j] = itw_B[j];
    if (lgA > 0 && tid < 128) {
        const u32 k1 = bitrev32(pb, lgA);
        const u64 base = dev_pow(w_n_inv, k1);
        if (tid < 64) T0[tid] = dev_pow(base, tid);
        else T1[tid - 64] = mul(dev_pow(base, (u32)(tid - 64) << 6), n_inv);
    }
    __syncthreads();
    const size_t off = (size_t)col * n + (size_t)pb * B;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int i2 = tid + TPB * e;
        if (i2 < B) {
            const u64 f = lgA > 0 ? mul(T1[i2 >> 6], T0[i2 & 63]) : n_inv;
            tile[i2] = mul(in[off + i2], f);
        }
    }
    __syncthreads();
    dif_stages(tile, tw, lgB, tid);
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int p = tid + TPB * e;
        if (p < B) out[off + p] = tile[p];
    }
}

__global__ void k_bitrev_copy(const u64 *__restrict__ in, u64 *__restrict__ out, int lg) {
    const size_t n = (size_t)1 << lg;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const size_t col = blockIdx.y;
    out[col * n + bitrev32((u32)p, lg)] = in[col * n + p];
}
// coset-major slot (r, q) -> natural index q * R + r
__global__ void k_lde_to_natural(const u64 *__restrict__ in, u64 *__restrict__ out, int lg, int rate_bits) {
    const size_t N = (size_t)1 << (lg + rate_bits);
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= N) return;
    const size_t col = blockIdx.y;
    const size_t r = pos >> lg, q = pos & (((size_t)1 << lg) - 1);
    out[col * N + (q << rate_bits) + r] = in[col * N + pos];
}

// ------------------------------------------------------------------------------------------
// radix-16 register kernels (B = 4096 = 16^3 contiguous, A = 256 = 16^2 strided)
//
// Each thread keeps 16 elements in registers and runs a 16-point DFT whose twiddles are the 16th
// roots of unity, i.e. +-2^(12k): shifts, no multiplier.  A 4096-point tile is three such steps
// with two LDS exchanges (Cooley-Tukey index map, general twiddles w_4096^(..) between steps),
// instead of 12 radix-2 stages with a barrier each.
// ------------------------------------------------------------------------------------------
template <bool INV, int J> struct W16 {     // w_16^(+-J) = (neg ? -1 : 1) * 2^sh
    static constexpr int e = ((INV ? 36 : 156) * J) % 192;
    static constexpr bool neg = e >= 96;
    static constexpr int sh = e % 96;
};
template <bool INV, int J> __device__ __forceinline__ void bf_dit(u64 &u, u64 &v) {   // (u, v) -> (u + w v, u - w v)
    u64 t;
    if constexpr (W16<INV, J>::sh == 0) t = v; else t = mul_pow2_c<W16<INV, J>::sh>(v);
    const u64 a = W16<INV, J>::neg ? sub(u, t) : add(u, t);
    const u64 b = W16<INV, J>::neg ? add(u, t) : sub(u, t);
    u = a; v = b;
}
template <bool INV, int J> __device__ __forceinline__ void bf_dif(u64 &u, u64 &v) {   // (u, v) -> (u + v, (u - v) w)
    const u64 a = add(u, v);
    const u64 d = W16<INV, J>::neg ? sub(v, u) : sub(u, v);
    if constexpr (W16<INV, J>::sh == 0) v = d; else v = mul_pow2_c<W16<INV, J>::sh>(d);
    u = a;
}
// 16-point DFT, decimation in time: x[bitrev4(k)] in -> X[q] out (natural)
template <bool INV> __device__ __forceinline__ void dft16_dit(u64 x[16]) {
#pragma unroll
    for (int j = 0; j < 16; j += 2) bf_dit<INV, 0>(x[j], x[j + 1]);
#pragma unroll
    for (int b = 0; b < 16; b += 4) { bf_dit<INV, 0>(x[b], x[b + 2]); bf_dit<INV, 4>(x[b + 1], x[b + 3]); }
#pragma unroll
    for (int b = 0; b < 16; b += 8) {
        bf_dit<INV, 0>(x[b], x[b + 4]); bf_dit<INV, 2>(x[b + 1], x[b + 5]);
        bf_dit<INV, 4>(x[b + 2], x[b + 6]); bf_dit<INV, 6>(x[b + 3], x[b + 7]);
    }
    bf_dit<INV, 0>(x[0], x[8]); bf_dit<INV, 1>(x[1], x[9]); bf_dit<INV, 2>(x[2], x[10]); bf_dit<INV, 3>(x[3], x[11]);
    bf_dit<INV, 4>(x[4], x[12]); bf_dit<INV, 5>(x[5], x[13]); bf_dit<INV, 6>(x[6], x[14]); bf_dit<INV, 7>(x[7], x[15]);
}
// 16-point DFT, decimation in frequency: x[i] in (natural) -> X[k] at x[bitrev4(k)]
template <bool INV> __device__ __forceinline__ void dft16_dif(u64 x[16]) {
    bf_dif<INV, 0>(x[0], x[8]); bf_dif<INV, 1>(x[1], x[9]); bf_dif<INV, 2>(x[2], x[10]); bf_dif<INV, 3>(x[3], x[11]);
    bf_dif<INV, 4>(x[4], x[12]); bf_dif<INV, 5>(x[5], x[13]); bf_dif<INV, 6>(x[6], x[14]); bf_dif<INV, 7>(x[7], x[15]);
#pragma unroll
    for (int b = 0; b < 16; b += 8) {
        bf_dif<INV, 0>(x[b], x[b + 4]); bf_dif<INV, 2>(x[b + 1], x[b + 5]);
        bf_dif<INV, 4>(x[b + 2], x[b + 6]); bf_dif<INV, 6>(x[b + 3], x[b + 7]);
    }
#pragma unroll
    for (int b = 0; b < 16; b += 4) { bf_dif<INV, 0>(x[b], x[b + 2]); bf_dif<INV, 4>(x[b + 1], x[b + 3]); }
#pragma unroll
    for (int j = 0; j < 16; j += 2) bf_dif<INV, 0>(x[j], x[j + 1]);
}
__device__ __forceinline__ int brev4(int x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); }

// 2^L-point DFTs (L <= 4) on registers: the first L stages of dft16_dit / the last L stages of dft16_dif -- the twiddle
// of a radix-2 stage depends on the stage, not on the transform length, and all of them are powers of w_16 (shifts).
template <bool INV, int L> __device__ __forceinline__ void dft_small_dit(u64 *x) {      // bit-reversed in -> natural out
    constexpr int S = 1 << L;
#pragma unroll
    for (int j = 0; j < S; j += 2) bf_dit<INV, 0>(x[j], x[j + 1]);
    if constexpr (L >= 2) {
#pragma unroll
        for (int b = 0; b < S; b += 4) { bf_dit<INV, 0>(x[b], x[b + 2]); bf_dit<INV, 4>(x[b + 1], x[b + 3]); }
    }
    if constexpr (L >= 3) {
#pragma unroll
        for (int b = 0; b < S; b += 8) {
            bf_dit<INV, 0>(x[b], x[b + 4]); bf_dit<INV, 2>(x[b + 1], x[b + 5]);
            bf_dit<INV, 4>(x[b + 2], x[b + 6]); bf_dit<INV, 6>(x[b + 3], x[b + 7]);
        }
    }
    if constexpr (L >= 4) {
        bf_dit<INV, 0>(x[0], x[8]); bf_dit<INV, 1>(x[1], x[9]); bf_dit<INV, 2>(x[2], x[10]); bf_dit<INV, 3>(x[3], x[11]);
        bf_dit<INV, 4>(x[4], x[12]); bf_dit<INV, 5>(x[5], x[13]); bf_dit<INV, 6>(x[6], x[14]); bf_dit<INV, 7>(x[7], x[15]);
    }
}
template <bool INV, int L> __device__ __forceinline__ void dft_small_dif(u64 *x) {      // natural in -> bit-reversed out
    constexpr int S = 1 << L;
    if constexpr (L >= 4) {
        bf_dif<INV, 0>(x[0], x[8]); bf_dif<INV, 1>(x[1], x[9]); bf_dif<INV, 2>(x[2], x[10]); bf_dif<INV, 3>(x[3], x[11]);
        bf_dif<INV, 4>(x[4], x[12]); bf_dif<INV, 5>(x[5], x[13]); bf_dif<INV, 6>(x[6], x[14]); bf_dif<INV, 7>(x[7], x[15]);
    }
    if constexpr (L >= 3) {
#pragma unroll
        for (int b = 0; b < S; b += 8) {
            bf_dif<INV, 0>(x[b], x[b + 4]); bf_dif<INV, 2>(x[b + 1], x[b + 5]);
            bf_dif<INV, 4>(x[b + 2], x[b + 6]); bf_dif<INV, 6>(x[b + 3], x[b + 7]);
        }
    }
    if constexpr (L >= 2) {
#pragma unroll
        for (int b = 0; b < S; b += 4) { bf_dif<INV, 0>(x[b], x[b + 2]); bf_dif<INV, 4>(x[b + 1], x[b + 3]); }
    }
#pragma unroll
    for (int j = 0; j < S; j += 2) bf_dif<INV, 0>(x[j], x[j + 1]);
}

// Outer pass of the three-pass transform (n = A' * 2^20, A' = 2^L <= 16): one thread owns position q of all A' blocks
// of a plane, so the outer twiddle and the A'-point transform happen in registers in ONE sweep over the data (the
// two-kernel form -- twiddle sweep, then 4-row LDS tiles -- moved the LDE through HBM twice and ran the DFT on 64-element
// tiles).  Forward (DIF = false): x[pbo] *= t1[r][pbo][q >> 10] * t0[pbo][q & 1023], then DIT over pbo (rows stored
// bit-reversed).  Inverse: DIF over pbo, then the twiddle (1/A' folded into t1).  grid = (M / 256, planes)
// TW = false: no twiddle, the plain strided pass of a two-pass transform with A <= 16 rows (lg 13..16).
template <bool DIF, int L, bool TW>
__global__ __launch_bounds__(256) void k_outer(const u64 *__restrict__ in, u64 *__restrict__ out, const u64 *__restrict__ t0,
                                               const u64 *__restrict__ t1, int lgM, int R) {
    constexpr int A = 1 << L;
    const size_t M = (size_t)1 << lgM;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 plane = blockIdx.y, r = plane % (u32)R;
    const size_t base = ((size_t)plane << L) * M + q;
    u64 x[A];
#pragma unroll
    for (int pbo = 0; pbo < A; pbo++) x[pbo] = in[base + (size_t)pbo * M];
    if (!DIF) {
        if constexpr (TW) {
#pragma unroll
            for (int pbo = 0; pbo < A; pbo++)
                x[pbo] = mul_c(x[pbo], mul_nc(t1[(((size_t)r << L) + pbo) * 1024 + (q >> 10)], t0[(size_t)pbo * 1024 + (q & 1023)]));
        }
        dft_small_dit<false, L>(x);
    } else {
        dft_small_dif<true, L>(x);
        if constexpr (TW) {
#pragma unroll
            for (int pbo = 0; pbo < A; pbo++)
                x[pbo] = mul_c(x[pbo], mul_nc(t1[(((size_t)r << L) + pbo) * 1024 + (q >> 10)], t0[(size_t)pbo * 1024 + (q & 1023)]));
        }
    }
#pragma unroll
    for (int pbo = 0; pbo < A; pbo++) out[base + (size_t)pbo * M] = x[pbo];
}
template <bool DIF, bool TW>
static void launch_outer(glp_ctx *c, const u64 *in, u64 *out, const u64 *t0, const u64 *t1, int lgAo, int lgM, int R, u32 planes) {
    const dim3 g((unsigned)(((size_t)1 << lgM) / 256), planes), b(256);
    switch (lgAo) {
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_outer<DIF, 1, TW>), g, b, 0, c->stream, in, out, t0, t1, lgM, R); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_outer<DIF, 2, TW>), g, b, 0, c->stream, in, out, t0, t1, lgM, R); break;
    case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_outer<DIF, 3, TW>), g, b, 0, c->stream, in, out, t0, t1, lgM, R); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_outer<DIF, 4, TW>), g, b, 0, c->stream, in, out, t0, t1, lgM, R); break;
    }
}

constexpr int R16_LDS = 17 * 256;       // 16 x 16 x 16 tile, rows of 16 padded to 17 (bank spread for stride-16 reads)

// Forward contiguous pass, B = 4096 (same contract as k_lde_contig).  k2 = 256 ka + 16 kb + kc is stored at
// tile slot 256 rc + 16 rb + ra (r* = bitrev4(k*)); q2 = qa + 16 qb + 256 qc.
__global__ __launch_bounds__(TPB) void k_lde_contig16(const u64 *__restrict__ coeffs, u64 *__restrict__ out,
                                                      const u64 *__restrict__ tw4096, const u64 *__restrict__ pre,
                                                      const u64 *__restrict__ s_r, u64 w_n, int lg, int lgA, int R, int lgAo) {
    __shared__ __attribute__((aligned(16))) u64 lds[R16_LDS];
    __shared__ u64 T0[64], T1[64], sk[MAXR_LDE];
    constexpr int B = 4096;
    const int tid = threadIdx.x;
    const u32 pb = blockIdx.x, col = blockIdx.y;
    const size_t n = (size_t)1 << lg;
    if (lgA > 0) {
        const u32 k1 = bitrev32(pb, lgA);
        if (tid < 128) {
            const u64 base = dev_pow(w_n, k1);
            if (tid < 64) T0[tid] = dev_pow(base, tid);
            else T1[tid - 64] = dev_pow(base, (u32)(tid - 64) << 6);
        } else if (tid < 128 + R) {
            sk[tid - 128] = dev_pow(s_r[tid - 128], k1);
        }
    }
    // step-1 role: tid = 16 rc + rb, owns slots 16 tid .. 16 tid + 15 (ra = 0..15).  The coefficients are read from
    // HBM once and held in 32 VGPRs across the coset loop (re-reading them per coset showed up as 7x the algorithmic
    // fetch bytes in the FETCH_SIZE counter).
    u64 c[16];
    {
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(coeffs + (size_t)col * n + (size_t)pb * B + 16 * tid);
#pragma unroll
        for (int e = 0; e < 8; e++) { const ulonglong2 v = src[e]; c[2 * e] = v.x; c[2 * e + 1] = v.y; }
    }
    __syncthreads();
    const int hi4 = tid >> 4, lo4 = tid & 15;
    u64 postP[16];                       // (w_n^k1)^q2 for this thread's 16 outputs, shared by all cosets
    if (lgA > 0) {
#pragma unroll
        for (int qc = 0; qc < 16; qc++) { const int q2 = tid + 256 * qc; postP[qc] = mul_nc(T1[q2 >> 6], T0[q2 & 63]); }
    }
    const int kb1_ = brev4(lo4);         // step 1: rb = lo4
    const int kc2_ = brev4(lo4);         // step 2: rc = lo4, qa = hi4
    for (int r = 0; r < R; r++) {
        u64 x[16];
#ifdef GLP_LDE_NO_HOIST
        int kb1 = kb1_, kc2 = kc2_;
        asm volatile("" : "+v"(kb1), "+v"(kc2));      // keep twiddle loads inside the loop (register pressure)
#else
        const int kb1 = kb1_, kc2 = kc2_;
#endif
        {
            const ulonglong2 *pr = reinterpret_cast<const ulonglong2 *>(pre + (size_t)r * B + 16 * tid);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const ulonglong2 v = pr[e];
                x[2 * e] = mul_c(c[2 * e], v.x);
                x[2 * e + 1] = mul_c(c[2 * e + 1], v.y);
            }
        }
        dft16_dit<false>(x);                                 // over ka -> qa
#pragma unroll
        for (int qa = 0; qa < 16; qa++) {                    // twiddle w_256^(qa kb); slot A1[qa][rc = hi4][rb = lo4]
            const u64 v = qa == 0 ? x[0] : mul_c(x[qa], tw4096[16 * qa * kb1]);
            lds[17 * (16 * qa + hi4) + lo4] = v;
        }
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 16; rb++) x[rb] = lds[17 * tid + rb];      // step-2 role: tid = 16 qa + rc
        __syncthreads();
        dft16_dit<false>(x);                                 // over kb -> qb
#pragma unroll
        for (int qb = 0; qb < 16; qb++) {                    // twiddle w_4096^((qa + 16 qb) kc); slot A2[qb][qa][rc]
            const u64 v = mul_c(x[qb], tw4096[(hi4 + 16 * qb) * kc2]);
            lds[17 * (hi4 + 16 * qb) + lo4] = v;
        }
        __syncthreads();
#pragma unroll
        for (int rc = 0; rc < 16; rc++) x[rc] = lds[17 * tid + rc];      // step-3 role: tid = qa + 16 qb
        __syncthreads();
        dft16_dit<false>(x);                                 // over kc -> qc ; q2 = tid + 256 qc
        // lgAo > 0: "col" is (column, outer block pbo); planes are laid out [column][r][pbo]
        u64 *dst = out + (((((size_t)(col >> lgAo) * R + r) << lgAo) + (col & ((1u << lgAo) - 1))) * n) + (size_t)pb * B + tid;
        if (lgA > 0) {
            const u64 skr = sk[r];
#pragma unroll
            for (int qc = 0; qc < 16; qc++) dst[256 * qc] = mul_c(x[qc], mul_nc(postP[qc], skr));
        } else {
#pragma unroll
            for (int qc = 0; qc < 16; qc++) dst[256 * qc] = x[qc];
        }
    }
}

// Inverse contiguous pass, B = 4096 (same contract as k_intt_contig).  i2 = 256 ia + 16 ib + ic natural in;
// k2 = ka + 16 kb + 256 kc out at slot 256 ra + 16 rb + rc.
__global__ __launch_bounds__(TPB) void k_intt_contig16(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                       const u64 *__restrict__ itw4096, u64 w_n_inv, u64 n_inv, int lg,
                                                       int lgA) {
    __shared__ __attribute__((aligned(16))) u64 lds[R16_LDS];
    __shared__ u64 T0[64], T1[64];
    constexpr int B = 4096;
    const int tid = threadIdx.x;
    const u32 pb = blockIdx.x, col = blockIdx.y;
    const size_t n = (size_t)1 << lg;
    if (lgA > 0 && tid < 128) {
        const u32 k1 = bitrev32(pb, lgA);
        const u64 base = dev_pow(w_n_inv, k1);
        if (tid < 64) T0[tid] = dev_pow(base, tid);
        else T1[tid - 64] = mul(dev_pow(base, (u32)(tid - 64) << 6), n_inv);
    }
    __syncthreads();
    const size_t off = (size_t)col * n + (size_t)pb * B;
    const int hi4 = tid >> 4, lo4 = tid & 15;
    u64 x[16];
    // step 1: tid = 16 ib + ic = i2 mod 256; elements ia = 0..15 at i2 = 256 ia + tid
#pragma unroll
    for (int ia = 0; ia < 16; ia++) {
        const int i2 = 256 * ia + tid;
        const u64 f = lgA > 0 ? mul_nc(T1[i2 >> 6], T0[i2 & 63]) : n_inv;
        x[ia] = mul_c(in[off + i2], f);
    }
    dft16_dif<true>(x);                                      // over ia -> ka at x[ra]
#pragma unroll
    for (int ra = 0; ra < 16; ra++) {                        // twiddle w^-(tid ka); slot [ra][ic = lo4][ib = hi4]
        const int ka = brev4(ra);
        const u64 v = ka == 0 ? x[ra] : mul_c(x[ra], itw4096[tid * ka]);
        lds[17 * (16 * ra + lo4) + hi4] = v;
    }
    __syncthreads();
#pragma unroll
    for (int ib = 0; ib < 16; ib++) x[ib] = lds[17 * tid + ib];          // step-2 role: tid = 16 ra + ic
    __syncthreads();
    dft16_dif<true>(x);                                      // over ib -> kb at x[rb]
#pragma unroll
    for (int rb = 0; rb < 16; rb++) {                        // twiddle w_256^-(ic kb); slot [ra = hi4][rb][ic = lo4]
        const int kb = brev4(rb);
        const u64 v = kb == 0 ? x[rb] : mul_c(x[rb], itw4096[16 * lo4 * kb]);
        lds[17 * (16 * hi4 + rb) + lo4] = v;
    }
    __syncthreads();
#pragma unroll
    for (int ic = 0; ic < 16; ic++) x[ic] = lds[17 * tid + ic];          // step-3 role: tid = 16 ra + rb
    dft16_dif<true>(x);                                      // over ic -> kc at x[rc]; slot 16 tid + rc
    ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(out + off + 16 * tid);
#pragma unroll
    for (int e = 0; e < 8; e++) { ulonglong2 v; v.x = x[2 * e]; v.y = x[2 * e + 1]; dst[e] = v; }
}

// Strided pass for A = 256: tile 256 rows x 16 columns, thread = (column w, group g).
//  DIT (forward): row pb = 16 rb + ra holds k1 = 16 ka + kb (r* = bitrev4(k*)); out row q1 = qa + 16 qb.
//  DIF (inverse): row i1 = 16 ia + ib natural in; out k1 = ka + 16 kb at row 16 ra + rb.
constexpr int S16_ROW = 272;            // 16 x 16 (+16 pad) words per first-index slab
template <bool DIF>
__global__ __launch_bounds__(TPB) void k_strided16(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                   const u64 *__restrict__ tw4096, int lg, int lgB) {
    __shared__ __attribute__((aligned(16))) u64 lds[16 * S16_ROW];
    const int tid = threadIdx.x, w = tid & 15, g = tid >> 4;
    const size_t n = (size_t)1 << lg, B = (size_t)1 << lgB;
    const size_t base = (size_t)blockIdx.y * n + (size_t)blockIdx.x * NTT_STRIDED_W + w;
    u64 x[16];
    if (!DIF) {
#pragma unroll
        for (int ra = 0; ra < 16; ra++) x[ra] = in[base + (size_t)(16 * g + ra) * B];     // g = rb
        dft16_dit<false>(x);                                                              // over ka -> qa
        const int kb = brev4(g);
#pragma unroll
        for (int qa = 0; qa < 16; qa++) lds[qa * S16_ROW + 16 * g + w] = qa == 0 ? x[0] : mul_c(x[qa], tw4096[16 * qa * kb]);
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 16; rb++) x[rb] = lds[g * S16_ROW + 16 * rb + w];           // g = qa
        dft16_dit<false>(x);                                                              // over kb -> qb
#pragma unroll
        for (int qb = 0; qb < 16; qb++) out[base + (size_t)(g + 16 * qb) * B] = x[qb];
    } else {
#pragma unroll
        for (int ia = 0; ia < 16; ia++) x[ia] = in[base + (size_t)(16 * ia + g) * B];     // g = ib
        dft16_dif<true>(x);                                                               // over ia -> ka at x[ra]
#pragma unroll
        for (int ra = 0; ra < 16; ra++) {
            const int ka = brev4(ra);
            lds[ra * S16_ROW + 16 * g + w] = ka == 0 ? x[ra] : mul_c(x[ra], tw4096[16 * g * ka]);   // itw table passed in
        }
        __syncthreads();
#pragma unroll
        for (int ib = 0; ib < 16; ib++) x[ib] = lds[g * S16_ROW + 16 * ib + w];           // g = ra
        dft16_dif<true>(x);                                                               // over ib -> kb at x[rb]
#pragma unroll
        for (int rb = 0; rb < 16; rb++) out[base + (size_t)(16 * g + rb) * B] = x[rb];
    }
}

// Strided pass for A = 16 E, E = 2^EL in {2, 4, 8} (traces of 2^17 .. 2^19 rows): one radix-16 step on registers, an LDS
// exchange with the twiddle w_A^(kb qa), then E-point transforms on registers (all remaining twiddles are powers of w_16).
// Tile = A rows x Wc columns, Wc = 256 / E (4096 elements, 16 per thread).
//  DIT (forward): row pb = 16 rb + ra holds k1 = E ka + kb (ra = bitrev4(ka), rb = bitrev_EL(kb)); out row q1 = qa + 16 qb.
//  DIF (inverse): row i1 = E ia + ib natural in; out k1 = ka + 16 kb at row E ra + rb.
template <bool DIF, int EL>
__global__ __launch_bounds__(TPB) void k_strided16e(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                    const u64 *__restrict__ tw4096, int lg, int lgB) {
    constexpr int E = 1 << EL, Wc = 256 >> EL;
    __shared__ __attribute__((aligned(16))) u64 lds[4096];
    const int tid = threadIdx.x, w = tid & (Wc - 1), g = tid >> (8 - EL);
    const size_t n = (size_t)1 << lg, B = (size_t)1 << lgB;
    const size_t base = (size_t)blockIdx.y * n + (size_t)blockIdx.x * Wc;
    u64 x[16];
    if (!DIF) {
#pragma unroll
        for (int ra = 0; ra < 16; ra++) x[ra] = in[base + (size_t)(16 * g + ra) * B + w];     // g = rb
        dft16_dit<false>(x);                                                                  // over ka -> qa
        const int kb = (int)(__brev((unsigned)g) >> (32 - EL));
#pragma unroll
        for (int qa = 0; qa < 16; qa++) lds[(qa * E + g) * Wc + w] = qa == 0 ? x[0] : mul_c(x[qa], tw4096[Wc * kb * qa]);
    } else {
#pragma unroll
        for (int ia = 0; ia < 16; ia++) x[ia] = in[base + (size_t)(E * ia + g) * B + w];      // g = ib
        dft16_dif<true>(x);                                                                   // over ia -> ka at x[ra]
#pragma unroll
        for (int ra = 0; ra < 16; ra++) {
            const int ka = brev4(ra);
            lds[(ra * E + g) * Wc + w] = ka == 0 ? x[ra] : mul_c(x[ra], tw4096[Wc * g * ka]);  // itw table passed in
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16 / E; j++) {
        const int id = j * TPB + tid, w2 = id & (Wc - 1), a16 = id >> (8 - EL);                 // a16 = qa (DIT) / ra (DIF)
        u64 y[E];
#pragma unroll
        for (int gg = 0; gg < E; gg++) y[gg] = lds[(a16 * E + gg) * Wc + w2];
        if (!DIF) {
            dft_small_dit<false, EL>(y);                                                       // over kb -> qb
#pragma unroll
            for (int qb = 0; qb < E; qb++) out[base + (size_t)(a16 + 16 * qb) * B + w2] = y[qb];
        } else {
            dft_small_dif<true, EL>(y);                                                        // over ib -> kb at y[rb]
#pragma unroll
            for (int rb = 0; rb < E; rb++) out[base + (size_t)(E * a16 + rb) * B + w2] = y[rb];
        }
    }
}
template <bool DIF>
static void launch_strided16e(glp_ctx *c, const u64 *in, u64 *out, const u64 *tw, int lg, int lgA, int lgB, u32 planes) {
    const int el = lgA - 4;
    const dim3 g((unsigned)(((size_t)1 << lgB) / (256u >> el)), planes), b(TPB);
    switch (el) {
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_strided16e<DIF, 1>), g, b, 0, c->stream, in, out, tw, lg, lgB); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_strided16e<DIF, 2>), g, b, 0, c->stream, in, out, tw, lg, lgB); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_strided16e<DIF, 3>), g, b, 0, c->stream, in, out, tw, lg, lgB); break;
    }
}

// ------------------------------------------------------------------------------------------
// Strided pass for A = 32 * 2^L2 rows, L2 in {4, 5} (traces of 2^21 / 2^22 rows): the tile of A rows x Wc columns is
// 64-128 KB -- it fits the 160 KB LDS of gfx950 and of no earlier CDNA part, and it is what keeps these sizes at TWO
// passes over HBM (the alternative, a third k_outer pass over 2^20-point blocks, moves the whole LDE through HBM once
// more).  Every 64th root of unity is a power of two (w_64 = 2^39), so the 32-point and 16/32-point register transforms
// are shifts; one general multiplication per element (the inter-step twiddle w_A^(qa kb)) as in k_strided16.
//  DIT (forward): row pb = 32 rb + ra holds k1 = A2 ka + kb (ra = bitrev5(ka), rb = bitrev_L2(kb)); out row q1 = qa + 32 qb.
//  DIF (inverse): row i1 = A2 ia + ib natural in; out k1 = ka + 32 kb at row A2 ra + rb.
// Threads = A2 * Wc (step 1: one (column, rb) each, 32 elements in registers; step 2: 32 / A2 items of A2 elements).
// LDS: 32 slabs [rb][w] of A2 * Wc words, padded by Wc words so that the step-2 reads of a 32-lane group (Wc columns of
// 32 / Wc consecutive slabs) fall on distinct banks.  grid = (B / Wc, planes); dynamic LDS = strided32_lds_bytes().
// ------------------------------------------------------------------------------------------
template <bool INV, int J> struct W64 {     // w_64^(+-J) = (neg ? -1 : 1) * 2^sh
    static constexpr int e = ((INV ? 153 : 39) * J) % 192;
    static constexpr bool neg = e >= 96;
    static constexpr int sh = e % 96;
};
template <bool INV, int J> __device__ __forceinline__ void bf64_dit(u64 &u, u64 &v) {   // (u, v) -> (u + w v, u - w v)
    u64 t;
    if constexpr (W64<INV, J>::sh == 0) t = v; else t = mul_pow2_c<W64<INV, J>::sh>(v);
    const u64 a = W64<INV, J>::neg ? sub(u, t) : add(u, t);
    const u64 b = W64<INV, J>::neg ? add(u, t) : sub(u, t);
    u = a; v = b;
}
template <bool INV, int J> __device__ __forceinline__ void bf64_dif(u64 &u, u64 &v) {   // (u, v) -> (u + v, (u - v) w)
    const u64 a = add(u, v);
    const u64 d = W64<INV, J>::neg ? sub(v, u) : sub(u, v);
    if constexpr (W64<INV, J>::sh == 0) v = d; else v = mul_pow2_c<W64<INV, J>::sh>(d);
    u = a;
}
// stage S of a 2^L-point transform on registers: butterflies at distance 2^S with twiddles w_(2^(S+1))^j = w_64^(j (32 >> S))
template <bool INV, int L, int S, int... Js>
__device__ __forceinline__ void reg_stage_dit(u64 *x, std::integer_sequence<int, Js...>) {
    constexpr int half = 1 << S, N = 1 << L;
#pragma unroll
    for (int b = 0; b < N; b += 2 * half) { (bf64_dit<INV, Js * (32 >> S)>(x[b + Js], x[b + Js + half]), ...); }
}
template <bool INV, int L, int S, int... Js>
__device__ __forceinline__ void reg_stage_dif(u64 *x, std::integer_sequence<int, Js...>) {
    constexpr int half = 1 << S, N = 1 << L;
#pragma unroll
    for (int b = 0; b < N; b += 2 * half) { (bf64_dif<INV, Js * (32 >> S)>(x[b + Js], x[b + Js + half]), ...); }
}
template <bool INV, int L> __device__ __forceinline__ void dft_reg_dit(u64 *x) {         // bit-reversed in -> natural out, L <= 6
    if constexpr (L >= 1) reg_stage_dit<INV, L, 0>(x, std::make_integer_sequence<int, 1>{});
    if constexpr (L >= 2) reg_stage_dit<INV, L, 1>(x, std::make_integer_sequence<int, 2>{});
    if constexpr (L >= 3) reg_stage_dit<INV, L, 2>(x, std::make_integer_sequence<int, 4>{});
    if constexpr (L >= 4) reg_stage_dit<INV, L, 3>(x, std::make_integer_sequence<int, 8>{});
    if constexpr (L >= 5) reg_stage_dit<INV, L, 4>(x, std::make_integer_sequence<int, 16>{});
    if constexpr (L >= 6) reg_stage_dit<INV, L, 5>(x, std::make_integer_sequence<int, 32>{});
}
template <bool INV, int L> __device__ __forceinline__ void dft_reg_dif(u64 *x) {         // natural in -> bit-reversed out, L <= 6
    if constexpr (L >= 6) reg_stage_dif<INV, L, 5>(x, std::make_integer_sequence<int, 32>{});
    if constexpr (L >= 5) reg_stage_dif<INV, L, 4>(x, std::make_integer_sequence<int, 16>{});
    if constexpr (L >= 4) reg_stage_dif<INV, L, 3>(x, std::make_integer_sequence<int, 8>{});
    if constexpr (L >= 3) reg_stage_dif<INV, L, 2>(x, std::make_integer_sequence<int, 4>{});
    if constexpr (L >= 2) reg_stage_dif<INV, L, 1>(x, std::make_integer_sequence<int, 2>{});
    if constexpr (L >= 1) reg_stage_dif<INV, L, 0>(x, std::make_integer_sequence<int, 1>{});
}
constexpr size_t strided32_lds_bytes(int L2, int LW) { return ((size_t)32 * (((size_t)1 << L2) + 1) * ((size_t)1 << LW) + ((size_t)32 << L2)) * sizeof(u64); }

// One tile = three phases (load, two register transforms around the LDS exchange, store).  With 64-128 KB of LDS per block
// only one or two blocks fit a CU, so the phases of DIFFERENT blocks cannot cover each other as they do for the 32 KB tiles of
// k_strided16 (measured: load + compute + store in sequence, 29 ms for the 2^22 wires LDE against 17 ms of HBM time).  The
// kernel therefore walks TL tiles (same columns, consecutive planes) per block and software-pipelines them: the loads of tile
// t + 1 are issued right after tile t's step-1 values have gone to LDS and stay in flight across the exchange, the second
// transform and the stores of tile t (~130 VGPRs at 2 waves per SIMD).
// Addresses are "uniform base (SGPR pair) + 32-bit per-thread byte offset": the row part of every load / store is the same for
// all threads, so it stays in scalar registers (global_load ... v_off, s[base]) instead of 32 + 32 per-thread 64-bit addresses
// that the compiler would otherwise keep live across the tile loop (the first form of this kernel needed > 256 VGPRs and spilled).
__device__ __forceinline__ u64 ld_su(const u64 *ubase, u32 byte_off) { return *reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(ubase) + byte_off); }
__device__ __forceinline__ void st_su(u64 *ubase, u32 byte_off, u64 v) { *reinterpret_cast<u64 *>(reinterpret_cast<char *>(ubase) + byte_off) = v; }
template <bool DIF, int L2, int LW>
__device__ __forceinline__ void s32_load(u64 *x, const u64 *__restrict__ tile /* uniform: first element of the tile */, size_t B, int g, int w) {
    constexpr int A1 = 32, A2 = 1 << L2;
    const u32 toff = (u32)(((size_t)(DIF ? g : A1 * g) * B + (size_t)w) * 8);                      // DIT: g = rb, r = ra; DIF: g = ib, r = ia
#pragma unroll
    for (int r = 0; r < A1; r++) x[r] = ld_su(tile + (size_t)(DIF ? A2 * r : r) * B, toff);
}
// step 1: 32-point transform on registers, inter-step twiddle, to LDS.  The twiddles w_A^(qa kb) come from a per-block LDS copy
// twl[g][32] (4-8 KB behind the tile), NOT from global memory: a global load issued here would sit behind the stores of the
// previous tile in the in-order vmcnt queue, and waiting for it would wait for those stores -- the pipelining would be gone.
template <bool DIF, int L2, int LW>
__device__ __forceinline__ void s32_step1(u64 *x, u64 *lds32, const u64 *twl, int g, int w) {
    constexpr int L1 = 5, A1 = 32, A2 = 1 << L2, Wc = 1 << LW, SLAB = (A2 + 1) * Wc;
    if (!DIF) dft_reg_dit<false, L1>(x);           // over ka -> qa
    else dft_reg_dif<true, L1>(x);                 // over ia -> ka at x[ra]
    const u64 *tw = twl + A1 * g;
#pragma unroll
    for (int q = 0; q < A1; q++) lds32[q * SLAB + g * Wc + w] = q == 0 ? x[0] : mul_c(x[q], tw[q]);      // q = qa (DIT) / ra (DIF); tw[0] = 1
}
template <bool DIF, int L2, int LW>
__device__ __forceinline__ void s32_step2(const u64 *lds32, u64 *__restrict__ tile, size_t B, int tid) {
    constexpr int A1 = 32, A2 = 1 << L2, Wc = 1 << LW, T = A2 * Wc, SLAB = (A2 + 1) * Wc;
#pragma unroll
    for (int j = 0; j < A1 / A2; j++) {
        const int id = j * T + tid, w2 = id & (Wc - 1), a = id >> LW;                              // a = qa (DIT) / ra (DIF)
        u64 y[A2];
#pragma unroll
        for (int gg = 0; gg < A2; gg++) y[gg] = lds32[a * SLAB + gg * Wc + w2];
        const u32 toff = (u32)(((size_t)(DIF ? A2 * a : a) * B + (size_t)w2) * 8);
        if (!DIF) {
            dft_reg_dit<false, L2>(y);                                                             // over kb -> qb; row a + 32 qb
#pragma unroll
            for (int qb = 0; qb < A2; qb++) st_su(tile + (size_t)(A1 * qb) * B, toff, y[qb]);
        } else {
            dft_reg_dif<true, L2>(y);                                                              // over ib -> kb at y[rb]; row A2 a + rb
#pragma unroll
            for (int rb = 0; rb < A2; rb++) st_su(tile + (size_t)rb * B, toff, y[rb]);
        }
    }
}
// grid = (B / Wc, ceil(planes / TL)); block (bx, by) transforms tiles (columns bx, planes by * TL .. min(planes, by * TL + TL) - 1)
template <bool DIF, int L2, int LW>
__global__ __launch_bounds__(1 << (L2 + LW)) void k_strided32(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                             const u64 *__restrict__ tw4096, int lg, int lgB, u32 planes, u32 TL) {
    constexpr int Wc = 1 << LW;
    constexpr int L1 = 5, A1 = 32, A2 = 1 << L2, T = A2 * Wc, SLAB = (A2 + 1) * Wc, TWS = 4096 >> (L1 + L2);
    extern __shared__ __attribute__((aligned(16))) u64 lds32[];
    u64 *twl = lds32 + A1 * SLAB;                                   // [A2][32] inter-step twiddles
    const int tid = threadIdx.x, w = tid & (Wc - 1), g = tid >> LW;
    const size_t n = (size_t)1 << lg, B = (size_t)1 << lgB;
    for (int e = tid; e < A2 * A1; e += T) {                        // DIT: w_A^(qa kb), kb = bitrev_L2(g); DIF: w_A^-(ib ka), ka = bitrev5(ra) (itw table passed in)
        const int gg = e >> L1, q = e & (A1 - 1);
        const int k = DIF ? gg * (int)(__brev((unsigned)q) >> (32 - L1)) : q * (int)(__brev((unsigned)gg) >> (32 - L2));
        twl[e] = tw4096[TWS * k];
    }
    const u32 p0 = blockIdx.y * TL, cnt = min(TL, planes - p0);
    const size_t base0 = (size_t)p0 * n + (size_t)blockIdx.x * Wc;      // uniform
    in += base0; out += base0;
    // Rotated loop: step 1 of tile i + 1 sits at the END of iteration i, behind "loads(i + 1), stores(i)" issued in that order in
    // the same iteration, so the compiler's wait for the loads is vmcnt(32 + ..): the 32 younger stores stay in flight.  (With
    // step 1 at the top of the loop the wait is merged with the prologue's state -- no stores yet -- and drains the stores of the
    // previous tile every iteration.)  One copy of each step in the loop body: two would not fit the instruction cache.
    u64 x[32];
    s32_load<DIF, L2, LW>(x, in, B, g, w);
    __syncthreads();                                                // twl
    s32_step1<DIF, L2, LW>(x, lds32, twl, g, w);
    for (u32 i = 0; i < cnt; i++) {
        const bool more = i + 1 < cnt;
        if (more) s32_load<DIF, L2, LW>(x, in + (size_t)(i + 1) * n, B, g, w);                     // in flight across the exchange and the stores of tile i
        __syncthreads();
        s32_step2<DIF, L2, LW>(lds32, out + (size_t)i * n, B, tid);
        __syncthreads();
        if (more) s32_step1<DIF, L2, LW>(x, lds32, twl, g, w);
    }
}
template <bool DIF, int L2, int LW>
static int launch_strided32_t(glp_ctx *c, const u64 *in, u64 *out, const u64 *tw, int lg, int lgB, u32 planes) {
    constexpr size_t bytes = strided32_lds_bytes(L2, LW);
    static bool attr_set[64] = {};               // per device: tiles above 64 KB need the dynamic-LDS attribute raised once
    if (!attr_set[c->device & 63]) {
        GLP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_strided32<DIF, L2, LW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_set[c->device & 63] = true;
    }
    const u32 TL = (u32)std::max(1, std::min<int>(c->strided32_tl, (int)planes));
    const dim3 g((unsigned)(((size_t)1 << lgB) >> LW), (planes + TL - 1) / TL), b(1u << (L2 + LW));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_strided32<DIF, L2, LW>), g, b, bytes, c->stream, in, out, tw, lg, lgB, planes, TL);
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}
template <bool DIF>
static int launch_strided32(glp_ctx *c, const u64 *in, u64 *out, const u64 *tw, int lg, int lgA, int lgB, u32 planes) {
    const bool narrow = c->strided32_lw == 3;
    if (lgA == 9) return narrow ? launch_strided32_t<DIF, 4, 3>(c, in, out, tw, lg, lgB, planes) : launch_strided32_t<DIF, 4, 4>(c, in, out, tw, lg, lgB, planes);
    return narrow ? launch_strided32_t<DIF, 5, 3>(c, in, out, tw, lg, lgB, planes) : launch_strided32_t<DIF, 5, 4>(c, in, out, tw, lg, lgB, planes);
}

// ------------------------------------------------------------------------------------------
// Transforms of at most 16 points (lg <= 4: the 2^3-row zkdsa trace and the last FRI layers): one thread per (column, coset),
// the whole transform on registers with shift twiddles.  The tiled kernels give such a column a 256-thread workgroup and an LDS
// tile for eight elements; a batch of 256 zkdsa proofs is 34 560 wire columns, and that was 0.5 ms of its 11.6 ms.
//   k_lde_small:  out[(col R + r) n + q] = sum_k c_k s_r^k w_n^(q k),  c_k at slot bitrev(k), pre[r][p] = s_r^bitrev(p)
//   k_intt_small: natural values in -> coefficients (times 1/n) at bit-reversed slots
// ------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(256) void k_lde_small(const u64 *__restrict__ coeffs, u64 *__restrict__ out, const u64 *__restrict__ pre, u32 total /* ncols * R */, int rate_bits) {
    constexpr int n = 1 << L;
    const u32 id = blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const u32 col = id >> rate_bits, r = id & ((1u << rate_bits) - 1);
    u64 x[n];
#pragma unroll
    for (int p = 0; p < n; p++) x[p] = mul_c(coeffs[(size_t)col * n + p], pre[(size_t)r * n + p]);
    dft_reg_dit<false, L>(x);
#pragma unroll
    for (int q = 0; q < n; q++) out[(size_t)id * n + q] = x[q];
}
template <int L>
__global__ __launch_bounds__(256) void k_intt_small(const u64 *__restrict__ in, u64 *__restrict__ out, u64 n_inv, u32 ncols) {
    constexpr int n = 1 << L;
    const u32 col = blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols) return;
    u64 x[n];
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = in[(size_t)col * n + i];
    dft_reg_dif<true, L>(x);
#pragma unroll
    for (int p = 0; p < n; p++) out[(size_t)col * n + p] = mul_c(x[p], n_inv);
}

// ------------------------------------------------------------------------------------------
// Transforms of 32..256 points (lg 5..8: the 2^7-row SMT trace, the middle FRI layers): ALL cosets of a column -- and several columns
// when that is still under 1024 elements -- in one LDS tile, every radix-2 stage over the whole tile.  k_lde_contig walks the cosets of
// a column one after the other (for 2^7 points: half the workgroup idle, nine barriers per coset) and needs a grid row per column:
// 256 SMT proofs are 34 560 wire columns, four launches and 0.72 ms; the output of a column group is one contiguous run.
// ------------------------------------------------------------------------------------------
template <int LGB>
__global__ __launch_bounds__(TPB) void k_lde_mid(const u64 *__restrict__ coeffs, u64 *__restrict__ out, const u64 *__restrict__ tw_B,
                                                 const u64 *__restrict__ pre, u32 ncols, int rate_bits, int cpb_lg) {
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    constexpr int B = 1 << LGB;
    const int tid = threadIdx.x, lgPC = LGB + rate_bits, E = 1 << (lgPC + cpb_lg);
    u64 *tile = smem, *tw = smem + E;
    const size_t col0 = (size_t)blockIdx.x << cpb_lg;
    for (int j = tid; j < (B >> 1); j += TPB) tw[j] = tw_B[j];
    for (int idx = tid; idx < E; idx += TPB) {
        const size_t col = col0 + (size_t)(idx >> lgPC);
        const int pl = idx & (B - 1), r = (idx >> LGB) & ((1 << rate_bits) - 1);
        tile[idx] = col < ncols ? mul(coeffs[col * B + pl], pre[(size_t)r * B + pl]) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < LGB; s++) {
        const int half = 1 << s;
        for (int bf = tid; bf < (E >> 1); bf += TPB) {
            const int j2 = bf & ((B >> 1) - 1), lo = j2 & (half - 1);
            const int j = ((bf >> (LGB - 1)) << LGB) | ((j2 >> s) << (s + 1)) | lo;
            const u64 u = tile[j];
            const u64 v = mul(tile[j + half], tw[lo << (LGB - 1 - s)]);
            tile[j] = add(u, v);
            tile[j + half] = sub(u, v);
        }
        __syncthreads();
    }
    const size_t live = (size_t)ncols << lgPC, base = col0 << lgPC;      // (col R + r) n + q of the group's first element
    for (int idx = tid; idx < E; idx += TPB)
        if (base + idx < live) out[base + idx] = tile[idx];
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static size_t contig_lds_bytes(int lgB) { return (((size_t)1 << lgB) + ((size_t)1 << lgB) / 2 + 1 + 64 + 64 + 16) * sizeof(u64); }

static int lde_coeffs_chunk(glp_ctx *c, const u64 *dev_coeffs, u64 *dev_lde, u32 ncols, int lg, int rate_bits, u64 shift);
static int lde_small(glp_ctx *c, const u64 *dev_coeffs, u64 *dev_lde, u32 ncols, int lg, int rate_bits, u64 shift) {
    if (ncols == 0) return GLP_OK;
    LdePlan *lp;
    GLP_TRY(get_lde_plan(c, lg, rate_bits, shift, &lp));
    const int R = 1 << rate_bits;
    if ((u64)ncols * R > 0x7FFFFFFFull) return set_error(GLP_ERR_UNSUPPORTED, "ncols*2^rate_bits=%llu too large", (unsigned long long)ncols * R);
    const u32 total = ncols * (u32)R;
    const dim3 g((total + 255) / 256), b(256);
    switch (lg) {
    case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lde_small<0>), g, b, 0, c->stream, dev_coeffs, dev_lde, lp->pre, total, rate_bits); break;
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lde_small<1>), g, b, 0, c->stream, dev_coeffs, dev_lde, lp->pre, total, rate_bits); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lde_small<2>), g, b, 0, c->stream, dev_coeffs, dev_lde, lp->pre, total, rate_bits); break;
    case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lde_small<3>), g, b, 0, c->stream, dev_coeffs, dev_lde, lp->pre, total, rate_bits); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lde_small<4>), g, b, 0, c->stream, dev_coeffs, dev_lde, lp->pre, total, rate_bits); break;
    }
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}
static int lde_mid(glp_ctx *c, const u64 *dev_coeffs, u64 *dev_lde, u32 ncols, int lg, int rate_bits, u64 shift) {
    if (ncols == 0) return GLP_OK;
    LdePlan *lp;
    GLP_TRY(get_lde_plan(c, lg, rate_bits, shift, &lp));
    const NttPlan *np = lp->ntt;
    const int cpb_lg = std::max(0, 10 - lg - rate_bits);                // columns per workgroup: up to 1024 elements in the tile
    const dim3 g((ncols + (1u << cpb_lg) - 1) >> cpb_lg), b(TPB);
    const size_t lds = (((size_t)1 << (lg + rate_bits + cpb_lg)) + ((size_t)1 << lg) / 2) * sizeof(u64);
    switch (lg) {
    case 5: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lde_mid<5>), g, b, lds, c->stream, dev_coeffs, dev_lde, np->tw_B, lp->pre, ncols, rate_bits, cpb_lg); break;
    case 6: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lde_mid<6>), g, b, lds, c->stream, dev_coeffs, dev_lde, np->tw_B, lp->pre, ncols, rate_bits, cpb_lg); break;
    case 7: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lde_mid<7>), g, b, lds, c->stream, dev_coeffs, dev_lde, np->tw_B, lp->pre, ncols, rate_bits, cpb_lg); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lde_mid<8>), g, b, lds, c->stream, dev_coeffs, dev_lde, np->tw_B, lp->pre, ncols, rate_bits, cpb_lg); break;
    }
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}
// grid.y carries (column, coset plane, outer block): at most 65535.  Wide inputs (the K-proof batches of glp_prove_batch:
// K * num_wires columns) go through in column chunks.
int lde_coeffs(glp_ctx *c, const u64 *dev_coeffs, u64 *dev_lde, u32 ncols, int lg, int rate_bits, u64 shift) {
    if (lg < 0 || lg > NTT_MAX_LG) return set_error(GLP_ERR_UNSUPPORTED, "log_n=%d outside the supported range 0..%d", lg, NTT_MAX_LG);
    if (rate_bits < 0 || rate_bits > 4) return set_error(GLP_ERR_UNSUPPORTED, "rate_bits=%d outside 0..4", rate_bits);
    if (lg <= 4) return lde_small(c, dev_coeffs, dev_lde, ncols, lg, rate_bits, shift);       // one thread per (column, coset)
    if (lg <= 8) return lde_mid(c, dev_coeffs, dev_lde, ncols, lg, rate_bits, shift);         // all cosets of a column in one tile
    const u32 per = 65535u >> (rate_bits + (lg > c->two_pass_lg ? lg - NTT_INNER_LG : 0));
    const size_t n = (size_t)1 << lg;
    for (u32 c0 = 0; c0 < ncols; c0 += per)
        GLP_TRY(lde_coeffs_chunk(c, dev_coeffs + (size_t)c0 * n, dev_lde + ((size_t)c0 * n << rate_bits), std::min(per, ncols - c0), lg, rate_bits, shift));
    return GLP_OK;
}
static int lde_coeffs_chunk(glp_ctx *c, const u64 *dev_coeffs, u64 *dev_lde, u32 ncols, int lg, int rate_bits, u64 shift) {
    if (ncols == 0) return GLP_OK;
    LdePlan *lp;
    GLP_TRY(get_lde_plan(c, lg, rate_bits, shift, &lp));
    const NttPlan *np = lp->ntt;
    const int R = 1 << rate_bits;
    if (ncols > 65535u || ((u64)ncols * R << np->lgAo) > 65535u)
        return set_error(GLP_ERR_UNSUPPORTED, "ncols*2^rate_bits*outer blocks=%llu exceeds grid.y", ((unsigned long long)ncols * R) << np->lgAo);
    if (np->lgAo > 0) {
        // three passes: per 2^20 block the two-pass coset transform with shift^A', the outer twiddle, then A' rows at stride 2^20
        const NttPlan *in = np->inner;
        const LdePlan *lin = lp->inner;
        const int lgAo = np->lgAo, lgM = NTT_INNER_LG;
        hipLaunchKernelGGL(k_lde_contig16, dim3(1u << in->lgA, ncols << lgAo), dim3(TPB), 0, c->stream, dev_coeffs, dev_lde, in->tw4096,
                           lin->pre, lin->s_r, in->w_n, lgM, in->lgA, R, lgAo);
        GLP_HIP(hipGetLastError());
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_strided16<false>), dim3((1u << in->lgB) / NTT_STRIDED_W, (ncols * R) << lgAo), dim3(TPB), 0,
                           c->stream, dev_lde, dev_lde, in->tw4096, lgM, in->lgB);
        GLP_HIP(hipGetLastError());
        launch_outer<false, true>(c, dev_lde, dev_lde, lp->t0, lp->t1, lgAo, lgM, R, ncols * R);
        GLP_HIP(hipGetLastError());
        return GLP_OK;
    }
    dim3 g1(1u << np->lgA, ncols);
    if (np->lgB == 12)
        hipLaunchKernelGGL(k_lde_contig16, g1, dim3(TPB), 0, c->stream, dev_coeffs, dev_lde, np->tw4096, lp->pre, lp->s_r,
                           np->w_n, lg, np->lgA, R, 0);
    else
        hipLaunchKernelGGL(k_lde_contig, g1, dim3(TPB), contig_lds_bytes(np->lgB), c->stream, dev_coeffs, dev_lde, np->tw_B,
                           lp->pre, lp->s_r, np->w_n, lg, np->lgA, np->lgB, R);
    GLP_HIP(hipGetLastError());
    if (np->lgA > 0) {
        dim3 g2((1u << np->lgB) / NTT_STRIDED_W, ncols * R);
        if (np->lgA == 8)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_strided16<false>), g2, dim3(TPB), 0, c->stream, dev_lde, dev_lde, np->tw4096, lg,
                               np->lgB);
        else if (np->lgA <= 4 && np->lgB >= 8)         // A <= 16 rows: register transform, no LDS
            launch_outer<false, false>(c, dev_lde, dev_lde, nullptr, nullptr, np->lgA, np->lgB, 1, ncols * R);
        else if (np->lgA >= 5 && np->lgA <= 7 && np->lgB == 12)
            launch_strided16e<false>(c, dev_lde, dev_lde, np->tw4096, lg, np->lgA, np->lgB, ncols * R);
        else if (np->lgA >= 9 && np->lgA <= 10 && np->lgB == 12)
            GLP_TRY(launch_strided32<false>(c, dev_lde, dev_lde, np->tw4096, lg, np->lgA, np->lgB, ncols * R));
        else if (np->lgA > NTT_LGA_MAX)
            return set_error(GLP_ERR_UNSUPPORTED, "internal: no strided kernel for 2^%d rows", np->lgA);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_strided<false>), g2, dim3(TPB), 0, c->stream, dev_lde, dev_lde, np->tw_A, lg,
                               np->lgA, np->lgB);
        GLP_HIP(hipGetLastError());
    }
    return GLP_OK;
}

int ntt_coeffs_to_values(glp_ctx *c, const u64 *dev_coeffs, u64 *dev_values, u32 ncols, int lg) {
    return lde_coeffs(c, dev_coeffs, dev_values, ncols, lg, 0, 1);
}

static int intt_chunk(glp_ctx *c, const u64 *dev_values, u64 *dev_coeffs, u32 ncols, int lg);
int intt_values_to_coeffs(glp_ctx *c, const u64 *dev_values, u64 *dev_coeffs, u32 ncols, int lg) {
    if (lg < 0 || lg > NTT_MAX_LG) return set_error(GLP_ERR_UNSUPPORTED, "log_n=%d outside the supported range 0..%d", lg, NTT_MAX_LG);
    if (lg <= 4) {                                       // one thread per column
        if (ncols == 0) return GLP_OK;
        NttPlan *np;
        GLP_TRY(get_ntt_plan(c, lg, &np));
        const dim3 g((ncols + 255) / 256), b(256);
        switch (lg) {
        case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intt_small<0>), g, b, 0, c->stream, dev_values, dev_coeffs, np->n_inv, ncols); break;
        case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intt_small<1>), g, b, 0, c->stream, dev_values, dev_coeffs, np->n_inv, ncols); break;
        case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intt_small<2>), g, b, 0, c->stream, dev_values, dev_coeffs, np->n_inv, ncols); break;
        case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intt_small<3>), g, b, 0, c->stream, dev_values, dev_coeffs, np->n_inv, ncols); break;
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intt_small<4>), g, b, 0, c->stream, dev_values, dev_coeffs, np->n_inv, ncols); break;
        }
        GLP_HIP(hipGetLastError());
        return GLP_OK;
    }
    const u32 per = 65535u >> (lg > c->two_pass_lg ? lg - NTT_INNER_LG : 0);
    const size_t n = (size_t)1 << lg;
    for (u32 c0 = 0; c0 < ncols; c0 += per)
        GLP_TRY(intt_chunk(c, dev_values + (size_t)c0 * n, dev_coeffs + (size_t)c0 * n, std::min(per, ncols - c0), lg));
    return GLP_OK;
}
static int intt_chunk(glp_ctx *c, const u64 *dev_values, u64 *dev_coeffs, u32 ncols, int lg) {
    if (ncols == 0) return GLP_OK;
    NttPlan *np;
    GLP_TRY(get_ntt_plan(c, lg, &np));
    if (ncols > 65535u || ((u64)ncols << np->lgAo) > 65535u) return set_error(GLP_ERR_UNSUPPORTED, "ncols=%u exceeds grid.y", ncols);
    if (np->lgAo > 0) {
        const NttPlan *in = np->inner;
        const int lgAo = np->lgAo, lgM = NTT_INNER_LG;
        launch_outer<true, true>(c, dev_values, dev_coeffs, np->it0, np->it1, lgAo, lgM, 1, ncols);
        GLP_HIP(hipGetLastError());
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_strided16<true>), dim3((1u << in->lgB) / NTT_STRIDED_W, ncols << lgAo), dim3(TPB), 0, c->stream,
                           dev_coeffs, dev_coeffs, in->itw4096, lgM, in->lgB);
        GLP_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_intt_contig16, dim3(1u << in->lgA, ncols << lgAo), dim3(TPB), 0, c->stream, dev_coeffs, dev_coeffs, in->itw4096,
                           in->w_n_inv, in->n_inv, lgM, in->lgA);
        GLP_HIP(hipGetLastError());
        return GLP_OK;
    }
    const u64 *src = dev_values;
    if (np->lgA > 0) {
        dim3 g1((1u << np->lgB) / NTT_STRIDED_W, ncols);
        if (np->lgA == 8)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_strided16<true>), g1, dim3(TPB), 0, c->stream, dev_values, dev_coeffs,
                               np->itw4096, lg, np->lgB);
        else if (np->lgA <= 4 && np->lgB >= 8)
            launch_outer<true, false>(c, dev_values, dev_coeffs, nullptr, nullptr, np->lgA, np->lgB, 1, ncols);
        else if (np->lgA >= 5 && np->lgA <= 7 && np->lgB == 12)
            launch_strided16e<true>(c, dev_values, dev_coeffs, np->itw4096, lg, np->lgA, np->lgB, ncols);
        else if (np->lgA >= 9 && np->lgA <= 10 && np->lgB == 12)
            GLP_TRY(launch_strided32<true>(c, dev_values, dev_coeffs, np->itw4096, lg, np->lgA, np->lgB, ncols));
        else if (np->lgA > NTT_LGA_MAX)
            return set_error(GLP_ERR_UNSUPPORTED, "internal: no strided kernel for 2^%d rows", np->lgA);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_strided<true>), g1, dim3(TPB), 0, c->stream, dev_values, dev_coeffs, np->itw_A,
                               lg, np->lgA, np->lgB);
        GLP_HIP(hipGetLastError());
        src = dev_coeffs;
    }
    dim3 g2(1u << np->lgA, ncols);
    if (np->lgB == 12)
        hipLaunchKernelGGL(k_intt_contig16, g2, dim3(TPB), 0, c->stream, src, dev_coeffs, np->itw4096, np->w_n_inv, np->n_inv, lg,
                           np->lgA);
    else
        hipLaunchKernelGGL(k_intt_contig, g2, dim3(TPB), contig_lds_bytes(np->lgB), c->stream, src, dev_coeffs, np->itw_B,
                           np->w_n_inv, np->n_inv, lg, np->lgA, np->lgB);
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}

int bitrev_copy(glp_ctx *c, const u64 *dev_in, u64 *dev_out, u32 ncols, int lg) {
    if (ncols == 0) return GLP_OK;
    const size_t n = (size_t)1 << lg;
    dim3 g((unsigned)((n + 255) / 256), ncols);
    hipLaunchKernelGGL(k_bitrev_copy, g, dim3(256), 0, c->stream, dev_in, dev_out, lg);
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}

int lde_to_natural(glp_ctx *c, const u64 *dev_lde, u64 *dev_out, u32 ncols, int lg, int rate_bits) {
    if (ncols == 0) return GLP_OK;
    const size_t N = (size_t)1 << (lg + rate_bits);
    dim3 g((unsigned)((N + 255) / 256), ncols);
    hipLaunchKernelGGL(k_lde_to_natural, g, dim3(256), 0, c->stream, dev_lde, dev_out, lg, rate_bits);
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}

}  // namespace glp
