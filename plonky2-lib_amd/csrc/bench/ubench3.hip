// ubench3.hip -- A/B of Goldilocks reduction formulations on gfx950 (one process, same data).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../glf.h"
using namespace glf;
constexpr int ITERS = 2048;

// V1: 64-bit form: t0 = lo - hh (borrow fix), r = hl*EPS + t0 via one multiply-add, carry fix by compare
__device__ __forceinline__ u64 fold128_v1(u64 lo, u64 hi) {
    const u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= EPS;
    u64 r = (u64)hl * (u32)EPS + t0;
    if (r < t0) r += EPS;
    return r;
}
__device__ __forceinline__ u64 mul_v1(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p00 = (u64)a0 * b0;
    const u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    const u64 p10 = (u64)a1 * b0 + (u32)p01;
    const u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    return fold128_v1(((u64)(u32)p10 << 32) | (u32)p00, p11);
}
// V2: fold to 96 bits with plain 64-bit arithmetic first: x = lo + hi*EPS = lo + (hi<<32) - hi
__device__ __forceinline__ u64 fold96_v2(u64 l, u32 h) {
    u64 r = (u64)h * (u32)EPS + l;
    if (r < l) r += EPS;
    return r;
}
__device__ __forceinline__ u64 mul_v2(u64 a, u64 b) {
    unsigned __int128 p = (unsigned __int128)a * b;
    const u64 lo = (u64)p, hi = (u64)(p >> 64);
    // y = lo + hi*(2^32 - 1) as a 97-bit number
    unsigned __int128 y = (unsigned __int128)lo + ((unsigned __int128)hi << 32) - hi;
    return fold96_v2((u64)y, (u32)(y >> 64));
}
__device__ __forceinline__ u64 mul_v3(u64 a, u64 b) {
    unsigned __int128 p = (unsigned __int128)a * b;
    const u64 lo = (u64)p, hi = (u64)(p >> 64);
    unsigned __int128 y = (unsigned __int128)lo + (unsigned __int128)hi * (u32)EPS;
    return fold96_v2((u64)y, (u32)(y >> 64));
}
// v4: like v2 but the 97-bit intermediate built from 32-bit pieces: y = lo + (hi << 32) - hi
__device__ __forceinline__ u64 mul_v4(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p00 = (u64)a0 * b0;
    const u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    const u64 p10 = (u64)a1 * b0 + (u32)p01;
    const u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);     // hi
    const u64 lo = ((u64)(u32)p10 << 32) | (u32)p00;
    // lo - hi (may borrow: then the true value is negative by < 2^64; add p instead: lo - hi + 2^64 - 2^32 + 1 handled below)
    const u64 d = lo - p11;
    const u32 borrow = lo < p11;
    // y = d + (hi << 32)  (97 bits) minus borrow * 2^64
    const u64 l = d + (p11 << 32);
    const u32 c = l < d;
    const u32 h = (u32)(p11 >> 32) + c - borrow;      // in [-1 .. 2^32]: borrow and the top part
    // h can be -1 only if p11>>32 == 0 && c == 0 && borrow == 1: then value = l - 2^64 = l - EPS (mod p)
    u64 r = (u64)h * (u32)EPS + l;                    // for h = 0xFFFFFFFF this is wrong; handled:
    if ((p11 >> 32) == 0 && c == 0 && borrow) return l - EPS - (l < EPS ? EPS : 0) + 0 * r;
    if (r < l) r += EPS;
    return r;
}
__device__ __forceinline__ u64 sqr_v2(u64 a, u64) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32);
    const u64 p00 = (u64)a0 * a0, p01 = (u64)a0 * a1, p11 = (u64)a1 * a1;
    unsigned __int128 p = (unsigned __int128)p00 + ((unsigned __int128)p01 << 33) + ((unsigned __int128)p11 << 64);
    const u64 lo = (u64)p, hi = (u64)(p >> 64);
    unsigned __int128 y = (unsigned __int128)lo + ((unsigned __int128)hi << 32) - hi;
    return fold96_v2((u64)y, (u32)(y >> 64));
}
// V5: fold96 with the multiply-add's own carry-out (inline asm: hipcc has no builtin for the carry of v_mad_u64_u32)
__device__ __forceinline__ u64 fold96_asm(u64 l, u32 h) {
    u64 r; u32 m;
    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, -1, vcc"
                 : "=&v"(r), "=v"(m) : "v"(h), "s"(0xFFFFFFFFu), "v"(l) : "vcc");
    return r + (u64)m;
}
__device__ __forceinline__ u64 mul_v5(u64 a, u64 b) {
    const unsigned __int128 p = (unsigned __int128)a * b;
    const u64 lo = (u64)p, hi = (u64)(p >> 64);
    const unsigned __int128 y = (unsigned __int128)lo + ((unsigned __int128)hi << 32) - hi;
    return fold96_asm((u64)y, (u32)(y >> 64));
}
// V6: the 97-bit y = lo + hi (2^32 - 1) from two more multiply-adds (carry-out of the first fed to the top limb)
// instead of a 128-bit shift/sub/add carry chain
__device__ __forceinline__ u64 mul_v6(u64 a, u64 b) {
    const unsigned __int128 p = (unsigned __int128)a * b;
    const u64 lo = (u64)p, hi = (u64)(p >> 64);
    const u32 h0 = (u32)hi, h1 = (u32)(hi >> 32);
    u64 t, cc;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(t), "=s"(cc) : "v"(h0), "v"(lo));
    const u64 u = (u64)h1 * 0xFFFFFFFFu + (t >> 32);
    u32 h; const u32 ulo = (u32)u, uhi = (u32)(u >> 32);
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, 0, %1" : "=v"(h), "+s"(cc) : "v"(uhi));
    return fold96_asm(((u64)ulo << 32) | (u32)t, h);
}
// V7: the middle sum a1 b0 + a0 b1 + hi(a0 b0) as one chain whose 65th bit is the second multiply-add's carry-out; the top
// product takes {m2.hi, carry} as its addend (one move + one select instead of three moves + a 64-bit add)
__device__ __forceinline__ u64 mul_v7(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 e = (u64)a0 * b0;
    const u64 m = (u64)a0 * b1 + (e >> 32);
    u64 m2, cc;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m2), "=s"(cc) : "v"(a1), "v"(b0), "v"(m));
    u32 chi;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(chi) : "s"(cc));
    const u64 hi = (u64)a1 * b1 + (((u64)chi << 32) | (m2 >> 32));
    const u64 lo = ((u64)(u32)m2 << 32) | (u32)e;
    const u32 h0 = (u32)hi, h1 = (u32)(hi >> 32);
    u64 t, c2;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(t), "=s"(c2) : "v"(h0), "v"(lo));
    const u64 u = (u64)h1 * 0xFFFFFFFFu + (t >> 32);
    u32 h; const u32 ulo = (u32)u, uhi = (u32)(u >> 32);
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, 0, %1" : "=v"(h), "+s"(c2) : "v"(uhi));
    return fold96_asm(((u64)ulo << 32) | (u32)t, h);
}
// V8: V7 with the carry of the middle chain folded AFTER the top product instead of inside its addend:
// 2^96 * carry = -carry (mod p): subtract it from the low word of y (no select, no pair building for it)
__device__ __forceinline__ u64 mul_v8(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 e = (u64)a0 * b0;
    const u64 m = (u64)a0 * b1 + (e >> 32);
    u64 m2, cc;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m2), "=s"(cc) : "v"(a1), "v"(b0), "v"(m));
    const u64 hi = (u64)a1 * b1 + (m2 >> 32);           // without the carry (worth 2^96 = -1)
    const u64 lo = ((u64)(u32)m2 << 32) | (u32)e;
    const u32 h0 = (u32)hi, h1 = (u32)(hi >> 32);
    u64 t, c2;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(t), "=s"(c2) : "v"(h0), "v"(lo));
    const u64 u = (u64)h1 * 0xFFFFFFFFu + (t >> 32);
    u32 h; const u32 ulo = (u32)u, uhi = (u32)(u >> 32);
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, 0, %1" : "=v"(h), "+s"(c2) : "v"(uhi));
    u64 r = fold96_asm(((u64)ulo << 32) | (u32)t, h);
    // r -= carry (mod p): r = r - c; on borrow add p back (r - c + 2^64 - EPS ... handled as: if r < c then r + p - c)
    u32 cb;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(cb) : "s"(cc));
    const u64 r2 = r - cb;
    return r2 > r ? r2 - EPS : r2;          // wrapped: subtract 2^64 mod p
}
#define MULK(NAME, F)                                                                 \
    __global__ void NAME(u64 *out, u64 a, u64 b) {                                    \
        u64 acc[4];                                                                   \
        for (int i = 0; i < 4; i++) acc[i] = canon(a + threadIdx.x + i);              \
        for (int it = 0; it < ITERS; it++) {                                          \
            _Pragma("unroll") for (int i = 0; i < 4; i++) acc[i] = F(acc[i], b ^ acc[(i + 1) & 3]); \
        }                                                                             \
        u64 s = 0; for (int i = 0; i < 4; i++) s ^= canon(acc[i]);                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                               \
    }
MULK(k_mul_v0, mul_nc)
MULK(k_mul_v1, mul_v1)
MULK(k_mul_v2, mul_v2)
MULK(k_mul_v3, mul_v3)
MULK(k_sqr_v2, sqr_v2)
MULK(k_mul_v5, mul_v5)
MULK(k_mul_v6, mul_v6)
MULK(k_mul_v7, mul_v7)
MULK(k_mul_v8, mul_v8)

// MDS row variants: 12 rows of 24 multiply-adds + fold
template <int V> __device__ __forceinline__ void mds_v(u64 s[12], const u64 *rc) {
    constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u64 al = (u32)rc[r], ah = rc[r] >> 32;
#pragma unroll
        for (int i = 0; i < 12; i++) { al += (u64)lo[(i + r) % 12] * C[i]; ah += (u64)hi[(i + r) % 12] * C[i]; }
        if (r == 0) { al += (u64)lo[0] * 8; ah += (u64)hi[0] * 8; }
        if (V == 2 || V == 3 || V == 4) {
            // chained: the high-half chain starts from the low half's overflow, so the 96-bit value needs no carry combine
            u64 al2 = (u32)rc[r];
#pragma unroll
            for (int i = 0; i < 12; i++) al2 += (u64)lo[(i + r) % 12] * C[i];
            if (r == 0) al2 += (u64)lo[0] * 8;
            u64 ah2 = (al2 >> 32) + (rc[r] >> 32);
#pragma unroll
            for (int i = 0; i < 12; i++) ah2 += (u64)hi[(i + r) % 12] * C[i];
            if (r == 0) ah2 += (u64)hi[0] * 8;
            const u64 l = (ah2 << 32) | (u32)al2;
            const u32 h = (u32)(ah2 >> 32);
            s[r] = V == 2 ? fold96_v2(l, h) : (V == 3 ? fold96_nc(l, h) : fold96_asm(l, h));
        } else if (V == 0) {
            u32 k;
            const u32 x1 = __builtin_addc((u32)(al >> 32), (u32)ah, 0u, &k);
            const u32 h = (u32)(ah >> 32) + k;
            s[r] = fold96_nc(((u64)x1 << 32) | (u32)al, h);
        } else {
            // value = al + ah 2^32: 64-bit add with the carry into h, then one multiply-add fold
            const u64 l = al + (ah << 32);
            const u32 h = (u32)(ah >> 32) + (l < al ? 1u : 0u);
            s[r] = fold96_v2(l, h);
        }
    }
}
template <int V> __global__ void k_mds(u64 *out, u64 a, const u64 *rc) {
    u64 s[12];
    for (int i = 0; i < 12; i++) s[i] = canon(a + threadIdx.x * 12 + i);
    for (int it = 0; it < 256; it++) mds_v<V>(s, rc);
    u64 x = 0; for (int i = 0; i < 12; i++) x ^= s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <class F> static float time_ms(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; }
    return best;
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount, blocks = cus * 8, threads = 256;
    u64 *out, *rc; hipMalloc(&out, (size_t)blocks * threads * 8); hipMalloc(&rc, 96); hipMemset(rc, 1, 96);
    double lanes = (double)blocks * threads;
#define RUNM(K) { float ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, out, (u64)12345, (u64)0xfedcba9876543210ull); }); \
                  u64 h4[4]; hipMemcpy(h4, out + 1000, 32, hipMemcpyDeviceToHost); \
                  printf("%-10s %7.3f ms  %7.1f lane-clk per mulmod   check %016llx\n", #K, ms, ms * 1e-3 * 2.4e9 * cus * 128 / (lanes * 4.0 * ITERS), (unsigned long long)(h4[0] ^ h4[1] * 3 ^ h4[2] * 5 ^ h4[3] * 7)); }
    RUNM(k_mul_v0) RUNM(k_mul_v1) RUNM(k_mul_v2) RUNM(k_mul_v3) RUNM(k_mul_v5) RUNM(k_mul_v6) RUNM(k_mul_v7) RUNM(k_mul_v8)
    { float ms = time_ms([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mds<0>), dim3(blocks), dim3(threads), 0, 0, out, (u64)5, rc); });
      printf("k_mds<0>   %7.3f ms  %7.1f lane-clk per MDS layer\n", ms, ms * 1e-3 * 2.4e9 * cus * 128 / (lanes * 256.0)); }
    { float ms = time_ms([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mds<1>), dim3(blocks), dim3(threads), 0, 0, out, (u64)5, rc); });
      printf("k_mds<1>   %7.3f ms  %7.1f lane-clk per MDS layer\n", ms, ms * 1e-3 * 2.4e9 * cus * 128 / (lanes * 256.0)); }
    { float ms = time_ms([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mds<2>), dim3(blocks), dim3(threads), 0, 0, out, (u64)5, rc); });
      printf("k_mds<2>   %7.3f ms  %7.1f lane-clk per MDS layer\n", ms, ms * 1e-3 * 2.4e9 * cus * 128 / (lanes * 256.0)); }
    { float ms = time_ms([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mds<3>), dim3(blocks), dim3(threads), 0, 0, out, (u64)5, rc); });
      printf("k_mds<3>   %7.3f ms  %7.1f lane-clk per MDS layer\n", ms, ms * 1e-3 * 2.4e9 * cus * 128 / (lanes * 256.0)); }
    { float ms = time_ms([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mds<4>), dim3(blocks), dim3(threads), 0, 0, out, (u64)5, rc); });
      printf("k_mds<4>   %7.3f ms  %7.1f lane-clk per MDS layer\n", ms, ms * 1e-3 * 2.4e9 * cus * 128 / (lanes * 256.0)); }
    return 0;
}
