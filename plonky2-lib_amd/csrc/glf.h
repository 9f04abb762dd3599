// glf.h -- Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic extension F[X]/(X^2-7) for
// gfx950 device code and the host-side transcript.  All values that cross a kernel boundary are
// canonical (< p); inside a function a value may be any u64 congruent to the element where noted.
//
// Replaces, on the prove() path of the plonky2 fork that /root/reference calls
// [REF src/ecdsa/gadgets/ecdsa.rs:349], plonky2_field's GoldilocksField / QuadraticExtension
// (crate absent from /root/reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;

#define GLF_HD __host__ __device__ __forceinline__

namespace glf {

constexpr u64 P = 0xFFFFFFFF00000001ull;
constexpr u64 EPS = 0xFFFFFFFFull;      // 2^64 mod p
constexpr u64 GEN = 7;                  // multiplicative generator == coset shift
constexpr u64 POW2_GEN = 1753635133440165772ull;  // order 2^32
constexpr u64 W = 7;                    // X^2 = W in the quadratic extension

GLF_HD u64 canon(u64 a) { return a >= P ? a - P : a; }

// a, b canonical -> canonical
GLF_HD u64 add(u64 a, u64 b) {
    u64 s = a + b;
    return (s < a || s >= P) ? s - P : s;
}
GLF_HD u64 sub(u64 a, u64 b) { return a >= b ? a - b : a + (P - b); }
GLF_HD u64 neg(u64 a) { return a ? P - a : 0; }
GLF_HD u64 dbl(u64 a) { return add(a, a); }

// x = lo + 2^64 * hi  (any 128-bit value) -> canonical.   2^64 = 2^32 - 1,  2^96 = -1 (mod p)
GLF_HD u64 reduce128(u64 lo, u64 hi) {
    u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= EPS;                       // borrow: subtract 2^64 mod p
    u64 t1 = ((u64)hl << 32) - hl;                // hl * (2^32 - 1), fits 64 bits
    u64 t2 = t0 + t1;
    if (t2 < t0) t2 += EPS;
    return canon(t2);
}
// x = lo + 2^64 * hi with hi < 2^32 -> canonical
GLF_HD u64 reduce96(u64 lo, u32 hi) {
    u64 t1 = ((u64)hi << 32) - hi;
    u64 t2 = lo + t1;
    if (t2 < lo) t2 += EPS;
    return canon(t2);
}

GLF_HD void mul128(u64 a, u64 b, u64& lo, u64& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    lo = a * b;
    hi = __umul64hi(a, b);
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    lo = (u64)p; hi = (u64)(p >> 64);
#endif
}
__device__ __forceinline__ u64 mul_nc(u64 a, u64 b);
// any u64 inputs -> canonical
GLF_HD u64 mul(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return canon(mul_nc(a, b));      // 32-bit limb form: fewer issue slots on gfx950 than 64-bit compare + select
#else
    u64 lo, hi;
    mul128(a, b, lo, hi);
    return reduce128(lo, hi);
#endif
}
GLF_HD u64 sqr(u64 a) { return mul(a, a); }

GLF_HD u64 pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = mul(r, b); b = sqr(b); e >>= 1; }
    return r;
}
GLF_HD u64 inv(u64 a) { return pow(a, P - 2); }
GLF_HD u64 root_of_unity(int n_log) {      // plonky2 `primitive_root_of_unity`
    u64 r = POW2_GEN;
    for (int i = n_log; i < 32; i++) r = sqr(r);
    return r;
}
// x * 2^k mod p for 0 <= k < 96 (x canonical or not) -> canonical.  Used for the 64th roots of unity,
// all of which are +-2^(3j).
GLF_HD u64 mul_2exp(u64 x, u32 k) {
    if (k == 0) return canon(x);
    if (k < 32) { u64 lo = x << k; u32 hi = (u32)(x >> (64 - k)); return reduce96(lo, hi); }
    if (k == 32) { return reduce96(x << 32, (u32)(x >> 32)); }
    if (k < 64) { u64 lo = x << k; u64 hi = x >> (64 - k); return reduce128(lo, hi); }
    if (k == 64) return reduce128(0, x);
    // 64 < k < 96: x*2^k = (x << (k-64)) * 2^64, x<<(k-64) is up to 96 bits: split
    u32 s = k - 64;
    u64 mid = x << s;            // bits 64.. of the product (low 64 of x<<s)
    u64 top = x >> (64 - s);     // bits 128..  -> 2^128 = 2^64*2^64 = (2^32-1)^2 ... reduce stepwise
    // value = mid*2^64 + top*2^128.  2^128 = 2^32 * 2^96 = -2^32
    u64 a = reduce128(0, mid);
    u64 b = reduce96(top << 32, (u32)(top >> 32));   // top * 2^32
    return sub(a, b);
}

// ---- quadratic extension ------------------------------------------------------------------
struct ext2 { u64 a, b; };      // a + b*X
GLF_HD ext2 e_make(u64 a, u64 b) { ext2 r; r.a = a; r.b = b; return r; }
GLF_HD ext2 e_from(u64 a) { return e_make(a, 0); }
GLF_HD ext2 e_add(ext2 x, ext2 y) { return e_make(add(x.a, y.a), add(x.b, y.b)); }
GLF_HD ext2 e_sub(ext2 x, ext2 y) { return e_make(sub(x.a, y.a), sub(x.b, y.b)); }
GLF_HD ext2 e_neg(ext2 x) { return e_make(neg(x.a), neg(x.b)); }
GLF_HD ext2 e_mul(ext2 x, ext2 y) {
    u64 bb = mul(x.b, y.b);
    u64 c0 = add(mul(x.a, y.a), mul(W, bb));
    u64 c1 = add(mul(x.a, y.b), mul(x.b, y.a));
    return e_make(c0, c1);
}
GLF_HD ext2 e_scale(ext2 x, u64 s) { return e_make(mul(x.a, s), mul(x.b, s)); }
GLF_HD ext2 e_sqr(ext2 x) { return e_mul(x, x); }
GLF_HD bool e_eq(ext2 x, ext2 y) { return x.a == y.a && x.b == y.b; }
GLF_HD ext2 e_inv(ext2 x) {
    u64 n = sub(sqr(x.a), mul(W, sqr(x.b)));
    u64 ni = inv(n);
    return e_make(mul(x.a, ni), mul(neg(x.b), ni));
}
GLF_HD ext2 e_pow(ext2 b, u64 e) {
    ext2 r = e_from(1);
    while (e) { if (e & 1) r = e_mul(r, b); b = e_sqr(b); e >>= 1; }
    return r;
}

// ---- gfx950 limb forms (device only) -------------------------------------------------------------
// l + 2^64 h -> some u64 congruent to it (NOT canonical).  32-bit limbs + carry builtins: on gfx950 a
// 64-bit compare + select costs more issue slots than a v_add_co/v_addc chain.
__device__ __forceinline__ u64 fold128_nc(u32 l0, u32 l1, u32 h0, u32 h1) {
    u32 c, c2, k, k2;
    u32 t0lo = __builtin_subc(l0, h1, 0u, &c);          // l - h1          (2^96 = -1)
    u32 t0hi = __builtin_subc(l1, 0u, c, &c);
    const u32 m = 0u - c;                               // borrow: subtract 2^64 mod p = 2^32 - 1
    t0lo = __builtin_subc(t0lo, m, 0u, &c2);
    t0hi = __builtin_subc(t0hi, 0u, c2, &c2);
    const u32 t1lo = 0u - h0, t1hi = h0 - (h0 != 0);    // h0 * (2^32 - 1)
    u32 rlo = __builtin_addc(t0lo, t1lo, 0u, &k);
    u32 rhi = __builtin_addc(t0hi, t1hi, k, &k);
    const u32 m2 = 0u - k;                              // carry: add 2^32 - 1
    rlo = __builtin_addc(rlo, m2, 0u, &k2);
    rhi = __builtin_addc(rhi, 0u, k2, &k2);
    return ((u64)rhi << 32) | rlo;
}
// l + 2^64 h, h < 2^32: one multiply-add by 2^32 - 1 and a carry fix-up (measured faster on gfx950 than the
// 32-bit carry chain: profiles/r01_ubench_variants.txt).  The carry is the v_mad_u64_u32's own carry-out, which
// hipcc has no builtin for (from C it re-derives it with a 64-bit compare and two selects: 6 VALU slots against
// 3 here); the s_nop covers the VCC write -> v_cndmask read hazard the compiler cannot see inside an asm block.
__device__ __forceinline__ u64 fold96_c(u64 l, u32 h) {   // plain-C form (lower register pressure under hipcc's scheduler)
    u64 r = (u64)h * (u32)EPS + l;
    if (r < l) r += EPS;
    return r;
}
__device__ __forceinline__ u64 fold96_nc(u64 l, u32 h) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 r, cc; u32 m;
    asm("v_mad_u64_u32 %0, %2, %3, -1, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, -1, %2"
        : "=v"(r), "=v"(m), "=&s"(cc) : "v"(h), "v"(l));
    return r + (u64)m;       // carry: add 2^64 mod p = 2^32 - 1 (cannot carry again: r < h (2^32 - 1) then)
#else
    u64 r = (u64)h * (u32)EPS + l;
    if (r < l) r += EPS;
    return r;
#endif
}
// any u64 inputs -> non-canonical product.  Two-level fold: y = lo + hi (2^32 - 1) as a 97-bit integer, then
// fold96.  y itself comes from two more multiply-adds by 2^32 - 1 (hi = h0 + 2^32 h1):
//     t + 2^64 c = h0 (2^32-1) + lo,    u = h1 (2^32-1) + (t >> 32)  (< 2^64),    y = t_lo + 2^32 u_lo + 2^64 (u_hi + c)
// which costs 3 half-rate VALU slots + 1 move against the 6 carry-chain slots of a 128-bit shift/sub/add
// (profiles/r01_ubench_variants.txt: 35.9 -> 32.5 lane-clk).  The carry c travels in an SGPR pair; the s_nop
// covers the VALU-writes-SGPR -> VALU-reads-SGPR hazard the compiler cannot see across asm blocks.
// lo + 2^64 hi (any 128-bit value) -> non-canonical u64, via the 97-bit y of the comment above
__device__ __forceinline__ u64 fold128_mad_nc(u64 lo, u64 hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 h0 = (u32)hi, h1 = (u32)(hi >> 32);
    u64 t, cc;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(t), "=s"(cc) : "v"(h0), "v"(lo));
    const u64 u = (u64)h1 * 0xFFFFFFFFu + (t >> 32);
    u32 h;
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, 0, %1" : "=v"(h), "+s"(cc) : "v"((u32)(u >> 32)));
    return fold96_nc((u << 32) | (u32)t, h);
#else
    const unsigned __int128 y = (unsigned __int128)lo + ((unsigned __int128)hi << 32) - hi;
    return fold96_nc((u64)y, (u32)(y >> 64));
#endif
}
__device__ __forceinline__ u64 mul_nc(u64 a, u64 b) {
    const unsigned __int128 p = (unsigned __int128)a * b;
    return fold128_mad_nc((u64)p, (u64)(p >> 64));
}
// The same product with the middle sum a1 b0 + a0 b1 + hi(a0 b0) as ONE multiply-add chain: its 65th bit is the second
// multiply-add's carry-out and enters the top product's addend as {m2.hi, carry}.  From C the compiler cuts that sum into 32-bit
// pieces and rebuilds even-aligned register pairs with moves (7 v_mov_b32 + 2 v_lshl_add_u64 per product against 5 + 1 here;
// profiles/r02_rejected_experiments.txt): 4.40 -> 4.11 ms in the multiplier microbenchmark, 27.4 k -> 26.5 k lane-clk per Poseidon
// permutation.  Used by the S-boxes; the NTT kernels keep mul_nc (the extra asm blocks cost k_lde_contig16 55 VGPRs and two
// waves of occupancy).
__device__ __forceinline__ u64 mul_nc_cc(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 e = (u64)a0 * b0;
    const u64 mm = (u64)a0 * b1 + (e >> 32);                // < 2^64: (2^32 - 1)^2 + 2^32 - 1
    u64 m2, mc;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m2), "=s"(mc) : "v"(a1), "v"(b0), "v"(mm));
    u32 chi;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(chi) : "s"(mc));
    const u64 hi = (u64)a1 * b1 + (((u64)chi << 32) | (m2 >> 32));     // < 2^64: the whole product is < 2^128
    return fold128_mad_nc(((u64)(u32)m2 << 32) | (u32)e, hi);
#else
    return mul_nc(a, b);
#endif
}
// a canonical (< p), b ANY u64 -> a + b as a non-canonical u64.  A wrapped sum is < p - 1, so adding 2^64 mod p back
// cannot wrap again.
__device__ __forceinline__ u64 add_cnc(u64 a, u64 b) {
    u32 c, c2;
    u32 lo = __builtin_addc((u32)a, (u32)b, 0u, &c);
    u32 hi = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c, &c);
    const u32 m = 0u - c;
    lo = __builtin_addc(lo, m, 0u, &c2);
    hi = __builtin_addc(hi, 0u, c2, &c2);
    return ((u64)hi << 32) | lo;
}
// x (any u64) times a 32-bit constant -> non-canonical: two multiply-adds and one fold
__device__ __forceinline__ u64 mul_small_nc(u64 x, u32 g) {
    const u64 lo = (u64)(u32)x * g;
    const u64 hi = (u64)(u32)(x >> 32) * g + (lo >> 32);
    return fold96_nc((hi << 32) | (u32)lo, (u32)(hi >> 32));
}
// canonical product through the limb form
__device__ __forceinline__ u64 mul_c(u64 a, u64 b) { return canon(mul_nc(a, b)); }
// x * 2^E mod p for a compile-time 0 < E < 96 (x any u64) -> canonical.  Every 64th root of unity of
// the field is +-2^(3j), so the small-radix butterflies of the NTT multiply by shifting.
template <int E>
__device__ __forceinline__ u64 mul_pow2_c(u64 x) {
    static_assert(E > 0 && E < 96, "shift out of range");
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    if constexpr (E < 32) {
        const u32 y0 = x0 << E, y1 = (x1 << E) | (x0 >> (32 - E)), y2 = x1 >> (32 - E);
        return canon(fold96_nc(((u64)y1 << 32) | y0, y2));
    } else if constexpr (E == 32) {
        return canon(fold128_nc(0u, x0, x1, 0u));
    } else if constexpr (E < 64) {
        constexpr int S = E - 32;
        const u32 y1 = x0 << S, y2 = (x1 << S) | (x0 >> (32 - S)), y3 = x1 >> (32 - S);
        return canon(fold128_nc(0u, y1, y2, y3));
    } else if constexpr (E == 64) {
        return canon(fold128_nc(0u, 0u, x0, x1));
    } else {
        // x 2^E = y' 2^64 with y' = x << (E-64) (96 bits):  y'0 (2^32-1) - y'1 - y'2 2^32;  -y'2 2^32 = (~y'2) 2^32 + 1 (mod p)
        constexpr int S = E - 64;
        const u32 y0 = x0 << S, y1 = (x1 << S) | (x0 >> (32 - S)), y2 = x1 >> (32 - S);
        return canon(fold128_nc(1u, ~y2, y0, y1));
    }
}

GLF_HD u32 bitrev32(u32 x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    u32 r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
#endif
}

}  // namespace glf
