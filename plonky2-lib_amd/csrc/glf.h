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
// any u64 inputs -> canonical
GLF_HD u64 mul(u64 a, u64 b) {
    u64 lo, hi;
    mul128(a, b, lo, hi);
    return reduce128(lo, hi);
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
