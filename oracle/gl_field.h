/*
 * oracle/gl_field.h -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of the Goldilocks field and its quadratic extension as used by the plonky2
 * prover that /root/reference drives through `data.prove(pw)` [REF src/ecdsa/gadgets/ecdsa.rs:349].
 * The plonky2 crate itself is absent from /root/reference; this follows the
 * published plonky2 0.1.4 algorithm (`field/src/goldilocks_field.rs`, `goldilocks_extensions.rs`,
 * `extension/quadratic.rs`).  Pinned by: the field identities in tests/test_oracle_field.py and,
 * end to end, by the reference's Poseidon known-answer vector [REF src/zkdsa/circuits/mod.rs:85-101].
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
 */
#ifndef GL_FIELD_H
#define GL_FIELD_H
#include <stdint.h>
#include <stddef.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
/* MULTIPLICATIVE_GROUP_GENERATOR = 7 = coset shift; POWER_OF_TWO_GENERATOR has order 2^32. */
#define GL_GEN 7ULL
#define GL_POW2_GEN 1753635133440165772ULL
#define GL_TWO_ADICITY 32
/* extension F[X]/(X^2 - 7) */
#define GL_W 7ULL

static inline u64 gl_canon(u64 a) { return a >= GL_P ? a - GL_P : a; }

static inline u64 gl_add(u64 a, u64 b) { /* a, b canonical */
    u64 s = a + b;
    if (s < a || s >= GL_P) s -= GL_P;
    return s;
}
static inline u64 gl_sub(u64 a, u64 b) { return a >= b ? a - b : a + (GL_P - b); }
static inline u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }

/* plonky2 `reduce128`: x = lo + 2^64*hi, 2^64 = 2^32 - 1, 2^96 = -1 (mod p). Returns canonical. */
static inline u64 gl_reduce128(u128 x) {
    u64 lo = (u64)x, hi = (u64)(x >> 64);
    u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    u64 t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= GL_EPS;
    u64 t1 = hi_lo * GL_EPS;
    u64 t2 = t0 + t1;
    if (t2 < t0) t2 += GL_EPS;
    return gl_canon(t2);
}
static inline u64 gl_mul(u64 a, u64 b) { return gl_reduce128((u128)a * b); }
static inline u64 gl_sqr(u64 a) { return gl_mul(a, a); }

static inline u64 gl_pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = gl_mul(r, b); b = gl_sqr(b); e >>= 1; }
    return r;
}
static inline u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }
/* primitive_root_of_unity(n_log) = POWER_OF_TWO_GENERATOR ^ (2^(32 - n_log)) */
static inline u64 gl_root_of_unity(int n_log) {
    u64 r = GL_POW2_GEN;
    for (int i = n_log; i < GL_TWO_ADICITY; i++) r = gl_sqr(r);
    return r;
}

/* ---- quadratic extension, element = a0 + a1*X, X^2 = 7 ---- */
typedef struct { u64 a[2]; } gl2;
static inline gl2 gl2_make(u64 a0, u64 a1) { gl2 r = {{a0, a1}}; return r; }
static inline gl2 gl2_from(u64 a0) { gl2 r = {{a0, 0}}; return r; }
static inline gl2 gl2_add(gl2 x, gl2 y) { return gl2_make(gl_add(x.a[0], y.a[0]), gl_add(x.a[1], y.a[1])); }
static inline gl2 gl2_sub(gl2 x, gl2 y) { return gl2_make(gl_sub(x.a[0], y.a[0]), gl_sub(x.a[1], y.a[1])); }
static inline gl2 gl2_neg(gl2 x) { return gl2_make(gl_neg(x.a[0]), gl_neg(x.a[1])); }
static inline gl2 gl2_mul(gl2 x, gl2 y) {
    u64 c0 = gl_add(gl_mul(x.a[0], y.a[0]), gl_mul(GL_W, gl_mul(x.a[1], y.a[1])));
    u64 c1 = gl_add(gl_mul(x.a[0], y.a[1]), gl_mul(x.a[1], y.a[0]));
    return gl2_make(c0, c1);
}
static inline gl2 gl2_scale(gl2 x, u64 s) { return gl2_make(gl_mul(x.a[0], s), gl_mul(x.a[1], s)); }
static inline int gl2_eq(gl2 x, gl2 y) { return x.a[0] == y.a[0] && x.a[1] == y.a[1]; }
static inline gl2 gl2_inv(gl2 x) {
    /* 1/(a+bX) = (a-bX)/(a^2 - 7 b^2) */
    u64 n = gl_sub(gl_sqr(x.a[0]), gl_mul(GL_W, gl_sqr(x.a[1])));
    u64 ni = gl_inv(n);
    return gl2_make(gl_mul(x.a[0], ni), gl_mul(gl_neg(x.a[1]), ni));
}
static inline gl2 gl2_pow(gl2 b, u64 e) {
    gl2 r = gl2_from(1);
    while (e) { if (e & 1) r = gl2_mul(r, b); b = gl2_mul(b, b); e >>= 1; }
    return r;
}
#endif
