/* oracle/gl_keccak.c -- TEST INFRASTRUCTURE, not product code.  See gl_keccak.h. */
#include "gl_keccak.h"
#include <string.h>
#define API __attribute__((visibility("default")))

static const u64 KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
    0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
    0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};   /* r[x + 5 y] */
static u64 rol(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

/* FIPS 202 section 3.2, lane A[x][y] at index x + 5 y */
API void glo_keccak_f1600(u64 a[25]) {
    for (int round = 0; round < 24; round++) {
        u64 c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], KROT[x + 5 * y]);      /* rho + pi */
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);   /* chi */
        a[0] ^= KRC[round];
    }
}

API void glo_keccak256(const unsigned char *msg, size_t len, unsigned char out[32]) {
    u64 st[25];
    unsigned char block[136];
    memset(st, 0, sizeof(st));
    size_t off = 0;
    for (;;) {
        size_t take = len - off < 136 ? len - off : 136;
        memset(block, 0, 136);
        if (take) memcpy(block, msg + off, take);
        off += take;
        int last = take < 136;
        if (last) { block[take] ^= 0x01; block[135] ^= 0x80; }
        for (int i = 0; i < 17; i++) { u64 w; memcpy(&w, block + 8 * i, 8); st[i] ^= w; }      /* little-endian host */
        glo_keccak_f1600(st);
        if (last) break;
    }
    memcpy(out, st, 32);
}

static void first25(const unsigned char h[32], u64 out[4]) {
    unsigned char b[32];
    memset(b, 0, 32);
    memcpy(b, h, 25);
    memcpy(out, b, 32);
}
API void glo_keccak_hash_no_pad(const u64 *in, size_t len, u64 out[4]) {
    unsigned char h[32];
    glo_keccak256((const unsigned char *)in, 8 * len, h);
    first25(h, out);
}
API void glo_keccak_hash_or_noop(const u64 *in, size_t len, u64 out[4]) {
    if (8 * len <= 25) { memset(out, 0, 32); memcpy(out, in, 8 * len); return; }
    glo_keccak_hash_no_pad(in, len, out);
}
API void glo_keccak_two_to_one(const u64 l[4], const u64 r[4], u64 out[4]) {
    unsigned char v[50], h[32];
    memcpy(v, l, 25);
    memcpy(v + 25, r, 25);
    glo_keccak256(v, 50, h);
    first25(h, out);
}
/* BytesHash<25>::to_vec: 7-byte chunks */
API void glo_keccak_hash_to_elements(const u64 h[4], u64 out[4]) {
    const unsigned char *b = (const unsigned char *)h;
    for (int i = 0; i < 4; i++) {
        u64 v = 0;
        int n = i < 3 ? 7 : 4;
        memcpy(&v, b + 7 * i, n);
        out[i] = v;
    }
}
/* KeccakPermutation::permute */
API void glo_keccak_permute(u64 st[12]) {
    unsigned char buf[96], h[32];
    memcpy(buf, st, 96);
    size_t blen = 96;
    int got = 0;
    u64 outv[12];
    while (got < 12) {
        glo_keccak256(buf, blen, h);
        memcpy(buf, h, 32);
        blen = 32;
        for (int w = 0; w < 4 && got < 12; w++) {
            u64 v;
            memcpy(&v, h + 8 * w, 8);
            if (v < GL_P) outv[got++] = v;
        }
    }
    memcpy(st, outv, 96);
}
