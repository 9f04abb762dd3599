// prover.hip -- `CircuitData::prove` after witness generation, on one MI355X.
//
// Replaces plonky2 0.1.4 `plonk/prover.rs::prove_with_partition_witness` (from "compute wires
// commitment" on), `plonk/vanishing_poly.rs`, `plonk/proof.rs::OpeningSet::new`,
// `fri/oracle.rs::prove_openings`, `fri/prover.rs::{fri_committed_trees, fri_proof_of_work,
// fri_prover_query_rounds}` and `iop/challenger.rs`, i.e. everything behind `data.prove(pw)`
// [REF src/ecdsa/gadgets/ecdsa.rs:349] except witness generation (CPU, the reference's generators).
// Gate bodies: plonky2 gates/{noop,constant,public_input,arithmetic_base}.rs and the reference's
// [REF src/u32/gates/interleave_u32.rs:84-135, uninterleave_to_u32.rs:93-150, uninterleave_to_b32.rs:95-150].
//
// Data stays on the GPU between stages; the host runs the Fiat-Shamir transcript (a few dozen
// Poseidon permutations) and sequences kernels.  Per proof the PCIe traffic is caps, openings,
// query paths (KBs) in and challenges out.
#include <algorithm>
#include <string.h>
#include "batch.h"
#include "common.h"
#include "merkle.h"
#include "ntt.h"
#include "poseidon.h"
#include "prover_types.h"

namespace {
void make_layout(const glp_circuit_desc &c, Layout &L) {
    const u32 cap = 1u << c.cap_height, nch = c.num_challenges;
    memset(&L, 0, sizeof(L));
    L.oracle_cols[0] = c.num_constants + c.num_routed_wires;
    L.oracle_cols[1] = c.num_wires;
    L.oracle_cols[2] = nch * (1 + c.num_partial_products);
    L.oracle_cols[3] = nch * c.quotient_degree_factor;
    L.nopen = (size_t)c.num_constants + c.num_routed_wires + c.num_wires + 2 * nch + nch * c.num_partial_products +
              nch * c.quotient_degree_factor;
    L.openings = 3 * (size_t)cap * 4;
    L.fri_caps = L.openings + 2 * L.nopen;
    L.queries = L.fri_caps + (size_t)c.num_reductions * cap * 4;
    const u32 lgN = c.degree_bits + c.rate_bits;
    L.depth0 = lgN - c.cap_height;
    size_t q = 0;
    for (int k = 0; k < 4; k++) q += L.oracle_cols[k] + 4 * (size_t)L.depth0;
    u32 lg = lgN;
    for (u32 i = 0; i < c.num_reductions; i++) {
        const u32 ab = c.reduction_arity_bits[i];
        lg -= ab;
        L.step_depth[i] = lg - c.cap_height;
        q += 2 * ((size_t)1 << ab) + 4 * (size_t)L.step_depth[i];
    }
    L.query_stride = q;
    L.final_len = 1u << (lg - c.rate_bits);
    L.final_poly = L.queries + q * c.num_query_rounds;
    L.pow = L.final_poly + 2 * (size_t)L.final_len;
    L.pis = L.pow + 1;
    L.total = L.pis + c.num_public_inputs;
}
}  // namespace

// ------------------------------------------------------------------------------------------ kernels
__device__ __forceinline__ u64 dpow(u64 b, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = mul(r, b); b = sqr(b); e >>= 1; }
    return r;
}

struct PPArgs {
    const u64 *wires, *sigmas, *k_is;
    u64 *zp, *dens;      // dens: scratch [nch][npp + 1][n]
    u64 betas[MAXCH], gammas[MAXCH];
    u64 w_n;
    u32 lg, nr, nch, npp, qdf;
    // many-proofs batch (blockIdx.y = proof): challenges from chal[proof][2 MAXCH] (betas, gammas), arrays strided per proof
    const u64 *chal;
    size_t wires_stride, zp_stride;
};
// K5a: per row, the running products of the quotient chunks  prod_{j in chunk} (w_j + beta k_j x + gamma)/(w_j + beta sigma_j + gamma)
template <int NCH>
__global__ __launch_bounds__(256) void k_pp_rows(PPArgs a) {
    const size_t n = (size_t)1 << a.lg;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 betas[MAXCH], gammas[MAXCH];
    _Pragma("unroll") for (int c = 0; c < NCH; c++) { betas[c] = a.betas[c]; gammas[c] = a.gammas[c]; }
    if (a.chal) {
        const size_t pk = blockIdx.y;
        a.wires += pk * a.wires_stride; a.zp += pk * a.zp_stride; a.dens += pk * a.zp_stride;
        _Pragma("unroll") for (int c = 0; c < NCH; c++) { betas[c] = a.chal[pk * 2 * MAXCH + c]; gammas[c] = a.chal[pk * 2 * MAXCH + MAXCH + c]; }
    }
    const u64 x = dpow(a.w_n, i);
    // Pass 1: prefix products of the chunk numerators (into the output columns) and the chunk denominators
    // (into `dens`); pass 2 walks back down with ONE field inversion per challenge instead of one per chunk:
    // 1/PD_k = (1/PD_{k+1}) * den_{k+1}.
    u64 pn[MAXCH], pd[MAXCH];
    _Pragma("unroll") for (int c = 0; c < NCH; c++) { pn[c] = 1; pd[c] = 1; }
    for (u32 chunk = 0; chunk <= a.npp; chunk++) {
        u64 num[MAXCH], den[MAXCH];
        _Pragma("unroll") for (int c = 0; c < NCH; c++) { num[c] = 1; den[c] = 1; }
        const u32 j1 = min((chunk + 1) * a.qdf, a.nr);
        for (u32 j = chunk * a.qdf; j < j1; j++) {
            const u64 w = a.wires[(size_t)j * n + i], s = a.sigmas[(size_t)j * n + i];
            const u64 kx = mul(a.k_is[j], x);
            _Pragma("unroll") for (int c = 0; c < NCH; c++) {
                num[c] = mul(num[c], add(add(w, mul(betas[c], kx)), gammas[c]));
                den[c] = mul(den[c], add(add(w, mul(betas[c], s)), gammas[c]));
            }
        }
        _Pragma("unroll") for (int c = 0; c < NCH; c++) {
            pn[c] = mul(pn[c], num[c]);
            pd[c] = mul(pd[c], den[c]);
            const u32 col = chunk < a.npp ? NCH + c * a.npp + chunk : c;   // Z column holds the row product for now
            a.zp[(size_t)col * n + i] = pn[c];
            a.dens[((size_t)c * (a.npp + 1) + chunk) * n + i] = den[c];
        }
    }
    u64 ipd[MAXCH];
    _Pragma("unroll") for (int c = 0; c < NCH; c++) ipd[c] = inv(pd[c]);
    for (int chunk = (int)a.npp; chunk >= 0; chunk--) {
        _Pragma("unroll") for (int c = 0; c < NCH; c++) {
            const u32 col = (u32)chunk < a.npp ? NCH + c * a.npp + chunk : c;
            const size_t o = (size_t)col * n + i;
            a.zp[o] = mul(a.zp[o], ipd[c]);
            ipd[c] = mul(ipd[c], a.dens[((size_t)c * (a.npp + 1) + chunk) * n + i]);
        }
    }
}

// The same for traces of at most 128 rows (a batch of small proofs: blockIdx.y = proof, one workgroup per proof).  k_pp_rows gives a row to a lane, and a
// lane then walks ~1000 dependent multiplications (80 wires x two challenges, one inversion per challenge) while 56 lanes of its wave idle: 120 us per 256
// zkdsa proofs, all of it latency.  Here a lane takes one (row, chunk, challenge): the chunk products in parallel through LDS, then one lane per
// (row, challenge) for the prefix products, the inversion and the walk back -- ~170 dependent multiplications -- and, since the whole trace is in this
// workgroup, the running product over the rows as well (k_pp_block_tot / k_pp_scan_tot / k_pp_apply of the large path).  Same values in the same places.
__global__ __launch_bounds__(256) void k_pp_rows_small(PPArgs a) {
    extern __shared__ __attribute__((aligned(16))) u64 pp_lds[];
    const u32 n = 1u << a.lg, nchunks = a.npp + 1, nch = a.nch, units = n * nchunks * nch;
    u64 *snum = pp_lds, *sden = pp_lds + units;              // [c][chunk][i]
    u64 *rowp = pp_lds + 2 * (size_t)units, *zrow = rowp + (size_t)nch * n;      // [c][i]: row products, running products
    const size_t pk = blockIdx.y;
    if (a.chal) { a.wires += pk * a.wires_stride; a.zp += pk * a.zp_stride; }
    const u64 *ch = a.chal ? a.chal + pk * 2 * MAXCH : nullptr;
    for (u32 u = threadIdx.x; u < units; u += 256) {
        const u32 i = u % n, chunk = (u / n) % nchunks, c = u / (n * nchunks);
        const u64 beta = ch ? ch[c] : a.betas[c], gamma = ch ? ch[MAXCH + c] : a.gammas[c];
        const u64 x = dpow(a.w_n, i);
        u64 num = 1, den = 1;
        const u32 j1 = min((chunk + 1) * a.qdf, a.nr);
        for (u32 j = chunk * a.qdf; j < j1; j++) {
            const u64 w = a.wires[(size_t)j * n + i], sg = a.sigmas[(size_t)j * n + i];
            num = mul(num, add(add(w, mul(beta, mul(a.k_is[j], x))), gamma));
            den = mul(den, add(add(w, mul(beta, sg)), gamma));
        }
        snum[u] = num; sden[u] = den;
    }
    __syncthreads();
    for (u32 u = threadIdx.x; u < n * nch; u += 256) {
        const u32 i = u % n, c = u / n;
        const u64 *nm = snum + (size_t)c * nchunks * n + i, *dn = sden + (size_t)c * nchunks * n + i;
        u64 pn = 1, pd = 1;
        for (u32 chunk = 0; chunk < nchunks; chunk++) {
            pn = mul(pn, nm[(size_t)chunk * n]);
            pd = mul(pd, dn[(size_t)chunk * n]);
            const u32 col = chunk < a.npp ? nch + c * a.npp + chunk : c;   // Z column holds the row product for now
            a.zp[(size_t)col * n + i] = pn;
        }
        u64 ipd = inv(pd);
        for (int chunk = (int)a.npp; chunk >= 0; chunk--) {
            const u32 col = (u32)chunk < a.npp ? nch + c * a.npp + chunk : c;
            const size_t o = (size_t)col * n + i;
            const u64 v = mul(a.zp[o], ipd);
            a.zp[o] = v;
            if ((u32)chunk == a.npp) rowp[(size_t)c * n + i] = v;          // the row's whole product
            ipd = mul(ipd, dn[(size_t)chunk * n]);
        }
    }
    __syncthreads();
    // Z_i = product of the rows before i (one lane per challenge walks the <= 128 rows), then every partial product of row i times Z_i
    if (threadIdx.x < nch) {
        const u32 c = threadIdx.x;
        u64 acc = 1;
        for (u32 i = 0; i < n; i++) { const u64 r = rowp[(size_t)c * n + i]; zrow[(size_t)c * n + i] = acc; acc = mul(acc, r); }
    }
    __syncthreads();
    for (u32 u = threadIdx.x; u < units; u += 256) {
        const u32 i = u % n, k = (u / n) % nchunks, c = u / (n * nchunks);
        const u64 z = zrow[(size_t)c * n + i];
        if (k < a.npp) { const size_t o = (size_t)(nch + c * a.npp + k) * n + i; a.zp[o] = mul(a.zp[o], z); }
        else a.zp[(size_t)c * n + i] = z;
    }
}
// K5b: product of each block of 256 row products
__global__ __launch_bounds__(256) void k_pp_block_tot(const u64 *zp, u64 *tot, u32 lg, u32 nblocks, size_t zp_stride) {
    __shared__ u64 sh[256];
    zp += (size_t)blockIdx.z * zp_stride; tot += (size_t)blockIdx.z * gridDim.y * nblocks;
    const size_t n = (size_t)1 << lg;
    const u32 c = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
    const size_t i = (size_t)b * 256 + t;
    sh[t] = i < n ? zp[(size_t)c * n + i] : 1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sh[t] = mul(sh[t], sh[t + s]); __syncthreads(); }
    if (t == 0) tot[(size_t)c * nblocks + b] = sh[0];
}
// K5c: exclusive prefix product of the block totals (one workgroup per challenge)
__global__ __launch_bounds__(256) void k_pp_scan_tot(u64 *tot, u32 nblocks) {
    __shared__ u64 sh[256];
    const u32 c = blockIdx.x, t = threadIdx.x;
    u64 *v = tot + ((size_t)blockIdx.y * gridDim.x + c) * nblocks;
    const u32 m = (nblocks + 255) / 256;
    u64 loc = 1;
    for (u32 k = t * m; k < min((t + 1) * m, nblocks); k++) loc = mul(loc, v[k]);
    sh[t] = loc;
    __syncthreads();
    if (t == 0) { u64 acc = 1; for (int k = 0; k < 256; k++) { u64 x = sh[k]; sh[k] = acc; acc = mul(acc, x); } }
    __syncthreads();
    u64 acc = sh[t];
    for (u32 k = t * m; k < min((t + 1) * m, nblocks); k++) { u64 x = v[k]; v[k] = acc; acc = mul(acc, x); }
}
// K5d: Z(x_i) = prefix(block) * in-block exclusive scan; partial products *= Z
__global__ __launch_bounds__(256) void k_pp_apply(u64 *zp, const u64 *tot, u32 lg, u32 nblocks, u32 nch, u32 npp, size_t zp_stride) {
    __shared__ u64 sh[2][256];
    zp += (size_t)blockIdx.z * zp_stride; tot += (size_t)blockIdx.z * gridDim.y * nblocks;
    const size_t n = (size_t)1 << lg;
    const u32 c = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
    const size_t i = (size_t)b * 256 + t;
    const u64 mine = i < n ? zp[(size_t)c * n + i] : 1;
    int cur = 0;
    sh[0][t] = mine;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {     // Hillis-Steele inclusive scan
        u64 v = sh[cur][t];
        if (t >= off) v = mul(sh[cur][t - off], v);
        sh[cur ^ 1][t] = v;
        cur ^= 1;
        __syncthreads();
    }
    const u64 excl = t ? sh[cur][t - 1] : 1;
    if (i >= n) return;
    const u64 z = mul(tot[(size_t)c * nblocks + b], excl);
    zp[(size_t)c * n + i] = z;
    for (u32 k = 0; k < npp; k++) {
        const size_t o = (size_t)(nch + c * npp + k) * n + i;
        zp[o] = mul(zp[o], z);
    }
}

// prod_{x < bound} (v - x), v canonical -> NON-canonical u64 (it only ever feeds acc_fma, which takes any u64)
__device__ __forceinline__ u64 range_product(u64 v, u32 bound) {
    if (bound == 4) {                       // v(v-3) * (v-1)(v-2) = u (u + 2): two multiplications instead of three
        // (mul_nc, not mul_nc_cc: in k_quotient_limbs the carry-chain form measured 7.33 -> 7.83 ms, in the permutation loop 5.85 -> 5.66)
        const u64 u = mul_nc(v, add_cnc(v, P - 3));       // v - 3 as v + (p - 3), left non-canonical; u any u64
        return mul_nc(u, add_cnc(2, u));
    }
    u64 p = v;
    for (u32 x = 1; x + 1 < bound; x++) p = mul(p, sub(v, (u64)x));
    return bound > 1 ? mul_nc(p, sub(v, (u64)(bound - 1))) : p;
}
// sum_j 4^j limb_j over up to 16 canonical limbs without a modular operation per limb: the 32-bit halves are
// accumulated separately (each sum < 2^32 (4^16 - 1) / 3 < 2^64 / 3) and folded once.
struct Base4Sum { u64 lo, hi; };
__device__ __forceinline__ void b4_zero(Base4Sum &b) { b.lo = 0; b.hi = 0; }
__device__ __forceinline__ void b4_add(Base4Sum &b, u64 limb, u32 j /* < 16 */) {
    const u32 w = 1u << (2 * j);
    b.lo += (u64)(u32)limb * w;
    b.hi += (u64)(u32)(limb >> 32) * w;
}
__device__ __forceinline__ u64 b4_value(const Base4Sum &b) {       // canonical
    const u64 l = b.lo + (b.hi << 32);
    const u32 h = (u32)(b.hi >> 32) + (l < b.lo ? 1u : 0u);
    return canon(fold96_nc(l, h));
}
// Unreduced accumulator for sum_k c_k * alpha^k: 128-bit products are added into five 32-bit words and folded
// once per gate instead of once per constraint (a modular multiply-add costs ~40 issue slots, this ~19).
struct Acc160 { u32 w0, w1, w2, w3, w4; };
__device__ __forceinline__ void acc_zero(Acc160 &a) { a.w0 = a.w1 = a.w2 = a.w3 = a.w4 = 0; }
__device__ __forceinline__ void acc_fma(Acc160 &a, u64 v, u64 m) {
    const u32 v0 = (u32)v, v1 = (u32)(v >> 32), m0 = (u32)m, m1 = (u32)(m >> 32);
    const u64 p00 = (u64)v0 * m0;
    const u64 p01 = (u64)v0 * m1 + (p00 >> 32);
    const u64 p10 = (u64)v1 * m0 + (u32)p01;
    const u64 p11 = (u64)v1 * m1 + (p01 >> 32) + (p10 >> 32);
    u32 c;
    a.w0 = __builtin_addc(a.w0, (u32)p00, 0u, &c);
    a.w1 = __builtin_addc(a.w1, (u32)p10, c, &c);
    a.w2 = __builtin_addc(a.w2, (u32)p11, c, &c);
    a.w3 = __builtin_addc(a.w3, (u32)(p11 >> 32), c, &c);
    a.w4 += c;
}
__device__ __forceinline__ u64 acc_reduce(const Acc160 &a) {      // canonical
    const u64 h = fold96_nc(((u64)a.w3 << 32) | a.w2, a.w4);       // (w2 + w3 2^32 + w4 2^64) mod p
    return canon(fold128_nc(a.w0, a.w1, (u32)h, (u32)(h >> 32)));
}

// Gate constraints: sum_k v_k alpha^k with NO carries per term.  v is cut into 22-bit limbs and alpha^k into 32-bit
// halves; each of the six limb products (< 2^54) is accumulated in its own 64-bit register by one v_mad_u64_u32, so up
// to 1024 terms fit before anything can overflow (glp_circuit_create rejects gates with more constraints).  6 issue
// slots per term against 14 for the 160-bit carry chain above; the limbs of v are shared by all challenges.
constexpr u32 ACC_MAX_TERMS = 1024;
struct AccLimb { u64 a00, a01, a10, a11, a20, a21; };     // a[i][j]: limb i of v (bits 22 i ..) times half j of m
__device__ __forceinline__ void acc2_zero(AccLimb &a) { a.a00 = a.a01 = a.a10 = a.a11 = a.a20 = a.a21 = 0; }
__device__ __forceinline__ void acc2_fma(AccLimb &a, u32 v0, u32 v1, u32 v2, u64 m) {
    const u32 m0 = (u32)m, m1 = (u32)(m >> 32);
    a.a00 += (u64)v0 * m0; a.a01 += (u64)v0 * m1;
    a.a10 += (u64)v1 * m0; a.a11 += (u64)v1 * m1;
    a.a20 += (u64)v2 * m0; a.a21 += (u64)v2 * m1;
}
template <int E> __device__ __forceinline__ void acc_add_shifted(Acc160 &w, u64 x) {   // w += x << E
    constexpr int idx = E / 32, sh = E % 32;
    const u64 lo = x << sh;
    const u32 t0 = (u32)lo, t1 = (u32)(lo >> 32);
    u32 t2 = 0;
    if constexpr (sh != 0) t2 = (u32)(x >> (64 - sh));
    u32 *W[5] = {&w.w0, &w.w1, &w.w2, &w.w3, &w.w4};
    u32 c;
    *W[idx] = __builtin_addc(*W[idx], t0, 0u, &c);
    *W[idx + 1] = __builtin_addc(*W[idx + 1], t1, c, &c);
    *W[idx + 2] = __builtin_addc(*W[idx + 2], t2, c, &c);
    if constexpr (idx + 3 < 5) *W[idx + 3] = __builtin_addc(*W[idx + 3], 0u, c, &c);
    if constexpr (idx + 4 < 5) *W[idx + 4] = __builtin_addc(*W[idx + 4], 0u, c, &c);
}
__device__ __forceinline__ u64 acc2_reduce(const AccLimb &a) {    // canonical
    Acc160 w;
    acc_zero(w);
    acc_add_shifted<0>(w, a.a00); acc_add_shifted<22>(w, a.a10); acc_add_shifted<32>(w, a.a01);
    acc_add_shifted<44>(w, a.a20); acc_add_shifted<54>(w, a.a11); acc_add_shifted<76>(w, a.a21);
    return acc_reduce(w);
}

// The same carry-free scheme with the roles swapped, for the quotient: the multiplier alpha^k comes from a table the host
// cuts into 22-bit limbs once per proof, so a constraint value enters as its two 32-bit halves -- the registers it already
// lives in -- instead of being cut into three limbs per term (5 shift / mask slots per constraint, 620 constraints per point).
// m enters TWICE, as m and as m' = m 2^32 mod p: then  v m = vlo m + vhi m'  and both products sit at the same limb weights, so
// three accumulators per sum are enough (six if the 2^32 is left to the weights).  Half the registers per gate in the quotient
// kernels -- what bounds how many gates share one pass over the wire planes -- for twice the (scalar) table loads.
struct AccHL { u64 c0, c1, c2; };                         // c[j]: limb j of m (bits 22 j ..) times vlo + limb j of m' times vhi
constexpr u32 ACC3_MAX_TERMS = 512;                       // 2 products < 2^54 per term and accumulator
inline void apl_words(u64 m, u64 out[4]) {                // host side of the table
    const u64 mp = glf::mul(m, 1ull << 32);
    out[0] = (m & 0x3FFFFFull) | (((m >> 22) & 0x3FFFFFull) << 32); out[1] = m >> 44;
    out[2] = (mp & 0x3FFFFFull) | (((mp >> 22) & 0x3FFFFFull) << 32); out[3] = mp >> 44;
}
__device__ __forceinline__ void acc3_zero(AccHL &a) { a.c0 = a.c1 = a.c2 = 0; }
__device__ __forceinline__ void acc3_fma(AccHL &a, u64 v, const u64 *ml) {
    const u32 vlo = (u32)v, vhi = (u32)(v >> 32);
    const u64 w0 = ml[0], w1 = ml[1], w2 = ml[2], w3 = ml[3];
    a.c0 += (u64)vlo * (u32)w0; a.c1 += (u64)vlo * (u32)(w0 >> 32); a.c2 += (u64)vlo * (u32)w1;
    a.c0 += (u64)vhi * (u32)w2; a.c1 += (u64)vhi * (u32)(w2 >> 32); a.c2 += (u64)vhi * (u32)w3;
}
__device__ __forceinline__ u64 acc3_reduce(const AccHL &a) {      // canonical
    Acc160 w;
    acc_zero(w);
    acc_add_shifted<0>(w, a.c0); acc_add_shifted<22>(w, a.c1); acc_add_shifted<44>(w, a.c2);
    return acc_reduce(w);
}

struct QArgs {                      // per circuit and FRI domain: the same for every proof of a batch
    const u64 *cs;                  // coset-major LDE [ncols][R][n] of constants ++ sigmas
    const DevGate *gates;
    const u64 *k_is;
    u64 shift_r[MAXR], zh[MAXR], zh_inv[MAXR];   // per evaluated plane
    u64 w_n, n_field;
    u32 lg, rb, step, nc, nsel, nr, nw, nch, npp, qdf, num_gates, nterms, many_selectors, gate_mode;
    u32 k_ratio;                    // != 0: k_is[j] = k_ratio^j (plonky2's get_unique_coset_shifts: powers of the generator 7)
    const u64 *l0;                  // [Rq][n]: L_0(x) = Z_H(x) / (n (x - 1)) on the evaluated planes (k_l0_table)
};
struct QProof {                     // per proof
    const u64 *wl, *zl;             // coset-major LDEs of the wires and of Z ++ partial products
    u64 *out;                       // [nch][Rq][n]
    const u64 *apow;                // [nch][nterms] powers of the alphas
    const u64 *apl;                 // the same powers as 22-bit limbs of m and of m 2^32, APL_WORDS words per power (AccHL)
    u64 betas[MAXCH], gammas[MAXCH], pih[4];
};
// many-proofs batch (glp_prove_batch): blockIdx.z = proof; arrays advance by a stride per proof, challenges and the public-input
// hash come from pp[proof][3 MAXCH] (betas, gammas, pih).  pp == nullptr: a single proof described by the QProof kernel argument.
struct QBatch { const u64 *pp; size_t wl_stride, zl_stride, out_stride, apow_stride; };
static_assert(MAXCH == 4, "pp layout: 4 betas, 4 gammas, 4 words of the public-input hash");
__device__ __forceinline__ QProof q_proof(const QProof &p0, const QBatch &b) {
    QProof p = p0;
    if (b.pp) {
        const size_t k = blockIdx.z;
        p.wl += k * b.wl_stride; p.zl += k * b.zl_stride; p.out += k * b.out_stride; p.apow += k * b.apow_stride; p.apl += APL_WORDS * k * b.apow_stride;
        const u64 *q = b.pp + k * 3 * MAXCH;
        _Pragma("unroll") for (int c = 0; c < MAXCH; c++) { p.betas[c] = q[c]; p.gammas[c] = q[MAXCH + c]; p.pih[c] = q[2 * MAXCH + c]; }
    }
    return p;
}
// L_0 on the evaluated planes.  One thread owns position q of every plane and inverts the Rq denominators n (x_rq - 1)
// with ONE field inversion (Montgomery's trick) instead of one per point inside k_quotient.
__global__ __launch_bounds__(256) void k_l0_table(QArgs a, u64 *out, u32 Rq) {
    const size_t n = (size_t)1 << a.lg;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const u64 wq = dpow(a.w_n, q);
    u64 d[MAXR], pre[MAXR];
    u64 run = 1;
    for (u32 rq = 0; rq < Rq; rq++) {            // x is never 1 on a coset g W^r H: every denominator is invertible
        d[rq] = mul(a.n_field, sub(mul(a.shift_r[rq], wq), 1));
        pre[rq] = run;
        run = mul(run, d[rq]);
    }
    u64 iv = inv(run);
    for (int rq = (int)Rq - 1; rq >= 0; rq--) {
        out[(size_t)rq * n + q] = mul(a.zh[rq], mul(iv, pre[rq]));
        iv = mul(iv, d[rq]);
    }
}
// selector filter of one gate at one point: prod_{i in group, i != row} (i - s) [* (UNUSED - s)]
__device__ __forceinline__ u64 gate_filter(const QArgs &a, const DevGate &g, size_t N, size_t slot) {
    const u64 s = a.cs[(size_t)g.selector_index * N + slot];
    u64 filter = 1;
    for (u32 i = g.group_start; i < g.group_end; i++)
        if (i != g.row) filter = mul(filter, sub((u64)i, s));
    if (a.many_selectors) filter = mul(filter, sub(0xFFFFFFFFull, s));
    return filter;
}
// The unfiltered constraints of one gate, multiplied into the carry-free accumulators ga[c] (+= constraint_k alpha_c^(k0 + k)).
// HEAD_ONLY (the four base-4 limb gates of plonky2_u32): skip the limb columns -- their range products, base-4 sums and the
// sum-equals-wire constraints -- which k_quotient_limbs evaluates for all fused gates from ONE read of the wire planes.
template <int NCH, int TYPE, bool HEAD_ONLY = false>
__device__ __forceinline__ void gate_terms(const QArgs &a, const QProof &p, const DevGate &g, size_t N, size_t slot, u32 k0, AccHL (&ga)[MAXCH]) {
    const u32 nt = a.nterms;
    const u64 *W = p.wl + slot;                       // wire j  -> W[j * N]
    const u64 *GC = a.cs + (size_t)a.nsel * N + slot; // gate constant i -> GC[i * N]
    {
        const u64 *ap = p.apl + APL_WORDS * (size_t)k0;
#define EMIT(k, v)                                                                     \
    do {                                                                               \
        const u64 _v = (v);                                                            \
        _Pragma("unroll") for (int c2 = 0; c2 < NCH; c2++) acc3_fma(ga[c2], _v, ap + APL_WORDS * ((size_t)c2 * nt + (k)));   \
    } while (0)
// Base-4 limb columns LIMBS[j*N], j = COUNT-1 .. 0: eight loads are issued before their values are used (the gate
// loops have run-time bounds, so the compiler cannot software-pipeline them itself).  Constraint index KIDX may use _j.
#define LIMBS4_DESC(LIMBS, COUNT, SPLIT, KIDX, ACCLO, ACCHI)                                                      \
    {                                                                                                             \
        Base4Sum _slo, _shi;                                                                                      \
        b4_zero(_slo); b4_zero(_shi);                                                                             \
        for (int _j0 = (int)(COUNT); _j0 > 0; _j0 -= 8) {                                                         \
            u64 _lv[8];                                                                                            \
            _Pragma("unroll") for (int _t = 0; _t < 8; _t++) if (_t < _j0) _lv[_t] = (LIMBS)[(size_t)(_j0 - 1 - _t) * N]; \
            _Pragma("unroll") for (int _t = 0; _t < 8; _t++) if (_t < _j0) {                                      \
                const int _j = _j0 - 1 - _t;                                                                      \
                EMIT((KIDX), range_product(_lv[_t], 4));                                                           \
                if (_j < (int)(SPLIT)) b4_add(_slo, _lv[_t], (u32)_j); else b4_add(_shi, _lv[_t], (u32)(_j - (int)(SPLIT))); \
            }                                                                                                     \
        }                                                                                                         \
        ACCLO = b4_value(_slo);                                                                                   \
        if ((int)(COUNT) > (int)(SPLIT)) ACCHI = b4_value(_shi);                                                  \
    }
        switch (TYPE >= 0 ? (u32)TYPE : g.type) {   // TYPE >= 0: the switch folds to one case at compile time
        case GLP_GATE_CONSTANT:
            for (u32 i = 0; i < g.p0; i++) EMIT(i, sub(GC[(size_t)i * N], W[(size_t)i * N]));
            break;
        case GLP_GATE_PUBLIC_INPUT:
            for (u32 i = 0; i < 4; i++) EMIT(i, sub(W[(size_t)i * N], p.pih[i]));
            break;
        case GLP_GATE_ARITHMETIC: {
            const u64 c0 = GC[0], c1 = GC[N];
            for (u32 i = 0; i < g.p0; i++) {
                const u64 m0 = W[(size_t)(4 * i) * N], m1 = W[(size_t)(4 * i + 1) * N];
                const u64 ad = W[(size_t)(4 * i + 2) * N], o = W[(size_t)(4 * i + 3) * N];
                EMIT(i, sub(o, add(mul(mul(m0, m1), c0), mul(ad, c1))));
            }
            break;
        }
        case GLP_GATE_POSEIDON: {
            // gates/poseidon.rs: wires = inputs 0..11, outputs 12..23, swap 24, delta 25..28, full_sbox_0(r=1..3)
            // from 29, partial_sbox from 65, full_sbox_1 from 87.  The S-box inputs are the only place wires enter, so any
            // schedule of the linear layers gives the constraints plonky2's sparse-matrix schedule gives: this is the
            // permutation's own gfx950 schedule (poseidon.h: non-canonical values between layers, the next round's constants
            // folded into the linear layer, partial rounds in blocks of 3 / 4 / 4 / 4 / 4 / 3) with the wire taking the
            // place of the state wherever an S-box is entered.
#if defined(__HIP_DEVICE_COMPILE__)
            u32 k = 0;
            u64 st[12];
            const u64 swap = W[(size_t)24 * N];
            EMIT(k, mul_nc(swap, sub(swap, 1))); k++;
            for (u32 i = 0; i < 4; i++) {
                const u64 lhs = W[(size_t)i * N], rhs = W[(size_t)(i + 4) * N], dl = W[(size_t)(25 + i) * N];
                EMIT(k, sub(mul(swap, sub(rhs, lhs)), dl)); k++;
                st[i] = add(lhs, dl); st[i + 4] = sub(rhs, dl);
            }
            for (u32 i = 8; i < 12; i++) st[i] = W[(size_t)i * N];
            for (u32 i = 0; i < 12; i++) st[i] = add(st[i], pos::RC[i]);
            pos::sbox_layer_nc(st);
            pos::mds_add_nc(st, pos::RCN.k[0]);
            for (u32 r = 1; r < 4; r++) {
                for (u32 i = 0; i < 12; i++) { const u64 in = W[(size_t)(29 + 12 * (r - 1) + i) * N]; EMIT(k, add_cnc(neg(in), st[i])); k++; st[i] = in; }
                pos::sbox_layer_nc(st);
                if (r < 3) pos::mds_add_nc(st, pos::RCN.k[r]);       // round 3's linear layer is part of the merged block
            }
            u32 pr = 0;                                              // partial round of the block's first S-box
            auto wire_in = [&](int j, u64 z) {
                const u64 in = W[(size_t)(65 + pr + (u32)j) * N];
                EMIT(k, add_cnc(neg(in), z)); k++;
                return in;
            };
            pos::partial_block_nc<3, true>(st, pos::PBM[0], wire_in); pr += 3;
            for (int b = 0; b < 4; b++) { pos::partial_block_nc<4, false>(st, pos::PB4[b], wire_in); pr += 4; }
            pos::partial_block_nc<3, false>(st, pos::PB3[0], wire_in);
            for (u32 r = 0; r < 4; r++) {
                for (u32 i = 0; i < 12; i++) { const u64 in = W[(size_t)(87 + 12 * r + i) * N]; EMIT(k, add_cnc(neg(in), st[i])); k++; st[i] = in; }
                pos::sbox_layer_nc(st);
                pos::mds_add_nc(st, r < 3 ? pos::RCN.k[4 + r] : pos::RC_ZERO);
            }
            for (u32 i = 0; i < 12; i++) { EMIT(k, add_cnc(neg(W[(size_t)(12 + i) * N]), st[i])); k++; }
#endif
            break;
        }
        case GLP_GATE_U32_INTERLEAVE: {
            u32 k = 0;
            for (u32 i = 0; i < g.p0; i++) {
                const u64 xw = W[(size_t)(2 * i) * N], xi = W[(size_t)(2 * i + 1) * N];
                const u64 *bits = W + (size_t)(2 * g.p0 + 32 * i) * N;
                u64 cx = 0, cxi = 0;
                const u32 kb = k + 2;
                for (u32 b = 0; b < 32; b++) {
                    const u64 bit = bits[(size_t)b * N];
                    cx = add(dbl(cx), bit);
                    cxi = add(dbl(dbl(cxi)), bit);
                    EMIT(kb + b, mul_nc(bit, sub(bit, 1)));
                }
                EMIT(k, sub(cx, xw));
                EMIT(k + 1, sub(cxi, xi));
                k += 34;
            }
            break;
        }
        case GLP_GATE_UNINTERLEAVE_U32:
        case GLP_GATE_UNINTERLEAVE_B32: {
            u32 k = 0;
            for (u32 i = 0; i < g.p0; i++) {
                const u64 xi = W[(size_t)(3 * i) * N], xe = W[(size_t)(3 * i + 1) * N], xo = W[(size_t)(3 * i + 2) * N];
                const u64 *bits = W + (size_t)(3 * g.p0 + 64 * i) * N;
                u64 cxi = 0, ce = 0, co = 0;
                const u32 kb = k + 3;
                for (u32 j = 0; j < 32; j++) {   // Horner from the most significant bit: coeff 2^(31-j) or 4^(31-j)
                    const u64 be = bits[(size_t)(2 * j) * N], bo = bits[(size_t)(2 * j + 1) * N];
                    cxi = add(dbl(add(dbl(cxi), be)), bo);
                    if (g.type == GLP_GATE_UNINTERLEAVE_U32) { ce = add(dbl(ce), be); co = add(dbl(co), bo); }
                    else { ce = add(dbl(dbl(ce)), be); co = add(dbl(dbl(co)), bo); }
                    EMIT(kb + 2 * j, mul_nc(be, sub(be, 1)));
                    EMIT(kb + 2 * j + 1, mul_nc(bo, sub(bo, 1)));
                }
                EMIT(k, sub(cxi, xi));
                EMIT(k + 1, sub(ce, xe));
                EMIT(k + 2, sub(co, xo));
                k += 67;
            }
            break;
        }
        case GLP_GATE_U32_ARITHMETIC: {
            u32 k = 0; const u32 nops = g.p0;
            for (u32 i = 0; i < nops; i++) {
                const u64 m0 = W[(size_t)(6 * i) * N], m1 = W[(size_t)(6 * i + 1) * N], ad = W[(size_t)(6 * i + 2) * N];
                const u64 lo = W[(size_t)(6 * i + 3) * N], hi = W[(size_t)(6 * i + 4) * N], iv = W[(size_t)(6 * i + 5) * N];
                const u64 hi_not_max = sub(mul(iv, sub(0xFFFFFFFFull, hi)), 1);
                EMIT(k, mul_nc(hi_not_max, lo)); k++;
                EMIT(k, sub(add(mul(hi, (u64)1 << 32), lo), add(mul(m0, m1), ad))); k++;
                if constexpr (!HEAD_ONLY) {
                    u64 cl = 0, chh = 0;
                    const u64 *limbs = W + (size_t)(6 * nops + 32 * i) * N;
                    LIMBS4_DESC(limbs, 32, 16, k + (31 - _j), cl, chh);
                    EMIT(k + 32, sub(cl, lo));
                    EMIT(k + 33, sub(chh, hi));
                }
                k += 34;
            }
            break;
        }
        case GLP_GATE_U32_ADD_MANY: {
            u32 k = 0; const u32 na = g.p0, nops = g.p1, wd = na + 3;
            for (u32 i = 0; i < nops; i++) {
                // addends + carry in as a 96-bit integer sum (three carry instructions per term against a modular addition's eight), folded once
                u64 slo = W[(size_t)(wd * i + na) * N];
                u32 shi = 0;
                for (u32 j = 0; j < na; j += 8) {          // eight loads in flight (na is a run-time value: no unrolling otherwise)
                    u64 t[8];
                    _Pragma("unroll") for (u32 e = 0; e < 8; e++) t[e] = j + e < na ? W[(size_t)(wd * i + j + e) * N] : 0;
                    _Pragma("unroll") for (u32 e = 0; e < 8; e++) { slo += t[e]; shi += slo < t[e] ? 1u : 0u; }
                }
                const u64 sum = canon(fold96_nc(slo, shi));
                const u64 res = W[(size_t)(wd * i + na + 1) * N], car = W[(size_t)(wd * i + na + 2) * N];
                EMIT(k, sub(add(mul(car, (u64)1 << 32), res), sum)); k++;
                if constexpr (!HEAD_ONLY) {
                    u64 cr = 0, cc = 0;
                    const u64 *limbs = W + (size_t)(wd * nops + 18 * i) * N;
                    LIMBS4_DESC(limbs, 18, 16, k + (17 - _j), cr, cc);
                    EMIT(k + 18, sub(cr, res));
                    EMIT(k + 19, sub(cc, car));
                }
                k += 20;
            }
            break;
        }
        case GLP_GATE_U32_SUBTRACTION: {
            u32 k = 0; const u32 nops = g.p0;
            for (u32 i = 0; i < nops; i++) {
                const u64 xx = W[(size_t)(5 * i) * N], yy = W[(size_t)(5 * i + 1) * N], bi = W[(size_t)(5 * i + 2) * N];
                const u64 res = W[(size_t)(5 * i + 3) * N], bo = W[(size_t)(5 * i + 4) * N];
                EMIT(k, sub(res, add(sub(sub(xx, yy), bi), mul(bo, (u64)1 << 32)))); k++;
                if constexpr (!HEAD_ONLY) {
                    u64 cl = 0, unused_hi = 0;
                    const u64 *limbs = W + (size_t)(5 * nops + 16 * i) * N;
                    LIMBS4_DESC(limbs, 16, 16, k + (15 - _j), cl, unused_hi);
                    (void)unused_hi;
                    EMIT(k + 16, sub(cl, res));
                }
                k += 17;
                EMIT(k, mul_nc(bo, sub(1, bo))); k++;
            }
            break;
        }
        case GLP_GATE_U32_RANGE_CHECK: {
            u32 k = 0; const u32 nin = g.p0;
            if constexpr (!HEAD_ONLY) {
                for (u32 i = 0; i < nin; i++) {
                    const u64 *aux = W + (size_t)(nin + 16 * i) * N;
                    u64 sum = 0, unused_hi = 0;
                    LIMBS4_DESC(aux, 16, 16, k + 1 + _j, sum, unused_hi);
                    (void)unused_hi;
                    EMIT(k, sub(sum, W[(size_t)i * N]));
                    k += 17;
                }
            }
            break;
        }
        case GLP_GATE_COMPARISON: {
            u32 k = 0; const u32 nb = g.p0, ncx = g.p1, cb = (nb + ncx - 1) / ncx, cs = 1u << cb;
            const u64 *ca = W + (size_t)4 * N, *cbp = ca + (size_t)ncx * N, *ed = cbp + (size_t)ncx * N;
            const u64 *ceq = ed + (size_t)ncx * N, *iv = ceq + (size_t)ncx * N, *mb = iv + (size_t)ncx * N;
            u64 fa = 0, fb = 0;
            for (int i = (int)ncx - 1; i >= 0; i--) { fa = add(mul(fa, cs), ca[(size_t)i * N]); fb = add(mul(fb, cs), cbp[(size_t)i * N]); }
            EMIT(k, sub(fa, W[0])); k++;
            EMIT(k, sub(fb, W[N])); k++;
            u64 msd = 0;
            for (u32 i0 = 0; i0 < ncx; i0 += 4) {     // 20 loads in flight per batch of four chunks
                u64 la[4], lb[4], le[4], li[4], ld[4];
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (i0 + t < ncx) {
                        const size_t o = (size_t)(i0 + t) * N;
                        la[t] = ca[o]; lb[t] = cbp[o]; le[t] = ceq[o]; li[t] = iv[o]; ld[t] = ed[o];
                    }
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (i0 + t < ncx) {
                        EMIT(k, range_product(la[t], cs)); k++;
                        EMIT(k, range_product(lb[t], cs)); k++;
                        const u64 diff = sub(lb[t], la[t]);
                        EMIT(k, sub(mul(diff, ld[t]), sub(1, le[t]))); k++;
                        EMIT(k, mul_nc(le[t], diff)); k++;
                        EMIT(k, sub(li[t], mul(le[t], msd))); k++;
                        msd = add(li[t], mul(sub(1, le[t]), diff));
                    }
            }
            const u64 msdw = W[(size_t)3 * N];
            EMIT(k, sub(msdw, msd)); k++;
            u64 bc = 0;
            for (u32 j = 0; j <= cb; j++) { const u64 bit = mb[(size_t)j * N]; EMIT(k, mul_nc(bit, sub(1, bit))); k++; }
            for (int j = (int)cb; j >= 0; j--) bc = add(dbl(bc), mb[(size_t)j * N]);
            EMIT(k, sub(add((u64)cs, msdw), bc)); k++;
            EMIT(k, sub(W[(size_t)2 * N], mb[(size_t)cb * N])); k++;
            break;
        }
        case GLP_GATE_BASE_SUM: {
            u32 k = 0; const u32 nl = g.p0, Bb = g.p1;
            u64 sum = 0;
            for (int j = (int)nl - 1; j >= 0; j--) sum = add(mul(sum, Bb), W[(size_t)(1 + j) * N]);
            EMIT(k, sub(sum, W[0])); k++;
            for (u32 j = 0; j < nl; j++) { EMIT(k, range_product(W[(size_t)(1 + j) * N], Bb)); k++; }
            break;
        }
        case GLP_GATE_RANDOM_ACCESS: {
            u32 k = 0; const u32 bits = g.p0, copies = g.p1 & 0xFFFF, nextra = g.p1 >> 16, vs = 1u << bits;
            const u32 routed = (2 + vs) * copies + nextra;
            for (u32 cpy = 0; cpy < copies; cpy++) {
                const u64 *bse = W + (size_t)((2 + vs) * cpy) * N, *bw = W + (size_t)(routed + bits * cpy) * N;
                u64 idx = 0;
                u64 sel;
                if (bits == 4) {           // the width the reference uses; folded in registers, every wire loaded once and the loads batched
                    const u64 b0 = bw[0], b1 = bw[N], b2 = bw[2 * N], b3 = bw[3 * N], claimed_idx = bse[0];
                    EMIT(k, mul_nc(b0, sub(b0, 1))); EMIT(k + 1, mul_nc(b1, sub(b1, 1))); EMIT(k + 2, mul_nc(b2, sub(b2, 1))); EMIT(k + 3, mul_nc(b3, sub(b3, 1)));
                    k += 4;
                    idx = add(dbl(add(dbl(add(dbl(b3), b2)), b1)), b0);
                    EMIT(k, sub(idx, claimed_idx)); k++;
                    u64 l2[4];
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) {
                        const u64 *it = bse + (size_t)(2 + 4 * q4) * N;
                        const u64 i0 = it[0], i1 = it[N], i2 = it[2 * N], i3 = it[3 * N];
                        const u64 f0 = add(i0, mul(b0, sub(i1, i0))), f1 = add(i2, mul(b0, sub(i3, i2)));
                        l2[q4] = add(f0, mul(b1, sub(f1, f0)));
                    }
                    const u64 g0 = add(l2[0], mul(b2, sub(l2[1], l2[0]))), g1 = add(l2[2], mul(b2, sub(l2[3], l2[2])));
                    sel = add(g0, mul(b3, sub(g1, g0)));
                } else {                   // generic width: select by recursion over the index bits (no local array)
                    for (u32 b = 0; b < bits; b++) { const u64 bit = bw[(size_t)b * N]; EMIT(k, mul_nc(bit, sub(bit, 1))); k++; }
                    for (int b = (int)bits - 1; b >= 0; b--) idx = add(dbl(idx), bw[(size_t)b * N]);
                    EMIT(k, sub(idx, bse[0])); k++;
                    sel = 0;
                    for (u32 j = 0; j < vs; j++) {
                        u64 ind = 1;       // product over bits of (bit or 1 - bit): Lagrange indicator of slot j
                        for (u32 b = 0; b < bits; b++) { const u64 bit = bw[(size_t)b * N]; ind = mul(ind, ((j >> b) & 1) ? bit : sub(1, bit)); }
                        sel = add(sel, mul(ind, bse[(size_t)(2 + j) * N]));
                    }
                }
                EMIT(k, sub(sel, bse[N])); k++;
            }
            for (u32 e = 0; e < nextra; e++) { EMIT(k, sub(GC[(size_t)e * N], W[(size_t)((2 + vs) * copies + e) * N])); k++; }
            break;
        }
        default: break;   // NOOP
        }
#undef LIMBS4_DESC
#undef EMIT
    }
}
// Contribution of ONE gate at one point: filter(selector) * sum_k constraint_k * alpha_c^(k0 + k), added into acc[c].
// TYPE >= 0 compiles a single gate body (per-gate kernels: small register footprint, high occupancy); TYPE = -1
// keeps the run-time switch (monolithic fallback).
template <int NCH, int TYPE>
__device__ __forceinline__ void gate_contrib(const QArgs &a, const QProof &p, const DevGate &g, size_t N, size_t slot, u32 k0, u64 (&acc)[MAXCH]) {
    const u64 filter = gate_filter(a, g, N, slot);
    AccHL ga[MAXCH];
    _Pragma("unroll") for (int c = 0; c < NCH; c++) acc3_zero(ga[c]);
    gate_terms<NCH, TYPE, false>(a, p, g, N, slot, k0, ga);
    _Pragma("unroll") for (int c = 0; c < NCH; c++) acc[c] = add(acc[c], mul(filter, acc3_reduce(ga[c])));
}

// Gates that ride along with another launch (indices into the gate table)
// arith_ops != 0: gate arith_gi is an ArithmeticGate whose first arith_ops operations read only routed wires; k_quotient
// evaluates them from the wire values its permutation loop has in registers anyway (no second read of those planes).
struct LightArgs { u32 count; u32 gi[8]; u32 arith_gi, arith_ops; };
// The HBM-bound gate types (Constant, PublicInput, Arithmetic, BaseSum, RandomAccess), evaluated one after the other
template <int NCH>
__device__ __forceinline__ void light_gates(const QArgs &a, const QProof &p, const LightArgs &la, size_t N, size_t slot, u32 k0, u64 (&acc)[MAXCH]) {
    for (u32 t = 0; t < la.count; t++) {
        const DevGate g = a.gates[la.gi[t]];
        switch (g.type) {                      // uniform: every lane runs the same gate
        case GLP_GATE_CONSTANT: gate_contrib<NCH, GLP_GATE_CONSTANT>(a, p, g, N, slot, k0, acc); break;
        case GLP_GATE_PUBLIC_INPUT: gate_contrib<NCH, GLP_GATE_PUBLIC_INPUT>(a, p, g, N, slot, k0, acc); break;
        case GLP_GATE_ARITHMETIC: gate_contrib<NCH, GLP_GATE_ARITHMETIC>(a, p, g, N, slot, k0, acc); break;
        case GLP_GATE_BASE_SUM: gate_contrib<NCH, GLP_GATE_BASE_SUM>(a, p, g, N, slot, k0, acc); break;
        case GLP_GATE_RANDOM_ACCESS: gate_contrib<NCH, GLP_GATE_RANDOM_ACCESS>(a, p, g, N, slot, k0, acc); break;
        default: break;
        }
    }
}
// K6: vanishing polynomial / Z_H on the planes r = 0, step, 2 step, ... of the coset-major LDE domain.
//   terms: [L_0 (Z_c - 1)]_c, [prev*num - next*den]_{c,chunk}, gate constraints; res_c = sum_k term_k alpha_c^k
// GATES: 0 = permutation terms only; 1 = every gate (monolithic, run-time switch); 2 = the light gates of `la`: the
// permutation terms are VALU-bound and the light gates HBM-bound, so in one launch the waves in one phase fill the other
// phase's idle unit (separately: 3.3 + 2.75 ms at the headline size)
template <int NCH, int GATES>
__global__ __launch_bounds__(256, GATES == 1 ? 3 : 4) void k_quotient(QArgs a, QProof p0, QBatch qb, LightArgs la) {
    const QProof p = q_proof(p0, qb);
    const size_t n = (size_t)1 << a.lg, N = n << a.rb;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const u32 rq = blockIdx.y, r = rq * a.step;
    const size_t slot = (size_t)r * n + q, slot_next = (size_t)r * n + ((q + 1) & (n - 1));
    const u64 x = mul(a.shift_r[rq], dpow(a.w_n, q));
    constexpr u32 nch = NCH; const u32 nchunks = a.npp + 1, nt = a.nterms;
    u64 acc[MAXCH], zx[MAXCH], zg[MAXCH];
    Acc160 pa[MAXCH];
    _Pragma("unroll") for (int c = 0; c < NCH; c++) { acc_zero(pa[c]); zx[c] = p.zl[(size_t)c * N + slot]; zg[c] = p.zl[(size_t)c * N + slot_next]; }
    const u64 l0 = a.l0[(size_t)rq * n + q];
    _Pragma("unroll") for (int c = 0; c < NCH; c++) {
        const u64 t = mul(l0, sub(zx[c], 1));
        _Pragma("unroll") for (int c2 = 0; c2 < NCH; c2++) acc_fma(pa[c2], t, p.apow[c2 * nt + c]);
    }
    u64 bkx[MAXCH];                                    // beta_c k_j x for the next wire j (k_ratio path)
    _Pragma("unroll") for (int c = 0; c < NCH; c++) bkx[c] = mul_nc(p.betas[c], x);
    const u32 k0 = nch + nch * nchunks;
    // ArithmeticGate riding on the permutation loop's wire loads (GATES == 2 only)
    const u32 ar_ops = GATES == 2 ? la.arith_ops : 0;
    AccHL gar[MAXCH];
    u64 ar_c0 = 0, ar_c1 = 0;
    if (GATES == 2) {
        _Pragma("unroll") for (int c = 0; c < NCH; c++) acc3_zero(gar[c]);
        if (ar_ops) { ar_c0 = a.cs[(size_t)a.nsel * N + slot]; ar_c1 = a.cs[(size_t)(a.nsel + 1) * N + slot]; }
    }
    for (u32 chunk = 0; chunk < nchunks; chunk++) {
        u64 num[MAXCH], den[MAXCH];
        _Pragma("unroll") for (int c = 0; c < NCH; c++) { num[c] = 1; den[c] = 1; }
        const u32 j0 = chunk * a.qdf, j1 = min((chunk + 1) * a.qdf, a.nr);
        for (u32 jb = j0; jb < j1; jb += 8) {          // eight wire + eight sigma loads in flight
            u64 w8[8], s8[8];
#pragma unroll
            for (int t = 0; t < 8; t++)
                if (jb + t < j1) { w8[t] = p.wl[(size_t)(jb + t) * N + slot]; s8[t] = a.cs[(size_t)(a.nc + jb + t) * N + slot]; }
            if (GATES == 2 && ar_ops) {                // jb is a multiple of 4 here (the host checks qdf % 4 == 0)
#pragma unroll
                for (int t = 0; t < 8; t += 4)
                    if (jb + t + 3 < j1 && (jb + t) / 4 < ar_ops) {
                        const u32 i = (jb + t) / 4;
                        const u64 v = sub(w8[t + 3], add(mul(mul(w8[t], w8[t + 1]), ar_c0), mul(w8[t + 2], ar_c1)));
                        _Pragma("unroll") for (int c2 = 0; c2 < NCH; c2++) acc3_fma(gar[c2], v, p.apl + APL_WORDS * ((size_t)c2 * nt + k0 + i));
                    }
            }
#pragma unroll
            for (int t = 0; t < 8; t++)
                if (jb + t < j1) {
                    // lazy chain: the running products and the beta terms stay non-canonical u64 (mul_nc takes any
                    // u64); only w + gamma is a canonical addition, shared by numerator and denominator
                    // beta k_j x: with k_j = g^j (g < 2^32, how plonky2 picks the coset shifts) it is the previous
                    // wire's value times g -- two multiply-adds and a fold instead of two full multiplications
                    u64 kx = 0;
                    if (!a.k_ratio) kx = mul_nc(a.k_is[jb + t], x);
                    _Pragma("unroll") for (int c = 0; c < NCH; c++) {
                        const u64 wg = add(w8[t], p.gammas[c]);
                        const u64 bk = a.k_ratio ? bkx[c] : mul_nc(p.betas[c], kx);
                        num[c] = mul_nc_cc(num[c], add_cnc(wg, bk));
                        den[c] = mul_nc_cc(den[c], add_cnc(wg, mul_nc_cc(p.betas[c], s8[t])));
                        if (a.k_ratio) bkx[c] = mul_small_nc(bkx[c], a.k_ratio);
                    }
                }
        }
        _Pragma("unroll") for (int c = 0; c < NCH; c++) {
            const u64 prev = chunk == 0 ? zx[c] : p.zl[(size_t)(nch + c * a.npp + chunk - 1) * N + slot];
            const u64 next = chunk == nchunks - 1 ? zg[c] : p.zl[(size_t)(nch + c * a.npp + chunk) * N + slot];
            const u64 t = sub(mul(prev, num[c]), mul(next, den[c]));
            const u32 k = nch + c * nchunks + chunk;
            _Pragma("unroll") for (int c2 = 0; c2 < NCH; c2++) acc_fma(pa[c2], t, p.apow[c2 * nt + k]);
        }
    }
    _Pragma("unroll") for (int c = 0; c < NCH; c++) acc[c] = acc_reduce(pa[c]);
    if (GATES == 2 && ar_ops) {
        const u64 filter = gate_filter(a, a.gates[la.arith_gi], N, slot);
        _Pragma("unroll") for (int c = 0; c < NCH; c++) acc[c] = add(acc[c], mul(filter, acc3_reduce(gar[c])));
    }
    if constexpr (GATES == 1) {                        // monolithic: every gate here
        for (u32 gi = 0; gi < a.num_gates; gi++) {
            const DevGate g = a.gates[gi];
            gate_contrib<NCH, -1>(a, p, g, N, slot, k0, acc);
        }
    }
    if constexpr (GATES == 2) light_gates<NCH>(a, p, la, N, slot, k0, acc);
    const size_t Rq = (size_t)gridDim.y;
    _Pragma("unroll") for (int c = 0; c < NCH; c++) p.out[((size_t)c * Rq + rq) * n + q] = mul(acc[c], a.zh_inv[rq]);
}

// One gate type per launch (gate_mode = 1): out[c][plane][q] += zh_inv * filter * sum_k constraint_k alpha_c^(k0 + k)
template <int NCH, int TYPE>
__global__ __launch_bounds__(256) void k_quotient_gate(QArgs a, QProof p0, QBatch qb, u32 gi) {
    const QProof p = q_proof(p0, qb);
    const size_t n = (size_t)1 << a.lg, N = n << a.rb;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const u32 rq = blockIdx.y, r = rq * a.step;
    const size_t slot = (size_t)r * n + q;
    u64 acc[MAXCH];
    _Pragma("unroll") for (int c = 0; c < NCH; c++) acc[c] = 0;
    const DevGate g = a.gates[gi];
    gate_contrib<NCH, TYPE>(a, p, g, N, slot, (u32)NCH + (u32)NCH * (a.npp + 1), acc);
    const size_t Rq = (size_t)gridDim.y;
    _Pragma("unroll") for (int c = 0; c < NCH; c++) {
        u64 *o = p.out + ((size_t)c * Rq + rq) * n + q;
        *o = add(*o, mul(acc[c], a.zh_inv[rq]));
    }
}

// The base-4 limb gates of plonky2_u32 (U32Arithmetic, U32AddMany, U32Subtraction, U32RangeCheck) in ONE launch.  Their
// limb columns overlap almost completely (wires 30..113 are limbs of all four in the secp256k1 circuit): every wire plane
// is read once, range_product(w_j, 4) is computed once per column and multiplied into each gate's own carry-free
// accumulators (the selector filters are applied after the reduction, as in the per-gate kernels).  The per-column work
// is driven by a table built at circuit creation (uniform control flow, scalar loads):
//   desc[j][s] for wire column j and fused gate slot s:
//     bit 0        the column is a base-4 limb of this gate
//     bits 1..4    position of the limb in its base-4 sum (weight 4^pos)
//     bit 5        this limb closes the sum: emit  (sum - W[ref])  at alpha index kf, then reset the sum
//     bits 6..15   alpha index of the limb's range-check constraint
//     bits 16..25  kf        bits 26..33  ref (wire column the sum must equal)
// The constraints that are not limb work (two per U32Arithmetic op, one per AddMany op, two per Subtraction op) come from
// gate_terms<.., HEAD_ONLY = true>.
// `extra`: HBM-bound gates without limb work of their own kind (ComparisonGate) evaluated in the same launch, for the same
// reason as the light gates in k_quotient: their loads overlap the limb gates' arithmetic.
// More limb gates than slots (the real secp256k1 circuit has ten: U32Arithmetic, seven U32AddMany parameter sets, U32RangeCheck,
// U32Subtraction) go through the same launch in GROUPS of LIMB_SLOTS (five: 123 VGPRs, four waves per SIMD; six cost a wave and
// measured slower): the accumulators are reused; the wire planes are read again per group (PMC: from HBM, the last-level cache does
// not hold them in between).
constexpr int LIMB_SLOTS = 5, LIMB_GROUPS = 4;
struct LimbArgs { const u64 *desc; u32 groups, num_wires; u32 count[LIMB_GROUPS], jlo[LIMB_GROUPS], jhi[LIMB_GROUPS]; u32 gi[LIMB_GROUPS][LIMB_SLOTS]; u32 extra_count, extra_gi[4]; };
inline void limb_args(const glp_circuit *cc, LimbArgs &la) {
    la.desc = cc->dev_limb_desc; la.groups = cc->limb_groups; la.num_wires = cc->d.num_wires;
    for (int g = 0; g < LIMB_GROUPS; g++) {
        la.count[g] = cc->limb_gcount[g]; la.jlo[g] = cc->limb_jlo[g]; la.jhi[g] = cc->limb_jhi[g];
        for (int i = 0; i < LIMB_SLOTS; i++) la.gi[g][i] = cc->limb_gi[g * LIMB_SLOTS + i];
    }
}
template <int NCH>
__global__ __launch_bounds__(256, 2) void k_quotient_limbs(QArgs a, QProof p0, QBatch qb, LimbArgs la) {
    const QProof p = q_proof(p0, qb);
    const size_t n = (size_t)1 << a.lg, N = n << a.rb;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const u32 rq = blockIdx.y, r = rq * a.step;
    const size_t slot = (size_t)r * n + q;
    const u32 k0 = (u32)NCH + (u32)NCH * (a.npp + 1), nt = a.nterms;
    const u64 *W = p.wl + slot;
    const u64 *ap = p.apl + APL_WORDS * (size_t)k0;
    u64 acc[MAXCH];
    _Pragma("unroll") for (int c = 0; c < NCH; c++) acc[c] = 0;
#define LIMB_EMIT(S, K, V)                                                                                     \
    do {                                                                                                       \
        const u64 _v = (V);                                                                                    \
        _Pragma("unroll") for (int c2 = 0; c2 < NCH; c2++) acc3_fma(ga[S][c2], _v, ap + APL_WORDS * ((size_t)c2 * nt + (K)));   \
    } while (0)
#pragma unroll 1
    for (u32 grp = 0; grp < la.groups; grp++) {
        const u32 gcount = la.count[grp], jlo = la.jlo[grp], jhi = la.jhi[grp];
        const u64 *desc = la.desc + (size_t)grp * la.num_wires * LIMB_SLOTS;
        AccHL ga[LIMB_SLOTS][MAXCH];
        Base4Sum bs[LIMB_SLOTS];
        _Pragma("unroll") for (int s = 0; s < LIMB_SLOTS; s++) {
            b4_zero(bs[s]);
            _Pragma("unroll") for (int c = 0; c < NCH; c++) acc3_zero(ga[s][c]);
        }
        // heads
        _Pragma("unroll") for (int s = 0; s < LIMB_SLOTS; s++) {
            if ((u32)s < gcount) {
                const DevGate g = a.gates[la.gi[grp][s]];
                switch (g.type) {
                case GLP_GATE_U32_ARITHMETIC: gate_terms<NCH, GLP_GATE_U32_ARITHMETIC, true>(a, p, g, N, slot, k0, ga[s]); break;
                case GLP_GATE_U32_ADD_MANY: gate_terms<NCH, GLP_GATE_U32_ADD_MANY, true>(a, p, g, N, slot, k0, ga[s]); break;
                case GLP_GATE_U32_SUBTRACTION: gate_terms<NCH, GLP_GATE_U32_SUBTRACTION, true>(a, p, g, N, slot, k0, ga[s]); break;
                default: break;                    // U32RangeCheck: limb work only
                }
            }
        }
        for (u32 j0 = jlo; j0 <= jhi; j0 += 8) {
            u64 lv[8];
            _Pragma("unroll") for (int t = 0; t < 8; t++) if (j0 + t <= jhi) lv[t] = W[(size_t)(j0 + t) * N];
            _Pragma("unroll") for (int t = 0; t < 8; t++) if (j0 + t <= jhi) {
                const u64 v = lv[t];
                const u64 *dj = desc + (size_t)(j0 + t) * LIMB_SLOTS;
                const u64 rp = range_product(v, 4);
                _Pragma("unroll") for (int s = 0; s < LIMB_SLOTS; s++) {
                    const u64 d = dj[s];
                    if (d & 1) {
                        const u32 kl = (u32)(d >> 6) & 0x3FFu;
                        _Pragma("unroll") for (int c2 = 0; c2 < NCH; c2++) acc3_fma(ga[s][c2], rp, ap + APL_WORDS * ((size_t)c2 * nt + kl));
                        b4_add(bs[s], v, (u32)(d >> 1) & 15u);
                        if (d & 32) {
                            const u32 kf = (u32)(d >> 16) & 0x3FFu, ref = (u32)(d >> 26) & 0xFFu;
                            LIMB_EMIT(s, kf, sub(b4_value(bs[s]), W[(size_t)ref * N]));
                            b4_zero(bs[s]);
                        }
                    }
                }
            }
        }
        _Pragma("unroll") for (int s = 0; s < LIMB_SLOTS; s++) {
            if ((u32)s < gcount) {
                const u64 filter = gate_filter(a, a.gates[la.gi[grp][s]], N, slot);
                _Pragma("unroll") for (int c = 0; c < NCH; c++) acc[c] = add(acc[c], mul(filter, acc3_reduce(ga[s][c])));
            }
        }
    }
#undef LIMB_EMIT
    for (u32 t = 0; t < la.extra_count; t++) {             // after the limb accumulators are dead (register budget)
        const DevGate g = a.gates[la.extra_gi[t]];
        if (g.type == GLP_GATE_COMPARISON) gate_contrib<NCH, GLP_GATE_COMPARISON>(a, p, g, N, slot, k0, acc);
    }
    const size_t Rq = (size_t)gridDim.y;
    _Pragma("unroll") for (int c = 0; c < NCH; c++) {
        u64 *o = p.out + ((size_t)c * Rq + rq) * n + q;
        *o = add(*o, mul(acc[c], a.zh_inv[rq]));
    }
}

// K6b: after the per-plane inverse NTT: undo the plane twist, inverse DFT across planes, undo the coset shift.
//   V [nch][Rq][n] (bit-reversed k')  ->  chunk coefficients [nch*Rq][n] (bit-reversed), chunk c = X^(c n) block
struct QCArgs { const u64 *V; u64 *out; u64 wM_inv, wR_inv, g_inv, rq_inv; u64 gn_inv_pow[MAXR]; u32 lg, Rq; };
__global__ __launch_bounds__(256) void k_quotient_combine(QCArgs a) {
    const size_t n = (size_t)1 << a.lg;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const u32 ch = blockIdx.y, Rq = a.Rq;
    const u32 kp = bitrev32((u32)p, a.lg);
    const u64 tw = dpow(a.wM_inv, kp);            // w_M^-k'
    const u64 gk = mul(dpow(a.g_inv, kp), a.rq_inv);
    u64 y[MAXR];
    u64 t = 1;
    for (u32 r = 0; r < Rq; r++) { y[r] = mul(a.V[((size_t)ch * Rq + r) * n + p], t); t = mul(t, tw); }
    u64 wc = 1;                                    // w_Rq^-c
    for (u32 c = 0; c < Rq; c++) {
        u64 s = 0, w = 1;
        for (u32 r = 0; r < Rq; r++) { s = add(s, mul(y[r], w)); w = mul(w, wc); }
        a.out[((size_t)ch * Rq + c) * n + p] = mul(s, mul(gk, a.gn_inv_pow[c]));
        wc = mul(wc, a.wR_inv);
    }
}

// zt[p] = z^bitrev(p)  (extension), from z^(2^b), b < lg
// batch (zeta_b != nullptr, blockIdx.y = proof): the point comes from zeta_b[proof][2] and its squarings are made here
struct ZTArgs { u64 *zt; ext2 zp2[24]; u32 lg; const u64 *zeta_b; size_t zeta_stride; };
template <bool BATCH>
__global__ __launch_bounds__(256) void k_zeta_table(ZTArgs a) {
    const size_t n = (size_t)1 << a.lg;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const u32 k = bitrev32((u32)p, a.lg);
    ext2 acc = e_from(1);
    if constexpr (BATCH) {
        const u64 *z = a.zeta_b + (size_t)blockIdx.y * a.zeta_stride;
        ext2 sq = e_make(z[0], z[1]);
        for (u32 b = 0; b < a.lg; b++) { if ((k >> b) & 1) acc = e_mul(acc, sq); sq = e_sqr(sq); }
        a.zt += (size_t)blockIdx.y * 2 * n;
    } else {
        for (u32 b = 0; b < a.lg; b++) if ((k >> b) & 1) acc = e_mul(acc, a.zp2[b]);
    }
    a.zt[2 * p] = acc.a; a.zt[2 * p + 1] = acc.b;
}
// K7: partial sums of  sum_p coeffs[col][p] * zt[p]   grid = (OPEN_BLOCKS, ncols)
constexpr int OPEN_BLOCKS = 32;      // at most; open_blocks(n) picks fewer for short polynomials (the stride is gridDim.x)
inline u32 open_blocks(size_t n) { return (u32)std::max<size_t>(1, std::min<size_t>(OPEN_BLOCKS, n / 256)); }
// blockIdx.z = proof of a batch: coefficients / table / partial sums advance by the given strides (0 = shared by all proofs)
__global__ __launch_bounds__(256) void k_open_dot(const u64 *coeffs, const u64 *zt, u64 *partial, u32 lg, size_t coeffs_bstride,
                                                  size_t zt_bstride, size_t partial_bstride) {
    __shared__ u64 sa[256], sb[256];
    coeffs += (size_t)blockIdx.z * coeffs_bstride; zt += (size_t)blockIdx.z * zt_bstride; partial += (size_t)blockIdx.z * partial_bstride;
    const size_t n = (size_t)1 << lg;
    const u32 col = blockIdx.y, t = threadIdx.x;
    // n / (OPEN_BLOCKS * 256) <= 2^11 terms per thread, flushed every ACC_MAX_TERMS: carry-free limb accumulators
    // (one reduction per flush instead of a modular multiply-add per coefficient)
    u64 a = 0, b = 0;
    AccLimb xa, xb;
    acc2_zero(xa); acc2_zero(xb);
    u32 terms = 0;
    for (size_t p = (size_t)blockIdx.x * 256 + t; p < n; p += (size_t)gridDim.x * 256) {
        const u64 c = coeffs[(size_t)col * n + p];
        const u32 c0 = (u32)c & 0x3FFFFFu, c1 = (u32)(c >> 22) & 0x3FFFFFu, c2 = (u32)(c >> 44);
        acc2_fma(xa, c0, c1, c2, zt[2 * p]);
        acc2_fma(xb, c0, c1, c2, zt[2 * p + 1]);
        if (++terms == ACC_MAX_TERMS) {
            a = add(a, acc2_reduce(xa)); b = add(b, acc2_reduce(xb));
            acc2_zero(xa); acc2_zero(xb); terms = 0;
        }
    }
    a = add(a, acc2_reduce(xa)); b = add(b, acc2_reduce(xb));
    sa[t] = a; sb[t] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) { sa[t] = add(sa[t], sa[t + s]); sb[t] = add(sb[t], sb[t + s]); } __syncthreads(); }
    if (t == 0) { partial[2 * ((size_t)col * gridDim.x + blockIdx.x)] = sa[0]; partial[2 * ((size_t)col * gridDim.x + blockIdx.x) + 1] = sb[0]; }
}

// K8: values of the FRI batch polynomial on the coset plane 0 (x_q = g w_n^q):
//   F(x) = alpha^nch * (sum_j alpha^j f_j(x) - red0)/(x - zeta) + (sum_{j<nch} alpha^j Z_j(x) - red1)/(x - g zeta)
struct FVArgs {
    const u64 *lde[4]; u32 ncols[4];
    const u64 *apow;            // ext alpha^j, j < total columns
    u64 *out;                   // [2][n]
    ext2 red0, red1, zeta, zeta_next, shift_acc;   // shift_acc = alpha^nch
    u64 w_n, g;
    u32 lg, rb, nch;
    // many-proofs batch (blockIdx.y = proof): pp[proof][10] = red0, red1, zeta, zeta_next, shift_acc; strides per proof
    const u64 *pp;
    size_t lde_stride[4], apow_stride, out_stride;
};
__global__ __launch_bounds__(256) void k_final_values(FVArgs a) {
    const size_t n = (size_t)1 << a.lg, N = n << a.rb;
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    if (a.pp) {
        const size_t pk = blockIdx.y;
        const u64 *v = a.pp + pk * 10;
        a.red0 = e_make(v[0], v[1]); a.red1 = e_make(v[2], v[3]); a.zeta = e_make(v[4], v[5]); a.zeta_next = e_make(v[6], v[7]);
        a.shift_acc = e_make(v[8], v[9]);
        _Pragma("unroll") for (int k = 0; k < 4; k++) a.lde[k] += pk * a.lde_stride[k];
        a.apow += pk * a.apow_stride; a.out += pk * a.out_stride;
    }
    // sum_j alpha^j f_j(x): the base-field value is cut into 22-bit limbs once and multiplied into carry-free
    // accumulators for the two extension coordinates (flushed every ACC_MAX_TERMS columns)
    ext2 acc0 = e_from(0), acc1 = e_from(0);
    AccLimb xa, xb;
    acc2_zero(xa); acc2_zero(xb);
    u32 j = 0, terms = 0;
    for (int k = 0; k < 4; k++) {
        const u64 *l = a.lde[k] + q;
        for (u32 c = 0; c < a.ncols[k]; c++, j++) {
            const u64 v = l[(size_t)c * N];
            const u32 v0 = (u32)v & 0x3FFFFFu, v1 = (u32)(v >> 22) & 0x3FFFFFu, v2 = (u32)(v >> 44);
            acc2_fma(xa, v0, v1, v2, a.apow[2 * j]);
            acc2_fma(xb, v0, v1, v2, a.apow[2 * j + 1]);
            if (++terms == ACC_MAX_TERMS) {
                acc0 = e_add(acc0, e_make(acc2_reduce(xa), acc2_reduce(xb)));
                acc2_zero(xa); acc2_zero(xb); terms = 0;
            }
            if (k == 2 && c < a.nch) {
                const ext2 ap1 = e_make(a.apow[2 * c], a.apow[2 * c + 1]);
                acc1 = e_add(acc1, e_scale(ap1, v));
            }
        }
    }
    acc0 = e_add(acc0, e_make(acc2_reduce(xa), acc2_reduce(xb)));
    const u64 x = mul(a.g, dpow(a.w_n, q));
    const ext2 d0 = e_inv(e_sub(e_from(x), a.zeta)), d1 = e_inv(e_sub(e_from(x), a.zeta_next));
    ext2 f = e_mul(e_mul(e_sub(acc0, a.red0), d0), a.shift_acc);
    f = e_add(f, e_mul(e_sub(acc1, a.red1), d1));
    a.out[q] = f.a; a.out[n + q] = f.b;
}

// The same for at most 128 points per proof (a batch of small proofs: blockIdx.y = proof, one workgroup per proof): 256 / n lanes share a point, each
// takes every (256 / n)-th column of the four oracles, an xor-butterfly adds the partial sums up, and lanes 0 and 1 of the group invert the two
// denominators side by side.  k_final_values walks ~250 columns and two extension inversions per lane with 8 lanes live: 105 us per 256 zkdsa proofs.
__global__ __launch_bounds__(256) void k_final_values_small(FVArgs a) {
    const u32 n = 1u << a.lg, lpp = 256u >> a.lg;            // lanes per point: 2 .. 64
    const size_t N = (size_t)n << a.rb;
    const u32 q = threadIdx.x / lpp, t = threadIdx.x % lpp;
    if (a.pp) {
        const size_t pk = blockIdx.y;
        const u64 *v = a.pp + pk * 10;
        a.red0 = e_make(v[0], v[1]); a.red1 = e_make(v[2], v[3]); a.zeta = e_make(v[4], v[5]); a.zeta_next = e_make(v[6], v[7]);
        a.shift_acc = e_make(v[8], v[9]);
        _Pragma("unroll") for (int k = 0; k < 4; k++) a.lde[k] += pk * a.lde_stride[k];
        a.apow += pk * a.apow_stride; a.out += pk * a.out_stride;
    }
    ext2 acc0 = e_from(0), acc1 = e_from(0);
    AccLimb xa, xb;
    acc2_zero(xa); acc2_zero(xb);
    u32 base = 0;
    for (int k = 0; k < 4; k++) {
        const u64 *l = a.lde[k] + q;
        for (u32 c = t; c < a.ncols[k]; c += lpp) {          // fewer than ACC_MAX_TERMS terms per lane: no flush
            const u32 j = base + c;
            const u64 v = l[(size_t)c * N];
            const u32 v0 = (u32)v & 0x3FFFFFu, v1 = (u32)(v >> 22) & 0x3FFFFFu, v2 = (u32)(v >> 44);
            acc2_fma(xa, v0, v1, v2, a.apow[2 * j]);
            acc2_fma(xb, v0, v1, v2, a.apow[2 * j + 1]);
            if (k == 2 && c < a.nch) acc1 = e_add(acc1, e_scale(e_make(a.apow[2 * c], a.apow[2 * c + 1]), v));
        }
        base += a.ncols[k];
    }
    acc0 = e_make(acc2_reduce(xa), acc2_reduce(xb));
    for (u32 m = lpp >> 1; m >= 1; m >>= 1) {                // lpp <= 64 here (n >= 4): the group lies inside one wavefront
        acc0 = e_add(acc0, e_make(pos::shfl_xor64(acc0.a, (int)m), pos::shfl_xor64(acc0.b, (int)m)));
        acc1 = e_add(acc1, e_make(pos::shfl_xor64(acc1.a, (int)m), pos::shfl_xor64(acc1.b, (int)m)));
    }
    const u64 x = mul(a.g, dpow(a.w_n, q));
    const ext2 dmine = e_inv(e_sub(e_from(x), t == 1 ? a.zeta_next : a.zeta));       // lane 0: 1 / (x - zeta), lane 1: 1 / (x - zeta_next)
    const int lane1 = (int)((threadIdx.x & 63u) - t + 1);
    const ext2 d1 = e_make(pos::shfl64(dmine.a, lane1), pos::shfl64(dmine.b, lane1));
    if (t == 0) {
        ext2 f = e_mul(e_mul(e_sub(acc0, a.red0), dmine), a.shift_acc);
        f = e_add(f, e_mul(e_sub(acc1, a.red1), d1));
        a.out[q] = f.a; a.out[n + q] = f.b;
    }
}
// data[c][p] *= base^bitrev(p)
__global__ __launch_bounds__(256) void k_scale_bitrev_pow(u64 *data, u64 base, u32 lg) {
    const size_t n = (size_t)1 << lg;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const u64 f = dpow(base, bitrev32((u32)p, lg));
    data[(size_t)blockIdx.y * n + p] = mul(data[(size_t)blockIdx.y * n + p], f);
}

// K9a: FRI commit-phase leaves.  vals = coset-major LDE [2][R][ncur] of the current polynomial (L = R*ncur
// points); leaf m = the `arity` extension values at natural indices bitrev_L(m*arity + t).  Lane = M' = bitrev(m).
__global__ __launch_bounds__(256, 4) void k_fri_leaf_hash(const u64 *vals, u64 *digests, u32 lgL, u32 rb, u32 ab, size_t vals_bstride,
                                                       size_t dig_bstride) {
    vals += (size_t)blockIdx.y * vals_bstride; digests += (size_t)blockIdx.y * dig_bstride;     // blockIdx.y = proof of a batch
    const size_t L = (size_t)1 << lgL, ncur = L >> rb, nleaves = L >> ab;
    const size_t Mp = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (Mp >= nleaves) return;
    const size_t m = bitrev32((u32)Mp, lgL - ab);
    const u32 arity = 1u << ab;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    const u64 *re = vals, *im = vals + L;
    u32 fill = 0;
    for (u32 t = 0; t < arity; t++) {
        const size_t i = (size_t)bitrev32(t, ab) * nleaves + Mp;
        const size_t pos = (i & (((size_t)1 << rb) - 1)) * ncur + (i >> rb);
        s[fill++] = re[pos];
        s[fill++] = im[pos];
        if (fill == 8) { if (2 * arity > 4) pos::permute(s); fill = 0; }
    }
    if (fill && 2 * arity > 4) pos::permute(s);
    ulonglong2 d0, d1;
    d0.x = s[0]; d0.y = s[1]; d1.x = s[2]; d1.y = s[3];
    reinterpret_cast<ulonglong2 *>(digests + 4 * m)[0] = d0;
    reinterpret_cast<ulonglong2 *>(digests + 4 * m)[1] = d1;
}
// K9a, latency form for small layers: one leaf per 16-lane group, sponge state on 12 lanes.
__global__ __launch_bounds__(256) void k_fri_leaf_hash_coop(const u64 *vals, u64 *digests, u32 lgL, u32 rb, u32 ab, size_t vals_bstride,
                                                            size_t dig_bstride) {
    vals += (size_t)blockIdx.y * vals_bstride; digests += (size_t)blockIdx.y * dig_bstride;
    const size_t L = (size_t)1 << lgL, ncur = L >> rb, nleaves = L >> ab;
    const int tid = threadIdx.x, l = tid & 15, lane = tid & 63, group_base = lane & ~15;
    const size_t Mp0 = (size_t)blockIdx.x * 16 + (tid >> 4);
    const bool live = Mp0 < nleaves;
    const size_t Mp = live ? Mp0 : 0;
    const size_t m = bitrev32((u32)Mp, lgL - ab);
    const u32 len = 2u << ab;                       // base-field elements per leaf
    u64 x = 0;
    for (u32 c = 0; c < len; c += 8) {
        if (l < 8 && c + l < len) {
            const u32 e = c + l, t = e >> 1;
            const size_t i = (size_t)bitrev32(t, ab) * nleaves + Mp;
            const size_t pos = (i & (((size_t)1 << rb) - 1)) * ncur + (i >> rb);
            x = vals[(e & 1 ? L : 0) + pos];
        }
        if (len > 4) x = pos::permute_coop(x, l, group_base);
    }
    if (live && l < 4) digests[4 * m + l] = x;
}
// K9a with KeccakHash<25>: hash_or_noop of the leaf's 2^(ab+1) elements (arity 2 already exceeds the 3 elements that are copied)
// one leaf per quad of lanes (pos::permute_quad): layers of 2^12..2^15 leaves, as for the initial trees (merkle.hip)
__global__ __launch_bounds__(256) void k_fri_leaf_hash_quad(const u64 *vals, u64 *digests, u32 lgL, u32 rb, u32 ab, size_t vals_bstride,
                                                            size_t dig_bstride) {
    vals += (size_t)blockIdx.y * vals_bstride; digests += (size_t)blockIdx.y * dig_bstride;
    const size_t L = (size_t)1 << lgL, ncur = L >> rb, nleaves = L >> ab;
    const int tid = threadIdx.x, q = tid & 3;
    const size_t Mp0 = (size_t)blockIdx.x * 64 + (tid >> 2);
    const bool live = Mp0 < nleaves;
    const size_t Mp = live ? Mp0 : 0;
    const size_t m = bitrev32((u32)Mp, lgL - ab);
    const u32 len = 2u << ab;                       // base-field elements per leaf
    u64 x[3] = {0, 0, 0};
    for (u32 c = 0; c < len; c += 8) {
#pragma unroll
        for (int s = 0; s < 3; s++) {
            const u32 e8 = 3 * q + s, e = c + e8;
            if (e8 < 8 && e < len) {
                const u32 t = e >> 1;
                const size_t i = (size_t)bitrev32(t, ab) * nleaves + Mp;
                const size_t pos = (i & (((size_t)1 << rb) - 1)) * ncur + (i >> rb);
                x[s] = vals[(e & 1 ? L : 0) + pos];
            }
        }
        if (len > 4) pos::permute_quad(x, q);
    }
    if (live) {
        if (q == 0) { digests[4 * m] = x[0]; digests[4 * m + 1] = x[1]; digests[4 * m + 2] = x[2]; }
        if (q == 1) digests[4 * m + 3] = x[0];
    }
}
__global__ __launch_bounds__(256) void k_fri_leaf_hash_keccak(const u64 *vals, u64 *digests, u32 lgL, u32 rb, u32 ab, size_t vals_bstride,
                                                              size_t dig_bstride) {
    vals += (size_t)blockIdx.y * vals_bstride; digests += (size_t)blockIdx.y * dig_bstride;
    const size_t L = (size_t)1 << lgL, ncur = L >> rb, nleaves = L >> ab;
    const size_t Mp = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (Mp >= nleaves) return;
    const size_t m = bitrev32((u32)Mp, lgL - ab);
    const u32 arity = 1u << ab;
    kec::Sponge s;
    kec::sponge_init(s);
    for (u32 t = 0; t < arity; t++) {
        const size_t i = (size_t)bitrev32(t, ab) * nleaves + Mp;
        const size_t pos = (i & (((size_t)1 << rb) - 1)) * ncur + (i >> rb);
        kec::sponge_absorb(s, vals[pos]);
        kec::sponge_absorb(s, vals[L + pos]);
    }
    kec::sponge_finish(s);
    u64 d[4];
    kec::sponge_digest25(s, d);
    ulonglong2 d0, d1;
    d0.x = d[0]; d0.y = d[1]; d1.x = d[2]; d1.y = d[3];
    reinterpret_cast<ulonglong2 *>(digests + 4 * m)[0] = d0;
    reinterpret_cast<ulonglong2 *>(digests + 4 * m)[1] = d1;
}
// K9b: fold coefficients (bit-reversed layout): new[p'] = sum_t beta^t old[bitrev(t) * nnew + p']
// batch (beta_b != nullptr, blockIdx.y = proof): beta from beta_b[proof][2]; coefficient arrays [proof][2][n]
__global__ __launch_bounds__(256) void k_fri_fold(const u64 *oldc, u64 *newc, ext2 beta, u32 lg_old, u32 ab, const u64 *beta_b) {
    const size_t nold = (size_t)1 << lg_old, nnew = nold >> ab;
    if (beta_b) {
        beta = e_make(beta_b[2 * blockIdx.y], beta_b[2 * blockIdx.y + 1]);
        oldc += (size_t)blockIdx.y * 2 * nold; newc += (size_t)blockIdx.y * 2 * nnew;
    }
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= nnew) return;
    ext2 acc = e_from(0);
    for (u32 t = 1u << ab; t-- > 0;) {
        const size_t o = (size_t)bitrev32(t, ab) * nnew + p;
        acc = e_add(e_mul(acc, beta), e_make(oldc[o], oldc[nold + o]));
    }
    newc[p] = acc.a; newc[nnew + p] = acc.b;
}
// leaf evals for the query phase: out[k][2*t..] = the arity values of leaf idx[k]
__global__ void k_fri_gather_leaf(const u64 *vals, u32 lgL, u32 rb, u32 ab, const u64 *idx, u32 idx_shift, u32 count, u64 *out,
                                  size_t out_stride, size_t vals_bstride, size_t out_bstride) {
    vals += (size_t)blockIdx.y * vals_bstride; idx += (size_t)blockIdx.y * count; out += (size_t)blockIdx.y * out_bstride;
    const size_t L = (size_t)1 << lgL, ncur = L >> rb, nleaves = L >> ab;
    const u32 arity = 1u << ab;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)count * arity) return;
    const u32 k = (u32)(gid / arity), t = (u32)(gid % arity);
    const size_t m = idx[k] >> idx_shift;
    const size_t Mp = bitrev32((u32)m, lgL - ab);
    const size_t i = (size_t)bitrev32(t, ab) * nleaves + Mp;
    const size_t pos = (i & (((size_t)1 << rb) - 1)) * ncur + (i >> rb);
    out[(size_t)k * out_stride + 2 * t] = vals[pos];
    out[(size_t)k * out_stride + 2 * t + 1] = vals[L + pos];
}

// K10: proof-of-work grinding; smallest candidate in [base, base + count) whose response has `bits` leading zeros
struct PowArgs { u64 st[12]; u32 pos, bits; u64 base; unsigned long long *best; };
template <int HASHER>
__global__ __launch_bounds__(256) void k_pow(PowArgs a) {
    const u64 cand = a.base + (u64)blockIdx.x * 256 + threadIdx.x;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = a.st[i];
#pragma unroll
    for (int i = 0; i < 8; i++) if ((u32)i == a.pos) s[i] = cand;        // candidates stay far below p
    if constexpr (HASHER == GLP_HASH_KECCAK25) kec::permute(s); else pos::permute(s);
    if (a.bits == 0 || (s[7] >> (64 - a.bits)) == 0) atomicMin(a.best, (unsigned long long)cand);
}

// K10 for a batch of K proofs, each with its own sponge state st_b[proof][12] and input position pos_b[proof].  Workgroups
// are persistent: a workgroup takes the next 256 candidates of a proof from that proof's counter (next[proof], handed out
// in increasing order), tests them, and records the smallest hit in best[proof]; it leaves a proof once a hit below its next
// chunk is known and moves on to the next unfinished proof, so the long tail of one unlucky search is shared by the whole
// GPU instead of idling it.  Every chunk below the final best[proof] was handed out and completed before the kernel ends,
// hence the result is the smallest witness regardless of scheduling.  Termination: a proof is finished once best <= next
// (a witness exists below 2^40 with overwhelming probability; the hand-out stops there in any case), and a workgroup exits
// after one full pass over the proofs finds none unfinished.
template <int HASHER>
__global__ __launch_bounds__(256) void k_pow_batch(const u64 *st_b, const u32 *pos_b, u32 bits, unsigned long long *best,
                                                   unsigned long long *next, u32 K) {
    __shared__ unsigned long long sh_base;
    u32 pk = blockIdx.x % K, idle = 0;
    while (idle < K) {
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long b = ~0ull;
            const unsigned long long cur = *(volatile unsigned long long *)(best + pk);
            if (*(volatile unsigned long long *)(next + pk) < cur) {
                b = atomicAdd(next + pk, 256ull);
                if (b >= cur || b >= (1ull << 40)) b = ~0ull;       // nothing below the known witness (or the cap) is left
            }
            sh_base = b;
        }
        __syncthreads();
        const unsigned long long base = sh_base;
        if (base == ~0ull) { pk = pk + 1 == K ? 0 : pk + 1; idle++; continue; }
        idle = 0;
        u64 s[12];
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = st_b[(size_t)pk * 12 + i];
        const u32 pos = pos_b[pk];
        const u64 cand = base + threadIdx.x;
        for (u32 i = 0; i < 8; i++) if (i == pos) s[i] = cand;      // candidates stay far below p
        if constexpr (HASHER == GLP_HASH_KECCAK25) kec::permute(s); else pos::permute(s);
        if (bits == 0 || (s[7] >> (64 - bits)) == 0) atomicMin(best + pk, (unsigned long long)cand);
    }
}

// The same search for PoseidonGoldilocksConfig, restructured (round 3):
//  * round 0 and the last linear layer collapse per candidate (poseidon.h permute_tail7; k_pow_prepare computes the twelve
//    per-proof constants once);
//  * work is dealt round robin over the UNFINISHED proofs: a workgroup draws a ticket (one global counter) and takes its next
//    chunk of 256 candidates from the (ticket mod U)-th of the U proofs still open -- every thread looks at the proofs
//    t, t + 256, ..., a wavefront scan ranks them.  What a finished search wastes is the chunks of that proof still in flight
//    beyond the witness, so the chunks in flight must be spread evenly: workgroups that stay on "their" proof and move to the
//    next open one when it finishes (the first form of this kernel) pile up behind runs of finished proofs, and 20 % of the
//    candidates hashed lay beyond a witness (profiles/r03_sq_pow_batch2.txt); dealt evenly it is the ~9 % that 2^18 lanes in
//    flight over U open proofs cost in any order;
//  * workgroups are NOT persistent: each takes at most `chunks` chunks of 256 candidates and leaves, so the launch drains as the
//    work runs out and the small latency-bound kernels of another sub-batch (own context and stream) find free slots between
//    them.  The grid is sized for several times the expected work; the last `tail_from`.. workgroups stay until every proof is
//    finished, so the search completes however unlucky it is.
// k_b[K][12] per-proof constants, then one word: the ticket counter (k_pow_prepare zeroes it)
__global__ __launch_bounds__(64) void k_pow_prepare(const u64 *st_b, const u32 *pos_b, u64 *k_b, u32 K) {
    const u32 k = blockIdx.x * 64 + threadIdx.x;
    if (k == 0) k_b[(size_t)K * 12] = 0;
    if (k >= K) return;
    u64 st[12], out[12];
    for (int i = 0; i < 12; i++) st[i] = st_b[(size_t)k * 12 + i];
    pos::pow_round0_consts(st, pos_b[k], out);
    for (int i = 0; i < 12; i++) k_b[(size_t)k * 12 + i] = out[i];
}
__device__ __forceinline__ bool pow_open(const unsigned long long *best, const unsigned long long *next, u32 q) {
    const unsigned long long b = *(volatile const unsigned long long *)(best + q), nx = *(volatile const unsigned long long *)(next + q);
    return nx < b && nx < (1ull << 40);
}
__global__ __launch_bounds__(256, 4) void k_pow_batch2(u64 *k_b, const u32 *pos_b, u32 bits, unsigned long long *best,
                                                        unsigned long long *next, u32 K, u32 chunks, u32 tail_from) {
    __shared__ unsigned long long sh_base, sh_ticket;
    __shared__ u32 sh_pick, sh_wsum[4];
    const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool persistent = blockIdx.x >= tail_from;
    unsigned long long *ticket = (unsigned long long *)(k_b + (size_t)K * 12);
    for (u32 done = 0; persistent || done < chunks; done++) {
        // rank the open proofs: thread t owns proofs t, t + 256, ...
        u32 mine = 0;
        for (u32 q = t; q < K; q += 256) mine += pow_open(best, next, q) ? 1u : 0u;
        u32 incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 v = (u32)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += v; }
        __syncthreads();                                            // the previous round's readers of sh_* are done
        if (lane == 63) sh_wsum[wave] = incl;
        if (t == 0) { sh_ticket = atomicAdd(ticket, 1ull); sh_pick = ~0u; }
        __syncthreads();
        u32 before = 0, open = 0;
#pragma unroll
        for (u32 w = 0; w < 4; w++) { if (w < wave) before += sh_wsum[w]; open += sh_wsum[w]; }
        if (open == 0) return;                                      // every proof has its witness (or its search is exhausted)
        const u32 want = (u32)sh_ticket % open, first = before + incl - mine;
        if (want >= first && want < first + mine) {                 // exactly one thread; proofs may have closed since the count
            u32 r = want - first;
            for (u32 q = t; q < K; q += 256)
                if (pow_open(best, next, q)) { if (r == 0) { sh_pick = q; break; } r--; }
        }
        __syncthreads();
        const u32 pk = sh_pick;
        if (pk == ~0u) continue;                                    // it closed in between: draw again
        if (t == 0) {
            unsigned long long b = atomicAdd(next + pk, 256ull);
            if (b >= *(volatile unsigned long long *)(best + pk) || b >= (1ull << 40)) b = ~0ull;       // taken by someone else in the meantime
            sh_base = b;
        }
        __syncthreads();
        const unsigned long long base = sh_base;
        if (base == ~0ull) continue;
        const u32 pos_ = pos_b[pk];
        const u64 cand = base + t;
        const u64 sp = pos::sbox7_nc(cand + pos::RC[pos_]);          // candidates stay far below p: the sum cannot wrap
        u64 s[12];
#pragma unroll
        for (int r = 0; r < 12; r++) s[r] = add_cnc(k_b[(size_t)pk * 12 + r], mul_small_nc(sp, pos::mds_entry(r, (int)pos_)));
        // a witness below this whole chunk may turn up while it is being hashed: then the rest of the permutation is wasted work
        const unsigned long long *bp = best + pk;
        const u64 e7 = pos::permute_tail7(s, [bp, base] { return *(volatile const unsigned long long *)bp < base; });
        if (bits == 0 || (e7 >> (64 - bits)) == 0) atomicMin(best + pk, (unsigned long long)cand);
    }
}

// ------------------------------------------------------------------------------------------ host side
namespace {
struct Tmp {   // pool-backed scratch for one prove() call
    glp_ctx *c;
    std::vector<void *> ptrs;
    explicit Tmp(glp_ctx *ctx) : c(ctx) {}
    ~Tmp() { (void)hipStreamSynchronize(c->stream); for (void *p : ptrs) c->release(p); }
    int get(u64 **p, size_t elems) {
        void *v = nullptr;
        int rc = c->alloc(&v, elems * sizeof(u64));
        if (rc == GLP_OK) { ptrs.push_back(v); *p = (u64 *)v; }
        return rc;
    }
};
struct BatchHolder {
    glp_batch *b = nullptr;
    ~BatchHolder() { batch_destroy(b); }
};
inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }
// the batch proof-of-work search for PoseidonGoldilocksConfig (k_pow_prepare + k_pow_batch2); st / pos / best / next as for k_pow_batch
int pow_batch_launch(glp_ctx *c, Tmp &tmp, const u64 *dev_pst, const u32 *dev_ppos, u32 bits, u64 *dev_best, u64 *dev_next, u32 K) {
    u64 *dev_k;
    GLP_TRY(tmp.get(&dev_k, (size_t)K * 12 + 1));              // + the ticket counter
    hipLaunchKernelGGL(k_pow_prepare, dim3((K + 63) / 64), dim3(64), 0, c->stream, dev_pst, dev_ppos, dev_k, K);
    GLP_HIP(hipGetLastError());
    constexpr u32 CHUNKS = 8;                                  // chunks of 256 candidates per non-persistent workgroup
    const double expected = (double)K * (bits >= 8 ? (double)(1ull << (bits - 8)) : 1.0);      // chunks: K 2^bits / 256
    const u32 tail = (u32)c->num_cus * 4;                      // these stay until every proof is finished
    const u32 body = (u32)std::min<double>(4.0 * expected / CHUNKS + 1.0, (double)(1u << 22));
    hipLaunchKernelGGL(k_pow_batch2, dim3(body + tail), dim3(256), 0, c->stream, dev_k, dev_ppos, bits,
                       (unsigned long long *)dev_best, (unsigned long long *)dev_next, K, CHUNKS, body);
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}
int d2h(glp_ctx *c, void *dst, const void *src, size_t bytes) {
    GLP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}
int h2d(glp_ctx *c, void *dst, const void *src, size_t bytes) {
    GLP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));   // the source is usually a stack/vector temporary
    return GLP_OK;
}
int batch_cap_host(glp_ctx *c, const glp_batch *b, std::vector<u64> &cap) {
    const size_t N = (size_t)1 << (b->lg + b->rate_bits);
    cap.resize((size_t)4 << b->cap_height);
    return d2h(c, cap.data(), b->digests + 4 * merkle_cap_offset(N, b->cap_height), cap.size() * 8);
}
// evaluate every polynomial of a batch at z: launch only; partial sums land in dev_partial [ncols][OPEN_BLOCKS][2]
int open_batch_launch(glp_ctx *c, const glp_batch *b, const u64 *dev_zt, u64 *dev_partial, u32 first_cols = 0 /* 0: all */) {
    dim3 g(open_blocks((size_t)1 << b->lg), first_cols ? first_cols : b->ncols);
    hipLaunchKernelGGL(k_open_dot, g, dim3(256), 0, c->stream, b->coeffs, dev_zt, dev_partial, (u32)b->lg, (size_t)0, (size_t)0, (size_t)0);
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}
// host side of the same: fold the OPEN_BLOCKS partial sums of each column
void open_batch_finish(const u64 *h, u32 ncols, u32 nob, std::vector<ext2> &out) {
    out.resize(ncols);
    for (u32 col = 0; col < ncols; col++) {
        u64 a = 0, bb = 0;
        for (u32 k = 0; k < nob; k++) { a = add(a, h[2 * ((size_t)col * nob + k)]); bb = add(bb, h[2 * ((size_t)col * nob + k) + 1]); }
        out[col] = e_make(a, bb);
    }
}
int zeta_table(glp_ctx *c, ext2 z, int lg, u64 *dev_zt) {
    ZTArgs za;
    za.zt = dev_zt; za.lg = (u32)lg; za.zeta_b = nullptr; za.zeta_stride = 0;
    ext2 p = z;
    for (int b = 0; b < 24; b++) { za.zp2[b] = p; p = e_sqr(p); }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_zeta_table<false>), dim3(nblk((size_t)1 << lg)), dim3(256), 0, c->stream, za);
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}
}  // namespace

// One proof in flight, cut at the points where the Fiat-Shamir transcript needs something from the device or the device
// needs a challenge.  glp_prove() drives it with the built-in Challenger; the glp_session_* entry points hand the same
// steps to a caller that keeps its own transcript (the Rust prover's `Challenger`): SURVEY.md section 8(b).
struct glp_session {
    glp_ctx *c;
    const glp_circuit *cc;
    const glp_circuit_desc &d;
    const Layout &L;
    const int lg, rb;
    const size_t n, N;
    const u32 nch, nr, nw, nc, qdf, npp, capn, nzp;
    const int hasher;
    Tmp tmp;
    u64 *owned_wires = nullptr;        // device copy made by begin() when the caller passed host memory
    const u64 *dev_wires = nullptr;
    u64 pih[4] = {0, 0, 0, 0};
    BatchHolder wb, zb, qb;
    u64 betas[MAXCH] = {}, gammas[MAXCH] = {}, alphas[MAXCH] = {};
    ext2 zeta = {0, 0}, zeta_next = {0, 0};
    const glp_batch *ob[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<ext2> open[4], zs_next;
    u64 *fcoef = nullptr;
    struct Layer { u64 *vals; u64 *dig; u32 lgL, ab; };
    std::vector<Layer> layers;
    u64 *cur = nullptr;
    int lgcur = 0;
    u64 shift = GEN;
    bool layer_open = false;           // a commit-phase layer has been committed and waits for its beta
    std::vector<u64> proof_words;      // the proof being assembled (glp_proof_words(circuit) words)
    std::vector<u64> cap;
    enum Stage { S_NEW, S_WIRES, S_ZS, S_QUOTIENT, S_OPEN, S_FRI, S_FINAL, S_DONE } stage = S_NEW;

    glp_session(glp_ctx *ctx, const glp_circuit *circ)
        : c(ctx), cc(circ), d(circ->d), L(circ->L), lg((int)circ->d.degree_bits), rb((int)circ->d.rate_bits),
          n((size_t)1 << circ->d.degree_bits), N(((size_t)1 << circ->d.degree_bits) << circ->d.rate_bits),
          nch(circ->d.num_challenges), nr(circ->d.num_routed_wires), nw(circ->d.num_wires), nc(circ->d.num_constants),
          qdf(circ->d.quotient_degree_factor), npp(circ->d.num_partial_products), capn(1u << circ->d.cap_height),
          nzp(circ->d.num_challenges * (1 + circ->d.num_partial_products)), hasher((int)circ->d.hasher), tmp(ctx) {}
    ~glp_session() { if (owned_wires) { (void)hipStreamSynchronize(c->stream); c->release(owned_wires); } }
    u64 *proof() { return proof_words.data(); }

    // K1-K4 over the witness; wires cap -> proof, cap
    // host_wires != nullptr: the witness is still in host memory and is uploaded into wires_dev chunk by chunk, overlapped
    // with the iNTT / LDE of the chunks already there
    int begin(const u64 *wires_dev, const u64 *public_inputs, const u64 *host_wires = nullptr) {
        GLP_REQUIRE(stage == S_NEW, "session already begun");
        proof_words.assign(L.total, 0);
        dev_wires = wires_dev;
        host_hash_no_pad(public_inputs, d.num_public_inputs, pih);
        if (d.num_public_inputs) memcpy(proof() + L.pis, public_inputs, (size_t)d.num_public_inputs * 8);
        GLP_TRY(batch_build(c, dev_wires, BATCH_VALUES, nw, lg, rb, (int)d.cap_height, &wb.b, host_wires, 1, hasher));
        GLP_TRY(batch_cap_host(c, wb.b, cap));
        memcpy(proof() + L.caps, cap.data(), capn * 32);
        stage = S_WIRES;
        return GLP_OK;
    }
    // K5 + commitment of Z and the partial products
    int partial_products(const u64 *betas_in, const u64 *gammas_in) {
        GLP_REQUIRE(stage == S_WIRES, "partial_products: call after begin");
        for (u32 i = 0; i < nch; i++) { betas[i] = betas_in[i]; gammas[i] = gammas_in[i]; }
        u64 *zp, *tot;
        const u32 nblocks = nblk(n);
        u64 *dens;
        GLP_TRY(tmp.get(&zp, (size_t)nzp * n));
        GLP_TRY(tmp.get(&dens, (size_t)nzp * n));
        GLP_TRY(tmp.get(&tot, (size_t)nch * nblocks));
        {
            StageScope st(c, "partial_products", 8.0 * n * (2.0 * nr + nzp));
            PPArgs a;
            a.wires = dev_wires; a.sigmas = cc->dev_sigmas; a.k_is = cc->dev_k_is; a.zp = zp; a.dens = dens;
            for (u32 i = 0; i < nch; i++) { a.betas[i] = betas[i]; a.gammas[i] = gammas[i]; }
            a.w_n = root_of_unity(lg); a.lg = (u32)lg; a.nr = nr; a.nch = nch; a.npp = npp; a.qdf = qdf;
            a.chal = nullptr; a.wires_stride = 0; a.zp_stride = 0;
            const size_t small_lds = (size_t)2 * nch * (npp + 2) * n * sizeof(u64);       // k_pp_rows_small: chunk products, row products and running products of one proof in LDS
            const bool small = lg <= 7 && small_lds <= 64 * 1024;
            if (small) hipLaunchKernelGGL(k_pp_rows_small, dim3(1, 1), dim3(256), small_lds, c->stream, a);      // rows AND the running product over them
            else switch (nch) {
            case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pp_rows<1>), dim3(nblocks), dim3(256), 0, c->stream, a); break;
            case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pp_rows<2>), dim3(nblocks), dim3(256), 0, c->stream, a); break;
            case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pp_rows<3>), dim3(nblocks), dim3(256), 0, c->stream, a); break;
            default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pp_rows<4>), dim3(nblocks), dim3(256), 0, c->stream, a); break;
            }
            GLP_HIP(hipGetLastError());
            if (!small) {
                hipLaunchKernelGGL(k_pp_block_tot, dim3(nblocks, nch), dim3(256), 0, c->stream, zp, tot, (u32)lg, nblocks, (size_t)0);
                hipLaunchKernelGGL(k_pp_scan_tot, dim3(nch), dim3(256), 0, c->stream, tot, nblocks);
                hipLaunchKernelGGL(k_pp_apply, dim3(nblocks, nch), dim3(256), 0, c->stream, zp, tot, (u32)lg, nblocks, nch, npp, (size_t)0);
                GLP_HIP(hipGetLastError());
            }
        }
        GLP_TRY(batch_build(c, zp, BATCH_VALUES, nzp, lg, rb, (int)d.cap_height, &zb.b, nullptr, 1, hasher));
        GLP_TRY(batch_cap_host(c, zb.b, cap));
        memcpy(proof() + L.caps + capn * 4, cap.data(), capn * 32);
        stage = S_ZS;
        return GLP_OK;
    }
    // K6 + commitment of the quotient chunks
    int quotient(const u64 *alphas_in) {
        GLP_REQUIRE(stage == S_ZS, "quotient: call after partial_products");
        for (u32 i = 0; i < nch; i++) alphas[i] = alphas_in[i];
        int qdb = 0;
        while ((1u << qdb) < qdf) qdb++;
        const u32 Rq = 1u << qdb, step = 1u << (rb - qdb);
        const u32 nchunks = npp + 1, nterms = nch + nch * nchunks + d.num_gate_constraints;
        // alpha powers twice: whole (permutation terms) and as 22-bit limbs of m and m 2^32 (gate constraints, AccHL): [1 + APL_WORDS][nch * nterms] words
        std::vector<u64> apow((size_t)(1 + APL_WORDS) * nch * nterms);
        for (u32 i = 0; i < nch; i++) {
            u64 x = 1;
            for (u32 k = 0; k < nterms; k++) {
                const size_t e = (size_t)i * nterms + k;
                apow[e] = x;
                apl_words(x, &apow[(size_t)nch * nterms + APL_WORDS * e]);
                x = mul(x, alphas[i]);
            }
        }
        u64 *dev_apow, *qv, *qV, *qc;
        GLP_TRY(tmp.get(&dev_apow, apow.size()));
        GLP_TRY(h2d(c, dev_apow, apow.data(), apow.size() * 8));
        GLP_TRY(tmp.get(&qv, (size_t)nch * Rq * n));
        GLP_TRY(tmp.get(&qV, (size_t)nch * Rq * n));
        GLP_TRY(tmp.get(&qc, (size_t)nch * Rq * n));
        QArgs a;
        QProof qp;
        QBatch qbt;
        memset(&qp, 0, sizeof(qp));
        memset(&qbt, 0, sizeof(qbt));                    // pp == nullptr: one proof, described by qp
        a.cs = cc->cs->lde; qp.wl = wb.b->lde; qp.zl = zb.b->lde; qp.out = qv;
        a.gates = cc->dev_gates; a.k_is = cc->dev_k_is; qp.apow = dev_apow; qp.apl = dev_apow + (size_t)nch * nterms; a.k_ratio = cc->k_ratio;
        for (u32 i = 0; i < nch; i++) { qp.betas[i] = betas[i]; qp.gammas[i] = gammas[i]; }
        memcpy(qp.pih, pih, 32);
        const u64 WN = root_of_unity(lg + rb), gn = pow(GEN, (u64)n), wR = root_of_unity(rb);
        for (u32 rq = 0; rq < Rq; rq++) {
            const u32 r = rq * step;
            a.shift_r[rq] = mul(GEN, pow(WN, (u64)r));
            a.zh[rq] = sub(mul(gn, pow(wR, (u64)r)), 1);      // Z_H(g W^(qR + r)) = g^n w_R^r - 1
            a.zh_inv[rq] = inv(a.zh[rq]);
        }
        a.w_n = root_of_unity(lg); a.n_field = (u64)n % P;
        a.lg = (u32)lg; a.rb = (u32)rb; a.step = step; a.nc = nc; a.nsel = d.num_selectors; a.nr = nr; a.nw = nw;
        a.nch = nch; a.npp = npp; a.qdf = qdf; a.num_gates = d.num_gates; a.nterms = nterms;
        a.many_selectors = d.num_selectors > 1;
        {
            StageScope st(c, "quotient_eval", 8.0 * n * Rq * (nc + nr + nw + nzp + 2.0 * nch));
            // two challenges (every preset the reference uses): permutation terms in one launch, then one launch per
            // gate type compiled on its own; other challenge counts take the monolithic kernel
            a.gate_mode = nch == 2 ? 1 : 0;
            u64 *l0t;
            GLP_TRY(tmp.get(&l0t, (size_t)Rq * n));
            a.l0 = l0t;
            hipLaunchKernelGGL(k_l0_table, dim3(nblk(n)), dim3(256), 0, c->stream, a, l0t, Rq);
            GLP_HIP(hipGetLastError());
            LightArgs lg_;
            lg_.count = cc->light_count; lg_.arith_gi = cc->arith_gi; lg_.arith_ops = cc->arith_ops;
            for (u32 i = 0; i < 8; i++) lg_.gi[i] = cc->light_gi[i];
            switch (nch) {
            case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_quotient<1, 1>), dim3(nblk(n), Rq), dim3(256), 0, c->stream, a, qp, qbt, lg_); break;
            case 2:
                if (cc->light_count || cc->arith_ops) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_quotient<2, 2>), dim3(nblk(n), Rq), dim3(256), 0, c->stream, a, qp, qbt, lg_);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_quotient<2, 0>), dim3(nblk(n), Rq), dim3(256), 0, c->stream, a, qp, qbt, lg_);
                break;
            case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_quotient<3, 1>), dim3(nblk(n), Rq), dim3(256), 0, c->stream, a, qp, qbt, lg_); break;
            default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_quotient<4, 1>), dim3(nblk(n), Rq), dim3(256), 0, c->stream, a, qp, qbt, lg_); break;
            }
            GLP_HIP(hipGetLastError());
            if (a.gate_mode == 1) {
                if (cc->limb_count) {
                    LimbArgs la;
                    limb_args(cc, la);
                    la.extra_count = cc->limb_extra_count;
                    for (int i = 0; i < 4; i++) la.extra_gi[i] = cc->limb_extra_gi[i];
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_quotient_limbs<2>), dim3(nblk(n), Rq), dim3(256), 0, c->stream, a, qp, qbt, la);
                    GLP_HIP(hipGetLastError());
                }
                for (u32 gi : cc->single_gates) {
#define GLP_GATE_LAUNCH(T) case T: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_quotient_gate<2, T>), dim3(nblk(n), Rq), dim3(256), 0, c->stream, a, qp, qbt, gi); break;
                    switch (cc->gates[gi].type) {
                        GLP_GATE_LAUNCH(GLP_GATE_CONSTANT) GLP_GATE_LAUNCH(GLP_GATE_PUBLIC_INPUT) GLP_GATE_LAUNCH(GLP_GATE_ARITHMETIC)
                        GLP_GATE_LAUNCH(GLP_GATE_POSEIDON) GLP_GATE_LAUNCH(GLP_GATE_U32_INTERLEAVE) GLP_GATE_LAUNCH(GLP_GATE_UNINTERLEAVE_U32)
                        GLP_GATE_LAUNCH(GLP_GATE_UNINTERLEAVE_B32) GLP_GATE_LAUNCH(GLP_GATE_U32_ARITHMETIC) GLP_GATE_LAUNCH(GLP_GATE_U32_ADD_MANY)
                        GLP_GATE_LAUNCH(GLP_GATE_U32_SUBTRACTION) GLP_GATE_LAUNCH(GLP_GATE_U32_RANGE_CHECK) GLP_GATE_LAUNCH(GLP_GATE_COMPARISON)
                        GLP_GATE_LAUNCH(GLP_GATE_BASE_SUM) GLP_GATE_LAUNCH(GLP_GATE_RANDOM_ACCESS)
                    default: break;   // NoopGate: no constraints
                    }
#undef GLP_GATE_LAUNCH
                    GLP_HIP(hipGetLastError());
                }
            }
            GLP_HIP(hipGetLastError());
        }
        {
            StageScope st(c, "quotient_intt", 16.0 * n * Rq * nch);
            GLP_TRY(intt_values_to_coeffs(c, qv, qV, nch * Rq, lg));
            QCArgs q;
            q.V = qV; q.out = qc; q.lg = (u32)lg; q.Rq = Rq;
            q.wM_inv = inv(root_of_unity(lg + qdb)); q.wR_inv = inv(root_of_unity(qdb)); q.g_inv = inv(GEN);
            q.rq_inv = inv((u64)Rq);
            const u64 gni = inv(gn);
            u64 x = 1;
            for (u32 cidx = 0; cidx < Rq; cidx++) { q.gn_inv_pow[cidx] = x; x = mul(x, gni); }
            hipLaunchKernelGGL(k_quotient_combine, dim3(nblk(n), nch), dim3(256), 0, c->stream, q);
            GLP_HIP(hipGetLastError());
        }
        GLP_TRY(batch_build(c, qc, BATCH_COEFFS_BITREV, nch * qdf, lg, rb, (int)d.cap_height, &qb.b, nullptr, 1, hasher));
        GLP_TRY(batch_cap_host(c, qb.b, cap));
        memcpy(proof() + L.caps + 2 * capn * 4, cap.data(), capn * 32);
        stage = S_QUOTIENT;
        return GLP_OK;
    }
    // K7: every committed polynomial at zeta, Z at g zeta; openings -> proof (OpeningSet order)
    int open_at(ext2 zeta_in) {
        GLP_REQUIRE(stage == S_QUOTIENT, "open: call after quotient");
        zeta = zeta_in;
        {
            ext2 zp = zeta;
            for (int i = 0; i < lg; i++) zp = e_sqr(zp);
            if (e_eq(zp, e_from(1))) return set_error(GLP_ERR_PROVE, "Opening point is in the subgroup.");
        }
        zeta_next = e_scale(zeta, root_of_unity(lg));
        ob[0] = cc->cs; ob[1] = wb.b; ob[2] = zb.b; ob[3] = qb.b;
        {
        StageScope st(c, "openings", 8.0 * n * (L.oracle_cols[0] + L.oracle_cols[1] + L.oracle_cols[2] + L.oracle_cols[3] + nch));
        // five evaluations (four batches at zeta, the Z batch at g zeta) queued back to back, one copy back
        u64 *zt, *partial;
        GLP_TRY(tmp.get(&zt, 2 * n));
        const u32 nob = open_blocks(n);
        size_t poff[6] = {0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 5; k++) poff[k + 1] = poff[k] + (size_t)(k < 4 ? ob[k]->ncols : nch) * nob * 2;    // at g zeta: only the Z columns
        GLP_TRY(tmp.get(&partial, poff[5]));
        GLP_TRY(zeta_table(c, zeta, lg, zt));
        for (int k = 0; k < 4; k++) GLP_TRY(open_batch_launch(c, ob[k], zt, partial + poff[k]));
        GLP_TRY(zeta_table(c, zeta_next, lg, zt));
        GLP_TRY(open_batch_launch(c, zb.b, zt, partial + poff[4], nch));
        std::vector<u64> hp(poff[5]);
        GLP_TRY(d2h(c, hp.data(), partial, hp.size() * 8));
        for (int k = 0; k < 4; k++) open_batch_finish(hp.data() + poff[k], ob[k]->ncols, nob, open[k]);
        std::vector<ext2> all;
        open_batch_finish(hp.data() + poff[4], nch, nob, all);
        zs_next.assign(all.begin(), all.begin() + nch);
    }
        {
            u64 *op = proof() + L.openings;
            size_t o = 0;
            auto put = [&](ext2 e) { op[o++] = e.a; op[o++] = e.b; };
            for (u32 k = 0; k < nc + nr; k++) put(open[0][k]);
            for (u32 k = 0; k < nw; k++) put(open[1][k]);
            for (u32 k = 0; k < nch; k++) put(open[2][k]);
            for (u32 k = 0; k < nch; k++) put(zs_next[k]);
            for (u32 k = 0; k < nch * npp; k++) put(open[2][nch + k]);
            for (u32 k = 0; k < nch * qdf; k++) put(open[3][k]);
        }
        stage = S_OPEN;
        return GLP_OK;
    }
    // K8: alpha-combination of all openings batches, quotient by (X - zeta) / (X - g zeta) -> FRI polynomial
    int fri_combine(ext2 alpha) {
        GLP_REQUIRE(stage == S_OPEN, "fri_combine: call after open");
        GLP_TRY(tmp.get(&fcoef, 2 * n));
    {
        StageScope st(c, "fri_combine", 8.0 * n * (L.oracle_cols[0] + L.oracle_cols[1] + L.oracle_cols[2] + L.oracle_cols[3]));
        size_t total_cols = 0;
        for (int k = 0; k < 4; k++) total_cols += ob[k]->ncols;
        std::vector<u64> ap(2 * total_cols);
        ext2 x = e_from(1), red0 = e_from(0), red1 = e_from(0);
        size_t j = 0;
        for (int k = 0; k < 4; k++)
            for (u32 col = 0; col < ob[k]->ncols; col++, j++) {
                ap[2 * j] = x.a; ap[2 * j + 1] = x.b;
                red0 = e_add(red0, e_mul(x, open[k][col]));
                x = e_mul(x, alpha);
            }
        x = e_from(1);
        for (u32 col = 0; col < nch; col++) { red1 = e_add(red1, e_mul(x, zs_next[col])); x = e_mul(x, alpha); }
        u64 *dev_ap, *fv;
        GLP_TRY(tmp.get(&dev_ap, ap.size()));
        GLP_TRY(h2d(c, dev_ap, ap.data(), ap.size() * 8));
        GLP_TRY(tmp.get(&fv, 2 * n));
        FVArgs a;
        for (int k = 0; k < 4; k++) { a.lde[k] = ob[k]->lde; a.ncols[k] = ob[k]->ncols; }
        a.apow = dev_ap; a.out = fv; a.red0 = red0; a.red1 = red1; a.zeta = zeta; a.zeta_next = zeta_next;
        a.shift_acc = e_pow(alpha, nch);
        a.w_n = root_of_unity(lg); a.g = GEN; a.lg = (u32)lg; a.rb = (u32)rb; a.nch = nch;
        a.pp = nullptr; a.apow_stride = a.out_stride = 0;
        for (int k = 0; k < 4; k++) a.lde_stride[k] = 0;
        if (lg >= 2 && lg <= 7) hipLaunchKernelGGL(k_final_values_small, dim3(1, 1), dim3(256), 0, c->stream, a);      // 4..128 points: 256 / n lanes per point
        else hipLaunchKernelGGL(k_final_values, dim3(nblk(n)), dim3(256), 0, c->stream, a);
        GLP_HIP(hipGetLastError());
        GLP_TRY(intt_values_to_coeffs(c, fv, fcoef, 2, lg));
        hipLaunchKernelGGL(k_scale_bitrev_pow, dim3(nblk(n), 2), dim3(256), 0, c->stream, fcoef, inv(GEN), (u32)lg);
        GLP_HIP(hipGetLastError());
    }

        cur = fcoef; lgcur = lg; shift = GEN;
        stage = S_FRI;
        return GLP_OK;
    }
    // K9, first half: LDE of the current polynomial on its coset, Merkle tree over arity-sized leaves; cap -> proof, cap
    int fri_commit_layer() {
        GLP_REQUIRE(stage == S_FRI && !layer_open && layers.size() < d.num_reductions, "fri_commit: no layer left or beta pending");
        StageScope st(c, "fri_commit", 0.0);
        const u32 r = (u32)layers.size();
        const u32 ab = d.reduction_arity_bits[r];
        const u32 lgL = (u32)(lgcur + rb);
        const size_t Lsz = (size_t)1 << lgL, nleaves = Lsz >> ab;
        Layer ly;
        ly.lgL = lgL; ly.ab = ab;
        GLP_TRY(tmp.get(&ly.vals, 2 * Lsz));
        GLP_TRY(tmp.get(&ly.dig, merkle_num_digests(nleaves, (int)d.cap_height) * 4));
        GLP_TRY(lde_coeffs(c, cur, ly.vals, 2, lgcur, rb, shift));
        if (hasher == GLP_HASH_KECCAK25)
            hipLaunchKernelGGL(k_fri_leaf_hash_keccak, dim3(nblk(nleaves)), dim3(256), 0, c->stream, ly.vals, ly.dig, lgL, (u32)rb, ab, (size_t)0,
                               (size_t)0);
        else if (nleaves <= c->merkle_coop_max)
            hipLaunchKernelGGL(k_fri_leaf_hash_coop, dim3((unsigned)((nleaves + 15) / 16)), dim3(256), 0, c->stream, ly.vals, ly.dig, lgL,
                               (u32)rb, ab, (size_t)0, (size_t)0);
        else if (nleaves <= c->merkle_quad_max)
            hipLaunchKernelGGL(k_fri_leaf_hash_quad, dim3((unsigned)((nleaves + 63) / 64)), dim3(256), 0, c->stream, ly.vals, ly.dig, lgL,
                               (u32)rb, ab, (size_t)0, (size_t)0);
        else
            hipLaunchKernelGGL(k_fri_leaf_hash, dim3(nblk(nleaves)), dim3(256), 0, c->stream, ly.vals, ly.dig, lgL, (u32)rb, ab, (size_t)0,
                               (size_t)0);
        GLP_HIP(hipGetLastError());
        GLP_TRY(merkle_levels(c, ly.dig, nleaves, (int)d.cap_height, 1, 0, hasher));
        cap.resize((size_t)capn * 4);
        GLP_TRY(d2h(c, cap.data(), ly.dig + 4 * merkle_cap_offset(nleaves, (int)d.cap_height), (size_t)capn * 32));
        memcpy(proof() + L.fri_caps + (size_t)r * capn * 4, cap.data(), (size_t)capn * 32);
        layers.push_back(ly);
        layer_open = true;
        return GLP_OK;
    }
    // K9, second half: fold the coefficients with beta (arity 2^ab), shift <- shift^arity
    int fri_fold(ext2 beta) {
        GLP_REQUIRE(stage == S_FRI && layer_open, "fri_fold: call after fri_commit");
        StageScope st(c, "fri_commit", 0.0);
        const u32 ab = layers.back().ab;
        u64 *nxt;
        const size_t nnew = ((size_t)1 << lgcur) >> ab;
        GLP_TRY(tmp.get(&nxt, 2 * nnew));
        hipLaunchKernelGGL(k_fri_fold, dim3(nblk(nnew)), dim3(256), 0, c->stream, cur, nxt, beta, (u32)lgcur, ab, (const u64 *)nullptr);
        GLP_HIP(hipGetLastError());
        cur = nxt; lgcur -= (int)ab;
        shift = pow(shift, (u64)1 << ab);
        layer_open = false;
        return GLP_OK;
    }
    // final polynomial (natural coefficient order) -> proof
    int fri_final_poly() {
        GLP_REQUIRE(stage == S_FRI && !layer_open && layers.size() == d.num_reductions, "fri_final_poly: reductions not finished");
        const size_t fl = (size_t)1 << lgcur;
        if (fl != L.final_len) return set_error(GLP_ERR_ARG, "reduction_arity_bits inconsistent with degree_bits");
        std::vector<u64> h(2 * fl);
        GLP_TRY(d2h(c, h.data(), cur, h.size() * 8));
        for (size_t p = 0; p < fl; p++) {
            const size_t k = bitrev32((u32)p, lgcur);
            proof()[L.final_poly + 2 * k] = h[p];
            proof()[L.final_poly + 2 * k + 1] = h[fl + p];
        }
        stage = S_FINAL;
        return GLP_OK;
    }
    // query phase: leaves and Merkle paths of the four initial oracles and of every commit-phase layer
    int queries(u64 pow_witness, const u64 *indices, u32 nq) {
        GLP_REQUIRE(stage == S_FINAL, "queries: call after fri_final_poly");
        GLP_REQUIRE(nq == d.num_query_rounds, "queries: %u indices, the circuit has %u query rounds", nq, d.num_query_rounds);
        proof()[L.pow] = pow_witness;
        StageScope st(c, "fri_queries", 0.0);
        std::vector<u64> xi(indices, indices + nq);
        for (u32 q = 0; q < nq; q++) GLP_REQUIRE(xi[q] < (u64)N, "query index %llu outside the LDE domain", (unsigned long long)xi[q]);
        // every gather writes straight into a device image of the proof's query section; one copy brings it back
        u64 *dev_idx, *dev_q;
        const size_t stride = L.query_stride;
        GLP_TRY(tmp.get(&dev_idx, nq));
        GLP_TRY(tmp.get(&dev_q, (size_t)nq * stride));
        GLP_TRY(h2d(c, dev_idx, xi.data(), nq * 8));
        size_t off = 0;   // word offset inside one query record
        for (int k = 0; k < 4; k++) {
            const u32 ncol = ob[k]->ncols;
            GLP_TRY(merkle_gather_lde_rows(c, ob[k]->lde, ncol, lg, rb, dev_idx, nq, dev_q + off, stride));
            off += ncol;
            GLP_TRY(merkle_gather_paths(c, ob[k]->digests, N, (int)d.cap_height, dev_idx, nq, dev_q + off, stride, 0));
            off += 4 * (size_t)L.depth0;
        }
        u32 shift_bits = 0;
        for (size_t r = 0; r < layers.size(); r++) {
            const Layer &ly = layers[r];
            const u32 arity = 1u << ly.ab;
            const size_t nleaves = ((size_t)1 << ly.lgL) >> ly.ab;
            shift_bits += ly.ab;
            hipLaunchKernelGGL(k_fri_gather_leaf, dim3(nblk((size_t)nq * arity)), dim3(256), 0, c->stream, ly.vals, ly.lgL, (u32)rb,
                               ly.ab, dev_idx, shift_bits, nq, dev_q + off, stride, (size_t)0, (size_t)0);
            GLP_HIP(hipGetLastError());
            off += 2 * (size_t)arity;
            GLP_TRY(merkle_gather_paths(c, ly.dig, nleaves, (int)d.cap_height, dev_idx, nq, dev_q + off, stride, shift_bits));
            off += 4 * (size_t)L.step_depth[r];
        }
        if (off != stride) return set_error(GLP_ERR_ARG, "internal: query record layout mismatch");
        GLP_TRY(d2h(c, proof() + L.queries, dev_q, (size_t)nq * stride * 8));
        stage = S_DONE;
        return GLP_OK;
    }
};

// One bound for glp_circuit_create and glp_pow_search: with the search capped at 2^40 candidates, 32 bits leaves a failure
// probability of exp(-2^8).
constexpr u32 POW_MAX_BITS = 32;
// K10: smallest witness w >= 0 such that the sponge (state + pending inputs + w) squeezes a value with `bits` leading
// zeros.  The search covers candidates in increasing order, so the result does not depend on launch geometry.
static int pow_search(glp_ctx *c, const u64 st[12], const u64 *pending, u32 npending, u32 bits, u64 *witness, int hasher = GLP_HASH_POSEIDON) {
    StageScope stg(c, "fri_pow", 0.0);
    GLP_REQUIRE(npending < 8, "proof of work: %u pending inputs (the rate is 8)", npending);
    PowArgs a;
    memcpy(a.st, st, 96);
    for (u32 i = 0; i < npending; i++) a.st[i] = pending[i];
    a.pos = npending; a.bits = bits;
    Tmp tmp(c);
    u64 *best;
    GLP_TRY(tmp.get(&best, 1));
    a.best = (unsigned long long *)best;
    const u64 none = ~0ull;
    u64 found = none;
    // expected 2^bits tries: size a launch at four times that (a 2^20-candidate launch is 0.6 ms of hashing)
    const u64 batch = 1ull << std::min<u32>(20, std::max<u32>(14, bits + 2));
    for (u64 base = 0; found == none; base += batch) {
        if (base >= (1ull << 40)) return set_error(GLP_ERR_PROVE, "Proof of work failed. This is highly unlikely!");
        GLP_TRY(h2d(c, best, &none, 8));
        a.base = base;
        if (hasher == GLP_HASH_KECCAK25) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pow<GLP_HASH_KECCAK25>), dim3((unsigned)(batch / 256)), dim3(256), 0, c->stream, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pow<GLP_HASH_POSEIDON>), dim3((unsigned)(batch / 256)), dim3(256), 0, c->stream, a);
        GLP_HIP(hipGetLastError());
        GLP_TRY(d2h(c, &found, best, 8));
    }
    *witness = found;
    return GLP_OK;
}

// prove(): the session driven by the library's own transcript (plonk/prover.rs order)
static int prove_impl(glp_ctx *c, const glp_circuit *cc, const u64 *dev_wires, const u64 *public_inputs, u64 *proof,
                      const u64 *host_wires = nullptr) {
    glp_session s(c, cc);
    const glp_circuit_desc &d = cc->d;
    const Layout &L = cc->L;
    const u32 nch = s.nch, nr = s.nr, nw = s.nw, nc = s.nc, qdf = s.qdf, npp = s.npp, capn = s.capn;
    GLP_TRY(s.begin(dev_wires, public_inputs, host_wires));
    Challenger ch((int)d.hasher);
    ch.observe_hashes(cc->digest, 1);
    ch.observe(s.pih, 4);                       // InnerHasher (Poseidon) HashOut
    ch.observe_hashes(s.cap.data(), capn);
    u64 betas[MAXCH], gammas[MAXCH], alphas[MAXCH];
    for (u32 i = 0; i < nch; i++) betas[i] = ch.get();
    for (u32 i = 0; i < nch; i++) gammas[i] = ch.get();
    GLP_TRY(s.partial_products(betas, gammas));
    ch.observe_hashes(s.cap.data(), capn);
    for (u32 i = 0; i < nch; i++) alphas[i] = ch.get();
    GLP_TRY(s.quotient(alphas));
    ch.observe_hashes(s.cap.data(), capn);
    GLP_TRY(s.open_at(ch.get_ext()));
    {
        const u64 *op = s.proof() + L.openings;
        const u64 *p_cs = op, *p_w = op + 2 * (nc + nr), *p_zs = p_w + 2 * nw, *p_zn = p_zs + 2 * nch;
        const u64 *p_pp = p_zn + 2 * nch, *p_q = p_pp + 2 * nch * npp;
        ch.observe(p_cs, 2 * (nc + nr)); ch.observe(p_w, 2 * nw); ch.observe(p_zs, 2 * nch);
        ch.observe(p_pp, 2 * (size_t)nch * npp); ch.observe(p_q, 2 * (size_t)nch * qdf); ch.observe(p_zn, 2 * nch);
    }
    GLP_TRY(s.fri_combine(ch.get_ext()));
    for (u32 r = 0; r < d.num_reductions; r++) {
        GLP_TRY(s.fri_commit_layer());
        ch.observe_hashes(s.cap.data(), capn);
        GLP_TRY(s.fri_fold(ch.get_ext()));
    }
    GLP_TRY(s.fri_final_poly());
    ch.observe(s.proof() + L.final_poly, 2 * L.final_len);
    u64 found;
    GLP_TRY(pow_search(c, ch.st, ch.in, (u32)ch.nin, d.proof_of_work_bits, &found, (int)d.hasher));
    ch.observe(&found, 1);
    const u64 resp = ch.get();
    if (d.proof_of_work_bits && (resp >> (64 - d.proof_of_work_bits)) != 0)
        return set_error(GLP_ERR_PROVE, "proof-of-work response check failed");
    std::vector<u64> xi(d.num_query_rounds);
    for (u32 q = 0; q < d.num_query_rounds; q++) xi[q] = ch.get() % (u64)s.N;
    GLP_TRY(s.queries(found, xi.data(), d.num_query_rounds));
    memcpy(proof, s.proof(), L.total * 8);
    return GLP_OK;
}

// ---- byte format (util/serialization.rs): words little-endian, u8 sibling count before each Merkle path
namespace {
// calls f(offset_words, count_words, kind) for the pieces of a proof in order.  PW_FIELD: field elements; PW_DIGESTS: digests of
// the proof's hasher, 4 words each (caps); PW_PATH: a Merkle path (digests, preceded by a one-byte sibling count on the wire)
enum { PW_FIELD = 0, PW_DIGESTS = 1, PW_PATH = 2 };
template <class F> void walk_proof(const glp_circuit *cc, F f) {
    const Layout &L = cc->L;
    const glp_circuit_desc &d = cc->d;
    f((size_t)0, L.openings, PW_DIGESTS);                                   // wires, Z / partial products, quotient caps
    f(L.openings, L.fri_caps - L.openings, PW_FIELD);                       // openings
    f(L.fri_caps, L.queries - L.fri_caps, PW_DIGESTS);                      // commit-phase caps
    for (u32 q = 0; q < d.num_query_rounds; q++) {
        size_t o = L.queries + (size_t)q * L.query_stride;
        for (int k = 0; k < 4; k++) {
            f(o, (size_t)L.oracle_cols[k], PW_FIELD); o += L.oracle_cols[k];
            f(o, 4 * (size_t)L.depth0, PW_PATH); o += 4 * (size_t)L.depth0;
        }
        for (u32 r = 0; r < d.num_reductions; r++) {
            const size_t ev = (size_t)2 << d.reduction_arity_bits[r];
            f(o, ev, PW_FIELD); o += ev;
            f(o, 4 * (size_t)L.step_depth[r], PW_PATH); o += 4 * (size_t)L.step_depth[r];
        }
    }
    f(L.final_poly, L.total - L.final_poly, PW_FIELD);     // final poly, pow witness, public inputs
}
}  // namespace


// Quotient launch plan: the base-4 limb gates (at most LIMB_SLOTS of them) share k_quotient_limbs, the HBM-bound light
// gates share k_quotient_light, every other gate type keeps its own launch.  Builds the column program of
// k_quotient_limbs (format: see the kernel).
static int build_quotient_plan(glp_ctx *c, glp_circuit *cc) {
    const glp_circuit_desc &d = cc->d;
    std::vector<u64> desc((size_t)LIMB_GROUPS * d.num_wires * LIMB_SLOTS, 0);
    for (int g = 0; g < LIMB_GROUPS; g++) { cc->limb_jlo[g] = d.num_wires; cc->limb_jhi[g] = 0; }
    auto put = [&](u32 s, u32 col, u32 pos, u32 kl, bool flush, u32 kf, u32 ref) {       // s = slot over all groups
        const u32 grp = s / LIMB_SLOTS, slot = s % LIMB_SLOTS;
        desc[((size_t)grp * d.num_wires + col) * LIMB_SLOTS + slot] = 1ull | ((u64)pos << 1) | (flush ? 32ull : 0ull) | ((u64)kl << 6) | ((u64)kf << 16) | ((u64)ref << 26);
        cc->limb_jlo[grp] = std::min(cc->limb_jlo[grp], col); cc->limb_jhi[grp] = std::max(cc->limb_jhi[grp], col);
    };
    std::vector<u32> limb_list;
    auto fill = [&](u32 s, u32 gi) {               // column program of gate gi in slot s (= 4 group + slot)
        const glp_gate &g = cc->gates[gi];
        cc->limb_gi[s] = gi;
        if (g.type == GLP_GATE_U32_ARITHMETIC) {
            for (u32 i = 0; i < g.p0; i++)
                for (u32 j = 0; j < 32; j++)
                    put(s, 6 * g.p0 + 32 * i + j, j & 15, 36 * i + 2 + (31 - j), (j & 15) == 15, 36 * i + 34 + (j >> 4), 6 * i + 3 + (j >> 4));
        } else if (g.type == GLP_GATE_U32_ADD_MANY) {
            const u32 na = g.p0, nops = g.p1, wd = na + 3;
            for (u32 i = 0; i < nops; i++)
                for (u32 j = 0; j < 18; j++)
                    put(s, wd * nops + 18 * i + j, j & 15, 21 * i + 1 + (17 - j), j == 15 || j == 17, 21 * i + 19 + (j >> 4), wd * i + na + 1 + (j >> 4));
        } else if (g.type == GLP_GATE_U32_SUBTRACTION) {
            for (u32 i = 0; i < g.p0; i++)
                for (u32 j = 0; j < 16; j++)
                    put(s, 5 * g.p0 + 16 * i + j, j, 19 * i + 1 + (15 - j), j == 15, 19 * i + 17, 5 * i + 3);
        } else {
            for (u32 i = 0; i < g.p0; i++)
                for (u32 j = 0; j < 16; j++)
                    put(s, g.p0 + 16 * i + j, j, 17 * i + 1 + j, j == 15, 17 * i, i);
        }
    };
    auto limb_weight = [&](u32 gi) -> u32 {        // limb columns of the gate = its share of the per-point work
        const glp_gate &g = cc->gates[gi];
        return g.type == GLP_GATE_U32_ARITHMETIC ? 32 * g.p0 : g.type == GLP_GATE_U32_ADD_MANY ? 18 * g.p1 : 16 * g.p0;
    };
    for (u32 gi = 0; gi < d.num_gates; gi++) {
        const glp_gate &g = cc->gates[gi];
        const bool limb_gate = g.type == GLP_GATE_U32_ARITHMETIC || g.type == GLP_GATE_U32_ADD_MANY ||
                               g.type == GLP_GATE_U32_SUBTRACTION || g.type == GLP_GATE_U32_RANGE_CHECK;
        const bool light = g.type == GLP_GATE_CONSTANT || g.type == GLP_GATE_PUBLIC_INPUT || g.type == GLP_GATE_ARITHMETIC ||
                           g.type == GLP_GATE_BASE_SUM || g.type == GLP_GATE_RANDOM_ACCESS;
        // alpha indices and wire columns must fit the descriptor fields (10 and 8 bits); glp_circuit_create has already
        // bounded num_constraints by ACC3_MAX_TERMS = 512
        if (limb_gate && limb_list.size() < (size_t)(LIMB_SLOTS * LIMB_GROUPS) && d.num_wires <= 256) {
            limb_list.push_back(gi);                       // slots are assigned below, once all limb gates are known
        } else if (g.type == GLP_GATE_ARITHMETIC && cc->arith_ops == 0 && 4 * g.p0 <= d.num_routed_wires && d.quotient_degree_factor % 4 == 0 &&
                   d.num_selectors + 2 <= d.num_constants) {
            cc->arith_gi = gi; cc->arith_ops = g.p0;      // evaluated inside the permutation loop of k_quotient
        } else if (light && cc->light_count < 8) {
            cc->light_gi[cc->light_count++] = gi;
        } else if (g.type != GLP_GATE_NOOP) {
            cc->single_gates.push_back(gi);
        }
    }
    if (limb_list.size() == 1) {               // nothing to share: the gate's own kernel is the better launch
        cc->single_gates.push_back(limb_list[0]);
        limb_list.clear();
    }
    if (!limb_list.empty()) {
        // Up to LIMB_SLOTS gates share a set of accumulators (one group); more gates go through the same launch group after group.  Every
        // group computes the range products of its own column range, so gates are grouped by where their limb columns START: the
        // union ranges of the groups then overlap least (secp256k1 circuit: [8,136) + [40,136) = 224 columns instead of 2 x 128).
        const u32 cnt = (u32)limb_list.size();
        const u32 G = (cnt + LIMB_SLOTS - 1) / LIMB_SLOTS;
        u32 used[LIMB_GROUPS] = {0, 0, 0, 0};
        auto limb_start = [&](u32 gi) -> u32 {
            const glp_gate &g = cc->gates[gi];
            return g.type == GLP_GATE_U32_ARITHMETIC ? 6 * g.p0 : g.type == GLP_GATE_U32_ADD_MANY ? (g.p0 + 3) * g.p1 : g.type == GLP_GATE_U32_SUBTRACTION ? 5 * g.p0 : g.p0;
        };
        std::stable_sort(limb_list.begin(), limb_list.end(), [&](u32 x, u32 y) {
            return limb_start(x) != limb_start(y) ? limb_start(x) < limb_start(y) : limb_weight(x) > limb_weight(y); });
        for (u32 t = 0; t < cnt; t++) {
            const u32 grp = t / LIMB_SLOTS;
            fill(grp * LIMB_SLOTS + used[grp], limb_list[t]);
            used[grp]++;
        }
        cc->limb_count = cnt; cc->limb_groups = G;
        for (u32 g = 0; g < G; g++) cc->limb_gcount[g] = used[g];
    }
    if (cc->limb_count) {                      // ComparisonGate (HBM-bound) rides with the VALU-bound limb launch
        std::vector<u32> keep;
        for (u32 gi : cc->single_gates) {
            if (cc->gates[gi].type == GLP_GATE_COMPARISON && cc->limb_extra_count < 4) cc->limb_extra_gi[cc->limb_extra_count++] = gi;
            else keep.push_back(gi);
        }
        cc->single_gates.swap(keep);
    }
    if (cc->limb_count) {
        GLP_TRY(c->alloc((void **)&cc->dev_limb_desc, desc.size() * 8));
        GLP_TRY(h2d(c, cc->dev_limb_desc, desc.data(), desc.size() * 8));
    }
    return GLP_OK;
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

void glp_circuit_free(glp_circuit *cc) {
    if (!cc) return;
    glp_ctx *c = cc->ctx;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    batch_destroy(cc->cs);
    c->release(cc->dev_sigmas);
    c->release(cc->dev_k_is);
    c->release(cc->dev_gates);
    c->release(cc->dev_limb_desc);
    c->release(cc->dev_consts);
    delete cc;
}

int glp_circuit_create(glp_ctx *c, const glp_circuit_desc *desc, glp_circuit **out) {
    GLP_REQUIRE(c && desc && out, "null argument");
    *out = nullptr;
    GLP_TRY(bind(c));
    const glp_circuit_desc &d = *desc;
    GLP_REQUIRE(d.gates && d.k_is && d.constants && d.sigmas, "null array in circuit description");
    GLP_REQUIRE(d.num_challenges >= 1 && d.num_challenges <= (u32)MAXCH, "num_challenges=%u outside 1..%d", d.num_challenges, MAXCH);
    if (d.hasher != GLP_HASH_POSEIDON && d.hasher != GLP_HASH_KECCAK25) return set_error(GLP_ERR_UNSUPPORTED, "hasher %u is not one of GLP_HASH_*", d.hasher);
    GLP_REQUIRE(d.rate_bits >= 1 && d.rate_bits <= 4, "rate_bits=%u outside 1..4", d.rate_bits);
    GLP_REQUIRE(d.num_routed_wires <= d.num_wires && d.num_routed_wires > 0, "bad wire counts");
    if ((int)d.degree_bits > NTT_MAX_LG) return set_error(GLP_ERR_UNSUPPORTED, "degree_bits=%u > %d", d.degree_bits, NTT_MAX_LG);
    const u32 qdf = d.quotient_degree_factor;
    if (qdf == 0 || (qdf & (qdf - 1)) || qdf > (1u << d.rate_bits))
        return set_error(GLP_ERR_UNSUPPORTED, "quotient_degree_factor=%u must be a power of two <= 2^rate_bits", qdf);
    GLP_REQUIRE(d.num_partial_products == (d.num_routed_wires + qdf - 1) / qdf - 1, "num_partial_products inconsistent");
    GLP_REQUIRE(d.num_reductions <= 16 && d.cap_height <= d.degree_bits + d.rate_bits, "bad FRI parameters");
    GLP_REQUIRE(d.proof_of_work_bits <= POW_MAX_BITS, "proof_of_work_bits=%u: this build searches at most 2^40 candidates and accepts up to %u bits",
                d.proof_of_work_bits, POW_MAX_BITS);
    u32 sum_ab = 0;
    for (u32 i = 0; i < d.num_reductions; i++) {
        GLP_REQUIRE(d.reduction_arity_bits[i] >= 1 && d.reduction_arity_bits[i] <= 5, "arity_bits outside 1..5");
        sum_ab += d.reduction_arity_bits[i];
        GLP_REQUIRE(sum_ab <= d.degree_bits && d.degree_bits + d.rate_bits - sum_ab >= d.cap_height, "FRI reduction deeper than the domain");
    }
    // Shapes the quotient kernel assumes, checked here so that a malformed description is an error and never an
    // out-of-bounds read on the device: wires / constants / constraints each gate type touches.
    auto gate_shape = [](const glp_gate &g, u32 &wires, u32 &consts, u32 &constraints) -> bool {
        const u32 p0 = g.p0, p1 = g.p1;
        consts = 0;
        switch (g.type) {
        case GLP_GATE_NOOP: wires = 0; constraints = 0; return true;
        case GLP_GATE_CONSTANT: wires = p0; consts = p0; constraints = p0; return true;
        case GLP_GATE_PUBLIC_INPUT: wires = 4; constraints = 4; return true;
        case GLP_GATE_ARITHMETIC: wires = 4 * p0; consts = 2; constraints = p0; return true;
        case GLP_GATE_POSEIDON: wires = 135; constraints = 123; return true;
        case GLP_GATE_U32_INTERLEAVE: wires = 34 * p0; constraints = 34 * p0; return true;
        case GLP_GATE_UNINTERLEAVE_U32: case GLP_GATE_UNINTERLEAVE_B32: wires = 67 * p0; constraints = 67 * p0; return true;
        case GLP_GATE_U32_ARITHMETIC: wires = 38 * p0; constraints = 36 * p0; return true;
        case GLP_GATE_U32_ADD_MANY: wires = (p0 + 3 + 18) * p1; constraints = 21 * p1; return p0 >= 1 && p0 <= 16;
        case GLP_GATE_U32_SUBTRACTION: wires = 21 * p0; constraints = 19 * p0; return true;
        case GLP_GATE_U32_RANGE_CHECK: wires = 17 * p0; constraints = 17 * p0; return true;
        case GLP_GATE_COMPARISON: {
            if (p1 == 0 || p0 == 0 || p0 > 64) return false;
            const u32 cb = (p0 + p1 - 1) / p1;
            if (cb > 4) return false;
            wires = 4 + 5 * p1 + cb + 1; constraints = 2 + 5 * p1 + 1 + (cb + 1) + 2; return true;
        }
        case GLP_GATE_BASE_SUM: wires = 1 + p0; constraints = 1 + p0; return p1 >= 2 && p1 <= 16;
        case GLP_GATE_RANDOM_ACCESS: {
            const u32 copies = p1 & 0xFFFF, nextra = p1 >> 16;
            if (p0 < 1 || p0 > 5) return false;
            wires = (2 + (1u << p0)) * copies + nextra + p0 * copies; consts = nextra; constraints = copies * (p0 + 2) + nextra;
            return true;
        }
        default: return false;
        }
    };
    u32 maxc = 0;
    for (u32 i = 0; i < d.num_gates; i++) {
        const glp_gate &g = d.gates[i];
        {
            u32 gw = 0, gcn = 0, gk = 0;
            if (!gate_shape(g, gw, gcn, gk))
                return set_error(g.type > GLP_GATE_RANDOM_ACCESS ? GLP_ERR_UNSUPPORTED : GLP_ERR_ARG,
                                 "gate %u: type %u with parameters (%u, %u) is not supported", i, g.type, g.p0, g.p1);
            GLP_REQUIRE(gw <= d.num_wires, "gate %u (type %u) needs %u wires, circuit has %u", i, g.type, gw, d.num_wires);
            GLP_REQUIRE(d.num_selectors + gcn <= d.num_constants, "gate %u (type %u) needs %u constants", i, g.type, gcn);
            GLP_REQUIRE(gk == g.num_constraints, "gate %u (type %u): num_constraints %u, expected %u", i, g.type, g.num_constraints, gk);
        }
        switch (g.type) {
        case GLP_GATE_NOOP: case GLP_GATE_CONSTANT: case GLP_GATE_PUBLIC_INPUT: case GLP_GATE_ARITHMETIC:
        case GLP_GATE_POSEIDON: if (g.type == GLP_GATE_POSEIDON && d.num_wires < 135) return set_error(GLP_ERR_ARG, "PoseidonGate needs 135 wires"); break;
        case GLP_GATE_U32_INTERLEAVE: case GLP_GATE_UNINTERLEAVE_U32: case GLP_GATE_UNINTERLEAVE_B32: break;
        case GLP_GATE_U32_ARITHMETIC: case GLP_GATE_U32_ADD_MANY: case GLP_GATE_U32_SUBTRACTION:
        case GLP_GATE_U32_RANGE_CHECK: case GLP_GATE_COMPARISON: case GLP_GATE_BASE_SUM: break;
        case GLP_GATE_RANDOM_ACCESS:
            GLP_REQUIRE(g.p0 >= 1 && g.p0 <= 5, "RandomAccessGate bits outside 1..5");
            break;
        default: break;
        }
        GLP_REQUIRE(g.selector_index < d.num_selectors && g.group_start <= g.row && g.row < g.group_end, "bad selector data for gate %u", i);
        GLP_REQUIRE(g.num_constraints <= ACC3_MAX_TERMS, "gate %u: %u constraints exceed the %u the quotient accumulators hold", i,
                    g.num_constraints, ACC3_MAX_TERMS);
        maxc = std::max(maxc, g.num_constraints);
    }
    GLP_REQUIRE(maxc <= d.num_gate_constraints, "num_gate_constraints smaller than a gate's constraint count");
    {   // field arrays from outside: canonical or rejected by name (one host pass, small against the uploads and the commitment below)
        const size_t nrows = (size_t)1 << d.degree_bits;
        const u64 *sec[3] = {d.k_is, d.constants, d.sigmas};
        const size_t cnt[3] = {d.num_routed_wires, (size_t)d.num_constants * nrows, (size_t)d.num_routed_wires * nrows};
        const char *names[3] = {"k_is", "constants", "sigmas"};
        for (int i = 0; i < 3; i++) {
            const size_t bad = first_noncanonical(sec[i], cnt[i]);
            GLP_REQUIRE(bad == cnt[i], "%s[%zu] = 0x%016llx is not a canonical field element (>= p)", names[i], bad, (unsigned long long)sec[i][bad]);
        }
    }

    std::unique_ptr<glp_circuit, void (*)(glp_circuit *)> cc(new glp_circuit(), glp_circuit_free);
    cc->ctx = c;
    cc->d = d;
    cc->gates.assign(d.gates, d.gates + d.num_gates);
    cc->k_is.assign(d.k_is, d.k_is + d.num_routed_wires);
    if (cc->k_is.size() >= 2 && cc->k_is[0] == 1 && cc->k_is[1] > 1 && cc->k_is[1] < (1ull << 32)) {
        cc->k_ratio = (u32)cc->k_is[1];
        for (size_t j = 1; j < cc->k_is.size(); j++)
            if (cc->k_is[j] != mul(cc->k_is[j - 1], (u64)cc->k_ratio)) { cc->k_ratio = 0; break; }
    }
    cc->d.gates = cc->gates.data(); cc->d.k_is = cc->k_is.data(); cc->d.constants = nullptr; cc->d.sigmas = nullptr;
    make_layout(cc->d, cc->L);
    const size_t n = (size_t)1 << d.degree_bits;
    const u32 nc = d.num_constants, nr = d.num_routed_wires;
    GLP_TRY(c->alloc((void **)&cc->dev_gates, sizeof(DevGate) * std::max<u32>(d.num_gates, 1)));
    GLP_TRY(c->alloc((void **)&cc->dev_k_is, (size_t)nr * 8));
    GLP_TRY(c->alloc((void **)&cc->dev_sigmas, (size_t)nr * n * 8));
    static_assert(sizeof(DevGate) == sizeof(glp_gate), "gate layout");
    GLP_TRY(h2d(c, cc->dev_gates, cc->gates.data(), sizeof(DevGate) * d.num_gates));
    GLP_TRY(h2d(c, cc->dev_k_is, cc->k_is.data(), (size_t)nr * 8));
    GLP_TRY(h2d(c, cc->dev_sigmas, d.sigmas, (size_t)nr * n * 8));
    GLP_TRY(c->alloc((void **)&cc->dev_consts, (size_t)nc * n * 8));
    GLP_TRY(h2d(c, cc->dev_consts, d.constants, (size_t)nc * n * 8));
    GLP_TRY(build_quotient_plan(c, cc.get()));
    {
        void *v = nullptr;
        GLP_TRY(c->alloc(&v, (size_t)(nc + nr) * n * 8));
        u64 *csv = (u64 *)v;
        int rc = GLP_OK;
        {
            hipError_t e = hipMemcpyAsync(csv, cc->dev_consts, (size_t)nc * n * 8, hipMemcpyDeviceToDevice, c->stream);
            if (e != hipSuccess) rc = set_error(GLP_ERR_HIP, "D2D copy: %s", hipGetErrorString(e));
        }
        if (rc == GLP_OK) {
            hipError_t e = hipMemcpyAsync(csv + (size_t)nc * n, cc->dev_sigmas, (size_t)nr * n * 8, hipMemcpyDeviceToDevice, c->stream);
            if (e != hipSuccess) rc = set_error(GLP_ERR_HIP, "D2D copy: %s", hipGetErrorString(e));
        }
        if (rc == GLP_OK) rc = batch_build(c, csv, BATCH_VALUES, nc + nr, (int)d.degree_bits, (int)d.rate_bits, (int)d.cap_height, &cc->cs, nullptr, 1, (int)d.hasher);
        (void)hipStreamSynchronize(c->stream);
        c->release(v);
        GLP_TRY(rc);
    }
    GLP_TRY(batch_cap_host(c, cc->cs, cc->cs_cap));
    bool zero = true;
    for (int i = 0; i < 4; i++) zero = zero && d.circuit_digest[i] == 0;
    if (zero && d.hasher == GLP_HASH_KECCAK25) {
        // the same recipe with C::Hasher = KeccakHash<25>: every hash enters as its four 7-byte chunks (BytesHash::to_vec)
        u64 pad[12] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1}, ds[4], e[4];
        kec::host_hash_no_pad(pad, 12, ds);
        std::vector<u64> parts;
        for (size_t i = 0; i < cc->cs_cap.size(); i += 4) { kec::digest_to_elements(&cc->cs_cap[i], e); parts.insert(parts.end(), e, e + 4); }
        kec::digest_to_elements(ds, e);
        parts.insert(parts.end(), e, e + 4);
        parts.push_back(d.degree_bits);
        kec::host_hash_no_pad(parts.data(), parts.size(), cc->digest);
    } else if (zero) {
        // hash_pad([]) = hash_no_pad([1, 0 x 10, 1]); digest = hash_no_pad(cap ++ that ++ [degree_bits])
        u64 pad[12] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1}, ds[4];
        host_hash_no_pad(pad, 12, ds);
        std::vector<u64> parts(cc->cs_cap);
        parts.insert(parts.end(), ds, ds + 4);
        parts.push_back(d.degree_bits);
        host_hash_no_pad(parts.data(), parts.size(), cc->digest);
    } else {
        memcpy(cc->digest, d.circuit_digest, 32);
    }
    memcpy(cc->d.circuit_digest, cc->digest, 32);
    *out = cc.release();
    return GLP_OK;
}

int glp_circuit_digest(const glp_circuit *cc, uint64_t out[4]) {
    GLP_REQUIRE(cc && out, "null argument");
    memcpy(out, cc->digest, 32);
    return GLP_OK;
}
int glp_circuit_constants_sigmas_cap(const glp_circuit *cc, uint64_t *cap_out) {
    GLP_REQUIRE(cc && cap_out, "null argument");
    memcpy(cap_out, cc->cs_cap.data(), cc->cs_cap.size() * 8);
    return GLP_OK;
}
size_t glp_proof_words(const glp_circuit *cc) { return cc ? cc->L.total : 0; }

// bytes of one digest on the wire: a Poseidon HashOut is 4 field elements, a KeccakHash<25> digest 25 bytes
static size_t digest_wire_bytes(const glp_circuit *cc) { return cc->d.hasher == GLP_HASH_KECCAK25 ? 25 : 32; }

size_t glp_proof_bytes_len(const glp_circuit *cc) {
    if (!cc) return 0;
    size_t bytes = 0;
    const size_t db = digest_wire_bytes(cc);
    walk_proof(cc, [&](size_t, size_t cnt, int kind) { bytes += kind == PW_FIELD ? cnt * 8 : (cnt / 4) * db + (kind == PW_PATH ? 1 : 0); });
    return bytes;
}

int glp_proof_to_bytes(const glp_circuit *cc, const uint64_t *words, uint8_t *out, size_t len) {
    GLP_REQUIRE(cc && words && out, "null argument");
    GLP_REQUIRE(len == glp_proof_bytes_len(cc), "bytes_len must equal glp_proof_bytes_len()");
    size_t o = 0;
    const bool kec25 = cc->d.hasher == GLP_HASH_KECCAK25;
    walk_proof(cc, [&](size_t off, size_t cnt, int kind) {
        if (kind == PW_PATH) out[o++] = (uint8_t)(cnt / 4);
        for (size_t i = 0; i < cnt; i++) {
            const u64 w = words[off + i];
            const int nb = (kind != PW_FIELD && kec25 && (i & 3) == 3) ? 1 : 8;        // last word of a 25-byte digest: one byte
            for (int b = 0; b < nb; b++) out[o++] = (uint8_t)(w >> (8 * b));
        }
    });
    return GLP_OK;
}

int glp_proof_from_bytes(const glp_circuit *cc, const uint8_t *in, size_t len, uint64_t *words) {
    GLP_REQUIRE(cc && words && in, "null argument");
    GLP_REQUIRE(len == glp_proof_bytes_len(cc), "byte length does not match this circuit");
    size_t o = 0;
    int bad = 0;
    const bool kec25 = cc->d.hasher == GLP_HASH_KECCAK25;
    walk_proof(cc, [&](size_t off, size_t cnt, int kind) {
        if (kind == PW_PATH && in[o++] != (uint8_t)(cnt / 4)) bad = 1;
        for (size_t i = 0; i < cnt; i++) {
            const bool dig = kind != PW_FIELD && kec25;
            const int nb = (dig && (i & 3) == 3) ? 1 : 8;
            u64 w = 0;
            for (int b = 0; b < nb; b++) w |= (u64)in[o++] << (8 * b);
            if (!dig && w >= glf::P) bad = 2;                // field elements and Poseidon digests are canonical; Keccak digests are bytes
            words[off + i] = w;
        }
    });
    if (bad == 1) return set_error(GLP_ERR_ARG, "Merkle path length byte does not match the circuit's FRI parameters");
    if (bad == 2) return set_error(GLP_ERR_ARG, "non-canonical field element in proof bytes");
    return GLP_OK;
}

int glp_session_begin(glp_ctx *c, const glp_circuit *cc, const uint64_t *wires, int wires_on_device, const uint64_t *public_inputs,
                      glp_session **out, uint64_t *wires_cap_out, uint64_t public_inputs_hash_out[4]) {
    GLP_REQUIRE(c && cc && wires && out && wires_cap_out && public_inputs_hash_out, "null argument");
    *out = nullptr;
    GLP_REQUIRE(cc->ctx == c, "circuit belongs to another context");
    GLP_REQUIRE(public_inputs || cc->d.num_public_inputs == 0, "public_inputs is null");
    GLP_TRY(bind(c));
    std::unique_ptr<glp_session> s(new glp_session(c, cc));
    const u64 *dw = wires;
    if (!wires_on_device) {
        const size_t tot = (size_t)cc->d.num_wires << cc->d.degree_bits;
        void *dv = nullptr;
        GLP_TRY(c->alloc(&dv, tot * 8));
        s->owned_wires = (u64 *)dv;
        dw = s->owned_wires;
    }
    GLP_TRY(s->begin(dw, public_inputs, wires_on_device ? nullptr : wires));
    memcpy(wires_cap_out, s->cap.data(), (size_t)s->capn * 32);
    memcpy(public_inputs_hash_out, s->pih, 32);
    *out = s.release();
    return GLP_OK;
}
#define GLP_SESSION_ENTER(S)                         \
    GLP_REQUIRE((S) != nullptr, "null session");     \
    GLP_TRY(bind((S)->c))
int glp_session_partial_products(glp_session *s, const uint64_t *betas, const uint64_t *gammas, uint64_t *zs_cap_out) {
    GLP_SESSION_ENTER(s);
    GLP_REQUIRE(betas && gammas && zs_cap_out, "null argument");
    for (u32 i = 0; i < s->nch; i++) GLP_REQUIRE(betas[i] < P && gammas[i] < P, "challenge %u is not a canonical field element", i);
    GLP_TRY(s->partial_products(betas, gammas));
    memcpy(zs_cap_out, s->cap.data(), (size_t)s->capn * 32);
    return GLP_OK;
}
int glp_session_quotient(glp_session *s, const uint64_t *alphas, uint64_t *quotient_cap_out) {
    GLP_SESSION_ENTER(s);
    GLP_REQUIRE(alphas && quotient_cap_out, "null argument");
    for (u32 i = 0; i < s->nch; i++) GLP_REQUIRE(alphas[i] < P, "challenge %u is not a canonical field element", i);
    GLP_TRY(s->quotient(alphas));
    memcpy(quotient_cap_out, s->cap.data(), (size_t)s->capn * 32);
    return GLP_OK;
}
size_t glp_num_openings(const glp_circuit *cc) { return cc ? cc->L.nopen : 0; }
size_t glp_final_poly_len(const glp_circuit *cc) { return cc ? cc->L.final_len : 0; }
int glp_session_open(glp_session *s, const uint64_t zeta[2], uint64_t *openings_out) {
    GLP_SESSION_ENTER(s);
    GLP_REQUIRE(zeta && openings_out, "null argument");
    GLP_REQUIRE(zeta[0] < P && zeta[1] < P, "zeta is not canonical");
    GLP_TRY(s->open_at(e_make(zeta[0], zeta[1])));
    memcpy(openings_out, s->proof() + s->L.openings, s->L.nopen * 16);
    return GLP_OK;
}
int glp_session_fri_combine(glp_session *s, const uint64_t alpha[2]) {
    GLP_SESSION_ENTER(s);
    GLP_REQUIRE(alpha && alpha[0] < P && alpha[1] < P, "alpha is null or not canonical");
    return s->fri_combine(e_make(alpha[0], alpha[1]));
}
int glp_session_fri_commit(glp_session *s, uint64_t *cap_out) {
    GLP_SESSION_ENTER(s);
    GLP_REQUIRE(cap_out, "null argument");
    GLP_TRY(s->fri_commit_layer());
    memcpy(cap_out, s->cap.data(), (size_t)s->capn * 32);
    return GLP_OK;
}
int glp_session_fri_fold(glp_session *s, const uint64_t beta[2]) {
    GLP_SESSION_ENTER(s);
    GLP_REQUIRE(beta && beta[0] < P && beta[1] < P, "beta is null or not canonical");
    return s->fri_fold(e_make(beta[0], beta[1]));
}
int glp_session_fri_final_poly(glp_session *s, uint64_t *coeffs_out) {
    GLP_SESSION_ENTER(s);
    GLP_REQUIRE(coeffs_out, "null argument");
    GLP_TRY(s->fri_final_poly());
    memcpy(coeffs_out, s->proof() + s->L.final_poly, s->L.final_len * 16);
    return GLP_OK;
}
int glp_pow_search(glp_ctx *c, const uint64_t sponge_state[12], const uint64_t *pending_inputs, uint32_t num_pending, uint32_t bits,
                   uint64_t *witness_out) {
    GLP_REQUIRE(c && sponge_state && witness_out && (pending_inputs || num_pending == 0), "null argument");
    GLP_REQUIRE(bits <= POW_MAX_BITS, "proof_of_work_bits=%u: this build searches at most 2^40 candidates and accepts up to %u bits", bits,
                POW_MAX_BITS);
    GLP_TRY(bind(c));
    return pow_search(c, sponge_state, pending_inputs, num_pending, bits, witness_out);
}
int glp_pow_search_h(glp_ctx *c, uint32_t hasher, const uint64_t sponge_state[12], const uint64_t *pending_inputs, uint32_t num_pending,
                     uint32_t bits, uint64_t *witness_out) {
    GLP_REQUIRE(c && sponge_state && witness_out && (pending_inputs || num_pending == 0), "null argument");
    GLP_REQUIRE(bits <= POW_MAX_BITS, "proof_of_work_bits=%u: this build searches at most 2^40 candidates and accepts up to %u bits", bits,
                POW_MAX_BITS);
    if (hasher != GLP_HASH_POSEIDON && hasher != GLP_HASH_KECCAK25) return set_error(GLP_ERR_UNSUPPORTED, "hasher %u is not one of GLP_HASH_*", hasher);
    GLP_TRY(bind(c));
    return pow_search(c, sponge_state, pending_inputs, num_pending, bits, witness_out, (int)hasher);
}
int glp_session_queries(glp_session *s, uint64_t pow_witness, const uint64_t *indices, uint32_t num_indices) {
    GLP_SESSION_ENTER(s);
    GLP_REQUIRE(indices, "null argument");
    return s->queries(pow_witness, indices, num_indices);
}
int glp_session_proof(glp_session *s, uint64_t *proof_out) {
    GLP_SESSION_ENTER(s);
    GLP_REQUIRE(proof_out, "null argument");
    GLP_REQUIRE(s->stage == glp_session::S_DONE, "the proof is not finished (call glp_session_queries first)");
    memcpy(proof_out, s->proof(), s->L.total * 8);
    return GLP_OK;
}
void glp_session_end(glp_session *s) {
    if (!s) return;
    (void)hipSetDevice(s->c->device);
    delete s;
}
int glp_prove_device(glp_ctx *c, const glp_circuit *cc, const uint64_t *dev_wires, const uint64_t *public_inputs, uint64_t *proof_out) {
    GLP_REQUIRE(c && cc && dev_wires && proof_out, "null argument");
    GLP_REQUIRE(cc->ctx == c, "circuit belongs to another context");
    GLP_REQUIRE(public_inputs || cc->d.num_public_inputs == 0, "public_inputs is null");
    GLP_TRY(bind(c));
    return prove_impl(c, cc, dev_wires, public_inputs, proof_out);
}

int glp_prove(glp_ctx *c, const glp_circuit *cc, const uint64_t *wires, const uint64_t *public_inputs, uint64_t *proof_out) {
    GLP_REQUIRE(c && cc && wires && proof_out, "null argument");
    GLP_REQUIRE(cc->ctx == c, "circuit belongs to another context");
    GLP_REQUIRE(public_inputs || cc->d.num_public_inputs == 0, "public_inputs is null");
    GLP_TRY(bind(c));
    const size_t tot = (size_t)cc->d.num_wires << cc->d.degree_bits;
    void *dv = nullptr;
    GLP_TRY(c->alloc(&dv, tot * 8));
    const int rc = prove_impl(c, cc, (const u64 *)dv, public_inputs, proof_out, wires);
    (void)hipStreamSynchronize(c->stream);
    c->release(dv);
    return rc;
}

// ---- staged witnesses (include/glp.h "witnesses that start in host memory, pipelined")
}  // extern "C"
struct glp_witness {
    glp_ctx *c = nullptr;
    const glp_circuit *cc = nullptr;
    u64 *dev = nullptr;            // [num_wires][n], from the context's pool
    hipEvent_t ready = nullptr;    // recorded on the copy stream behind the upload
    bool routed_only = false, filled = false;
};
extern "C" {
int glp_host_alloc(glp_ctx *c, size_t bytes, void **host_out) {
    GLP_REQUIRE(c && host_out && bytes > 0, "null argument or zero size");
    *host_out = nullptr;
    GLP_TRY(bind(c));
    GLP_HIP(hipHostMalloc(host_out, bytes, hipHostMallocDefault));
    return GLP_OK;
}
int glp_host_free(glp_ctx *c, void *host) {
    GLP_REQUIRE(c, "null context");
    if (!host) return GLP_OK;
    GLP_TRY(bind(c));
    GLP_HIP(hipHostFree(host));
    return GLP_OK;
}
void glp_witness_free(glp_witness *w) {
    if (!w) return;
    (void)hipSetDevice(w->c->device);
    if (w->ready) { (void)hipEventSynchronize(w->ready); (void)hipEventDestroy(w->ready); }
    if (w->dev) { (void)hipStreamSynchronize(w->c->stream); w->c->release(w->dev); }
    delete w;
}
int glp_witness_stage(glp_ctx *c, const glp_circuit *cc, const uint64_t *host_wires, uint32_t flags, glp_witness **out) {
    GLP_REQUIRE(c && cc && host_wires && out, "null argument");
    *out = nullptr;
    GLP_REQUIRE(cc->ctx == c, "circuit belongs to another context");
    GLP_REQUIRE((flags & ~GLP_WITNESS_ROUTED_ONLY) == 0, "unknown flags 0x%x", flags);
    GLP_TRY(bind(c));
    const size_t n = (size_t)1 << cc->d.degree_bits;
    const u32 nw = cc->d.num_wires, nr = cc->d.num_routed_wires;
    std::unique_ptr<glp_witness, void (*)(glp_witness *)> w(new glp_witness(), glp_witness_free);
    w->c = c; w->cc = cc; w->routed_only = (flags & GLP_WITNESS_ROUTED_ONLY) != 0;
    void *dv = nullptr;
    GLP_TRY(c->alloc(&dv, (size_t)nw * n * 8));
    w->dev = (u64 *)dv;
    GLP_HIP(hipEventCreateWithFlags(&w->ready, hipEventDisableTiming));
    const u32 ncopy = w->routed_only ? nr : nw;
    GLP_HIP(hipMemcpyAsync(w->dev, host_wires, (size_t)ncopy * n * 8, hipMemcpyHostToDevice, c->copy_stream));
    if (ncopy < nw) GLP_HIP(hipMemsetAsync(w->dev + (size_t)ncopy * n, 0, (size_t)(nw - ncopy) * n * 8, c->copy_stream));
    GLP_HIP(hipEventRecord(w->ready, c->copy_stream));
    *out = w.release();
    return GLP_OK;
}
int glp_prove_staged(glp_ctx *c, const glp_circuit *cc, glp_witness *w, const uint64_t *public_inputs, uint64_t *proof_out) {
    GLP_REQUIRE(c && cc && w && proof_out, "null argument");
    GLP_REQUIRE(cc->ctx == c && w->c == c && w->cc == cc, "witness, circuit and context do not belong together");
    GLP_REQUIRE(public_inputs || cc->d.num_public_inputs == 0, "public_inputs is null");
    GLP_TRY(bind(c));
    GLP_HIP(hipStreamWaitEvent(c->stream, w->ready, 0));
    if (w->routed_only && !w->filled) {
        GLP_TRY(glp_witness_fill(c, cc, w->dev, 1));
        w->filled = true;
    }
    return prove_impl(c, cc, w->dev, public_inputs, proof_out);
}
}  // extern "C"

#include <stdlib.h>
#include "prover_batch.inc"
#include "prover_batch_dev.inc"
