// api.hip -- the extern "C" boundary of libglprover.so (declared in include/glp.h): context,
// device-memory pool, stage timers, primitive entry points and PolynomialBatch.
#include <stdarg.h>
#include <string.h>
#include "batch.h"
#include "common.h"
#include "keccak.h"
#include "merkle.h"
#include "ntt.h"

namespace glp {
thread_local std::string g_last_error;
int set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
}  // namespace glp
using namespace glp;

// ------------------------------------------------------------------------------------------ ctx
int glp_ctx::alloc(void **p, size_t bytes) {
    if (bytes == 0) bytes = 8;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = pool.find(bytes);
    if (it != pool.end()) {
        *p = it->second;
        pool.erase(it);
        pool_bytes -= bytes;
        live[*p] = bytes;
        return GLP_OK;
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        trim();  // drop cached blocks of other sizes and retry once
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) return set_error(GLP_ERR_HIP, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    live[*p] = bytes;
    return GLP_OK;
}
void glp_ctx::release(void *p) {
    if (!p) return;
    auto it = live.find(p);
    if (it == live.end()) return;
    pool.insert({it->second, p});
    pool_bytes += it->second;
    live.erase(it);
}
void glp_ctx::trim() {
    (void)hipStreamSynchronize(stream);
    for (auto &kv : pool) (void)hipFree(kv.second);
    pool.clear();
    pool_bytes = 0;
}
int glp_ctx::stage_begin(const char *name, double bytes) {
    if (!profiling) return GLP_OK;
    Stage s;
    s.name = name; s.bytes = bytes;
    if (hipEventCreate(&s.beg) != hipSuccess || hipEventCreate(&s.end) != hipSuccess) return GLP_ERR_HIP;
    (void)hipEventRecord(s.beg, stream);
    stages.push_back(s);
    return GLP_OK;
}
int glp_ctx::stage_end() {
    if (!profiling || stages.empty()) return GLP_OK;
    (void)hipEventRecord(stages.back().end, stream);
    return GLP_OK;
}

extern "C" {

const char *glp_last_error(void) { return g_last_error.c_str(); }
const char *glp_version(void) { return "glprover 0.1 (gfx950)"; }

int glp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int glp_ctx_create(int device_id, glp_ctx **out) {
    GLP_REQUIRE(out != nullptr, "glp_ctx_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return set_error(GLP_ERR_NOGPU, "no HIP device visible (%s); libglprover has no CPU fallback",
                         e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    GLP_REQUIRE(device_id >= 0 && device_id < n, "device_id %d out of range (0..%d)", device_id, n - 1);
    GLP_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    GLP_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(GLP_ERR_NOGPU, "device %d is %s; this library carries gfx950 code objects only", device_id,
                         prop.gcnArchName);
    std::unique_ptr<glp_ctx> c(new glp_ctx());
    c->device = device_id;
    c->num_cus = prop.multiProcessorCount;
    // tuning switches (profiles/ A-B runs; documented in include/glp.h): the defaults are what ships
    if (const char *e1 = getenv("GLP_NTT_2PASS_LG")) { const int v = atoi(e1); if (v >= NTT_INNER_LG && v <= NTT_2PASS_LG) c->two_pass_lg = v; }
    if (const char *e2 = getenv("GLP_NTT_STRIDED32_TL")) { const int v = atoi(e2); if (v >= 1 && v <= 64) c->strided32_tl = v; }
    if (const char *e3 = getenv("GLP_NTT_STRIDED32_LW")) { const int v = atoi(e3); if (v == 3 || v == 4) c->strided32_lw = v; }
    if (const char *e4 = getenv("GLP_MERKLE_COOP_MAX")) c->merkle_coop_max = (size_t)strtoull(e4, nullptr, 10);
    if (const char *e5 = getenv("GLP_MERKLE_QUAD_MAX")) c->merkle_quad_max = (size_t)strtoull(e5, nullptr, 10);
    GLP_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        hipError_t e2 = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
        if (e2 != hipSuccess) {
            (void)hipStreamDestroy(c->stream);
            return set_error(GLP_ERR_HIP, "hipStreamCreateWithFlags (copy stream): %s", hipGetErrorString(e2));
        }
    }
    *out = c.release();
    return GLP_OK;
}

void glp_ctx_destroy(glp_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    glp_ctx_stage_reset(c);
    if (c->host_pool && c->host_pool_free) c->host_pool_free(c->host_pool);
    free_plans(c);
    c->trim();
    for (auto &kv : c->live) (void)hipFree(kv.first);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int glp_ctx_synchronize(glp_ctx *c) {
    GLP_REQUIRE(c, "null ctx");
    GLP_TRY(bind(c));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}
void *glp_ctx_stream(glp_ctx *c) { return c ? (void *)c->stream : nullptr; }

int glp_dev_alloc(glp_ctx *c, size_t bytes, void **dev_out) {
    GLP_REQUIRE(c && dev_out && bytes > 0, "null argument or zero size");
    *dev_out = nullptr;
    GLP_TRY(bind(c));
    return c->alloc(dev_out, bytes);
}
int glp_dev_free(glp_ctx *c, void *dev) {
    GLP_REQUIRE(c, "null context");
    if (!dev) return GLP_OK;
    GLP_TRY(bind(c));
    GLP_REQUIRE(c->live.count(dev), "pointer was not allocated by glp_dev_alloc on this context");
    GLP_HIP(hipStreamSynchronize(c->stream));
    c->release(dev);
    return GLP_OK;
}
// the live block of this context that contains [p, p + bytes), or an error: a copy never runs past an allocation
static int check_dev_range(glp_ctx *c, const void *p, size_t bytes) {
    auto it = c->live.upper_bound(const_cast<void *>(p));
    GLP_REQUIRE(it != c->live.begin(), "device pointer is not inside an allocation of this context");
    --it;
    const char *b0 = (const char *)it->first, *q = (const char *)p;
    GLP_REQUIRE(q >= b0 && (size_t)(q - b0) <= it->second && bytes <= it->second - (size_t)(q - b0),
                "%zu bytes at offset %zu run past the end of a %zu-byte device block", bytes, (size_t)(q - b0), it->second);
    return GLP_OK;
}
int glp_dev_upload(glp_ctx *c, void *dev_dst, const void *host_src, size_t bytes) {
    GLP_REQUIRE(c && dev_dst && host_src, "null argument");
    GLP_TRY(bind(c));
    GLP_TRY(check_dev_range(c, dev_dst, bytes));
    GLP_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}
int glp_dev_download(glp_ctx *c, void *host_dst, const void *dev_src, size_t bytes) {
    GLP_REQUIRE(c && host_dst && dev_src, "null argument");
    GLP_TRY(bind(c));
    GLP_TRY(check_dev_range(c, dev_src, bytes));
    GLP_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}
int glp_ctx_set_profiling(glp_ctx *c, int on) {
    GLP_REQUIRE(c, "null ctx");
    c->profiling = on != 0;
    return GLP_OK;
}
int glp_ctx_stage_reset(glp_ctx *c) {
    GLP_REQUIRE(c, "null ctx");
    (void)hipStreamSynchronize(c->stream);
    for (auto &s : c->stages) { (void)hipEventDestroy(s.beg); (void)hipEventDestroy(s.end); }
    c->stages.clear();
    return GLP_OK;
}
int glp_ctx_stage_count(glp_ctx *c) { return c ? (int)c->stages.size() : 0; }
int glp_ctx_stage_get(glp_ctx *c, int index, const char **name, float *ms, double *bytes) {
    GLP_REQUIRE(c && index >= 0 && index < (int)c->stages.size(), "stage index out of range");
    Stage &s = c->stages[index];
    GLP_HIP(hipEventSynchronize(s.end));
    float t = 0;
    GLP_HIP(hipEventElapsedTime(&t, s.beg, s.end));
    if (name) *name = s.name.c_str();
    if (ms) *ms = t;
    if (bytes) *bytes = s.bytes;
    return GLP_OK;
}

// ------------------------------------------------------------------------------------------ primitives
struct Scratch {  // RAII for pool blocks inside one API call
    glp_ctx *c;
    std::vector<void *> ptrs;
    explicit Scratch(glp_ctx *ctx) : c(ctx) {}
    ~Scratch() { (void)hipStreamSynchronize(c->stream); for (void *p : ptrs) c->release(p); }
    int get(u64 **p, size_t elems) {
        void *v = nullptr;
        int rc = c->alloc(&v, elems * sizeof(u64));
        if (rc == GLP_OK) { ptrs.push_back(v); *p = (u64 *)v; }
        return rc;
    }
};

int glp_poseidon_permute(glp_ctx *c, uint64_t *states, size_t count) {
    GLP_REQUIRE(c && (states || !count), "null argument");
    GLP_TRY(bind(c));
    if (!count) return GLP_OK;
    Scratch s(c);
    u64 *d;
    GLP_TRY(s.get(&d, count * 12));
    GLP_HIP(hipMemcpyAsync(d, states, count * 96, hipMemcpyHostToDevice, c->stream));
    GLP_TRY(poseidon_permute_states(c, d, count));
    GLP_HIP(hipMemcpyAsync(states, d, count * 96, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

int glp_fft(glp_ctx *c, uint64_t *cols, uint32_t ncols, uint32_t log_n) {
    GLP_REQUIRE(c && (cols || !ncols), "null argument");
    GLP_TRY(bind(c));
    if (!ncols) return GLP_OK;
    if (log_n > (uint32_t)NTT_MAX_LG) return set_error(GLP_ERR_UNSUPPORTED, "log_n=%u > %d", log_n, NTT_MAX_LG);
    const size_t tot = (size_t)ncols << log_n;
    Scratch s(c);
    u64 *a, *b;
    GLP_TRY(s.get(&a, tot));
    GLP_TRY(s.get(&b, tot));
    GLP_HIP(hipMemcpyAsync(a, cols, tot * 8, hipMemcpyHostToDevice, c->stream));
    GLP_TRY(bitrev_copy(c, a, b, ncols, (int)log_n));
    GLP_TRY(ntt_coeffs_to_values(c, b, a, ncols, (int)log_n));
    GLP_HIP(hipMemcpyAsync(cols, a, tot * 8, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

int glp_ifft(glp_ctx *c, uint64_t *cols, uint32_t ncols, uint32_t log_n) {
    GLP_REQUIRE(c && (cols || !ncols), "null argument");
    GLP_TRY(bind(c));
    if (!ncols) return GLP_OK;
    if (log_n > (uint32_t)NTT_MAX_LG) return set_error(GLP_ERR_UNSUPPORTED, "log_n=%u > %d", log_n, NTT_MAX_LG);
    const size_t tot = (size_t)ncols << log_n;
    Scratch s(c);
    u64 *a, *b;
    GLP_TRY(s.get(&a, tot));
    GLP_TRY(s.get(&b, tot));
    GLP_HIP(hipMemcpyAsync(a, cols, tot * 8, hipMemcpyHostToDevice, c->stream));
    GLP_TRY(intt_values_to_coeffs(c, a, b, ncols, (int)log_n));
    GLP_TRY(bitrev_copy(c, b, a, ncols, (int)log_n));
    GLP_HIP(hipMemcpyAsync(cols, a, tot * 8, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

int glp_lde(glp_ctx *c, const uint64_t *coeffs, uint32_t ncols, uint32_t log_n, uint32_t rate_bits, uint64_t shift,
            uint64_t *out) {
    GLP_REQUIRE(c && ((coeffs && out) || !ncols), "null argument");
    GLP_REQUIRE(shift != 0 && shift < glf::P, "shift must be a nonzero canonical field element");
    if (rate_bits > 4) return set_error(GLP_ERR_UNSUPPORTED, "rate_bits=%u outside 0..4", rate_bits);
    GLP_TRY(bind(c));
    if (!ncols) return GLP_OK;
    if (log_n > (uint32_t)NTT_MAX_LG) return set_error(GLP_ERR_UNSUPPORTED, "log_n=%u > %d", log_n, NTT_MAX_LG);
    const size_t tot = (size_t)ncols << log_n, TOT = tot << rate_bits;
    Scratch s(c);
    u64 *a, *b, *l, *o;
    GLP_TRY(s.get(&a, tot));
    GLP_TRY(s.get(&b, tot));
    GLP_TRY(s.get(&l, TOT));
    GLP_TRY(s.get(&o, TOT));
    GLP_HIP(hipMemcpyAsync(a, coeffs, tot * 8, hipMemcpyHostToDevice, c->stream));
    GLP_TRY(bitrev_copy(c, a, b, ncols, (int)log_n));
    GLP_TRY(lde_coeffs(c, b, l, ncols, (int)log_n, (int)rate_bits, shift));
    GLP_TRY(lde_to_natural(c, l, o, ncols, (int)log_n, (int)rate_bits));
    GLP_HIP(hipMemcpyAsync(out, o, TOT * 8, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

}  // extern "C"

__global__ void k_fill_random(u64 *out, size_t count, u64 seed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    out[i] = glf::canon(z);
}

// one message per thread, bytes packed into lanes on the fly
__global__ void k_keccak256(const uint8_t *msgs, size_t count, size_t len, uint8_t *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint8_t *m = msgs + i * len;
    kec::Sponge s;
    kec::sponge_init(s);
    size_t off = 0;
    for (; off + 8 <= len; off += 8) {
        u64 lane = 0;
        for (int b = 0; b < 8; b++) lane |= (u64)m[off + b] << (8 * b);
        kec::sponge_absorb(s, lane);
    }
    u64 extra = 0;
    for (size_t b = 0; off + b < len; b++) extra |= (u64)m[off + b] << (8 * b);
    kec::sponge_finish(s, extra, (int)(len - off));
    for (int w = 0; w < 4; w++) for (int b = 0; b < 8; b++) out[i * 32 + 8 * w + b] = (uint8_t)(s.a[w] >> (8 * b));
}
extern "C" int glp_keccak256(glp_ctx *c, const uint8_t *msgs, size_t count, size_t len, uint8_t *digests_out) {
    GLP_REQUIRE(c && digests_out && (msgs || !len), "null argument");
    GLP_REQUIRE(len == 0 || count <= ((size_t)1 << 40) / len, "count * len = %zu * %zu bytes is more than this entry point stages (2^40)", count, len);
    GLP_TRY(bind(c));
    if (!count) return GLP_OK;
    Scratch s(c);
    u64 *dm, *dd;
    GLP_TRY(s.get(&dm, (count * len + 7) / 8 + 1));
    GLP_TRY(s.get(&dd, count * 4));
    if (len) GLP_HIP(hipMemcpyAsync(dm, msgs, count * len, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_keccak256, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, c->stream, (const uint8_t *)dm, count, len, (uint8_t *)dd);
    GLP_HIP(hipGetLastError());
    GLP_HIP(hipMemcpyAsync(digests_out, dd, count * 32, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

extern "C" int glp_fill_random_device(glp_ctx *c, uint64_t *dev_out, size_t count, uint64_t seed) {
    GLP_REQUIRE(c && (dev_out || !count), "null argument");
    GLP_TRY(bind(c));
    if (!count) return GLP_OK;
    hipLaunchKernelGGL(k_fill_random, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, dev_out, count, seed);
    GLP_HIP(hipGetLastError());
    return GLP_OK;
}

// ------------------------------------------------------------------------------------------ PolynomialBatch

namespace glp {

void batch_destroy(glp_batch *b) {
    if (!b) return;
    glp_ctx *c = b->ctx;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->release(b->coeffs);
    c->release(b->lde);
    c->release(b->digests);
    delete b;
}

// dev_in: values (natural) if from_values, else coefficients in natural order.
// host_src != nullptr (BATCH_VALUES only): the values are still in host memory; they are copied into dev_in in column
// chunks on the copy stream while the transforms of the chunks already on the device run on the compute stream.
int batch_build(glp_ctx *c, const u64 *dev_in, int input_kind, u32 ncols, int lg, int rate_bits, int cap_height,
                glp_batch **out, const u64 *host_src, u32 K, int hasher) {
    GLP_REQUIRE(out, "out is null");
    *out = nullptr;
    if (hasher != GLP_HASH_POSEIDON && hasher != GLP_HASH_KECCAK25) return set_error(GLP_ERR_UNSUPPORTED, "hasher %d is not one of GLP_HASH_*", hasher);
    GLP_REQUIRE(ncols > 0, "ncols must be positive");
    GLP_REQUIRE(K >= 1 && (K == 1 || host_src == nullptr), "bad batch arguments");
    if (K > 1) {
        // K oracles of one shape: the transforms see K * ncols independent columns, the trees get a proof index
        GLP_REQUIRE(rate_bits >= 0 && rate_bits <= 4 && cap_height >= 0 && cap_height <= lg + rate_bits && lg <= NTT_MAX_LG, "bad batch shape");
        const size_t n = (size_t)1 << lg, N = n << rate_bits;
        std::unique_ptr<glp_batch, void (*)(glp_batch *)> b(new glp_batch(), batch_destroy);
        b->ctx = c; b->ncols = ncols; b->lg = lg; b->rate_bits = rate_bits; b->cap_height = cap_height; b->K = K; b->hasher = hasher;
        b->ndigests = merkle_num_digests(N, cap_height);
        const size_t tot = (size_t)K * ncols;
        GLP_REQUIRE(tot <= 0x7FFFFFFFu, "batch too wide");
        GLP_TRY(c->alloc((void **)&b->coeffs, tot * n * 8));
        GLP_TRY(c->alloc((void **)&b->lde, tot * N * 8));
        GLP_TRY(c->alloc((void **)&b->digests, (size_t)K * b->ndigests * 32));
        if (input_kind == BATCH_VALUES) GLP_TRY(intt_values_to_coeffs(c, dev_in, b->coeffs, (u32)tot, lg));
        else if (input_kind == BATCH_COEFFS_NATURAL) GLP_TRY(bitrev_copy(c, dev_in, b->coeffs, (u32)tot, lg));
        else GLP_HIP(hipMemcpyAsync(b->coeffs, dev_in, tot * n * 8, hipMemcpyDeviceToDevice, c->stream));
        GLP_TRY(lde_coeffs(c, b->coeffs, b->lde, (u32)tot, lg, rate_bits, glf::GEN));
        GLP_TRY(merkle_from_lde(c, b->lde, ncols, lg, rate_bits, cap_height, b->digests, K, (size_t)ncols * N, b->ndigests * 4, hasher));
        *out = b.release();
        return GLP_OK;
    }
    if (lg > NTT_MAX_LG) return set_error(GLP_ERR_UNSUPPORTED, "log_n=%d > %d (three-pass NTT not built yet)", lg, NTT_MAX_LG);
    GLP_REQUIRE(rate_bits >= 0 && rate_bits <= 4, "rate_bits=%d outside 0..4", rate_bits);
    GLP_REQUIRE(cap_height >= 0 && cap_height <= lg + rate_bits, "cap_height=%d should be at most log2(leaves)=%d", cap_height,
                lg + rate_bits);
    const size_t n = (size_t)1 << lg, N = n << rate_bits;
    std::unique_ptr<glp_batch, void (*)(glp_batch *)> b(new glp_batch(), batch_destroy);
    b->ctx = c; b->ncols = ncols; b->lg = lg; b->rate_bits = rate_bits; b->cap_height = cap_height; b->hasher = hasher;
    b->ndigests = merkle_num_digests(N, cap_height);
    GLP_TRY(c->alloc((void **)&b->coeffs, (size_t)ncols * n * 8));
    GLP_TRY(c->alloc((void **)&b->lde, (size_t)ncols * N * 8));
    GLP_TRY(c->alloc((void **)&b->digests, b->ndigests * 32));
    if (input_kind == BATCH_VALUES && host_src != nullptr) {
        // chunking pays when a chunk's copy is long against a few launches: below 64 MB the witness goes up in one piece
        const u32 nchunks = ((size_t)ncols * n * 8 < ((size_t)64 << 20)) ? 1u : std::min<u32>(8, ncols), per = (ncols + nchunks - 1) / nchunks;
        int rc = GLP_OK;
        std::vector<hipEvent_t> evs;
        for (u32 c0 = 0; c0 < ncols && rc == GLP_OK; c0 += per) {
            const u32 cn = std::min(per, ncols - c0);
            hipEvent_t ev = nullptr;
            hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) evs.push_back(ev);
            if (e == hipSuccess) e = hipMemcpyAsync(const_cast<u64 *>(dev_in) + (size_t)c0 * n, host_src + (size_t)c0 * n, (size_t)cn * n * 8,
                                                    hipMemcpyHostToDevice, c->copy_stream);
            if (e == hipSuccess) e = hipEventRecord(ev, c->copy_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ev, 0);
            if (e != hipSuccess) { rc = set_error(GLP_ERR_HIP, "witness upload: %s", hipGetErrorString(e)); break; }
            {
                StageScope st(c, "intt", 16.0 * n * cn);
                rc = intt_values_to_coeffs(c, dev_in + (size_t)c0 * n, b->coeffs + (size_t)c0 * n, cn, lg);
            }
            if (rc == GLP_OK) {
                StageScope st(c, "lde", (8.0 * n + 8.0 * N) * cn);
                rc = lde_coeffs(c, b->coeffs + (size_t)c0 * n, b->lde + (size_t)c0 * N, cn, lg, rate_bits, glf::GEN);
            }
        }
        if (rc != GLP_OK || !evs.empty()) (void)hipStreamSynchronize(c->copy_stream);
        for (hipEvent_t ev : evs) (void)hipEventDestroy(ev);
        GLP_TRY(rc);
        GLP_TRY(merkle_from_lde(c, b->lde, ncols, lg, rate_bits, cap_height, b->digests, 1, 0, 0, hasher));
        *out = b.release();
        return GLP_OK;
    }
    if (input_kind == BATCH_VALUES) {
        StageScope st(c, "intt", 16.0 * n * ncols);
        GLP_TRY(intt_values_to_coeffs(c, dev_in, b->coeffs, ncols, lg));
    } else if (input_kind == BATCH_COEFFS_NATURAL) {
        StageScope st(c, "bitrev_coeffs", 16.0 * n * ncols);
        GLP_TRY(bitrev_copy(c, dev_in, b->coeffs, ncols, lg));
    } else {
        StageScope st(c, "copy_coeffs", 16.0 * n * ncols);
        GLP_HIP(hipMemcpyAsync(b->coeffs, dev_in, (size_t)ncols * n * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    {
        StageScope st(c, "lde", (8.0 * n + 8.0 * N) * ncols);
        GLP_TRY(lde_coeffs(c, b->coeffs, b->lde, ncols, lg, rate_bits, glf::GEN));
    }
    GLP_TRY(merkle_from_lde(c, b->lde, ncols, lg, rate_bits, cap_height, b->digests, 1, 0, 0, hasher));
    *out = b.release();
    return GLP_OK;
}

static int batch_from_host(glp_ctx *c, const u64 *host, bool from_values, u32 ncols, u32 log_n, u32 rate_bits, u32 cap_height,
                           glp_batch **out, int hasher = GLP_HASH_POSEIDON) {
    GLP_REQUIRE(c && host && out, "null argument");
    GLP_TRY(bind(c));
    if (log_n > (u32)NTT_MAX_LG) return set_error(GLP_ERR_UNSUPPORTED, "log_n=%u > %d", log_n, NTT_MAX_LG);
    const size_t tot = (size_t)ncols << log_n;
    void *d = nullptr;
    GLP_TRY(c->alloc(&d, tot * 8));
    int rc = GLP_OK;
    if (from_values) {       // upload pipelined with the transforms
        rc = batch_build(c, (const u64 *)d, BATCH_VALUES, ncols, (int)log_n, (int)rate_bits, (int)cap_height, out, host, 1, hasher);
    } else {
        hipError_t e = hipMemcpyAsync(d, host, tot * 8, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) rc = set_error(GLP_ERR_HIP, "H2D copy: %s", hipGetErrorString(e));
        if (rc == GLP_OK) rc = batch_build(c, (const u64 *)d, BATCH_COEFFS_NATURAL, ncols, (int)log_n, (int)rate_bits, (int)cap_height, out, nullptr, 1, hasher);
    }
    (void)hipStreamSynchronize(c->stream);
    c->release(d);
    return rc;
}

}  // namespace glp

extern "C" {

int glp_batch_from_values(glp_ctx *c, const uint64_t *values, uint32_t ncols, uint32_t log_n, uint32_t rate_bits,
                          uint32_t cap_height, glp_batch **out) {
    return batch_from_host(c, values, true, ncols, log_n, rate_bits, cap_height, out);
}
int glp_batch_from_coeffs(glp_ctx *c, const uint64_t *coeffs, uint32_t ncols, uint32_t log_n, uint32_t rate_bits,
                          uint32_t cap_height, glp_batch **out) {
    return batch_from_host(c, coeffs, false, ncols, log_n, rate_bits, cap_height, out);
}
int glp_batch_from_values_h(glp_ctx *c, const uint64_t *values, uint32_t ncols, uint32_t log_n, uint32_t rate_bits, uint32_t cap_height,
                            uint32_t hasher, glp_batch **out) {
    return batch_from_host(c, values, true, ncols, log_n, rate_bits, cap_height, out, (int)hasher);
}
int glp_batch_from_coeffs_h(glp_ctx *c, const uint64_t *coeffs, uint32_t ncols, uint32_t log_n, uint32_t rate_bits, uint32_t cap_height,
                            uint32_t hasher, glp_batch **out) {
    return batch_from_host(c, coeffs, false, ncols, log_n, rate_bits, cap_height, out, (int)hasher);
}
int glp_batch_from_values_device(glp_ctx *c, const uint64_t *dev_values, uint32_t ncols, uint32_t log_n, uint32_t rate_bits,
                                 uint32_t cap_height, glp_batch **out) {
    GLP_REQUIRE(c && dev_values && out, "null argument");
    GLP_TRY(bind(c));
    return batch_build(c, dev_values, BATCH_VALUES, ncols, (int)log_n, (int)rate_bits, (int)cap_height, out);
}
int glp_batch_from_coeffs_device(glp_ctx *c, const uint64_t *dev_coeffs, uint32_t ncols, uint32_t log_n, uint32_t rate_bits,
                                 uint32_t cap_height, glp_batch **out) {
    GLP_REQUIRE(c && dev_coeffs && out, "null argument");
    GLP_TRY(bind(c));
    return batch_build(c, dev_coeffs, BATCH_COEFFS_NATURAL, ncols, (int)log_n, (int)rate_bits, (int)cap_height, out);
}
void glp_batch_free(glp_batch *b) { batch_destroy(b); }

int glp_batch_info(const glp_batch *b, uint32_t *ncols, uint32_t *log_n, uint32_t *rate_bits, uint32_t *cap_height) {
    GLP_REQUIRE(b, "null batch");
    if (ncols) *ncols = b->ncols;
    if (log_n) *log_n = (u32)b->lg;
    if (rate_bits) *rate_bits = (u32)b->rate_bits;
    if (cap_height) *cap_height = (u32)b->cap_height;
    return GLP_OK;
}

int glp_batch_cap(const glp_batch *b, uint64_t *cap_out) {
    GLP_REQUIRE(b && cap_out, "null argument");
    glp_ctx *c = b->ctx;
    GLP_TRY(bind(c));
    const size_t N = (size_t)1 << (b->lg + b->rate_bits);
    const size_t off = merkle_cap_offset(N, b->cap_height);
    GLP_HIP(hipMemcpyAsync(cap_out, b->digests + 4 * off, ((size_t)32) << b->cap_height, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

int glp_batch_coeffs(const glp_batch *b, uint32_t col_begin, uint32_t ncols, uint64_t *out) {
    GLP_REQUIRE(b && (out || !ncols), "null argument");
    GLP_REQUIRE((u64)col_begin + ncols <= b->ncols, "column range out of bounds");
    if (!ncols) return GLP_OK;
    glp_ctx *c = b->ctx;
    GLP_TRY(bind(c));
    const size_t n = (size_t)1 << b->lg;
    Scratch s(c);
    u64 *t;
    GLP_TRY(s.get(&t, (size_t)ncols * n));
    GLP_TRY(bitrev_copy(c, b->coeffs + (size_t)col_begin * n, t, ncols, b->lg));
    GLP_HIP(hipMemcpyAsync(out, t, (size_t)ncols * n * 8, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

int glp_batch_leaf(const glp_batch *b, uint64_t leaf_index, uint64_t *out) {
    GLP_REQUIRE(b && out, "null argument");
    const size_t N = (size_t)1 << (b->lg + b->rate_bits);
    GLP_REQUIRE(leaf_index < N, "leaf_index out of range");
    glp_ctx *c = b->ctx;
    GLP_TRY(bind(c));
    Scratch s(c);
    u64 *idx, *t;
    GLP_TRY(s.get(&idx, 1));
    GLP_TRY(s.get(&t, b->ncols));
    GLP_HIP(hipMemcpyAsync(idx, &leaf_index, 8, hipMemcpyHostToDevice, c->stream));
    GLP_TRY(merkle_gather_lde_rows(c, b->lde, b->ncols, b->lg, b->rate_bits, idx, 1, t));
    GLP_HIP(hipMemcpyAsync(out, t, (size_t)b->ncols * 8, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

int glp_batch_merkle_proof(const glp_batch *b, uint64_t leaf_index, uint64_t *siblings_out) {
    GLP_REQUIRE(b, "null argument");
    const size_t N = (size_t)1 << (b->lg + b->rate_bits);
    GLP_REQUIRE(leaf_index < N, "leaf_index out of range");
    const int depth = b->lg + b->rate_bits - b->cap_height;
    if (depth == 0) return GLP_OK;
    GLP_REQUIRE(siblings_out, "null argument");
    glp_ctx *c = b->ctx;
    GLP_TRY(bind(c));
    Scratch s(c);
    u64 *idx, *t;
    GLP_TRY(s.get(&idx, 1));
    GLP_TRY(s.get(&t, (size_t)depth * 4));
    GLP_HIP(hipMemcpyAsync(idx, &leaf_index, 8, hipMemcpyHostToDevice, c->stream));
    GLP_TRY(merkle_gather_paths(c, b->digests, N, b->cap_height, idx, 1, t));
    GLP_HIP(hipMemcpyAsync(siblings_out, t, (size_t)depth * 32, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

size_t glp_batch_num_digests(const glp_batch *b) { return b ? b->ndigests : 0; }

int glp_batch_digests(const glp_batch *b, uint64_t *out) {
    GLP_REQUIRE(b && out, "null argument");
    glp_ctx *c = b->ctx;
    GLP_TRY(bind(c));
    GLP_HIP(hipMemcpyAsync(out, b->digests, b->ndigests * 32, hipMemcpyDeviceToHost, c->stream));
    GLP_HIP(hipStreamSynchronize(c->stream));
    return GLP_OK;
}

}  // extern "C"
