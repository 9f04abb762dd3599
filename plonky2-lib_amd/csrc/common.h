// common.h -- context, device-memory pool, error plumbing and stage timers shared by the
// translation units of libglprover.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../include/glp.h"
#include "glf.h"

namespace glp {

extern thread_local std::string g_last_error;
int set_error(int code, const char *fmt, ...);

#define GLP_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return glp::set_error(GLP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                  __FILE__, __LINE__);                                         \
    } while (0)
#define GLP_TRY(expr)              \
    do {                           \
        int _rc = (expr);          \
        if (_rc != GLP_OK) return _rc; \
    } while (0)
#define GLP_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return glp::set_error(GLP_ERR_ARG, __VA_ARGS__); \
    } while (0)

struct NttPlan;
struct LdePlan;

// index of the first word >= p, or count if every word is a canonical field element (host; the inner loop vectorises).
// The kernels assume canonical inputs: a word >= p gives a silently wrong proof, so the entry points that take field arrays
// from outside (circuit description, hand-off file) reject it by name.
inline size_t first_noncanonical(const u64 *v, size_t count) {
    for (size_t i0 = 0; i0 < count; i0 += 4096) {
        const size_t i1 = i0 + 4096 < count ? i0 + 4096 : count;
        u64 bad = 0;
        for (size_t i = i0; i < i1; i++) bad |= (u64)(v[i] >= glf::P);
        if (bad) for (size_t i = i0; i < i1; i++) if (v[i] >= glf::P) return i;
    }
    return count;
}

struct Stage {
    std::string name;
    hipEvent_t beg, end;
    double bytes;
};

}  // namespace glp

struct glp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // host->device witness chunks, overlapped with the transforms of earlier chunks
    int num_cus = 256;
    // size-keyed free lists: commit buffers are GB-sized and recur with identical sizes every proof
    std::multimap<size_t, void *> pool;
    std::map<void *, size_t> live;
    size_t pool_bytes = 0;
    std::map<int, glp::NttPlan *> ntt_plans;                       // key: log_n
    std::map<std::pair<std::pair<int, int>, u64>, glp::LdePlan *> lde_plans;  // key: ((log_n, rate_bits), shift)
    unsigned long long lde_clock = 0;                              // LRU stamps for lde_plans
    int two_pass_lg = 22;                // largest log_n transformed in two passes (ntt.h NTT_2PASS_LG; GLP_NTT_2PASS_LG overrides, 20..22)
    int strided32_tl = 8;                // tiles (planes) a k_strided32 block walks, software-pipelined (GLP_NTT_STRIDED32_TL: 1..64; 1 = no pipelining)
    int strided32_lw = 4;                // log2 columns of a k_strided32 tile (GLP_NTT_STRIDED32_LW: 3 = 64-byte row segments, 4 = 128-byte)
    size_t merkle_coop_max = 4096;       // leaf hash: up to this many leaves per launch one leaf per 16-lane group (GLP_MERKLE_COOP_MAX), ...
    size_t merkle_quad_max = 32768;      // ... up to this many one leaf per quad of lanes (GLP_MERKLE_QUAD_MAX), above it one leaf per lane (merkle.hip)
    void *host_pool = nullptr;           // HostPool of prover_batch.inc (host threads for the transcripts of a batch), made on first use
    void (*host_pool_free)(void *) = nullptr;
    bool profiling = false;
    std::vector<glp::Stage> stages;

    int alloc(void **p, size_t bytes);
    void release(void *p);
    void trim();
    int stage_begin(const char *name, double bytes);
    int stage_end();
};

namespace glp {
struct StageScope {
    glp_ctx *c;
    StageScope(glp_ctx *ctx, const char *name, double bytes) : c(ctx) { c->stage_begin(name, bytes); }
    ~StageScope() { c->stage_end(); }
};
inline int bind(glp_ctx *c) {
    hipError_t e = hipSetDevice(c->device);
    return e == hipSuccess ? GLP_OK : set_error(GLP_ERR_HIP, "hipSetDevice(%d): %s", c->device, hipGetErrorString(e));
}
}  // namespace glp
