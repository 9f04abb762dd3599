// ubench.hip -- gfx950 integer-issue microbenchmarks that size the VALU roofline of the Poseidon
// and NTT kernels (the MI355X guides give no integer-multiply rates).  Standalone program:
//   hipcc --offload-arch=gfx950 -O3 -I.. ubench.hip -o ubench && ./ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../glf.h"
#include "../poseidon.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

__global__ void k_mad64(u64 *out, u32 a, u32 b) {
    u64 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i;
    u32 x = a + threadIdx.x, y = b;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y) : "vcc");
    }
    u64 s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mullo(u64 *out, u32 a, u32 b) {
    u32 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    u32 y = b | 1;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
    }
    u32 s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi(u64 *out, u32 a, u32 b) {
    u32 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    u32 y = b | 0x80000001u;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(y));
    }
    u32 s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add32(u64 *out, u32 a, u32 b) {
    u32 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
    }
    u32 s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_lshladd64(u64 *out, u32 a, u32 b) {
    u64 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    u64 y = b;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(y));
    }
    u64 s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_addc(u64 *out, u32 a, u32 b) {
    u32 lo[8], hi[8];
    for (int i = 0; i < 8; i++) { lo[i] = threadIdx.x + i + a; hi[i] = i; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(lo[i]), "+v"(hi[i]) : "v"(b) : "vcc");
    }
    u32 s = 0; for (int i = 0; i < 8; i++) s ^= lo[i] ^ hi[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dot2(u64 *out, u32 a, u32 b) {
    u32 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    u32 x = a * 0x10001u + threadIdx.x, y = b | 0x00110017u;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
    }
    u32 s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dot4(u64 *out, u32 a, u32 b) {
    u32 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    u32 x = a * 0x01010101u + threadIdx.x, y = b | 0x11170f29u;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
    }
    u32 s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_perm(u64 *out, u32 a, u32 b) {
    u32 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    u32 y = b + threadIdx.x, sel = 0x07060100u;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(y), "v"(sel));
    }
    u32 s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad24(u64 *out, u32 a, u32 b) {
    u32 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    u32 x = a + threadIdx.x, y = b | 17u;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
    }
    u32 s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulmod(u64 *out, u64 a, u64 b) {
    u64 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = glf::canon(a + threadIdx.x + i);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = glf::mul(acc[i], b);
    }
    u64 s = 0; for (int i = 0; i < 4; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_poseidon(u64 *out, u64 a, int reps) {
    u64 s[12];
    for (int i = 0; i < 12; i++) s[i] = glf::canon(a + threadIdx.x * 12 + i);
    for (int r = 0; r < reps; r++) pos::permute(s);
    u64 x = 0; for (int i = 0; i < 12; i++) x ^= s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <class F> static float time_ms(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    int blocks = cus * 8, threads = 256;
    u64 *out; CK(hipMalloc(&out, (size_t)blocks * threads * 8));
    double lanes = (double)blocks * threads;
    auto rep = [&](const char *name, float ms, double ops_per_lane) {
        double ops = lanes * ops_per_lane;
        double rate = ops / (ms * 1e-3);
        printf("%-22s %8.3f ms  %8.2f Gop/s  %6.2f lane-ops/clk/CU (peak 128 @ %.0f MHz)\n", name, ms, rate / 1e9, rate / cus / clk, clk / 1e6);
    };
    rep("v_add_u32", time_ms([&] { hipLaunchKernelGGL(k_add32, dim3(blocks), dim3(threads), 0, 0, out, 1u, 3u); }), 8.0 * ITERS);
    rep("v_mul_lo_u32", time_ms([&] { hipLaunchKernelGGL(k_mullo, dim3(blocks), dim3(threads), 0, 0, out, 1u, 3u); }), 8.0 * ITERS);
    rep("v_mul_hi_u32", time_ms([&] { hipLaunchKernelGGL(k_mulhi, dim3(blocks), dim3(threads), 0, 0, out, 1u, 3u); }), 8.0 * ITERS);
    rep("v_dot2_u32_u16", time_ms([&] { hipLaunchKernelGGL(k_dot2, dim3(blocks), dim3(threads), 0, 0, out, 1u, 3u); }), 8.0 * ITERS);
    rep("v_dot4_u32_u8", time_ms([&] { hipLaunchKernelGGL(k_dot4, dim3(blocks), dim3(threads), 0, 0, out, 1u, 3u); }), 8.0 * ITERS);
    rep("v_perm_b32", time_ms([&] { hipLaunchKernelGGL(k_perm, dim3(blocks), dim3(threads), 0, 0, out, 1u, 3u); }), 8.0 * ITERS);
    rep("v_mad_u32_u24", time_ms([&] { hipLaunchKernelGGL(k_mad24, dim3(blocks), dim3(threads), 0, 0, out, 1u, 3u); }), 8.0 * ITERS);
    rep("v_mad_u64_u32", time_ms([&] { hipLaunchKernelGGL(k_mad64, dim3(blocks), dim3(threads), 0, 0, out, 1u, 3u); }), 8.0 * ITERS);
    rep("v_lshl_add_u64", time_ms([&] { hipLaunchKernelGGL(k_lshladd64, dim3(blocks), dim3(threads), 0, 0, out, 1u, 3u); }), 8.0 * ITERS);
    rep("v_add_co+v_addc (pair)", time_ms([&] { hipLaunchKernelGGL(k_addc, dim3(blocks), dim3(threads), 0, 0, out, 1u, 3u); }), 8.0 * ITERS);
    rep("glf::mul (mulmod)", time_ms([&] { hipLaunchKernelGGL(k_mulmod, dim3(blocks), dim3(threads), 0, 0, out, (u64)12345, (u64)0xfedcba9876543210ull); }), 4.0 * ITERS);
    int reps = 64;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_poseidon, dim3(blocks), dim3(threads), 0, 0, out, (u64)99, reps); });
    printf("%-22s %8.3f ms  %8.2f Mperm/s  (%.0f lane-clk per permutation per lane at 128 lanes/clk/CU)\n", "pos::permute", ms,
           lanes * reps / (ms * 1e-3) / 1e6, (ms * 1e-3) * clk * cus * 128.0 / (lanes * reps));
    hipFree(out);
    return 0;
}
