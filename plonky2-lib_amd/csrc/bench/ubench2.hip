// ubench2.hip -- operand-source sensitivity of gfx950 VALU issue (VGPR vs SGPR/inline operands).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint64_t u64; typedef uint32_t u32;
constexpr int ITERS = 4096;
#define KERNEL(NAME, DECL, ASM, ...)                                                          \
    __global__ void NAME(u64 *out, u32 a, u32 b) {                                            \
        DECL;                                                                                 \
        u32 x = a + threadIdx.x, y = b | 0x00110017u; (void)x; (void)y;                       \
        for (int it = 0; it < ITERS; it++) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : __VA_ARGS__);    \
        }                                                                                     \
        u64 s = 0; for (int i = 0; i < 8; i++) s ^= (u64)acc[i];                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                       \
    }
#define ACC32 u32 acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a
#define ACC64 u64 acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a
KERNEL(k_add_vv, ACC32, "v_add_u32 %0, %0, %1", "+v"(acc[i]) : "v"(y))
KERNEL(k_add3_vvv, ACC32, "v_add3_u32 %0, %0, %1, %2", "+v"(acc[i]) : "v"(x), "v"(y))
KERNEL(k_add3_vsv, ACC32, "v_add3_u32 %0, %0, %1, %2", "+v"(acc[i]) : "s"(b), "v"(y))
KERNEL(k_xor_vv, ACC32, "v_xor_b32 %0, %0, %1", "+v"(acc[i]) : "v"(y))
KERNEL(k_lshl_or, ACC32, "v_lshl_or_b32 %0, %0, 3, %1", "+v"(acc[i]) : "v"(y))
KERNEL(k_mad24_vvv, ACC32, "v_mad_u32_u24 %0, %1, %2, %0", "+v"(acc[i]) : "v"(x), "v"(y))
KERNEL(k_mad24_vcv, ACC32, "v_mad_u32_u24 %0, %1, 17, %0", "+v"(acc[i]) : "v"(x))
KERNEL(k_mad24_vsv, ACC32, "v_mad_u32_u24 %0, %1, %2, %0", "+v"(acc[i]) : "v"(x), "s"(b))
KERNEL(k_dot2_vvv, ACC32, "v_dot2_u32_u16 %0, %1, %2, %0", "+v"(acc[i]) : "v"(x), "v"(y))
KERNEL(k_dot2_vsv, ACC32, "v_dot2_u32_u16 %0, %1, %2, %0", "+v"(acc[i]) : "v"(x), "s"(b))
KERNEL(k_dot4_vsv, ACC32, "v_dot4_u32_u8 %0, %1, %2, %0", "+v"(acc[i]) : "v"(x), "s"(b))
KERNEL(k_perm_vvs, ACC32, "v_perm_b32 %0, %0, %1, %2", "+v"(acc[i]) : "v"(y), "s"(b))
KERNEL(k_mad64_vvv, ACC64, "v_mad_u64_u32 %0, vcc, %1, %2, %0", "+v"(acc[i]) : "v"(x), "v"(y) : "vcc")
KERNEL(k_mad64_vcv, ACC64, "v_mad_u64_u32 %0, vcc, %1, 17, %0", "+v"(acc[i]) : "v"(x) : "vcc")
KERNEL(k_mad64_vsv, ACC64, "v_mad_u64_u32 %0, vcc, %1, %2, %0", "+v"(acc[i]) : "v"(x), "s"(b) : "vcc")
KERNEL(k_mullo_vc, ACC32, "v_mul_lo_u32 %0, %0, 17", "+v"(acc[i]) : )
KERNEL(k_mul24_vc, ACC32, "v_mul_u32_u24 %0, %0, %1", "+v"(acc[i]) : "v"(y))
KERNEL(k_lshladd64_vv, ACC64, "v_lshl_add_u64 %0, %0, 0, %1", "+v"(acc[i]) : "v"((u64)y))
KERNEL(k_addco_sgpr, ACC32, "v_add_co_u32 %0, s[10:11], %0, %1", "+v"(acc[i]) : "v"(y) : "s10", "s11")
KERNEL(k_mov, ACC32, "v_mov_b32 %0, %1", "+v"(acc[i]) : "v"(y))
KERNEL(k_sub_vv, ACC32, "v_sub_u32 %0, %0, %1", "+v"(acc[i]) : "v"(y))
KERNEL(k_and_vv, ACC32, "v_and_b32 %0, %0, %1", "+v"(acc[i]) : "v"(y))
KERNEL(k_lshl_vc, ACC32, "v_lshlrev_b32 %0, 3, %0", "+v"(acc[i]) : )
KERNEL(k_lshr_vc, ACC32, "v_lshrrev_b32 %0, 3, %0", "+v"(acc[i]) : )
KERNEL(k_cndmask32, ACC32, "v_cndmask_b32 %0, %0, %1, vcc", "+v"(acc[i]) : "v"(y) : )
KERNEL(k_cndmask64, ACC32, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]", "+v"(acc[i]) : "v"(y) : )
KERNEL(k_cmp_u32, ACC32, "v_cmp_lt_u32 vcc, %0, %1", "+v"(acc[i]) : "v"(y) : "vcc")
KERNEL(k_cmp_u64, ACC64, "v_cmp_lt_u64 vcc, %0, %1", "+v"(acc[i]) : "v"((u64)y) : "vcc")
KERNEL(k_addco_vcc, ACC32, "v_add_co_u32 %0, vcc, %0, %1", "+v"(acc[i]) : "v"(y) : "vcc")
KERNEL(k_addc_vcc, ACC32, "v_addc_co_u32 %0, vcc, %0, %1, vcc", "+v"(acc[i]) : "v"(y) : "vcc")
KERNEL(k_alignbit, ACC32, "v_alignbit_b32 %0, %0, %1, 7", "+v"(acc[i]) : "v"(y))
KERNEL(k_lshl64, ACC64, "v_lshlrev_b64 %0, 3, %0", "+v"(acc[i]) : )
KERNEL(k_mov64, ACC64, "v_mov_b64 %0, %1", "+v"(acc[i]) : "v"((u64)y))
KERNEL(k_or3, ACC32, "v_or3_b32 %0, %0, %1, %2", "+v"(acc[i]) : "v"(x), "v"(y))
template <class F> static float time_ms(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount, blocks = cus * 8, threads = 256;
    u64 *out; hipMalloc(&out, (size_t)blocks * threads * 8);
    double lanes = (double)blocks * threads, clk = 2.4e9;
#define RUN(K) { float ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, out, 1u, 3u); }); \
                 printf("%-18s %7.3f ms  %6.2f lane-ops/clk/CU\n", #K, ms, lanes * 8.0 * ITERS / (ms * 1e-3) / cus / clk); }
    RUN(k_add_vv) RUN(k_xor_vv) RUN(k_add3_vvv) RUN(k_add3_vsv) RUN(k_lshl_or) RUN(k_mad24_vvv) RUN(k_mad24_vcv) RUN(k_mad24_vsv)
    RUN(k_mul24_vc) RUN(k_dot2_vvv) RUN(k_dot2_vsv) RUN(k_dot4_vsv) RUN(k_perm_vvs) RUN(k_mad64_vvv) RUN(k_mad64_vcv) RUN(k_mad64_vsv)
    RUN(k_mullo_vc) RUN(k_lshladd64_vv) RUN(k_addco_sgpr)
    RUN(k_mov) RUN(k_sub_vv) RUN(k_and_vv) RUN(k_lshl_vc) RUN(k_lshr_vc) RUN(k_cndmask32) RUN(k_cndmask64) RUN(k_cmp_u32) RUN(k_cmp_u64)
    RUN(k_addco_vcc) RUN(k_addc_vcc) RUN(k_alignbit) RUN(k_lshl64) RUN(k_mov64) RUN(k_or3)
    return 0;
}
