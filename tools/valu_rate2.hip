// Issue rate of more vector opcodes (companion of valu_rate.hip): SIMD cycles per wave-instruction with 4 and 8
// resident waves per SIMD, from wall time at an assumed 2.4 GHz.  Eight independent accumulators per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define OPS(X)                                                                   \
    X(0, "v_add_u32 %0, %0, %1", "v_add_u32")                                    \
    X(1, "v_sub_u32 %0, %0, %1", "v_sub_u32")                                    \
    X(2, "v_or_b32 %0, %0, %1", "v_or_b32")                                      \
    X(3, "v_xor_b32 %0, %0, %1", "v_xor_b32")                                    \
    X(4, "v_lshlrev_b32 %0, 3, %0", "v_lshlrev_b32")                             \
    X(5, "v_ashrrev_i32 %0, 3, %0", "v_ashrrev_i32")                             \
    X(6, "v_min_u32 %0, %0, %1", "v_min_u32")                                    \
    X(7, "v_max_i32 %0, %0, %1", "v_max_i32")                                    \
    X(8, "v_mov_b32 %0, %1", "v_mov_b32")                                        \
    X(9, "v_mul_f32 %0, %0, %1", "v_mul_f32")                                    \
    X(10, "v_add_f32 %0, %0, %1", "v_add_f32")                                   \
    X(11, "v_max_f32 %0, %0, %1", "v_max_f32")                                   \
    X(12, "v_fmac_f32 %0, %1, %2", "v_fmac_f32")                                 \
    X(13, "v_mul_u32_u24 %0, %0, %1", "v_mul_u32_u24")                           \
    X(14, "v_mad_u32_u24 %0, %0, %1, %2", "v_mad_u32_u24")                       \
    X(15, "v_and_or_b32 %0, %0, %1, %2", "v_and_or_b32")                         \
    X(16, "v_or3_b32 %0, %0, %1, %2", "v_or3_b32")                               \
    X(17, "v_lshl_add_u32 %0, %0, 2, %1", "v_lshl_add_u32")                      \
    X(18, "v_add_lshl_u32 %0, %0, %1, 1", "v_add_lshl_u32")                      \
    X(19, "v_xad_u32 %0, %0, %1, %2", "v_xad_u32")                               \
    X(20, "v_mad_i32_i16 %0, %1, %2, %0", "v_mad_i32_i16")                       \
    X(21, "v_add_u16 %0, %0, %1", "v_add_u16")                                   \
    X(22, "v_sub_u16 %0, %0, %1", "v_sub_u16")                                   \
    X(23, "v_mul_lo_u16 %0, %0, %1", "v_mul_lo_u16")                             \
    X(24, "v_bcnt_u32_b32 %0, %1, %0", "v_bcnt_u32_b32")                         \
    X(25, "v_cvt_f32_ubyte0 %0, %0", "v_cvt_f32_ubyte0")                         \
    X(26, "v_cvt_f32_u32 %0, %0", "v_cvt_f32_u32")                               \
    X(27, "v_cvt_u32_f32 %0, %0", "v_cvt_u32_f32")                               \
    X(28, "v_subrev_u32 %0, %1, %0", "v_subrev_u32")                             \
    X(29, "v_lshrrev_b32 %0, %1, %0", "v_lshrrev_b32 vreg")                      \
    X(30, "v_and_b32 %0, 0xff00ff, %0", "v_and_b32 lit")                         \
    X(31, "v_add_u32 %0, 0x800080, %0", "v_add_u32 lit")                         \
    X(32, "v_add_u32 %0, s4, %0", "v_add_u32 sgpr")                              \
    X(33, "v_pk_add_u16 %0, %0, %1", "v_pk_add_u16")                             \
    X(34, "v_add_co_u32 %0, vcc, %0, %1", "v_add_co_u32")                        \
    X(35, "v_sub_f32 %0, %0, %1", "v_sub_f32")                                   \
    X(36, "v_min_f32 %0, %0, %1", "v_min_f32")                                   \
    X(37, "v_fma_f32 %0, %0, %1, %2", "v_fma_f32")                               \
    X(38, "v_mul_i32_i24 %0, %0, %1", "v_mul_i32_i24")                           \
    X(39, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD src0_sel:WORD_0 src1_sel:WORD_0", "v_sub_u32 sdwa") \
    X(40, "v_add_u32_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1", "v_add_u32 sdwa w1") \
    X(41, "v_cndmask_b32 %0, %0, %1, s[6:7]", "v_cndmask sgpr")                  \
    X(42, "v_cmp_lt_u32 s[8:9], %0, %1", "v_cmp_lt_u32 sgpr")                    \
    X(43, "v_cmp_lt_u32 vcc, %0, %1", "v_cmp_lt_u32 vcc (vopc)")                 \
    X(44, "v_readlane_b32 s10, %0, 5", "v_readlane_b32")                         \
    X(45, "v_mbcnt_lo_u32_b32 %0, %1, %0", "v_mbcnt_lo")                         \
    X(46, "v_med3_i32 %0, %0, %1, %2", "v_med3_i32")                             \
    X(47, "v_max3_u32 %0, %0, %1, %2", "v_max3_u32")                             \
    X(48, "v_msad_u8 %0, %1, %2, %0", "v_msad_u8")                               \
    X(49, "v_lerp_u8 %0, %0, %1, %2", "v_lerp_u8")                               \
    X(50, "v_cvt_pk_u8_f32 %0, %1, 1, %0", "v_cvt_pk_u8_f32")                    \
    X(51, "v_pk_mul_f32 %0, %0, %1", "v_pk_mul_f32 (64b regs misuse)")           \
    X(52, "v_log_f32 %0, %0", "v_log_f32")                                       \
    X(53, "v_rcp_f32 %0, %0", "v_rcp_f32")                                       \
    X(54, "v_mul_f64 %0, %0, %0", "v_mul_f64 (needs 64b)")

constexpr int NOPS = 51;  // 51.. use 64-bit register pairs and are skipped here

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = threadIdx.x * 2654435761u + 12345u, c = threadIdx.x ^ 0x5bd1e995u;
    asm volatile("s_mov_b32 s4, 17\n s_mov_b64 s[6:7], 0x5555" ::: "s4", "s6", "s7");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#define X(id, str, name)                                                                                       \
    if (OP == id) {                                                                                            \
        asm volatile(str : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s8", "s9", "s10");                              \
        asm volatile(str : "+v"(a1) : "v"(b), "v"(c) : "vcc", "s8", "s9", "s10");                              \
        asm volatile(str : "+v"(a2) : "v"(b), "v"(c) : "vcc", "s8", "s9", "s10");                              \
        asm volatile(str : "+v"(a3) : "v"(b), "v"(c) : "vcc", "s8", "s9", "s10");                              \
        asm volatile(str : "+v"(a4) : "v"(b), "v"(c) : "vcc", "s8", "s9", "s10");                              \
        asm volatile(str : "+v"(a5) : "v"(b), "v"(c) : "vcc", "s8", "s9", "s10");                              \
        asm volatile(str : "+v"(a6) : "v"(b), "v"(c) : "vcc", "s8", "s9", "s10");                              \
        asm volatile(str : "+v"(a7) : "v"(b), "v"(c) : "vcc", "s8", "s9", "s10");                              \
    }
            OPS(X)
#undef X
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

static const char* NAMES[] = {
#define X(id, str, name) name,
    OPS(X)
#undef X
};

template <int OP>
void run(uint32_t* out) {
    const int iters = 2000;
    printf("%-28s", NAMES[OP]);
    for (int wps : {2, 4, 8}) {
        const int grid = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  w%d: %5.2f cyc", wps, ms * 1e-3 * 2.4e9 / (iters * 64.0 * wps));
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n");
    fflush(stdout);
}

template <int OP>
void run_all(uint32_t* out) {
    run<OP>(out);
    if constexpr (OP + 1 < NOPS) run_all<OP + 1>(out);
}

int main() {
    uint32_t* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    run_all<0>(out);
    return 0;
}
