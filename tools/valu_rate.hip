// Issue rate of the vector instructions the stencil kernel is made of: cycles per wave-instruction on one SIMD
// at 1, 2 and 4 resident waves per SIMD (in-kernel cycle counter, so independent of the clock the chip holds).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* cyc, int iters) {
    uint32_t a[8], b = threadIdx.x * 2654435761u + 12345u, c = threadIdx.x ^ 0x5bd1e995u;
    double d[8];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[8], pb = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; d[i] = (double)i; p[i] = v2f{(float)i, 1.0f}; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#define ONE(i)                                                                                              \
    if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                \
    if (OP == 1) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                             \
    if (OP == 2) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                 \
    if (OP == 3) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));               \
    if (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                   \
    if (OP == 5) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));                    \
    if (OP == 6) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b));                       \
    if (OP == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                    \
    if (OP == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));               \
    if (OP == 9) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                    \
    if (OP == 10) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b)); \
    if (OP == 11) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                  \
    if (OP == 12) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i]));                                    \
    if (OP == 13) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                  \
    if (OP == 14) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                         \
    if (OP == 15) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                            \
    if (OP == 16) asm volatile("v_pk_lshrrev_b16 %0, 7, %0 op_sel_hi:[0,1]" : "+v"(a[i]));                  \
    if (OP == 17) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                            \
    if (OP == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                               \
    if (OP == 19) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));                                     \
    if (OP == 20) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[i]));                                        \
    if (OP == 21) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[i]));                                        \
    if (OP == 22) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));                                           \
    if (OP == 23) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                            \
    if (OP == 24) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));                           \
    if (OP == 25) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));                      \
    if (OP == 26) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");                   \
    if (OP == 27) asm volatile("v_bfe_u32 %0, %0, 4, 4" : "+v"(a[i]));                                      \
    if (OP == 28) asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(a[i]) : "v"(b));                        \
    if (OP == 29) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));               \
    if (OP == 30) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(ONE)
#undef ONE
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)d[i] + (uint32_t)p[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static const char* NAMES[] = {"v_add_u32", "v_pk_add_u16", "v_pk_mad_u16", "v_dot2_u32_u16", "v_perm_b32", "v_bfi_b32",
                              "v_alignbit_b32", "v_fma_f32", "v_pk_fma_f32", "v_sad_u16", "v_mov_dpp wave_shr", "v_min3_u32",
                              "v_fma_f64", "v_add3_u32", "v_pk_mul_lo_u16", "v_pk_max_i16", "v_pk_lshrrev_b16", "v_pk_sub_i16",
                              "v_and_b32", "v_lshrrev_b32", "v_mul_f64", "v_add_f64", "v_sqrt_f32", "v_mul_lo_u32", "v_pk_add_f32",
                              "v_cndmask_b32", "v_cmp_lt_u32", "v_bfe_u32", "v_lshl_or_b32", "v_dot4_u32_u8", "v_sad_u8"};
constexpr int NOPS = 31;

template <int OP>
void run(uint32_t* out, unsigned long long* cyc, std::vector<unsigned long long>& h) {
    const int iters = 2000;
    printf("%-20s", NAMES[OP]);
    for (int wps : {1, 2, 4, 8}) {
        const int grid = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> v(h.begin(), h.begin() + grid * 4);
        std::sort(v.begin(), v.end());
        const double med = (double)v[v.size() / 2];
        const double per_wave = med / (iters * 64.0);          // counter ticks per instruction of one wave
        const double wall_cyc = ms * 1e-3 * 2.4e9 / (iters * 64.0 * wps);  // SIMD cycles per instruction at 2.4 GHz
        printf("  w%d: %6.2f tick/instr/wave (%5.2f per SIMD), wall %5.2f cyc@2.4GHz", wps, per_wave, per_wave / wps, wall_cyc);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n");
}

template <int OP>
void run_all(uint32_t* out, unsigned long long* cyc, std::vector<unsigned long long>& h) {
    run<OP>(out, cyc, h);
    if constexpr (OP + 1 < NOPS) run_all<OP + 1>(out, cyc, h);
}

int main() {
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&cyc, 256 * 8 * 4 * 8);
    std::vector<unsigned long long> h(256 * 8 * 4);
    run_all<0>(out, cyc, h);
    return 0;
}
