// Issue rate of mixed instruction streams at 4 waves per SIMD: does a plain 32-bit op keep its double rate between
// packed / three-operand instructions, and with how many independent chains?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define SLOW(r, b) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(b));
#define FAST(r, b) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b));

// MIX 0: all slow, 1: all fast, 2: S S F (one chain each of 8 regs), 3: S F alternating, 4: SSF with only 2 chains (dependent),
// 5: SSF with 4 chains
template <int MIX>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    uint32_t b = threadIdx.x * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MIX == 0) { SLOW(a[0], b) SLOW(a[1], b) SLOW(a[2], b) SLOW(a[3], b) SLOW(a[4], b) SLOW(a[5], b) }
            if (MIX == 1) { FAST(a[0], b) FAST(a[1], b) FAST(a[2], b) FAST(a[3], b) FAST(a[4], b) FAST(a[5], b) }
            if (MIX == 2) { SLOW(a[0], b) SLOW(a[1], b) FAST(a[2], b) SLOW(a[3], b) SLOW(a[4], b) FAST(a[5], b) }
            if (MIX == 3) { SLOW(a[0], b) FAST(a[1], b) SLOW(a[2], b) FAST(a[3], b) SLOW(a[4], b) FAST(a[5], b) }
            if (MIX == 4) { SLOW(a[0], b) SLOW(a[1], b) FAST(a[0], b) SLOW(a[1], b) SLOW(a[0], b) FAST(a[1], b) }
            if (MIX == 5) { SLOW(a[0], b) SLOW(a[1], b) FAST(a[2], b) SLOW(a[3], b) SLOW(a[0], b) FAST(a[1], b) }
            if (MIX == 6) { FAST(a[0], b) FAST(a[1], b) FAST(a[0], b) FAST(a[1], b) FAST(a[0], b) FAST(a[1], b) }
            if (MIX == 7) { FAST(a[0], b) FAST(a[0], b) FAST(a[0], b) FAST(a[0], b) FAST(a[0], b) FAST(a[0], b) }
            if (MIX == 8) { SLOW(a[0], b) SLOW(a[0], b) SLOW(a[0], b) SLOW(a[0], b) SLOW(a[0], b) SLOW(a[0], b) }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static const char* NAMES[] = {"6 slow, 6 chains", "6 fast, 6 chains", "SSF SSF, 6 chains", "SF SF SF, 6 chains", "SSF SSF, 2 chains",
                              "SSF SSF, 4 chains", "6 fast, 2 chains", "6 fast, 1 chain", "6 slow, 1 chain"};

template <int MIX>
void run(uint32_t* out) {
    const int iters = 2000;
    printf("%-22s", NAMES[MIX]);
    for (int wps : {1, 2, 4, 8}) {
        const int grid = 256 * wps;
        hipLaunchKernelGGL(k<MIX>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MIX>, dim3(grid), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  w%d: %5.2f cyc/instr", wps, ms * 1e-3 * 2.4e9 / (iters * 48.0 * wps));
    }
    printf("\n");
}

int main() {
    uint32_t* out;
    (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>(out); run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<5>(out); run<6>(out); run<7>(out); run<8>(out);
    return 0;
}
