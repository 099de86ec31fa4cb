// write-bandwidth microbenchmark: coalesced vs the row-FFT store pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_coalesced8(float2* out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = make_float2((float)i, 1.f);
}
__global__ __launch_bounds__(256) void k_coalesced16(float4* out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
// layout [frame][kb=121][row=1080][cbw=8] float2; one wave = one row pair, 15 k-iterations of 64 lanes
__global__ __launch_bounds__(64) void k_rows_pattern(float2* out, int frames, int units_per_wave) {
    const int lane = threadIdx.x;
    const int fh = 1080, cbw = 8, nkb = 121;
    for (int u = 0; u < units_per_wave; ++u) {
        const int unit = blockIdx.x + u * gridDim.x;
        const int frame = unit / 540, ra = 2 * (unit % 540);
        if (frame >= frames) return;
        float2* base = out + (size_t)frame * nkb * fh * cbw;
        for (int k = lane; k <= 960; k += 64) {
            const int kb = k >> 3, kc = k & 7;
            float2* o = base + ((size_t)kb * fh + ra) * cbw + kc;
            o[0] = make_float2((float)k, 1.f);
            o[cbw] = make_float2((float)k, 2.f);
        }
    }
}
int main() {
    const int frames = 150;
    const size_t n = (size_t)frames * 121 * 1080 * 8;  // float2 elements, 1.25 GB
    float2* buf;
    CK(hipMalloc(&buf, n * sizeof(float2)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_coalesced8, dim3(256 * 16), dim3(256), 0, 0, buf, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("coalesced 8B/lane : %.3f ms  %.2f TB/s\n", ms / 5, n * 8.0 / (ms / 5 * 1e-3) / 1e12);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_coalesced16, dim3(256 * 16), dim3(256), 0, 0, (float4*)buf, n / 2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("coalesced 16B/lane: %.3f ms  %.2f TB/s\n", ms / 5, n * 8.0 / (ms / 5 * 1e-3) / 1e12);
        const int grid = 2560, upw = (frames * 540 + grid - 1) / grid;
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_rows_pattern, dim3(grid), dim3(64), 0, 0, buf, frames, upw);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rows pattern 2560w: %.3f ms  %.2f TB/s\n", ms / 5, (double)frames * 1080 * 961 * 8.0 / (ms / 5 * 1e-3) / 1e12);
        const int grid2 = 2560 * 4, upw2 = (frames * 540 + grid2 - 1) / grid2;
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_rows_pattern, dim3(grid2), dim3(64), 0, 0, buf, frames, upw2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rows pattern 10240w: %.3f ms  %.2f TB/s\n", ms / 5, (double)frames * 1080 * 961 * 8.0 / (ms / 5 * 1e-3) / 1e12);
    }
    return 0;
}
