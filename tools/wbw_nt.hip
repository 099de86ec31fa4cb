// write-bandwidth microbenchmark: plain against nontemporal stores, coalesced and in the row-FFT store pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void k_coalesced16(f4* out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const f4 v = {(float)i, 1.f, 2.f, 3.f};
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
template <bool NT>
__global__ __launch_bounds__(64) void k_rows_pattern(f2* out, int frames, int units_per_wave) {
    const int lane = threadIdx.x;
    const int fh = 1080, cbw = 8, nkb = 121;
    for (int u = 0; u < units_per_wave; ++u) {
        const int unit = blockIdx.x + u * gridDim.x;
        const int frame = unit / 540, ra = 2 * (unit % 540);
        if (frame >= frames) return;
        f2* base = out + (size_t)frame * nkb * fh * cbw;
        for (int k = lane; k <= 960; k += 64) {
            const int kb = k >> 3, kc = k & 7;
            f2* o = base + ((size_t)kb * fh + ra) * cbw + kc;
            const f2 a = {(float)k, 1.f}, b = {(float)k, 2.f};
            if (NT) { __builtin_nontemporal_store(a, o); __builtin_nontemporal_store(b, o + cbw); }
            else { o[0] = a; o[cbw] = b; }
        }
    }
}
int main() {
    const int frames = 150;
    const size_t n = (size_t)frames * 121 * 1080 * 8;  // float2 elements, 1.25 GB
    f2* buf;
    CK(hipMalloc(&buf, n * sizeof(f2)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
#define RUN(name, launch, bytes) \
        CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) { launch; } \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); \
        printf("%-28s %.3f ms  %.2f TB/s\n", name, ms / 5, (bytes) / (ms / 5 * 1e-3) / 1e12);
        RUN("coalesced16 plain", (k_coalesced16<false><<<dim3(256 * 16), dim3(256)>>>((f4*)buf, n / 2)), n * 8.0)
        RUN("coalesced16 nt", (k_coalesced16<true><<<dim3(256 * 16), dim3(256)>>>((f4*)buf, n / 2)), n * 8.0)
        const int grid = 2048, upw = (frames * 540 + grid - 1) / grid;
        const double rb = (double)frames * 1080 * 961 * 8.0;
        RUN("rows pattern plain", (k_rows_pattern<false><<<dim3(grid), dim3(64)>>>(buf, frames, upw)), rb)
        RUN("rows pattern nt", (k_rows_pattern<true><<<dim3(grid), dim3(64)>>>(buf, frames, upw)), rb)
        // a buffer that fits the Infinity Cache: 15 frames
        const int f2 = 15, upw2 = (f2 * 540 + grid - 1) / grid;
        const double rb2 = (double)f2 * 1080 * 961 * 8.0;
        RUN("rows pattern plain 15 fr", (k_rows_pattern<false><<<dim3(grid), dim3(64)>>>(buf, f2, upw2)), rb2)
        RUN("rows pattern nt 15 fr", (k_rows_pattern<true><<<dim3(grid), dim3(64)>>>(buf, f2, upw2)), rb2)
    }
    return 0;
}
