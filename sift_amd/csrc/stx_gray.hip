// BGR -> gray (fixed point, as cv2.cvtColor(BGR2GRAY) at
// /root/reference/app/passive_checker.py:100) into a padded plane whose two-pixel
// frame already holds the BORDER_REFLECT_101 copies the 3x3 and 5x5 filters need.
// A pure streaming kernel: each lane reads 12 contiguous BGR bytes (three dwords)
// and writes one dword of gray, so a wave reads 768 and writes 256 contiguous bytes.
#include "stx_internal.h"

namespace stx {

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}

__device__ __forceinline__ uint32_t to_gray(uint32_t b, uint32_t g, uint32_t r, int shift15) {
    return shift15 ? ((b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15)
                   : ((b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14);
}

struct __attribute__((aligned(4))) Bgr4 {
    uint32_t a, b, c;  // 12 bytes = four BGR pixels
};

// grid: x = 256-dword column blocks of a padded gray row, y = padded row, z = frame
__global__ __launch_bounds__(256) void gray_kernel(const uint8_t* __restrict__ bgr, int64_t row_stride,
                                                   int64_t frame_stride, int H, int W, int gpitch, int grows,
                                                   int shift15, int aligned4, uint8_t* __restrict__ gray) {
    const int frame = blockIdx.z;
    const int ry = blockIdx.y;
    const int dpr = gpitch >> 2;
    const int dc = blockIdx.x * 256 + threadIdx.x;
    if (dc < dpr) {
        const int sy = reflect101(ry - GRAY_YOFF, H);
        const int x = 4 * dc - GRAY_XOFF;
        const uint8_t* src = bgr + (int64_t)frame * frame_stride + (int64_t)sy * row_stride;
        uint32_t out = 0;
        if (aligned4 && x >= 0 && x + 3 < W) {
            const Bgr4 v = *reinterpret_cast<const Bgr4*>(src + 3 * x);
            const uint32_t a = v.a, b = v.b, c = v.c;
            // bytes: a = B0 G0 R0 B1 | b = G1 R1 B2 G2 | c = R2 B3 G3 R3
            const uint32_t g0 = to_gray(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u, shift15);
            const uint32_t g1 = to_gray(a >> 24, b & 255u, (b >> 8) & 255u, shift15);
            const uint32_t g2 = to_gray((b >> 16) & 255u, b >> 24, c & 255u, shift15);
            const uint32_t g3 = to_gray((c >> 8) & 255u, (c >> 16) & 255u, c >> 24, shift15);
            out = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = x + j;
                if (xx >= -2 && xx <= W + 1) {
                    const int sx = reflect101(xx, W);
                    const uint8_t* p = src + 3 * sx;
                    const uint32_t gv = to_gray(p[0], p[1], p[2], shift15);
                    out |= gv << (8 * j);
                }
            }
        }
        reinterpret_cast<uint32_t*>(gray + ((int64_t)frame * grows + ry) * gpitch)[dc] = out;
    }
}

void launch_gray(hipStream_t s, const uint8_t* bgr, int64_t row_stride, int64_t frame_stride, int n_frames,
                 const Geom& g, int shift, uint8_t* gray) {
    const int dpr = g.gpitch >> 2;
    const int aligned4 = ((reinterpret_cast<uintptr_t>(bgr) | (uintptr_t)row_stride | (uintptr_t)frame_stride) & 3u) == 0;
    dim3 grid((dpr + 255) / 256, g.grows, n_frames);
    hipLaunchKernelGGL(gray_kernel, grid, dim3(256), 0, s, bgr, row_stride, frame_stride, g.H, g.W, g.gpitch, g.grows,
                       shift == 15 ? 1 : 0, aligned4, gray);
}

}  // namespace stx
