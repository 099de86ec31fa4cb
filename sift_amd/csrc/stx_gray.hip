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

constexpr int GRAY_ROWS_PER_BLOCK = 4;

// grid, wide frames: x = 256-dword column blocks of a padded row, y = groups of 4 padded rows, z = frame;
// narrow frames (FLAT): x = 256-thread blocks over (row group, dword column) pairs, so that crops fill their
// workgroups too; the dword columns that lie fully inside the image are numbered first (row group major) and
// the ones that touch the reflected border after them, so that a wave runs one of the two code paths.
// A thread converts the same four-pixel column of four consecutive rows: four independent 12-byte
// loads in flight per lane, and a quarter of the workgroups to dispatch.
template <bool FLAT>
__global__ __launch_bounds__(256) void gray_kernel(const uint8_t* __restrict__ bgr, int64_t row_stride,
                                                   int64_t frame_stride, int H, int W, int gpitch, int grows,
                                                   int shift15, int aligned4, uint8_t* __restrict__ gray) {
    const int frame = blockIdx.z;
    const int dpr = gpitch >> 2;
    const int id = blockIdx.x * 256 + threadIdx.x;
    int rgroup, dc;
    if (FLAT) {
        // dword columns 1 .. W/4 hold four image pixels each
        const int nfast = W >> 2, nslow = dpr - nfast;
        const int groups = (grows + GRAY_ROWS_PER_BLOCK - 1) / GRAY_ROWS_PER_BLOCK;
        if (id < groups * nfast) {
            rgroup = id / nfast;
            dc = 1 + id - rgroup * nfast;
        } else {
            const int k = id - groups * nfast;
            rgroup = k / nslow;
            const int q = k - rgroup * nslow;
            dc = q == 0 ? 0 : nfast + q;
        }
    } else {
        rgroup = (int)blockIdx.y;
        dc = id;
    }
    if (dc >= dpr || rgroup * GRAY_ROWS_PER_BLOCK >= grows) return;
    const int x = 4 * dc - GRAY_XOFF;
    const uint8_t* fsrc = bgr + (int64_t)frame * frame_stride;
    const bool fast = aligned4 && x >= 0 && x + 3 < W;
    uint32_t out[GRAY_ROWS_PER_BLOCK];
    if (fast) {
        Bgr4 v[GRAY_ROWS_PER_BLOCK];
#pragma unroll
        for (int j = 0; j < GRAY_ROWS_PER_BLOCK; ++j) {
            int ry = rgroup * GRAY_ROWS_PER_BLOCK + j;
            ry = ry < grows ? ry : grows - 1;
            const int sy = reflect101(ry - GRAY_YOFF, H);
            v[j] = *reinterpret_cast<const Bgr4*>(fsrc + (int64_t)sy * row_stride + 3 * x);
        }
#pragma unroll
        for (int j = 0; j < GRAY_ROWS_PER_BLOCK; ++j) {
            const uint32_t a = v[j].a, b = v[j].b, c = v[j].c;
            // bytes: a = B0 G0 R0 B1 | b = G1 R1 B2 G2 | c = R2 B3 G3 R3
            const uint32_t g0 = to_gray(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u, shift15);
            const uint32_t g1 = to_gray(a >> 24, b & 255u, (b >> 8) & 255u, shift15);
            const uint32_t g2 = to_gray((b >> 16) & 255u, b >> 24, c & 255u, shift15);
            const uint32_t g3 = to_gray((c >> 8) & 255u, (c >> 16) & 255u, c >> 24, shift15);
            out[j] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        }
    } else {
#pragma unroll
        for (int j = 0; j < GRAY_ROWS_PER_BLOCK; ++j) {
            int ry = rgroup * GRAY_ROWS_PER_BLOCK + j;
            ry = ry < grows ? ry : grows - 1;
            const uint8_t* src = fsrc + (int64_t)reflect101(ry - GRAY_YOFF, H) * row_stride;
            uint32_t o = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int xx = x + q;
                if (xx >= -2 && xx <= W + 1) {
                    const uint8_t* p = src + 3 * reflect101(xx, W);
                    o |= to_gray(p[0], p[1], p[2], shift15) << (8 * q);
                }
            }
            out[j] = o;
        }
    }
#pragma unroll
    for (int j = 0; j < GRAY_ROWS_PER_BLOCK; ++j) {
        const int ry = rgroup * GRAY_ROWS_PER_BLOCK + j;
        if (ry < grows) reinterpret_cast<uint32_t*>(gray + ((int64_t)frame * grows + ry) * gpitch)[dc] = out[j];
    }
}

void launch_gray(hipStream_t s, const uint8_t* bgr, int64_t row_stride, int64_t frame_stride, int n_frames,
                 const Geom& g, int shift, uint8_t* gray) {
    const int dpr = g.gpitch >> 2;
    const int aligned4 = ((reinterpret_cast<uintptr_t>(bgr) | (uintptr_t)row_stride | (uintptr_t)frame_stride) & 3u) == 0;
    const int groups = (g.grows + GRAY_ROWS_PER_BLOCK - 1) / GRAY_ROWS_PER_BLOCK;
    if (dpr >= 256) {
        dim3 grid((dpr + 255) / 256, groups, n_frames);
        hipLaunchKernelGGL(gray_kernel<false>, grid, dim3(256), 0, s, bgr, row_stride, frame_stride, g.H, g.W, g.gpitch,
                           g.grows, shift == 15 ? 1 : 0, aligned4, gray);
    } else {
        dim3 grid((unsigned)(((int64_t)groups * dpr + 255) / 256), 1, n_frames);
        hipLaunchKernelGGL(gray_kernel<true>, grid, dim3(256), 0, s, bgr, row_stride, frame_stride, g.H, g.W, g.gpitch,
                           g.grows, shift == 15 ? 1 : 0, aligned4, gray);
    }
}

}  // namespace stx
