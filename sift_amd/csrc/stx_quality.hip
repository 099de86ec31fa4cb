// Frame-quality sums for the callers of the texture path: the fallback checkers score a frame from
//   cv2.Laplacian(gray, cv2.CV_64F).var()  and  gray.var()
// (/root/reference/app/main.py:120-123 and :209-210).  Aperture 1 selects the kernel
// [0 1 0; 1 -4 1; 0 1 0] with BORDER_REFLECT_101, which the padded gray plane already holds, so
// this is a plain streaming pass over the plane: per pixel one Laplacian value (an exact integer),
// accumulated as sum, sum of squares, and the sum of squared gray levels.  Only queued when the
// caller asks for it (stx_params.quality_stats); one partial per workgroup, summed in a fixed order.
#include "stx_internal.h"

namespace stx {

constexpr int Q_ROWS = 4;  // rows per workgroup (one per wave), 64 x 4 pixels per thread block column

// grid: x = 256-pixel column blocks, y = groups of Q_ROWS*4 rows, z = frame; a thread walks 4 rows
__global__ __launch_bounds__(256) void quality_kernel(const uint8_t* __restrict__ gray, int H, int W, int gpitch, int grows,
                                                      unsigned long long* __restrict__ part /*[frame][blocks][4]*/) {
    __shared__ unsigned long long red[4][3];
    const int frame = blockIdx.z;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const uint8_t* gp = gray + (int64_t)frame * grows * gpitch;
    long long s = 0;
    unsigned long long q = 0, g2 = 0;
    if (x < W) {
#pragma unroll
        for (int i = 0; i < Q_ROWS; ++i) {
            const int y = (blockIdx.y * 4 + wv) * Q_ROWS + i;
            if (y < H) {
                const uint8_t* p = gp + (int64_t)(y + GRAY_YOFF) * gpitch + x + GRAY_XOFF;
                const int c = p[0];
                const int lap = (int)p[-1] + (int)p[1] + (int)p[-gpitch] + (int)p[gpitch] - 4 * c;
                s += lap;
                q += (unsigned long long)(lap * lap);
                g2 += (unsigned long long)(c * c);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
        g2 += __shfl_xor(g2, o, 64);
    }
    if (lane == 0) { red[wv][0] = (unsigned long long)s; red[wv][1] = q; red[wv][2] = g2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int64_t blk = ((int64_t)frame * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        part[blk * 4 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void quality_reduce_kernel(const unsigned long long* __restrict__ part, int blocks,
                                                             FrameAcc* __restrict__ acc) {
    __shared__ unsigned long long red[3][4];
    const int frame = blockIdx.x, tid = threadIdx.x;
    unsigned long long v[3] = {0, 0, 0};
    for (int b = tid; b < blocks; b += 256) {
        const unsigned long long* p = part + ((int64_t)frame * blocks + b) * 4;
        v[0] += p[0]; v[1] += p[1]; v[2] += p[2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        if ((tid & 63) == 0) red[k][tid >> 6] = v[k];
    }
    __syncthreads();
    if (tid == 0) {
        acc[frame].lap_sum = (long long)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        acc[frame].lap_sumsq = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        acc[frame].gray_sumsq = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    }
}

static dim3 quality_grid(const Geom& g, int n_frames) {
    return dim3((g.W + 63) / 64, (g.H + 4 * Q_ROWS - 1) / (4 * Q_ROWS), n_frames);
}
size_t quality_part_bytes(const Geom& g, int n_frames) {
    const dim3 grid = quality_grid(g, n_frames);
    return (size_t)grid.x * grid.y * grid.z * 4 * sizeof(unsigned long long);
}

void launch_quality(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, unsigned long long* part, FrameAcc* acc) {
    const dim3 grid = quality_grid(g, n_frames);
    hipLaunchKernelGGL(quality_kernel, grid, dim3(256), 0, s, gray, g.H, g.W, g.gpitch, g.grows, part);
    hipLaunchKernelGGL(quality_reduce_kernel, dim3(n_frames), dim3(256), 0, s, part, (int)(grid.x * grid.y), acc);
}

}  // namespace stx
