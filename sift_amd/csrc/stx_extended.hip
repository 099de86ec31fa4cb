// Extended per-frame statistics that the reference does not compute (no counterpart in
// /root/reference/app/passive_checker.py): gradient statistics on the Sobel images that cv2.Canny
// builds at passive_checker.py:116, and an 8x8 blockiness measure of the kind used to detect JPEG /
// video-codec block structure.  They are only reported through the extended entry points and never
// alter the reference's result dict.
//
// All of it is exact integer work on the gray plane, one streaming pass:
//   mag(y,x)  = |gx| + |gy|, Sobel 3x3 with replicated borders        -> sum, sum of squares,
//               count(mag > canny_high)
//   dh(y,x)   = |g(y,x) - g(y,x-1)|, x >= 1: "boundary" when x % 8 == 0, else "interior"
//   dv(y,x)   = |g(y,x) - g(y-1,x)|, y >= 1: likewise with y
// A wave covers 64 columns, i.e. eight 8-pixel blocks: the boundary lanes are lane % 8 == 0, and the
// per-wave sums come from one cross-lane reduction per statistic at the end.
#include "stx_internal.h"

namespace stx {

constexpr int X_ROWS = 16;  // rows per wave
constexpr int X_STATS = 8;  // partial sums per workgroup (seven used)

// A lane walks down one column: the three pixels (x-1, x, x+1) of the rows above and below stay in
// registers, so a pixel costs three byte loads; the sums of a lane's 16 rows fit 32 bits.
__global__ __launch_bounds__(256) void extended_kernel(const uint8_t* __restrict__ gray, int H, int W, int gpitch, int grows,
                                                       int high, unsigned long long* __restrict__ part) {
    __shared__ unsigned long long red[4][X_STATS];
    const int frame = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: row offsets stay scalar
    const int x = blockIdx.x * 64 + lane;
    const uint8_t* gp = gray + (int64_t)frame * grows * gpitch + (int64_t)GRAY_YOFF * gpitch + GRAY_XOFF;
    uint32_t v[X_STATS] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int y0 = (blockIdx.y * 4 + wv) * X_ROWS;
    if (x < W && y0 < H) {
        const int xl = x > 0 ? x - 1 : 0, xr = x + 1 < W ? x + 1 : W - 1;
        const bool xedge = (x & 7) == 0;
        // every byte of the strip first (three columns of X_ROWS + 2 rows): the loads are all in flight together
        // and the strip waits for memory once, not once per row
        // (buffer loads: the row is a scalar offset, clamped on the scalar side, the column a per-lane offset that
        // never changes -- no address arithmetic per load)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(gp), 0, (H - 1) * gpitch + W, 0x00020000);
        uint8_t pl[X_ROWS + 2], pm[X_ROWS + 2], pr[X_ROWS + 2];
#pragma unroll
        for (int i = 0; i < X_ROWS + 2; ++i) {
            const int y = y0 - 1 + i;
            const int ro = (y < 0 ? 0 : (y < H ? y : H - 1)) * gpitch;
            pl[i] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, xl, ro, 0);
            pm[i] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, x, ro, 0);
            pr[i] = __builtin_amdgcn_raw_buffer_load_b8(rsrc, xr, ro, 0);
        }
        // which sum a difference belongs to never changes for a lane (columns) or is known when the loop is unrolled
        // (rows: y0 is a multiple of 8): masks and fixed accumulators, no indexed register array
        static_assert(X_ROWS % 8 == 0, "row boundaries by the unrolled index");
        const uint32_t in_h = x > 0 ? 0xFFFFFFFFu : 0u, on_h = (x > 0 && xedge) ? 0xFFFFFFFFu : 0u;
        uint32_t h_all = 0, h_edge = 0;
        int a = pl[0], b = pm[0], c = pr[0], d = pl[1], e = pm[1], f = pr[1], g, h, k;
#pragma unroll
        for (int i = 0; i < X_ROWS; ++i) {
            const int y = y0 + i;
            g = pl[i + 2]; h = pm[i + 2]; k = pr[i + 2];
            if (y < H) {
                const int gx = (c + 2 * f + k) - (a + 2 * d + g);
                const int gy = (g + 2 * h + k) - (a + 2 * b + c);
                const int mag = abs(gx) + abs(gy);
                v[0] += (uint32_t)mag;
                v[1] += (uint32_t)(mag * mag);
                v[2] += mag > high ? 1u : 0u;
                const uint32_t dh = (uint32_t)abs(e - d);
                h_all += dh & in_h;
                h_edge += dh & on_h;
                if (y > 0) v[(i & 7) == 0 ? 5 : 6] += (uint32_t)abs(e - b);
            }
            a = d; b = e; c = f;
            d = g; e = h; f = k;
        }
        v[3] = h_edge;
        v[4] = h_all - h_edge;
    }
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        unsigned long long t = v[s];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) red[wv][s] = t;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int64_t blk = ((int64_t)frame * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        part[blk * X_STATS + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void extended_reduce_kernel(const unsigned long long* __restrict__ part, int blocks,
                                                              FrameAcc* __restrict__ acc) {
    __shared__ unsigned long long red[7][4];
    const int frame = blockIdx.x, tid = threadIdx.x;
    unsigned long long v[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = tid; b < blocks; b += 256) {
        const unsigned long long* p = part + ((int64_t)frame * blocks + b) * X_STATS;
#pragma unroll
        for (int k = 0; k < 7; ++k) v[k] += p[k];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        if ((tid & 63) == 0) red[k][tid >> 6] = v[k];
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long t[7];
        for (int k = 0; k < 7; ++k) t[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
        acc[frame].sobel_sum = t[0];
        acc[frame].sobel_sumsq = t[1];
        acc[frame].sobel_strong = (uint32_t)t[2];
        acc[frame].blk[0] = t[3];
        acc[frame].blk[1] = t[4];
        acc[frame].blk[2] = t[5];
        acc[frame].blk[3] = t[6];
    }
}

static dim3 extended_grid(const Geom& g, int n_frames) {
    return dim3((g.W + 63) / 64, (g.H + 4 * X_ROWS - 1) / (4 * X_ROWS), n_frames);
}
size_t extended_part_bytes(const Geom& g, int n_frames) {
    const dim3 grid = extended_grid(g, n_frames);
    return (size_t)grid.x * grid.y * grid.z * X_STATS * sizeof(unsigned long long);
}

void launch_extended(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, int high, unsigned long long* part,
                     FrameAcc* acc) {
    const dim3 grid = extended_grid(g, n_frames);
    hipLaunchKernelGGL(extended_kernel, grid, dim3(256), 0, s, gray, g.H, g.W, g.gpitch, g.grows, high, part);
    hipLaunchKernelGGL(extended_reduce_kernel, dim3(n_frames), dim3(256), 0, s, part, (int)(grid.x * grid.y), acc);
}

}  // namespace stx
