// Uniform LBP for any (P <= 24, R): one thread per pixel, the scikit-image algorithm
// (skimage.feature.local_binary_pattern(..., 'uniform'), called with P=8, R=1 at
// /root/reference/app/passive_checker.py:103) with no specialisation.  The fused
// stencil kernel covers the reference's (8,1) case; this kernel serves the
// multi-radius configurations.  fp64, one rounding per operation: compile with
// -ffp-contract=off.
#include "stx_internal.h"

namespace stx {

#pragma clang fp contract(off)

struct LbpGenArgs {
    const uint8_t* gray;
    int H, W, gpitch, grows, P;
    LbpOffsets off;
    FrameAcc* acc;
    uint8_t* codes;
};

__device__ __forceinline__ double gpx(const uint8_t* gp, int gpitch, int H, int W, long r, long c) {
    if (r < 0 || r >= H || c < 0 || c >= W) return 0.0;
    return (double)gp[(r + GRAY_YOFF) * (long)gpitch + c + GRAY_XOFF];
}

__global__ __launch_bounds__(256) void lbp_generic_kernel(LbpGenArgs a) {
    __shared__ uint32_t hist[STX_LBP_MAX_BINS];
    const int frame = blockIdx.z;
    if (threadIdx.x < STX_LBP_MAX_BINS) hist[threadIdx.x] = 0;
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < a.W && y < a.H) {
        const uint8_t* gp = a.gray + (int64_t)frame * a.grows * a.gpitch;
        const double centre = gpx(gp, a.gpitch, a.H, a.W, y, x);
        uint32_t bits = 0;
        for (int i = 0; i < a.P; ++i) {
            const double rf = (double)y + a.off.rp[i];
            const double cf = (double)x + a.off.cp[i];
            const double fr = floor(rf), fc = floor(cf);
            const long minr = (long)fr, minc = (long)fc;
            const long maxr = (long)ceil(rf), maxc = (long)ceil(cf);
            const double dr = rf - fr;
            const double dc = cf - fc;
            const double tl = gpx(gp, a.gpitch, a.H, a.W, minr, minc), tr = gpx(gp, a.gpitch, a.H, a.W, minr, maxc);
            const double bl = gpx(gp, a.gpitch, a.H, a.W, maxr, minc), br = gpx(gp, a.gpitch, a.H, a.W, maxr, maxc);
            const double top = (1.0 - dc) * tl + dc * tr;
            const double bottom = (1.0 - dc) * bl + dc * br;
            const double t = (1.0 - dr) * top + dr * bottom;
            bits |= ((t - centre >= 0.0) ? 1u : 0u) << i;
        }
        const uint32_t pairmask = (a.P >= 2) ? ((1u << (a.P - 1)) - 1u) : 0u;
        const int changes = __popc((bits ^ (bits >> 1)) & pairmask);
        const int code = (changes <= 2) ? __popc(bits) : a.P + 1;
        atomicAdd(&hist[code], 1u);
        if (a.codes) a.codes[((int64_t)frame * a.H + y) * a.W + x] = (uint8_t)code;
    }
    __syncthreads();
    if (threadIdx.x < a.P + 2 && hist[threadIdx.x]) atomicAdd(&a.acc[frame].lbp_hist[threadIdx.x], hist[threadIdx.x]);
}

void launch_lbp_generic(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, int P, const LbpOffsets& off,
                        int halo, FrameAcc* acc, uint8_t* codes_out) {
    (void)halo;
    LbpGenArgs a;
    a.gray = gray; a.H = g.H; a.W = g.W; a.gpitch = g.gpitch; a.grows = g.grows; a.P = P;
    a.off = off; a.acc = acc; a.codes = codes_out;
    dim3 grid((g.W + 63) / 64, (g.H + 3) / 4, n_frames);
    hipLaunchKernelGGL(lbp_generic_kernel, grid, dim3(256), 0, s, a);
}

}  // namespace stx
