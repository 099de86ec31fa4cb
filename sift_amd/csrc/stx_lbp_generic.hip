// Uniform LBP for any (P <= 24, R): the scikit-image algorithm
// (skimage.feature.local_binary_pattern(..., 'uniform'), called with P=8, R=1 at
// /root/reference/app/passive_checker.py:103) with no specialisation of the sample
// positions.  The fused stencil kernel covers the reference's (8,1) case; this kernel serves
// the multi-radius configurations.
//
// A 256-thread workgroup owns a 64x64 pixel tile held in LDS with a zero-filled halo (pixels
// outside the image are 0.0 for scikit-image).  Every sample is a bilinear blend of a 2x2
// cell whose position relative to the centre and whose four weights are the same for every
// pixel, so they are tabulated once per launch: the blend runs in 16.16 fixed point (four
// 24-bit multiply-adds) and decides the sign whenever the result is further from the centre
// than the rounding of the weights can explain; the remaining samples (ties, mostly) are
// replayed in scikit-image's fp64 operation order, one rounding per operation: compile with
// -ffp-contract=off.  Histograms leave the kernel as one partial per tile (no global atomics).
#include "stx_internal.h"

namespace stx {

#pragma clang fp contract(off)

constexpr int GL_TILE = 64;        // tile width and height
constexpr int GL_ROWS = GL_TILE / 4;  // rows per wave
constexpr int GL_BINS = 32;        // partial histogram slots per tile (P + 2 <= 26 used)

struct LbpGenArgs {
    const uint8_t* gray;
    int H, W, gpitch, grows, P;
    int halo, lpitch;     // zero-filled border of the LDS tile, bytes per tile row
    uint32_t margin;      // bound on |fixed-point sample - exact sample| in 1/65536 gray levels, plus a guard
    int ofs[24];          // LDS byte offset of the cell's top-left pixel relative to the centre pixel
    int dcol[24];         // 0 or 1: the cell's right column
    int drow[24];         // 0 or lpitch: the cell's lower row
    uint32_t w[24][4];    // 16.16 weights of top-left, top-right, bottom-left, bottom-right
    uint32_t lim[24];     // 2*margin, or 0 for samples that sit exactly on a pixel (always decided)
    LbpOffsets off;       // the exact sample offsets for the fp64 path
    uint32_t* part;       // [frame][tiles][GL_BINS]
    uint8_t* codes;       // optional per-pixel codes
};

__global__ __launch_bounds__(256) void lbp_generic_kernel(LbpGenArgs a) {
    extern __shared__ uint8_t gl_tile[];  // (GL_TILE + 2*halo) rows of lpitch bytes
    __shared__ uint32_t hist[GL_BINS];
    __shared__ double offs[2][24];
    const int frame = blockIdx.z;
    const int x0 = blockIdx.x * GL_TILE, y0 = blockIdx.y * GL_TILE;
    const int tid = threadIdx.x;
    const int halo = a.halo, lpitch = a.lpitch;
    const int trows = GL_TILE + 2 * halo;

    const uint8_t* gp = a.gray + (int64_t)frame * a.grows * a.gpitch;
    for (int i = tid; i < trows * lpitch; i += 256) {
        const int r = i / lpitch, cidx = i - r * lpitch;
        const int gy = y0 - halo + r, gx = x0 - halo + cidx;
        uint8_t v = 0;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = gp[(int64_t)(gy + GRAY_YOFF) * a.gpitch + gx + GRAY_XOFF];
        gl_tile[i] = v;
    }
    if (tid < GL_BINS) hist[tid] = 0;
    if (tid < 24) { offs[0][tid] = a.off.rp[tid]; offs[1][tid] = a.off.cp[tid]; }
    __syncthreads();

    const int lane = tid & 63, wv = tid >> 6;
    const int x = x0 + lane;
    const int P = a.P;
    const uint32_t pmask = (P >= 32) ? 0xFFFFFFFFu : ((1u << P) - 1u);
    if (x < a.W) {
        for (int i = 0; i < GL_ROWS; ++i) {
            const int y = y0 + wv * GL_ROWS + i;
            if (y >= a.H) break;
            const int pa = (halo + wv * GL_ROWS + i) * lpitch + halo + lane;
            const uint32_t c = gl_tile[pa];
            const uint32_t init = a.margin - (c << 16);
            uint32_t signs = 0, near = 0;  // sample p at bit P-1-p
            for (int p = 0; p < P; ++p) {
                const int a00 = pa + a.ofs[p];
                const int a10 = a00 + a.drow[p];
                const uint32_t v00 = gl_tile[a00], v01 = gl_tile[a00 + a.dcol[p]];
                const uint32_t v10 = gl_tile[a10], v11 = gl_tile[a10 + a.dcol[p]];
                // t = (sample - centre) + margin in 1/65536 gray levels
                const uint32_t t = v00 * a.w[p][0] + v01 * a.w[p][1] + v10 * a.w[p][2] + v11 * a.w[p][3] + init;
                signs = (signs << 1) | (t >> 31);
                near = (near << 1) | (t <= a.lim[p] ? 1u : 0u);
            }
            uint32_t bits = ~signs & pmask;  // LBP bit = sample >= centre
            while (near != 0) {
                const int b = __ffs(near) - 1;
                near &= near - 1;
                const int p = P - 1 - b;
                // skimage's bilinear_interpolation, operation by operation
                const double rf = (double)y + offs[0][p], cf = (double)x + offs[1][p];
                const double fr = floor(rf), fc = floor(cf);
                const int minr = (int)fr, minc = (int)fc;
                const int maxr = (int)ceil(rf), maxc = (int)ceil(cf);
                const double dr = rf - fr, dc = cf - fc;
                auto px = [&](int yy, int xx) -> double {
                    return (double)gl_tile[(yy - y0 + halo) * lpitch + (xx - x0 + halo)];
                };
                const double tl = px(minr, minc), tr = px(minr, maxc), bl = px(maxr, minc), br = px(maxr, maxc);
                const double top = (1.0 - dc) * tl + dc * tr;
                const double bottom = (1.0 - dc) * bl + dc * br;
                const double t = (1.0 - dr) * top + dr * bottom;
                const uint32_t bit = (t - (double)c >= 0.0) ? 1u : 0u;
                bits = (bits & ~(1u << b)) | (bit << b);
            }
            // 0/1 changes between consecutive samples (not circular), as scikit-image counts them; the
            // reversed bit order changes neither that count nor the number of ones
            const uint32_t pairmask = (P >= 2) ? ((1u << (P - 1)) - 1u) : 0u;
            const int changes = __popc((bits ^ (bits >> 1)) & pairmask);
            const int code = (changes <= 2) ? __popc(bits) : P + 1;
            atomicAdd(&hist[code], 1u);
            if (a.codes) a.codes[((int64_t)frame * a.H + y) * a.W + x] = (uint8_t)code;
        }
    }
    __syncthreads();
    if (tid < GL_BINS) {
        const int64_t tileid = ((int64_t)frame * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.part[tileid * GL_BINS + tid] = hist[tid];
    }
}

// Per-frame sum of the tile histograms, in a fixed order.
__global__ __launch_bounds__(256) void lbp_hist_reduce_kernel(const uint32_t* __restrict__ part, int tiles, int bins,
                                                              FrameAcc* __restrict__ acc, int hist_slot) {
    __shared__ uint32_t red[8][GL_BINS];
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int q = tid & (GL_BINS - 1), grp = tid >> 5;
    uint32_t s = 0;
    for (int t = grp; t < tiles; t += 8) s += part[((int64_t)frame * tiles + t) * GL_BINS + q];
    red[grp][q] = s;
    __syncthreads();
    if (tid < bins) {
        uint32_t v = 0;
        for (int g = 0; g < 8; ++g) v += red[g][tid];
        uint32_t* h = hist_slot == 0 ? acc[frame].lbp_hist : (hist_slot == 1 ? acc[frame].lbp_hist2 : acc[frame].lbp_hist3);
        h[tid] = v;
    }
}

int lbp_generic_tiles(const Geom& g) { return ((g.W + GL_TILE - 1) / GL_TILE) * ((g.H + GL_TILE - 1) / GL_TILE); }
size_t lbp_generic_part_bytes(const Geom& g, int n_frames) {
    return (size_t)n_frames * lbp_generic_tiles(g) * GL_BINS * sizeof(uint32_t);
}

bool launch_lbp_generic(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, int P, const LbpOffsets& off,
                        uint32_t* part, FrameAcc* acc, uint8_t* codes_out, int hist_slot) {
    LbpGenArgs a;
    a.gray = gray; a.H = g.H; a.W = g.W; a.gpitch = g.gpitch; a.grows = g.grows; a.P = P;
    a.off = off; a.part = part; a.codes = codes_out;
    // cell geometry and weights from the nominal offsets (pixel (0,0)): the per-pixel fp64 weights differ from
    // these by rounding in the 13th digit, which the guard in the margin covers
    int halo = 1;
    double worst = 0.0;
    int fr[24], fc[24], dr1[24], dc1[24];
    double wr[24], wc[24];
    for (int p = 0; p < P; ++p) {
        const double rp = off.rp[p], cp = off.cp[p];
        const double flr = floor(rp), flc = floor(cp);
        fr[p] = (int)flr; fc[p] = (int)flc;
        wr[p] = rp - flr; wc[p] = cp - flc;
        dr1[p] = wr[p] > 0.0 ? 1 : 0;
        dc1[p] = wc[p] > 0.0 ? 1 : 0;
        const int need_r = fr[p] < 0 ? -fr[p] : fr[p] + dr1[p];
        const int need_c = fc[p] < 0 ? -fc[p] : fc[p] + dc1[p];
        if (need_r > halo) halo = need_r;
        if (need_c > halo) halo = need_c;
    }
    const int lpitch = ((GL_TILE + 2 * halo + 3) / 4) * 4;
    const size_t lds = (size_t)(GL_TILE + 2 * halo) * lpitch;
    if (lds > 60 * 1024) return false;  // radius beyond what one tile with its halo can hold
    a.halo = halo; a.lpitch = lpitch;
    for (int p = 0; p < 24; ++p) {
        a.ofs[p] = 0; a.dcol[p] = 0; a.drow[p] = 0; a.lim[p] = 0;
        for (int k = 0; k < 4; ++k) a.w[p][k] = 0;
    }
    for (int p = 0; p < P; ++p) {
        a.ofs[p] = fr[p] * lpitch + fc[p];
        a.dcol[p] = dc1[p];
        a.drow[p] = dr1[p] * lpitch;
        const double wex[4] = {(1.0 - wr[p]) * (1.0 - wc[p]), (1.0 - wr[p]) * wc[p], wr[p] * (1.0 - wc[p]), wr[p] * wc[p]};
        double ep = 0.0, en = 0.0;
        for (int k = 0; k < 4; ++k) {
            const double exact = 65536.0 * wex[k];
            const long long q = llround(exact);
            const double e = (double)q - exact;
            if (e > 0) ep += e; else en -= e;
            a.w[p][k] = (uint32_t)q;
        }
        worst = fmax(worst, 255.0 * fmax(ep, en));
    }
    a.margin = (uint32_t)ceil(worst) + 2u;
    for (int p = 0; p < P; ++p) a.lim[p] = (dr1[p] == 0 && dc1[p] == 0) ? 0u : 2u * a.margin;
    dim3 grid((g.W + GL_TILE - 1) / GL_TILE, (g.H + GL_TILE - 1) / GL_TILE, n_frames);
    hipLaunchKernelGGL(lbp_generic_kernel, grid, dim3(256), lds, s, a);
    hipLaunchKernelGGL(lbp_hist_reduce_kernel, dim3(n_frames), dim3(256), 0, s, part, lbp_generic_tiles(g), P + 2, acc, hist_slot);
    return true;
}

}  // namespace stx
