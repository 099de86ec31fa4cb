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
// A wave covers 256 columns, four per lane: the boundary columns are the first pixel of every other lane, and the
// per-wave sums come from one cross-lane reduction per statistic at the end.
#include "stx_internal.h"

namespace stx {

constexpr int X_ROWS = 16;  // rows per wave
constexpr int X_STATS = 8;  // partial sums per workgroup (seven used)

// A lane walks down four adjacent columns, kept as two registers of packed 16-bit pairs.  A row is one aligned
// 32-bit load per lane; the pixel left of a lane's four and the pixel right of them come from the neighbouring lanes'
// words (DPP shifts), and the first and last lane of the wave load theirs with a second, mostly idle, load.  Every
// load of a strip (X_ROWS + 2 rows) is issued before the first use.  Borders are replicated: rows by clamping the
// (scalar) row offset, columns by per-lane byte selectors that pick a lane's own edge pixel where the neighbour lies
// outside the image.  The sums of a lane's 16 x 4 pixels fit 32 bits.
constexpr int X_COLS = 256;  // columns per wave

__device__ __forceinline__ uint32_t xperm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ uint32_t xpk_mad2(uint32_t a, uint32_t c) {  // 2 a + c per 16-bit half
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t xpk_sub(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t xpk_add(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t xpk_abs(uint32_t a) {
    uint32_t n, d;
    asm("v_pk_sub_i16 %0, 0, %1" : "=v"(n) : "v"(a));
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(n));
    return d;
}
__device__ __forceinline__ uint32_t xsad(uint32_t a, uint32_t b, uint32_t acc) {  // |a.lo-b.lo| + |a.hi-b.hi| + acc
    uint32_t d;
    asm("v_sad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}

__global__ __launch_bounds__(256) void extended_kernel(const uint8_t* __restrict__ gray, int H, int W, int gpitch, int grows,
                                                       int high, unsigned long long* __restrict__ part) {
    __shared__ unsigned long long red[4][X_STATS];
    typedef unsigned short xus2 __attribute__((ext_vector_type(2)));
    const int frame = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: row offsets stay scalar
    const int x = blockIdx.x * X_COLS + 4 * lane;
    const uint8_t* gp = gray + (int64_t)frame * grows * gpitch + (int64_t)GRAY_YOFF * gpitch + GRAY_XOFF;
    uint32_t v[X_STATS] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int y0 = (blockIdx.y * 4 + wv) * X_ROWS;
    if (y0 < H) {  // wave-uniform; lanes past the right edge stay in (their neighbours read them) and add nothing
        constexpr uint32_t NOWHERE = 0x7FFF0000u;  // past the end of the buffer: such a load returns zero
        const int nv = W - x < 0 ? 0 : (W - x > 4 ? 4 : W - x);  // valid pixels of this lane
        // buffer loads: the row is a scalar offset, clamped on the scalar side, the column a per-lane offset
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(gp), 0, H * gpitch, 0x00020000);
        const uint32_t off_c = nv > 0 ? (uint32_t)x : NOWHERE;
        const uint32_t off_h = (lane == 0 && x > 0) ? (uint32_t)(x - 4) : (lane == 63 && x + 4 < W) ? (uint32_t)(x + 4) : NOWHERE;
        uint32_t cw[X_ROWS + 2], hw[X_ROWS + 2];
#pragma unroll
        for (int i = 0; i < X_ROWS + 2; ++i) {
            const int y = y0 - 1 + i;
            const int ro = (y < 0 ? 0 : (y < H ? y : H - 1)) * gpitch;
            cw[i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off_c, ro, 0);
            hw[i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off_h, ro, 0);
        }
        // byte selectors over {left word (bytes 4..7), own word (0..3)} and {right word (4..7), own word (0..3)}:
        // pairs (left, p0) (p1, p2) and (right of p0, right of p1) (right of p2, right of p3), edges replicated
        const int last = nv > 0 ? nv - 1 : 0;
        auto own = [&](int q) -> uint32_t { return (uint32_t)(q < last ? q : last); };
        const uint32_t sel_l01 = (x > 0 ? 7u : 0u) | 0x0c00u | (0u << 16) | 0x0c000000u;
        const uint32_t sel_r01 = own(1) | 0x0c00u | (own(2) << 16) | 0x0c000000u;
        const uint32_t sel_r23 = own(3) | 0x0c00u | ((nv == 4 && x + 4 < W ? 4u : own(3)) << 16) | 0x0c000000u;
        // validity of the pairs (p0,p1) and (p2,p3)
        const uint32_t vm01 = (nv > 0 ? 0xFFFFu : 0u) | (nv > 1 ? 0xFFFF0000u : 0u);
        const uint32_t vm23 = (nv > 2 ? 0xFFFFu : 0u) | (nv > 3 ? 0xFFFF0000u : 0u);
        const uint32_t edge0 = ((x & 7) == 0) ? 0x0000FFFFu : 0u;  // pixel 0 of every other lane sits on a block boundary
        const uint32_t high2 = ((uint32_t)(high < -1 ? -1 : (high > 4095 ? 4095 : high)) & 0xFFFFu) * 0x00010001u;

        // per image row: the pixel pairs (p0,p1) (p2,p3) and their left neighbours, zero where the pixel is outside
        // the image (they only feed differences), and the horizontal Sobel sums l + 2m + r and differences r - l
        struct Row { uint32_t m01, m23, l01, l23, s01, s23, d01, d23; };
        auto make_row = [&](int i, Row& o) {
            const uint32_t c = cw[i];
            const uint32_t lw = (uint32_t)__builtin_amdgcn_update_dpp((int)hw[i], (int)c, 0x138, 0xf, 0xf, false);  // wave_shr:1
            const uint32_t rw = (uint32_t)__builtin_amdgcn_update_dpp((int)hw[i], (int)c, 0x130, 0xf, 0xf, false);  // wave_shl:1
            const uint32_t m01 = xperm(0, c, 0x0c010c00u), m23 = xperm(0, c, 0x0c030c02u);
            const uint32_t l01 = xperm(lw, c, sel_l01), l23 = xperm(0, c, 0x0c020c01u);
            const uint32_t r01 = xperm(rw, c, sel_r01), r23 = xperm(rw, c, sel_r23);
            o.s01 = xpk_mad2(m01, l01 + r01);  // halves of at most 510: a plain add
            o.s23 = xpk_mad2(m23, l23 + r23);
            o.d01 = xpk_sub(r01, l01);
            o.d23 = xpk_sub(r23, l23);
            o.m01 = m01 & vm01;
            o.m23 = m23 & vm23;
            o.l01 = l01 & vm01;
            o.l23 = l23 & vm23;
        };
        static_assert(X_ROWS % 8 == 0, "row boundaries by the unrolled index");
        uint32_t mag_s = 0, mag_q = 0, strong2 = 0, h_all = 0, h_edge = 0, v_edge = 0, v_in = 0;
        Row ru, rc, rb;
        make_row(0, ru);
        make_row(1, rc);
#pragma unroll
        for (int i = 0; i < X_ROWS; ++i) {
            const int y = y0 + i;
            make_row(i + 2, rb);
            if (y < H) {
                const uint32_t gx01 = xpk_mad2(rc.d01, xpk_add(ru.d01, rb.d01)), gx23 = xpk_mad2(rc.d23, xpk_add(ru.d23, rb.d23));
                const uint32_t gy01 = xpk_sub(rb.s01, ru.s01), gy23 = xpk_sub(rb.s23, ru.s23);
                const uint32_t mag01 = (xpk_abs(gx01) + xpk_abs(gy01)) & vm01;  // halves of at most 2040
                const uint32_t mag23 = (xpk_abs(gx23) + xpk_abs(gy23)) & vm23;
                mag_s = xsad(mag01, 0u, xsad(mag23, 0u, mag_s));
                mag_q = __builtin_amdgcn_udot2(__builtin_bit_cast(xus2, mag01), __builtin_bit_cast(xus2, mag01), mag_q, false);
                mag_q = __builtin_amdgcn_udot2(__builtin_bit_cast(xus2, mag23), __builtin_bit_cast(xus2, mag23), mag_q, false);
                // mag > high: the sign of high - mag per half, counted in two 16-bit counters
                strong2 += ((xpk_sub(high2, mag01) >> 15) & 0x00010001u & vm01) + ((xpk_sub(high2, mag23) >> 15) & 0x00010001u & vm23);
                // |p - left neighbour|: all of them, and apart the ones on a block boundary (the leftmost pixel of the
                // image has itself as neighbour and adds nothing)
                h_all = xsad(rc.m01, rc.l01, xsad(rc.m23, rc.l23, h_all));
                h_edge = xsad(rc.m01 & edge0, rc.l01 & edge0, h_edge);
                // |p - upper neighbour| (row 0 has itself as neighbour)
                if ((i & 7) == 0) v_edge = xsad(rc.m01, ru.m01, xsad(rc.m23, ru.m23, v_edge));
                else v_in = xsad(rc.m01, ru.m01, xsad(rc.m23, ru.m23, v_in));
            }
            ru = rc; rc = rb;
        }
        v[0] = mag_s;
        v[1] = mag_q;
        v[2] = (strong2 & 0xFFFFu) + (strong2 >> 16);
        v[3] = h_edge;
        v[4] = h_all - h_edge;
        v[5] = v_edge;
        v[6] = v_in;
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
    return dim3((g.W + X_COLS - 1) / X_COLS, (g.H + 4 * X_ROWS - 1) / (4 * X_ROWS), n_frames);
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
