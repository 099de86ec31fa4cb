// The frame preparation of the CNN branch: PassiveChecker._preprocess_frame
// (/root/reference/app/passive_checker.py:393-418) = cv2.cvtColor(BGR2RGB), cv2.resize to the
// model's input size (bilinear, the default), astype(float32) / 255, HWC -> CHW.
//
// cv2.resize on 8-bit images works in fixed point: horizontal taps as 11-bit weights
// (saturate_cast<short>(w * 2048)), the two row sums S0 / S1 combined as
//   (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2,
// columns and rows outside the image replaced by the nearest inside.  An exact 2:1 reduction
// in both directions is the one special case: OpenCV then averages 2x2 blocks, (a+b+c+d+2)>>2.
// The tap tables are computed on the host, once per call.  One thread per output pixel: the
// output is a few KB per frame, the input rows it touches are a small part of the frame.
#include <math.h>

#include <vector>

#include "stx_internal.h"

namespace stx {

struct PrepTap {
    int32_t i0, i1;   // source column (or row) of the two taps
    int32_t w0, w1;   // their 11-bit weights
};

__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ bgr, int64_t row_stride,
                                                         int64_t frame_stride, int out_h, int out_w,
                                                         const PrepTap* __restrict__ xt, const PrepTap* __restrict__ yt,
                                                         int area2, float* __restrict__ out) {
    const int frame = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= out_h * out_w) return;
    const int dy = i / out_w, dx = i - dy * out_w;
    const PrepTap tx = xt[dx], ty = yt[dy];
    const uint8_t* r0 = bgr + frame * frame_stride + ty.i0 * row_stride;
    const uint8_t* r1 = bgr + frame * frame_stride + ty.i1 * row_stride;
    float* o = out + (int64_t)frame * 3 * out_h * out_w + i;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const int p00 = r0[3 * tx.i0 + ch], p01 = r0[3 * tx.i1 + ch];
        const int p10 = r1[3 * tx.i0 + ch], p11 = r1[3 * tx.i1 + ch];
        int v;
        if (area2) {
            v = (p00 + p01 + p10 + p11 + 2) >> 2;
        } else {
            const int s0 = p00 * tx.w0 + p01 * tx.w1, s1 = p10 * tx.w0 + p11 * tx.w1;
            v = (((ty.w0 * (s0 >> 4)) >> 16) + ((ty.w1 * (s1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        // BGR -> RGB: source channel ch lands in plane 2 - ch
        o[(int64_t)(2 - ch) * out_h * out_w] = __fdiv_rn((float)v, 255.0f);
    }
}

// Taps of one axis, as cv::resize sets them up for INTER_LINEAR.
void preprocess_taps(int src, int dst, bool horizontal, PrepTap* t) {
    const double inv_scale = (double)dst / (double)src;
    const double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        int s1;
        if (horizontal) {
            // the weights are reset where a tap would fall outside
            bool single = false;
            if (s < 0) { f = 0.f; s = 0; }
            if (s + 1 >= src) {
                single = true;  // the row pass uses S[s] * 2048 alone from here on
                if (s >= src - 1) { f = 0.f; s = src - 1; }
            }
            s1 = single ? s : s + 1;
            if (single) f = 0.f;
        } else {
            // rows are clamped, the weights stay
            s1 = s + 1;
            s = s < 0 ? 0 : (s >= src ? src - 1 : s);
            s1 = s1 < 0 ? 0 : (s1 >= src ? src - 1 : s1);
        }
        const float c0 = 1.f - f;
        t[d].i0 = s;
        t[d].i1 = s1;
        t[d].w0 = (int)lrintf(c0 * 2048.f);
        t[d].w1 = (int)lrintf(f * 2048.f);
    }
}

bool preprocess_is_area2(int h, int w, int out_h, int out_w) {
    const double sx = 1.0 / ((double)out_w / (double)w), sy = 1.0 / ((double)out_h / (double)h);
    return fabs(sx - 2.0) < 2.220446049250313e-16 && fabs(sy - 2.0) < 2.220446049250313e-16;
}

void launch_preprocess(hipStream_t s, const uint8_t* bgr, int64_t row_stride, int64_t frame_stride, int n_frames,
                       int out_h, int out_w, const void* xt, const void* yt, bool area2, float* out) {
    dim3 grid((out_h * out_w + 255) / 256, n_frames);
    hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, s, bgr, row_stride, frame_stride, out_h, out_w,
                       (const PrepTap*)xt, (const PrepTap*)yt, area2 ? 1 : 0, out);
}

size_t preprocess_tap_bytes(int n) { return (size_t)n * sizeof(PrepTap); }

void preprocess_fill_taps(int h, int w, int out_h, int out_w, void* xt_host, void* yt_host) {
    PrepTap* xt = (PrepTap*)xt_host;
    PrepTap* yt = (PrepTap*)yt_host;
    if (preprocess_is_area2(h, w, out_h, out_w)) {
        for (int d = 0; d < out_w; ++d) xt[d] = PrepTap{2 * d, 2 * d + 1, 0, 0};
        for (int d = 0; d < out_h; ++d) yt[d] = PrepTap{2 * d, 2 * d + 1, 0, 0};
        return;
    }
    preprocess_taps(w, out_w, true, xt);
    preprocess_taps(h, out_h, false, yt);
}

}  // namespace stx
