// 2-D real FFT of the gray plane and the log-magnitude statistic
// (np.fft.fft2 -> fftshift -> log(abs+1) -> np.std at
// /root/reference/app/passive_checker.py:120-123; the shift is a permutation and is skipped).
//
// Two kernels, both LDS-resident Stockham autosort transforms of arbitrary length
// (radix 4/2/3/5 butterflies, any other prime factor by a direct small DFT):
//   rows:  two image rows are packed into one complex transform (real-input trick),
//          split, and only the W/2+1 non-redundant columns are written (fp32 complex);
//   cols:  a block of adjacent spectrum columns is transformed along H and reduced on the
//          spot to sum / sum-of-squares of ln(|F|+1); the mirrored half of the spectrum is
//          accounted for by weighting interior columns twice.  Nothing is written back.
// The input is gray - m, m = the frame's gray level at the origin of the transformed region: it keeps
// the DC term out of the fp32 dynamic range, and only bin (0,0) differs, by exactly m*H*W, which the
// column kernel adds back.  Taking m from the frame (rather than a fixed 128) makes a constant frame
// transform to exact zeros, as it does in the reference's fp64.
#include "stx_internal.h"

namespace stx {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i
__device__ __forceinline__ float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }

// One Stockham pass of radix R over nb transforms of length n held in LDS.
// in/out: [nb][n]; ns = product of the radices already applied; tw[j] = exp(-2 pi i j / n).
__device__ void fft_pass(const float2* __restrict__ in, float2* __restrict__ out, int n, int R, int ns, int nb,
                         const float2* __restrict__ tw, int tid, int nthreads) {
    const int m = n / R;            // butterflies per transform
    const int twstep = n / (ns * R);
    if (R != 2 && R != 3 && R != 4 && R != 5) {
        // direct DFT of size R on the twiddled inputs (rare sizes only): one output per thread and step, so
        // that a prime length (a single pass with one butterfly) still spreads over the workgroup
        for (int b = 0; b < nb; ++b) {
            const float2* ib = in + b * n;
            float2* ob = out + b * n;
            for (int idx = tid; idx < m * R; idx += nthreads) {
                const int q = idx / m, j = idx - q * m;
                const int k = j % ns;
                const int o = (j - k) * R + k;
                const int t1 = k * twstep;
                float2 accv = make_float2(0.f, 0.f);
                long e = 0;  // (r*t1 + r*q*m) mod n
                const long step = ((long)t1 + (long)q * m) % n;
                for (int r = 0; r < R; ++r) {
                    accv = cadd(accv, cmul(ib[j + r * m], tw[e]));
                    e += step;
                    if (e >= n) e -= n;
                }
                ob[o + q * ns] = accv;
            }
        }
        return;
    }
    for (int b = 0; b < nb; ++b) {
        const float2* ib = in + b * n;
        float2* ob = out + b * n;
        for (int j = tid; j < m; j += nthreads) {
            const int k = j % ns;
            const int o = (j - k) * R + k;
            const int t1 = k * twstep;
            if (R == 4) {
                float2 v0 = ib[j], v1 = ib[j + m], v2 = ib[j + 2 * m], v3 = ib[j + 3 * m];
                if (k) {
                    v1 = cmul(v1, tw[t1]);
                    v2 = cmul(v2, tw[2 * t1]);
                    v3 = cmul(v3, tw[3 * t1]);
                }
                const float2 a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), a3 = cmul_mi(csub(v1, v3));
                ob[o] = cadd(a0, a2);
                ob[o + ns] = cadd(a1, a3);
                ob[o + 2 * ns] = csub(a0, a2);
                ob[o + 3 * ns] = csub(a1, a3);
            } else if (R == 2) {
                float2 v0 = ib[j], v1 = ib[j + m];
                if (k) v1 = cmul(v1, tw[t1]);
                ob[o] = cadd(v0, v1);
                ob[o + ns] = csub(v0, v1);
            } else if (R == 3) {
                float2 v0 = ib[j], v1 = ib[j + m], v2 = ib[j + 2 * m];
                if (k) {
                    v1 = cmul(v1, tw[t1]);
                    v2 = cmul(v2, tw[2 * t1]);
                }
                const float2 s = cadd(v1, v2), d = csub(v1, v2);
                const float2 h = make_float2(v0.x - 0.5f * s.x, v0.y - 0.5f * s.y);
                const float c = 0.86602540378443864676f;
                const float2 r = make_float2(c * d.y, -c * d.x);
                ob[o] = cadd(v0, s);
                ob[o + ns] = cadd(h, r);
                ob[o + 2 * ns] = csub(h, r);
            } else if (R == 5) {
                float2 v0 = ib[j], v1 = ib[j + m], v2 = ib[j + 2 * m], v3 = ib[j + 3 * m], v4 = ib[j + 4 * m];
                if (k) {
                    v1 = cmul(v1, tw[t1]);
                    v2 = cmul(v2, tw[2 * t1]);
                    v3 = cmul(v3, tw[3 * t1]);
                    v4 = cmul(v4, tw[4 * t1]);
                }
                const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
                const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
                const float2 a1 = cadd(v1, v4), a2 = cadd(v2, v3), b1 = csub(v1, v4), b2 = csub(v2, v3);
                const float2 p1 = make_float2(v0.x + c1 * a1.x + c2 * a2.x, v0.y + c1 * a1.y + c2 * a2.y);
                const float2 p2 = make_float2(v0.x + c2 * a1.x + c1 * a2.x, v0.y + c2 * a1.y + c1 * a2.y);
                const float2 q1 = cmul_mi(make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
                const float2 q2 = cmul_mi(make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
                ob[o] = make_float2(v0.x + a1.x + a2.x, v0.y + a1.y + a2.y);
                ob[o + ns] = cadd(p1, q1);
                ob[o + 2 * ns] = cadd(p2, q2);
                ob[o + 3 * ns] = csub(p2, q2);
                ob[o + 4 * ns] = csub(p1, q1);
            }
        }
    }
}

// runs the whole plan; returns the buffer that holds the result
__device__ float2* fft_run(float2* a, float2* b, const FftPlan& plan, int nb, const float2* tw, int tid, int nthreads) {
    int ns = 1;
    for (int p = 0; p < plan.npass; ++p) {
        const int R = plan.radix[p];
        fft_pass(a, b, plan.n, R, ns, nb, tw, tid, nthreads);
        __syncthreads();
        float2* t = a; a = b; b = t;
        ns *= R;
    }
    return a;
}

struct FftRowsArgs {
    const uint8_t* gray;
    int gpitch, grows;
    int foy, fox, fh, fw, fwh;
    int cbw, nkb;  // spectrum layout [frame][nkb][fh][cbw]
    int nb;  // complex transforms (row pairs) per workgroup
    FftPlan plan;
    const float2* tw;
    float2* spec;
};

__global__ __launch_bounds__(256) void fft_rows_kernel(FftRowsArgs a) {
    extern __shared__ float2 lds[];
    const int n = a.fw, nb = a.nb, tid = threadIdx.x;
    float2* buf0 = lds;
    float2* buf1 = lds + nb * n;
    const int frame = blockIdx.y;
    const int row0 = blockIdx.x * 2 * nb;
    const uint8_t* gp = a.gray + (int64_t)frame * a.grows * a.gpitch;
    const int m = gp[(int64_t)(a.foy + GRAY_YOFF) * a.gpitch + a.fox + GRAY_XOFF];
    for (int b = 0; b < nb; ++b) {
        const int ra = row0 + 2 * b, rb = ra + 1;
        const uint8_t* pa = gp + (int64_t)(a.foy + ra + GRAY_YOFF) * a.gpitch + a.fox + GRAY_XOFF;
        const uint8_t* pb = pa + a.gpitch;
        for (int x = tid; x < n; x += 256) {
            const float va = ra < a.fh ? (float)((int)pa[x] - m) : 0.f;
            const float vb = rb < a.fh ? (float)((int)pb[x] - m) : 0.f;
            buf0[b * n + x] = make_float2(va, vb);
        }
    }
    __syncthreads();
    const float2* z = fft_run(buf0, buf1, a.plan, nb, a.tw, tid, 256);
    for (int b = 0; b < nb; ++b) {
        const int ra = row0 + 2 * b, rb = ra + 1;
        if (ra >= a.fh) break;
        float2* base = a.spec + (int64_t)frame * a.nkb * a.fh * a.cbw;
        for (int k = tid; k < a.fwh; k += 256) {
            const float2 zk = z[b * n + k];
            const float2 zc = z[b * n + (k ? n - k : 0)];
            const int kb = k / a.cbw, kc = k - kb * a.cbw;
            float2* o = base + ((int64_t)kb * a.fh + ra) * a.cbw + kc;
            o[0] = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
            if (rb < a.fh) o[a.cbw] = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
        }
    }
}

struct FftColsArgs {
    const float2* spec;
    int fh, fw, fwh;
    int cb;  // columns per workgroup
    int nblocks;
    FftPlan plan;
    const float2* tw;
    double2* part;
    const uint8_t* gray0;  // gray level at the origin of frame 0's transformed region; frames gstride apart
    int64_t gstride;
};

__global__ __launch_bounds__(256) void fft_cols_kernel(FftColsArgs a) {
    extern __shared__ float2 lds[];
    __shared__ double red[2][4];
    const int n = a.fh, cb = a.cb, tid = threadIdx.x;
    float2* buf0 = lds;
    float2* buf1 = lds + cb * n;
    const int frame = blockIdx.y;
    const int k0 = blockIdx.x * cb;
    // one column block [n][cb] is contiguous
    const float2* sp = a.spec + ((int64_t)frame * a.nblocks + blockIdx.x) * (int64_t)n * cb;
    for (int i = tid; i < cb * n; i += 256) {
        const int y = i / cb, c = i - y * cb;
        const int k = k0 + c;
        buf0[c * n + y] = k < a.fwh ? sp[i] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    const float2* f = fft_run(buf0, buf1, a.plan, cb, a.tw, tid, 256);
    double s = 0.0, s2 = 0.0;
    const float dc_fix = (float)a.gray0[(int64_t)frame * a.gstride] * (float)a.fh * (float)a.fw;
    for (int c = 0; c < cb; ++c) {
        const int k = k0 + c;
        if (k >= a.fwh) break;
        const double wgt = (k == 0 || ((a.fw & 1) == 0 && k == a.fw / 2)) ? 1.0 : 2.0;
        // accumulate in fp64: the variance is a difference of two nearly equal sums
        double ls = 0.0, ls2 = 0.0;
        for (int y = tid; y < n; y += 256) {
            float2 v = f[c * n + y];
            if (k == 0 && y == 0) v.x += dc_fix;
            const float mag = sqrtf(v.x * v.x + v.y * v.y);
            const double l = (double)logf(mag + 1.0f);
            ls += l;
            ls2 += l * l;
        }
        s += wgt * ls;
        s2 += wgt * ls2;
    }
    // deterministic block reduction: fixed shuffle tree, then waves in order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        a.part[(int64_t)frame * a.nblocks + blockIdx.x] =
            make_double2(((red[0][0] + red[0][1]) + red[0][2]) + red[0][3], ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]);
    }
}

bool make_fft_plan(int n, FftPlan* plan) {
    plan->n = n;
    plan->npass = 0;
    int r = n;
    auto push = [&](int f) {
        if (plan->npass >= FFT_MAX_PASSES) return false;
        plan->radix[plan->npass++] = f;
        return true;
    };
    while (r % 4 == 0) { if (!push(4)) return false; r /= 4; }
    while (r % 2 == 0) { if (!push(2)) return false; r /= 2; }
    while (r % 3 == 0) { if (!push(3)) return false; r /= 3; }
    while (r % 5 == 0) { if (!push(5)) return false; r /= 5; }
    for (int p = 7; (long)p * p <= r; p += 2)
        while (r % p == 0) { if (!push(p)) return false; r /= p; }
    if (r > 1 && !push(r)) return false;
    return true;
}

constexpr int FFT_LDS_BUDGET = 64 * 1024;
constexpr int FFT_LDS_MAX = 159 * 1024;  // dynamic part; 1 KiB of the CU's 160 KiB is left for static LDS

// LDS bytes for the generic row kernel and the number of row pairs per workgroup
static int fft_lds_bytes_rows(int n, int* nb_out) {
    int nb = FFT_LDS_BUDGET / (2 * n * (int)sizeof(float2));
    if (nb < 1) nb = 1;
    if (nb > 8) nb = 8;
    *nb_out = nb;
    return 2 * nb * n * (int)sizeof(float2);
}

// Spectrum layout [frame][nkb][fh][cbw]: blocks of cbw adjacent columns, each block contiguous,
// so the column pass streams its input.  cbw is set by what the column kernel can hold in LDS.
FftLayout fft_choose_layout(const Geom& g) {
    FftLayout l;
    l.fast_rows = fft_fast_available(g.fw);
    l.fast_cols = fft_fast_available(g.fh);
    int cbw;
    if (l.fast_cols) {
        const int stride = fft_fast_stride(g.fh);
        // two workgroups per CU (each at most half of the LDS), so that one can load its block while
        // the other computes
        cbw = 16;
        while (cbw > 1 && (size_t)cbw * stride * sizeof(float2) > (size_t)FFT_LDS_MAX / 2) cbw >>= 1;
        const int tpw = fft_fast_tpw(g.fh);
        if (cbw < tpw) cbw = tpw;
    } else {
        cbw = FFT_LDS_MAX / (2 * g.fh * (int)sizeof(float2));
        if (cbw > 8) cbw = 8;
        if (cbw < 1) cbw = 1;
        while (cbw & (cbw - 1)) cbw &= cbw - 1;  // the block width is always a power of two
    }
    l.cbw = cbw;
    l.nkb = (g.fwh + cbw - 1) / cbw;
    l.parts_per_frame = l.fast_cols ? l.nkb * cbw : l.nkb;
    return l;
}

void launch_fft_rows(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, const FftLayout& lay,
                     const FftPlan& plan, const float2* tw, float2* spec) {
    if (lay.fast_rows && launch_fft_rows_fast(s, gray, n_frames, g, lay, tw, spec)) return;
    FftRowsArgs a;
    a.gray = gray; a.gpitch = g.gpitch; a.grows = g.grows;
    a.foy = g.foy; a.fox = g.fox; a.fh = g.fh; a.fw = g.fw; a.fwh = g.fwh;
    a.cbw = lay.cbw; a.nkb = lay.nkb;
    int nb;
    const int lds = fft_lds_bytes_rows(g.fw, &nb);
    a.nb = nb; a.plan = plan; a.tw = tw; a.spec = spec;
    dim3 grid((g.fh + 2 * nb - 1) / (2 * nb), n_frames);
    hipLaunchKernelGGL(fft_rows_kernel, grid, dim3(256), lds, s, a);
}

void launch_fft_cols(hipStream_t s, const uint8_t* gray, const float2* spec, int n_frames, const Geom& g,
                     const FftLayout& lay, const FftPlan& plan, const float2* tw, double2* part) {
    const uint8_t* gray0 = gray + (int64_t)(g.foy + GRAY_YOFF) * g.gpitch + g.fox + GRAY_XOFF;
    const int64_t gstride = (int64_t)g.grows * g.gpitch;
    if (lay.fast_cols && launch_fft_cols_fast(s, gray0, gstride, spec, n_frames, g, lay, tw, part)) return;
    FftColsArgs a;
    a.spec = spec; a.fh = g.fh; a.fw = g.fw; a.fwh = g.fwh;
    a.cb = lay.cbw;
    a.nblocks = lay.nkb;
    a.plan = plan; a.tw = tw; a.part = part;
    a.gray0 = gray0; a.gstride = gstride;
    const int lds = 2 * a.cb * g.fh * (int)sizeof(float2);
    dim3 grid(a.nblocks, n_frames);
    hipLaunchKernelGGL(fft_cols_kernel, grid, dim3(256), lds, s, a);
}

hipError_t fft_set_lds_limits() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft_rows_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, FFT_LDS_MAX);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft_cols_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, FFT_LDS_MAX);
    if (e != hipSuccess) return e;
    return fft_fast_init();
}

}  // namespace stx
