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
#include "stx_fft_generic.h"

namespace stx {

typedef Cpx<float> cf;

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
    cf* buf0 = reinterpret_cast<cf*>(lds);
    cf* buf1 = buf0 + nb * n;  // unused by the chirp-z transform (nb = 1, one buffer of bz_m elements)
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
            buf0[b * n + x] = cf{va, vb};
        }
    }
    __syncthreads();
    const cf* z = buf0;
    if (a.plan.bz_m) fft_bluestein<float>(buf0, n, a.plan.bz_m, reinterpret_cast<const cf*>(a.tw), tid, 256);
    else z = fft_run<float>(buf0, buf1, a.plan, nb, reinterpret_cast<const cf*>(a.tw), tid, 256);
    for (int b = 0; b < nb; ++b) {
        const int ra = row0 + 2 * b, rb = ra + 1;
        if (ra >= a.fh) break;
        float2* base = a.spec + (int64_t)frame * a.nkb * a.fh * a.cbw;
        for (int k = tid; k < a.fwh; k += 256) {
            const cf zk = z[b * n + k];
            const cf zc = z[b * n + (k ? n - k : 0)];
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
    FftPart* part;
    float small;
    const uint8_t* gray0;  // gray level at the origin of frame 0's transformed region; frames gstride apart
    int64_t gstride;
};

__global__ __launch_bounds__(256) void fft_cols_kernel(FftColsArgs a) {
    extern __shared__ float2 lds[];
    __shared__ double red[5][4];
    const int n = a.fh, cb = a.cb, tid = threadIdx.x;
    cf* buf0 = reinterpret_cast<cf*>(lds);
    cf* buf1 = buf0 + cb * n;
    const int frame = blockIdx.y;
    const int k0 = blockIdx.x * cb;
    // one column block [n][cb] is contiguous
    const float2* sp = a.spec + ((int64_t)frame * a.nblocks + blockIdx.x) * (int64_t)n * cb;
    for (int i = tid; i < cb * n; i += 256) {
        const int y = i / cb, c = i - y * cb;
        const int k = k0 + c;
        const float2 v = k < a.fwh ? sp[i] : make_float2(0.f, 0.f);
        buf0[c * n + y] = cf{v.x, v.y};
    }
    __syncthreads();
    const cf* f = buf0;
    if (a.plan.bz_m) fft_bluestein<float>(buf0, n, a.plan.bz_m, reinterpret_cast<const cf*>(a.tw), tid, 256);
    else f = fft_run<float>(buf0, buf1, a.plan, cb, reinterpret_cast<const cf*>(a.tw), tid, 256);
    double s = 0.0, s2 = 0.0, t1 = 0.0, t2 = 0.0, e2 = 0.0;
    const float dc_fix = (float)a.gray0[(int64_t)frame * a.gstride] * (float)a.fh * (float)a.fw;
    for (int c = 0; c < cb; ++c) {
        const int k = k0 + c;
        if (k >= a.fwh) break;
        const double wgt = (k == 0 || ((a.fw & 1) == 0 && k == a.fw / 2)) ? 1.0 : 2.0;
        // accumulate in fp64: the variance is a difference of two nearly equal sums
        double ls = 0.0, ls2 = 0.0;
        float lt1 = 0.f, lt2 = 0.f, le2 = 0.f;
        for (int y = tid; y < n; y += 256) {
            cf v = f[c * n + y];
            le2 += v.x * v.x + v.y * v.y;
            if (k == 0 && y == 0) v.x += dc_fix;
            const float mag = sqrtf(v.x * v.x + v.y * v.y);
            const float lf = logf(mag + 1.0f);
            const double l = (double)lf;
            ls += l;
            ls2 += l * l;
            if (mag < a.small) {
                const float r = 1.0f / (mag + 1.0f);
                lt1 += r;
                lt2 += r * lf;
            }
        }
        s += wgt * ls;
        s2 += wgt * ls2;
        t1 += wgt * (double)lt1;
        t2 += wgt * (double)lt2;
        e2 += wgt * (double)le2;
    }
    // deterministic block reduction: fixed shuffle tree, then waves in order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        s2 += __shfl_xor(s2, o, 64);
        t1 += __shfl_xor(t1, o, 64);
        t2 += __shfl_xor(t2, o, 64);
        e2 += __shfl_xor(e2, o, 64);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = s2; red[2][tid >> 6] = t1; red[3][tid >> 6] = t2; red[4][tid >> 6] = e2; }
    __syncthreads();
    if (tid == 0) {
        auto tot = [&](int q) { return ((red[q][0] + red[q][1]) + red[q][2]) + red[q][3]; };
        a.part[(int64_t)frame * a.nblocks + blockIdx.x] = FftPart{tot(0), tot(1), tot(2), tot(3), tot(4)};
    }
}

bool make_fft_plan(int n, FftPlan* plan) {
    plan->n = n;
    plan->npass = 0;
    plan->bz_m = 0;
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
    int largest = 1;
    for (int p = 0; p < plan->npass; ++p) largest = plan->radix[p] > largest ? plan->radix[p] : largest;
    if (largest > FFT_BLUESTEIN_MIN_PRIME) {
        int m = 1;
        while (m < 2 * n - 1) m <<= 1;
        plan->bz_m = m;
    }
    return true;
}

// exp(-i pi (k^2 mod 2n) / n): the square is reduced in integers first, the angle stays small
static void chirp_at(long long k, int n, double* re, double* im) {
    const long long q = (k * k) % (2LL * n);
    const double a = -M_PI * (double)q / (double)n;
    *re = cos(a);
    *im = sin(a);
}

void fft_fill_table(const FftPlan& p, double* out, double* work) {
    const int n = p.n;
    if (!p.bz_m) {
        for (int k = 0; k < n; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)n;
            out[2 * k] = cos(a);
            out[2 * k + 1] = sin(a);
        }
        return;
    }
    const int m = p.bz_m;
    double* chirp = out;
    double* twm = out + 2 * (size_t)n;
    double* bhat = twm + m;  // m/2 complex entries before it
    for (int k = 0; k < n; ++k) chirp_at(k, n, &chirp[2 * k], &chirp[2 * k + 1]);
    for (int j = 0; j < m / 2; ++j) {
        const double a = -2.0 * M_PI * (double)j / (double)m;
        twm[2 * j] = cos(a);
        twm[2 * j + 1] = sin(a);
    }
    // the convolution kernel conj(c[j]) at j and m-j, transformed (decimation in frequency, as on the device, so the
    // result is already in bit-reversed order) and divided by m
    std::vector<double> own;
    if (!work) {
        own.resize(2 * (size_t)m);
        work = own.data();
    }
    double* b = work;
    for (size_t j = 0; j < 2 * (size_t)m; ++j) b[j] = 0.0;
    for (int j = 0; j < n; ++j) {
        double re, im;
        chirp_at(j, n, &re, &im);
        b[2 * (size_t)j] = re;
        b[2 * (size_t)j + 1] = -im;
        if (j) {
            b[2 * (size_t)(m - j)] = re;
            b[2 * (size_t)(m - j) + 1] = -im;
        }
    }
    for (int half = m >> 1, step = 1; half >= 1; half >>= 1, step <<= 1)
        for (int bfly = 0; bfly < m / 2; ++bfly) {
            const int i = bfly & (half - 1);
            const size_t lo = (size_t)((bfly - i) << 1) + i, hi = lo + half;
            const double ur = b[2 * lo], ui = b[2 * lo + 1], vr = b[2 * hi], vi = b[2 * hi + 1];
            const double wr = twm[2 * (size_t)(i * step)], wi = twm[2 * (size_t)(i * step) + 1];
            b[2 * lo] = ur + vr;
            b[2 * lo + 1] = ui + vi;
            const double dr = ur - vr, di = ui - vi;
            b[2 * hi] = dr * wr - di * wi;
            b[2 * hi + 1] = dr * wi + di * wr;
        }
    for (size_t j = 0; j < 2 * (size_t)m; ++j) bhat[j] = b[j] / (double)m;
}

constexpr int FFT_LDS_BUDGET = 64 * 1024;
constexpr int FFT_LDS_MAX = 159 * 1024;  // dynamic part; 1 KiB of the CU's 160 KiB is left for static LDS

// LDS bytes for the generic row kernel and the number of row pairs per workgroup
static int fft_lds_bytes_rows(const FftPlan& plan, int* nb_out) {
    const int n = plan.n;
    if (plan.bz_m) {  // one transform per workgroup in one buffer of the convolution length
        *nb_out = 1;
        return plan.bz_m * (int)sizeof(float2);
    }
    int nb = FFT_LDS_BUDGET / (2 * n * (int)sizeof(float2));
    if (nb < 1) nb = 1;
    if (nb > 8) nb = 8;
    *nb_out = nb;
    return 2 * nb * n * (int)sizeof(float2);
}

// Spectrum layout [frame][nkb][fh][cbw]: blocks of cbw adjacent columns, each block contiguous,
// so the column pass streams its input.  cbw is set by what the column kernel can hold in LDS.
FftLayout fft_choose_layout(const Geom& g, const FftPlan& plan_h) {
    FftLayout l;
    l.fast_rows = fft_fast_available(g.fw);
    l.fast_cols = fft_fast_available(g.fh);
    int cbw;
    if (l.fast_cols) {
        cbw = fft_fast_cbw(g.fh);
    } else if (plan_h.bz_m) {
        cbw = 1;  // the chirp-z transform takes the workgroup's whole buffer for one column
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
    const int lds = fft_lds_bytes_rows(plan, &nb);
    a.nb = nb; a.plan = plan; a.tw = tw; a.spec = spec;
    dim3 grid((g.fh + 2 * nb - 1) / (2 * nb), n_frames);
    hipLaunchKernelGGL(fft_rows_kernel, grid, dim3(256), lds, s, a);
}

void launch_fft_cols(hipStream_t s, const uint8_t* gray, const float2* spec, int n_frames, const Geom& g,
                     const FftLayout& lay, const FftPlan& plan, const float2* tw, FftPart* part, float small) {
    const uint8_t* gray0 = gray + (int64_t)(g.foy + GRAY_YOFF) * g.gpitch + g.fox + GRAY_XOFF;
    const int64_t gstride = (int64_t)g.grows * g.gpitch;
    if (lay.fast_cols && launch_fft_cols_fast(s, gray0, gstride, spec, n_frames, g, lay, tw, part, small)) return;
    FftColsArgs a;
    a.spec = spec; a.fh = g.fh; a.fw = g.fw; a.fwh = g.fwh;
    a.cb = lay.cbw;
    a.nblocks = lay.nkb;
    a.plan = plan; a.tw = tw; a.part = part; a.small = small;
    a.gray0 = gray0; a.gstride = gstride;
    const int lds = plan.bz_m ? plan.bz_m * (int)sizeof(float2) : 2 * a.cb * g.fh * (int)sizeof(float2);
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
    e = fft64_set_lds_limits();
    if (e != hipSuccess) return e;
    return fft_fast_init();
}

// Rounding noise of the fp32 transform: every butterfly stage adds a relative 2^-24 or so of the running values,
// whose root-mean-square over the bins is sqrt(sum x^2) <= 255 sqrt(N) (Parseval); 3e-7 leaves room for the
// log2(N)/2 stages' accumulation and the twiddle errors.
// A chirp-z dimension goes through two transforms four times its length and three products: four times the noise.
double fft_f32_bin_error_rms(int fh, int fw, double rms) {
    FftPlan ph, pw;
    const bool bz = (make_fft_plan(fh, &ph) && ph.bz_m) || (make_fft_plan(fw, &pw) && pw.bz_m);
    return (bz ? 4.0 : 1.0) * 3e-7 * rms * sqrt((double)fh * (double)fw);
}
double fft_f32_bin_error(int fh, int fw) { return fft_f32_bin_error_rms(fh, fw, 255.0); }

}  // namespace stx
