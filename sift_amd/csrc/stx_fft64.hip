// The 2-D transform and log-magnitude statistic of /root/reference/app/passive_checker.py:120-123 in double
// precision, as numpy computes them (np.fft.fft2 of the uint8 plane is a complex128 transform).
//
// The fp32 kernels are what runs by default; their rounding noise is about 1e-7 of the spectrum's
// root-mean-square per bin.  That is invisible in ln(|F|+1) wherever |F| is well above it (any frame with sensor
// noise), but an image whose exact spectrum is zero in many bins (ramps, step edges, flat synthetic content) gets
// ln(1 + noise) there instead of 0.  The host decides per frame from the sums the fp32 column pass leaves
// (FftPart::t1, t2) and sends such frames through these kernels: same structure as the generic fp32 pair
// (row pairs packed into one complex transform, then columns), up to FFT64_BATCH frames per launch, any size.
#include "stx_fft_generic.h"

namespace stx {

typedef Cpx<double> cd;

constexpr int F64_THREADS = 256;
constexpr int F64_LDS_MAX = 156 * 1024;  // dynamic LDS a workgroup may take (the rest is left for static LDS)
constexpr int F64_MAX_GRID = 512;        // workgroups per launch: each walks over its transforms, and owns one scratch slice

// One transform at a time per workgroup.  Both pass buffers live in LDS when they fit (n <= 4992); longer transforms
// keep the first in LDS and the second in the workgroup's slice of global scratch.  The chirp-z transform of a length
// with a large prime factor works in one buffer of its convolution length: in LDS up to 8192 elements, else in the
// scratch slice.
static size_t f64_lds_bytes(const FftPlan& p) {
    if (p.bz_m) {
        const size_t one = (size_t)p.bz_m * sizeof(cd);
        return one <= (size_t)F64_LDS_MAX ? one : 0;
    }
    const size_t two = 2 * (size_t)p.n * sizeof(cd);
    return two <= (size_t)F64_LDS_MAX ? two : (size_t)p.n * sizeof(cd);
}
// elements of global scratch per workgroup
static size_t f64_scratch_elems(const FftPlan& p) {
    if (p.bz_m) return (size_t)p.bz_m * sizeof(cd) <= (size_t)F64_LDS_MAX ? 0 : (size_t)p.bz_m;
    return 2 * (size_t)p.n * sizeof(cd) > (size_t)F64_LDS_MAX ? (size_t)p.n : 0;
}
static int f64_grid(int units) { return units < F64_MAX_GRID ? units : F64_MAX_GRID; }

struct Rows64Args {
    const uint8_t* gray;  // padded planes of the chunk
    int64_t gstride;      // bytes from one frame's plane to the next
    int frames[FFT64_BATCH];  // blockIdx.y -> frame of the chunk
    int64_t spec_stride, scratch_stride;  // elements from one batch slot to the next
    int gpitch;
    int foy, fox, fh, fw, fwh;
    FftPlan plan;
    const cd* tw;
    cd* spec;     // [fh][fwh]
    cd* scratch;  // [workgroups][scratch_elems] or null
    int scratch_elems;
};

__global__ __launch_bounds__(F64_THREADS) void fft64_rows_kernel(Rows64Args a) {
    extern __shared__ double2 lds64[];
    const int n = a.fw, tid = threadIdx.x;
    cd* slice = a.scratch ? a.scratch + (int64_t)blockIdx.y * a.scratch_stride + (int64_t)blockIdx.x * a.scratch_elems : nullptr;
    const bool bz = a.plan.bz_m != 0;
    // chirp-z: the one work buffer, in LDS or (too long for it) in the scratch slice
    cd* buf0 = (bz && slice) ? slice : reinterpret_cast<cd*>(lds64);
    cd* buf1 = slice ? slice : buf0 + n;
    a.gray += (int64_t)a.frames[blockIdx.y] * a.gstride;
    a.spec += (int64_t)blockIdx.y * a.spec_stride;
    const double m = (double)a.gray[(int64_t)(a.foy + GRAY_YOFF) * a.gpitch + a.fox + GRAY_XOFF];
    for (int unit = blockIdx.x; 2 * unit < a.fh; unit += gridDim.x) {
        const int ra = 2 * unit, rb = ra + 1;
        const uint8_t* pa = a.gray + (int64_t)(a.foy + ra + GRAY_YOFF) * a.gpitch + a.fox + GRAY_XOFF;
        const uint8_t* pb = pa + a.gpitch;
        for (int x = tid; x < n; x += F64_THREADS)
            buf0[x] = cd{(double)pa[x] - m, rb < a.fh ? (double)pb[x] - m : 0.0};
        __syncthreads();
        const cd* z = buf0;
        if (bz) fft_bluestein<double>(buf0, n, a.plan.bz_m, a.tw, tid, F64_THREADS);
        else z = fft_run<double>(buf0, buf1, a.plan, 1, a.tw, tid, F64_THREADS);
        // rows ra and rb from the packed transform: A[k] = (Z[k] + conj Z[n-k]) / 2, B[k] = (Z[k] - conj Z[n-k]) / 2i
        for (int k = tid; k < a.fwh; k += F64_THREADS) {
            const cd zk = z[k], zc = z[k ? n - k : 0];
            a.spec[(int64_t)ra * a.fwh + k] = cd{0.5 * (zk.x + zc.x), 0.5 * (zk.y - zc.y)};
            if (rb < a.fh) a.spec[(int64_t)rb * a.fwh + k] = cd{0.5 * (zk.y + zc.y), -0.5 * (zk.x - zc.x)};
        }
        __syncthreads();  // the next signal overwrites what other threads have just read
    }
}

struct Cols64Args {
    int64_t spec_stride, scratch_stride, part_stride, gray0_stride;  // from one batch slot (frame, for gray0) to the next
    int frames[FFT64_BATCH];
    const cd* spec;
    int fh, fw, fwh;
    FftPlan plan;
    const cd* tw;
    cd* scratch;
    int scratch_elems;
    double2* part;  // [fwh] (sum, sum of squares) of ln(|F|+1), weighted twice where the column has a mirror image
    const uint8_t* gray0;  // first pixel of the transformed region: its level times fh*fw goes back into bin (0,0)
};

__global__ __launch_bounds__(F64_THREADS) void fft64_cols_kernel(Cols64Args a) {
    extern __shared__ double2 lds64[];
    __shared__ double red[2][F64_THREADS / 64];
    const int n = a.fh, tid = threadIdx.x;
    cd* slice = a.scratch ? a.scratch + (int64_t)blockIdx.y * a.scratch_stride + (int64_t)blockIdx.x * a.scratch_elems : nullptr;
    const bool bz = a.plan.bz_m != 0;
    cd* buf0 = (bz && slice) ? slice : reinterpret_cast<cd*>(lds64);
    cd* buf1 = slice ? slice : buf0 + n;
    a.spec += (int64_t)blockIdx.y * a.spec_stride;
    a.part += (int64_t)blockIdx.y * a.part_stride;
    a.gray0 += (int64_t)a.frames[blockIdx.y] * a.gray0_stride;
    for (int k = blockIdx.x; k < a.fwh; k += gridDim.x) {
        for (int y = tid; y < n; y += F64_THREADS) buf0[y] = a.spec[(int64_t)y * a.fwh + k];
        __syncthreads();
        const cd* f = buf0;
        if (bz) fft_bluestein<double>(buf0, n, a.plan.bz_m, a.tw, tid, F64_THREADS);
        else f = fft_run<double>(buf0, buf1, a.plan, 1, a.tw, tid, F64_THREADS);
        double s = 0.0, s2 = 0.0;
        for (int y = tid; y < n; y += F64_THREADS) {
            cd v = f[y];
            if (k == 0 && y == 0) v.x += (double)*a.gray0 * (double)a.fh * (double)a.fw;
            const double l = log(sqrt(v.x * v.x + v.y * v.y) + 1.0);
            s += l;
            s2 += l * l;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_xor(s, o, 64);
            s2 += __shfl_xor(s2, o, 64);
        }
        if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = s2; }
        __syncthreads();
        if (tid == 0) {
            const double wgt = (k == 0 || ((a.fw & 1) == 0 && k == a.fw / 2)) ? 1.0 : 2.0;
            a.part[k] = make_double2(wgt * (((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]),
                                     wgt * (((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]));
        }
        __syncthreads();  // the next column overwrites the buffers and the partial sums
    }
}

void fft64_sizes(const Geom& g, const FftPlan& plan_w, const FftPlan& plan_h, size_t* spec_bytes, size_t* scratch_bytes,
                 size_t* part_bytes) {
    *spec_bytes = (size_t)g.fh * g.fwh * sizeof(cd);
    const size_t rows = f64_scratch_elems(plan_w) * (size_t)f64_grid((g.fh + 1) / 2) * sizeof(cd);
    const size_t cols = f64_scratch_elems(plan_h) * (size_t)f64_grid(g.fwh) * sizeof(cd);
    *scratch_bytes = rows > cols ? rows : cols;
    *part_bytes = (size_t)g.fwh * sizeof(double2);
}

void launch_fft64(hipStream_t s, const uint8_t* gray_chunk, const int* frames, int n_batch, const Geom& g,
                  const FftPlan& plan_w, const FftPlan& plan_h, const void* tw_w, const void* tw_h, const Fft64Work& w) {
    const int64_t spec_stride = (int64_t)(w.spec_bytes / sizeof(cd));
    const int64_t scratch_stride = (int64_t)(w.scratch_bytes / sizeof(cd));
    Rows64Args r;
    r.gray = gray_chunk; r.gstride = (int64_t)g.grows * g.gpitch; r.gpitch = g.gpitch;
    r.spec_stride = spec_stride; r.scratch_stride = scratch_stride;
    r.foy = g.foy; r.fox = g.fox; r.fh = g.fh; r.fw = g.fw; r.fwh = g.fwh;
    r.plan = plan_w; r.tw = (const cd*)tw_w; r.spec = (cd*)w.spec;
    r.scratch_elems = (int)f64_scratch_elems(plan_w);
    r.scratch = r.scratch_elems ? (cd*)w.scratch : nullptr;
    Cols64Args c;
    c.spec_stride = spec_stride; c.scratch_stride = scratch_stride;
    c.part_stride = (int64_t)(w.part_bytes / sizeof(double2));
    c.gray0_stride = r.gstride;
    for (int i = 0; i < FFT64_BATCH; ++i) r.frames[i] = c.frames[i] = i < n_batch ? frames[i] : 0;
    hipLaunchKernelGGL(fft64_rows_kernel, dim3(f64_grid((g.fh + 1) / 2), n_batch), dim3(F64_THREADS), f64_lds_bytes(plan_w), s, r);
    c.spec = (const cd*)w.spec; c.fh = g.fh; c.fw = g.fw; c.fwh = g.fwh;
    c.plan = plan_h; c.tw = (const cd*)tw_h;
    c.scratch_elems = (int)f64_scratch_elems(plan_h);
    c.scratch = c.scratch_elems ? (cd*)w.scratch : nullptr;
    c.part = (double2*)w.part;
    c.gray0 = gray_chunk + (int64_t)(g.foy + GRAY_YOFF) * g.gpitch + g.fox + GRAY_XOFF;
    hipLaunchKernelGGL(fft64_cols_kernel, dim3(f64_grid(g.fwh), n_batch), dim3(F64_THREADS), f64_lds_bytes(plan_h), s, c);
}

hipError_t fft64_set_lds_limits() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft64_rows_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, F64_LDS_MAX);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(fft64_cols_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, F64_LDS_MAX);
}

}  // namespace stx
