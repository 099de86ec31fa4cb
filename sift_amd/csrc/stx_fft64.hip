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
// The spectrum between the two passes is stored in blocks of F64_CB adjacent columns, [block][row][F64_CB]: the row
// pass writes 64-byte pieces, and the column pass, which takes as many columns of a block at a time as fit LDS twice
// over, reads 64-byte pieces instead of one 16-byte element per 15 KB (row-major, a column's elements are a row apart).
constexpr int F64_CB = 4;
__host__ __device__ __forceinline__ int64_t f64_spec_index(int k, int y, int fh) {
    return ((int64_t)(k / F64_CB) * fh + y) * F64_CB + (k % F64_CB);
}
// columns a workgroup of the column pass transforms together: 4, 2 or 1
static int f64_cols_nb(const FftPlan& p) {
    if (p.bz_m) return 1;
    // (two workgroups per compute unit at least: the passes are chains of LDS round trips and barriers, and what
    // hides them is other workgroups)
    int nb = F64_CB;
    while (nb > 1 && 2 * (size_t)nb * p.n * sizeof(cd) > (size_t)F64_LDS_MAX / 2) nb >>= 1;
    return nb;
}

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

// Frame dimensions whose plan runs with compile-time index arithmetic (fixed: 1 .. 4; 0 = the run-time plan)
static int f64_fixed_plan(const FftPlan& p) {
    if (fft_plan_is(p, 1920, 4, 4, 4, 2, 3, 5)) return 1;
    if (fft_plan_is(p, 1080, 4, 2, 3, 3, 3, 5)) return 2;
    if (fft_plan_is(p, 1280, 4, 4, 4, 4, 5, 1)) return 3;
    if (fft_plan_is(p, 720, 4, 4, 3, 3, 5, 1)) return 4;
    return 0;
}
__device__ __forceinline__ cd* f64_run(int fixed, cd* a, cd* b, const FftPlan& plan, int nb, const cd* tw, int tid) {
    switch (fixed) {  // (uniform over the launch)
        case 1: return fft_run_fixed<double, 1920, 4, 4, 4, 2, 3, 5>(a, b, nb, tw, tid, F64_THREADS);
        case 2: return fft_run_fixed<double, 1080, 4, 2, 3, 3, 3, 5>(a, b, nb, tw, tid, F64_THREADS);
        case 3: return fft_run_fixed<double, 1280, 4, 4, 4, 4, 5, 1>(a, b, nb, tw, tid, F64_THREADS);
        case 4: return fft_run_fixed<double, 720, 4, 4, 3, 3, 5, 1>(a, b, nb, tw, tid, F64_THREADS);
        default: return fft_run<double>(a, b, plan, nb, tw, tid, F64_THREADS);
    }
}

struct Rows64Args {
    const uint8_t* gray;  // padded planes of the chunk
    int64_t gstride;      // bytes from one frame's plane to the next
    int frames[FFT64_BATCH];  // blockIdx.y -> frame of the chunk
    int64_t spec_stride, scratch_stride;  // elements from one batch slot to the next
    int gpitch;
    int foy, fox, fh, fw, fwh;
    FftPlan plan;
    int fixed;    // f64_fixed_plan(plan)
    const cd* tw;
    cd* spec;     // [ceil(fwh / F64_CB)][fh][F64_CB]
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
        else z = f64_run(a.fixed, buf0, buf1, a.plan, 1, a.tw, tid);
        // rows ra and rb from the packed transform: A[k] = (Z[k] + conj Z[n-k]) / 2, B[k] = (Z[k] - conj Z[n-k]) / 2i
        for (int k = tid; k < a.fwh; k += F64_THREADS) {
            const cd zk = z[k], zc = z[k ? n - k : 0];
            const int64_t o = f64_spec_index(k, ra, a.fh);
            a.spec[o] = cd{0.5 * (zk.x + zc.x), 0.5 * (zk.y - zc.y)};
            if (rb < a.fh) a.spec[o + F64_CB] = cd{0.5 * (zk.y + zc.y), -0.5 * (zk.x - zc.x)};
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
    int fixed;
    const cd* tw;
    cd* scratch;
    int scratch_elems;
    int nb;         // columns per workgroup and step (a power of two, at most F64_CB)
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
    const int nb = a.nb, nb_log2 = nb == 4 ? 2 : (nb == 2 ? 1 : 0);
    if (nb > 1) buf1 = buf0 + nb * n;  // (several columns at a time only when both buffers fit LDS)
    const int per = F64_THREADS >> nb_log2;  // threads that sum one column
    for (int k0 = blockIdx.x * nb; k0 < a.fwh; k0 += gridDim.x * nb) {
        for (int idx = tid; idx < nb * n; idx += F64_THREADS) {
            const int c = idx & (nb - 1), y = idx >> nb_log2;
            buf0[c * n + y] = k0 + c < a.fwh ? a.spec[f64_spec_index(k0 + c, y, n)] : cd{0.0, 0.0};
        }
        __syncthreads();
        const cd* f = buf0;
        if (bz) fft_bluestein<double>(buf0, n, a.plan.bz_m, a.tw, tid, F64_THREADS);
        else f = f64_run(a.fixed, buf0, buf1, a.plan, nb, a.tw, tid);
        // column c of the step is summed by the threads c * per .. (c + 1) * per - 1 (whole waves), in a fixed order
        const int c = tid / per, t = tid - c * per, k = k0 + c;
        double s = 0.0, s2 = 0.0;
        for (int y = t; y < n; y += per) {
            cd v = f[c * n + y];
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
        if (tid < nb && k0 + tid < a.fwh) {
            const int kk = k0 + tid, wpc = (F64_THREADS / 64) >> nb_log2;  // waves per column
            double r0 = 0.0, r1 = 0.0;
            for (int w = 0; w < wpc; ++w) { r0 += red[0][tid * wpc + w]; r1 += red[1][tid * wpc + w]; }
            const double wgt = (kk == 0 || ((a.fw & 1) == 0 && kk == a.fw / 2)) ? 1.0 : 2.0;
            a.part[kk] = make_double2(wgt * r0, wgt * r1);
        }
        __syncthreads();  // the next columns overwrite the buffers and the partial sums
    }
}


// ---------------------------------------------------------------- compile-time plans for the common frame dimensions
// The run-time plan above takes six barrier-separated passes of radix 2..5 for a 1920- or 1080-point transform.  Here
// a transform is three passes whose radix-R butterflies (R up to 16, composed at compile time from radix 2/3/4/5
// kernels with constant twiddles) live in registers, in place in ONE LDS buffer (every thread reads the inputs of all
// its butterflies before anybody writes), with all index arithmetic folded at compile time: 30 KB of LDS per
// 1920-point transform instead of 61, five workgroups of two waves per compute unit instead of two of four.
namespace f64fast {

constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double ct_sin(double x) {
    double term = x, sum = x;
    for (int i = 1; i < 30; ++i) { term *= -x * x / (double)((2 * i) * (2 * i + 1)); sum += term; }
    return sum;
}
constexpr double ct_cos(double x) {
    double term = 1.0, sum = 1.0;
    for (int i = 1; i < 30; ++i) { term *= -x * x / (double)((2 * i - 1) * (2 * i)); sum += term; }
    return sum;
}
template <int R>
struct Tw { double re[R], im[R]; };
template <int R>
constexpr Tw<R> make_tw() {  // w^m = exp(-2 pi i m / R)
    Tw<R> t{};
    for (int m = 0; m < R; ++m) {
        const int mm = (2 * m > R) ? m - R : m;
        const double ang = -2.0 * kPi * (double)mm / (double)R;
        t.re[m] = ct_cos(ang);
        t.im[m] = ct_sin(ang);
    }
    return t;
}
__device__ __forceinline__ cd add(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd sub(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd mul(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cd mul_mi(cd a) { return cd{a.y, -a.x}; }  // times -i
template <int R>
__device__ __forceinline__ cd mul_tw(cd v, int m, const Tw<R>& tw) {  // m folds at compile time
    if (m == 0) return v;
    if (4 * m == R) return cd{v.y, -v.x};
    if (4 * m == 3 * R) return cd{-v.y, v.x};
    if (2 * m == R) return cd{-v.x, -v.y};
    return mul(v, cd{tw.re[m], tw.im[m]});
}
template <int R>
struct Split {
    static constexpr int A = (R % 4 == 0 && R > 4) ? 4 : (R % 3 == 0 && R > 3) ? 3 : (R % 2 == 0 && R > 2) ? 2 : (R % 5 == 0 && R > 5) ? 5 : 1;
};
template <int R>
struct Dft {
    // R = A*B Cooley-Tukey in registers, output in natural order
    static __device__ __forceinline__ void run(cd (&v)[R]) {
        constexpr int A = Split<R>::A, B = R / A;
        static_assert(A > 1 && B > 1, "unsupported radix");
        constexpr Tw<R> tw = make_tw<R>();
        cd y[R];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            cd t[B];
#pragma unroll
            for (int b = 0; b < B; ++b) t[b] = v[a + A * b];
            Dft<B>::run(t);
#pragma unroll
            for (int kb = 0; kb < B; ++kb) y[a + A * kb] = mul_tw<R>(t[kb], a * kb, tw);
        }
#pragma unroll
        for (int kb = 0; kb < B; ++kb) {
            cd t[A];
#pragma unroll
            for (int a = 0; a < A; ++a) t[a] = y[a + A * kb];
            Dft<A>::run(t);
#pragma unroll
            for (int ka = 0; ka < A; ++ka) v[kb + B * ka] = t[ka];
        }
    }
};
template <>
struct Dft<2> {
    static __device__ __forceinline__ void run(cd (&v)[2]) {
        const cd a = v[0], b = v[1];
        v[0] = add(a, b);
        v[1] = sub(a, b);
    }
};
template <>
struct Dft<3> {
    static __device__ __forceinline__ void run(cd (&v)[3]) {
        const double c = 0.86602540378443864676;
        const cd s = add(v[1], v[2]), d = sub(v[1], v[2]);
        const cd h = cd{v[0].x - 0.5 * s.x, v[0].y - 0.5 * s.y};
        const cd r = cd{c * d.y, -c * d.x};
        v[0] = add(v[0], s);
        v[1] = add(h, r);
        v[2] = sub(h, r);
    }
};
template <>
struct Dft<4> {
    static __device__ __forceinline__ void run(cd (&v)[4]) {
        const cd a0 = add(v[0], v[2]), a1 = sub(v[0], v[2]), a2 = add(v[1], v[3]), a3 = mul_mi(sub(v[1], v[3]));
        v[0] = add(a0, a2);
        v[1] = add(a1, a3);
        v[2] = sub(a0, a2);
        v[3] = sub(a1, a3);
    }
};
template <>
struct Dft<5> {
    static __device__ __forceinline__ void run(cd (&v)[5]) {
        const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
        const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
        const cd v0 = v[0];
        const cd a1 = add(v[1], v[4]), a2 = add(v[2], v[3]), b1 = sub(v[1], v[4]), b2 = sub(v[2], v[3]);
        const cd p1 = cd{v0.x + c1 * a1.x + c2 * a2.x, v0.y + c1 * a1.y + c2 * a2.y};
        const cd p2 = cd{v0.x + c2 * a1.x + c1 * a2.x, v0.y + c2 * a1.y + c1 * a2.y};
        const cd q1 = mul_mi(cd{s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y});
        const cd q2 = mul_mi(cd{s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y});
        v[0] = cd{v0.x + a1.x + a2.x, v0.y + a1.y + a2.y};
        v[1] = add(p1, q1);
        v[2] = add(p2, q2);
        v[3] = sub(p2, q2);
        v[4] = sub(p1, q1);
    }
};

constexpr int T = 128;  // threads per transform (one workgroup)
// N = R0 R1 R2.  R0 is odd wherever N has an odd factor: the first pass writes R0 elements apart, and an even
// distance would put a wave's stores on a few LDS banks.
template <int N_, int R0_, int R1_, int R2_>
struct Plan {
    static constexpr int N = N_, R0 = R0_, R1 = R1_, R2 = R2_;
    static_assert(R0_ * R1_ * R2_ == N_, "radices must multiply to N");
};

// One Stockham pass in place: NS = product of the radices already applied; tw[j] = exp(-2 pi i j / N).
template <int N, int R, int NS>
__device__ __forceinline__ void pass(cd* __restrict__ buf, const cd* __restrict__ tw, int tid) {
    constexpr int M = N / R, ITER = (M + T - 1) / T;
    cd v[ITER][R];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int j = tid + it * T;
        if (ITER * T == M || j < M) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[it][r] = buf[j + r * M];
        }
    }
    __syncthreads();  // every thread has read its inputs before anybody overwrites them
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int j = tid + it * T;
        if (ITER * T == M || j < M) {
            const int k = (NS > 1) ? (j % NS) : 0;
            if constexpr (NS > 1) {
                // w^(r k), w = exp(-2 pi i / (NS R)): one table read, the other powers by products
                cd w[R];
                w[1] = tw[k * (N / (NS * R))];
#pragma unroll
                for (int r = 2; r < R; ++r) w[r] = mul(w[r / 2], w[r - r / 2]);
#pragma unroll
                for (int r = 1; r < R; ++r) v[it][r] = mul(v[it][r], w[r]);
            }
            Dft<R>::run(v[it]);
            const int o = (j - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) buf[o + r * NS] = v[it][r];
        }
    }
    __syncthreads();
}
template <class P>
__device__ __forceinline__ void transform(cd* buf, const cd* tw, int tid) {
    pass<P::N, P::R0, 1>(buf, tw, tid);
    pass<P::N, P::R1, P::R0>(buf, tw, tid);
    pass<P::N, P::R2, P::R0 * P::R1>(buf, tw, tid);
}

// grid (row pairs, batch); Rows64Args as for the run-time kernel (fw == P::N)
template <class P>
__global__ __launch_bounds__(T) void rows_kernel(Rows64Args a) {
    __shared__ cd buf[P::N];
    constexpr int n = P::N;
    const int tid = threadIdx.x;
    // (locals, not updates of the argument block: a modified copy of it would live in scratch memory)
    const uint8_t* gray = a.gray + (int64_t)a.frames[blockIdx.y] * a.gstride;
    cd* spec = a.spec + (int64_t)blockIdx.y * a.spec_stride;
    const double m = (double)gray[(int64_t)(a.foy + GRAY_YOFF) * a.gpitch + a.fox + GRAY_XOFF];
    const int ra = 2 * blockIdx.x, rb = ra + 1;
    const uint8_t* pa = gray + (int64_t)(a.foy + ra + GRAY_YOFF) * a.gpitch + a.fox + GRAY_XOFF;
    const uint8_t* pb = pa + a.gpitch;
    const bool inb = rb < a.fh;
    for (int x = tid; x < n; x += T) buf[x] = cd{(double)pa[x] - m, inb ? (double)pb[x] - m : 0.0};
    __syncthreads();
    transform<P>(buf, a.tw, tid);
    // rows ra and rb from the packed transform: A[k] = (Z[k] + conj Z[n-k]) / 2, B[k] = (Z[k] - conj Z[n-k]) / 2i
    // A block row of the spectrum is four bins, 64 bytes; rows ra and rb of a block are the two halves of one 128-byte
    // line.  Eight consecutive threads write one line (four bins of row ra, the same four of row rb): whole lines from
    // one instruction stream at 5.9 TB/s, half lines from two at 3.7.
    static_assert(F64_CB == 4, "eight threads per line");
    const int items = 2 * ((a.fwh + F64_CB - 1) / F64_CB * F64_CB);
    for (int e = tid; e < items; e += T) {
        const int l = e & 7, h = l >> 2, k = (e >> 3) * F64_CB + (l & 3);
        if (k < a.fwh && (h == 0 || inb)) {
            const cd zk = buf[k], zc = buf[k ? n - k : 0];
            spec[f64_spec_index(k, ra, a.fh) + h * F64_CB] =
                h ? cd{0.5 * (zk.y + zc.y), -0.5 * (zk.x - zc.x)} : cd{0.5 * (zk.x + zc.x), 0.5 * (zk.y - zc.y)};
        }
    }
}

// grid (columns rounded up to a multiple of 32, batch); Cols64Args as for the run-time kernel (fh == P::N)
template <class P>
__global__ __launch_bounds__(T) void cols_kernel(Cols64Args a) {
    __shared__ cd buf[P::N];
    __shared__ double red[2][T / 64];
    constexpr int n = P::N;
    const int tid = threadIdx.x;
    // Workgroups go to the eight dies round-robin by their number.  The four columns of a spectrum block share their
    // 64-byte pieces: workgroups b, b+8, b+16, b+24 -- same die, launched together -- take the four columns of one
    // block, so three of them find those pieces in the die's L2.
    const int b = blockIdx.x;
    const int k = ((b >> 3) & 3) + F64_CB * ((b & 7) + 8 * (b >> 5));
    if (k >= a.fwh) return;
    const cd* spec = a.spec + (int64_t)blockIdx.y * a.spec_stride;
    double2* part = a.part + (int64_t)blockIdx.y * a.part_stride;
    const uint8_t* gray0 = a.gray0 + (int64_t)a.frames[blockIdx.y] * a.gray0_stride;
    for (int y = tid; y < n; y += T) buf[y] = spec[f64_spec_index(k, y, n)];
    __syncthreads();
    transform<P>(buf, a.tw, tid);
    double s = 0.0, s2 = 0.0;
    for (int y = tid; y < n; y += T) {
        cd v = buf[y];
        if (k == 0 && y == 0) v.x += (double)*gray0 * (double)a.fh * (double)a.fw;
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
        part[k] = make_double2(wgt * (red[0][0] + red[0][1]), wgt * (red[1][0] + red[1][1]));
    }
}

#define STX_F64_SIZES(X) X(1920, 15, 16, 8) X(1280, 5, 16, 16) X(1080, 15, 9, 8) X(720, 9, 16, 5)

static bool launch_rows(hipStream_t s, const Rows64Args& r, int pairs, int n_batch) {
    switch (r.fw) {
#define X(N, A, B, C)                                                                                        \
    case N:                                                                                                  \
        hipLaunchKernelGGL((rows_kernel<Plan<N, A, B, C>>), dim3(pairs, n_batch), dim3(T), 0, s, r);          \
        return true;
        STX_F64_SIZES(X)
#undef X
        default:
            return false;
    }
}
static bool launch_cols(hipStream_t s, const Cols64Args& c, int n_batch) {
    const int grid = (c.fwh + 31) / 32 * 32;
    switch (c.fh) {
#define X(N, A, B, C)                                                                                        \
    case N:                                                                                                  \
        hipLaunchKernelGGL((cols_kernel<Plan<N, A, B, C>>), dim3(grid, n_batch), dim3(T), 0, s, c);           \
        return true;
        STX_F64_SIZES(X)
#undef X
        default:
            return false;
    }
}

}  // namespace f64fast

void fft64_sizes(const Geom& g, const FftPlan& plan_w, const FftPlan& plan_h, size_t* spec_bytes, size_t* scratch_bytes,
                 size_t* part_bytes) {
    *spec_bytes = (size_t)g.fh * (size_t)((g.fwh + F64_CB - 1) / F64_CB * F64_CB) * sizeof(cd);
    const size_t rows = f64_scratch_elems(plan_w) * (size_t)f64_grid((g.fh + 1) / 2) * sizeof(cd);
    const size_t cols = f64_scratch_elems(plan_h) * (size_t)f64_grid((g.fwh + f64_cols_nb(plan_h) - 1) / f64_cols_nb(plan_h)) * sizeof(cd);
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
    r.plan = plan_w; r.fixed = f64_fixed_plan(plan_w); r.tw = (const cd*)tw_w; r.spec = (cd*)w.spec;
    r.scratch_elems = (int)f64_scratch_elems(plan_w);
    r.scratch = r.scratch_elems ? (cd*)w.scratch : nullptr;
    Cols64Args c;
    c.spec_stride = spec_stride; c.scratch_stride = scratch_stride;
    c.part_stride = (int64_t)(w.part_bytes / sizeof(double2));
    c.gray0_stride = r.gstride;
    for (int i = 0; i < FFT64_BATCH; ++i) r.frames[i] = c.frames[i] = i < n_batch ? frames[i] : 0;
    // (the plain table of n twiddles is what both kinds of kernel read; a chirp-z length has another table and no fast kernel)
    if (plan_w.bz_m || !f64fast::launch_rows(s, r, (g.fh + 1) / 2, n_batch))
        hipLaunchKernelGGL(fft64_rows_kernel, dim3(f64_grid((g.fh + 1) / 2), n_batch), dim3(F64_THREADS), f64_lds_bytes(plan_w), s, r);
    c.spec = (const cd*)w.spec; c.fh = g.fh; c.fw = g.fw; c.fwh = g.fwh;
    c.plan = plan_h; c.fixed = f64_fixed_plan(plan_h); c.tw = (const cd*)tw_h;
    c.scratch_elems = (int)f64_scratch_elems(plan_h);
    c.scratch = c.scratch_elems ? (cd*)w.scratch : nullptr;
    c.part = (double2*)w.part;
    c.gray0 = gray_chunk + (int64_t)(g.foy + GRAY_YOFF) * g.gpitch + g.fox + GRAY_XOFF;
    c.nb = f64_cols_nb(plan_h);
    const size_t cols_lds = c.nb > 1 ? 2 * (size_t)c.nb * plan_h.n * sizeof(cd) : f64_lds_bytes(plan_h);
    if (plan_h.bz_m || !f64fast::launch_cols(s, c, n_batch))
        hipLaunchKernelGGL(fft64_cols_kernel, dim3(f64_grid((g.fwh + c.nb - 1) / c.nb), n_batch), dim3(F64_THREADS), cols_lds, s, c);
}

hipError_t fft64_set_lds_limits() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft64_rows_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, F64_LDS_MAX);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(fft64_cols_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, F64_LDS_MAX);
}

}  // namespace stx
