// Generic LDS Stockham passes shared by the fp32 kernels (stx_fft.hip) and the fp64 kernels (stx_fft64.hip):
// radix 4/2/3/5 butterflies, any other prime factor by a direct small DFT.
#pragma once

#include "stx_internal.h"

namespace stx {

template <class T>
struct alignas(2 * sizeof(T)) Cpx {
    T x, y;
};
template <class T>
__device__ __forceinline__ Cpx<T> mk(T x, T y) { return Cpx<T>{x, y}; }
template <class T>
__device__ __forceinline__ Cpx<T> cmul(Cpx<T> a, Cpx<T> b) { return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <class T>
__device__ __forceinline__ Cpx<T> cadd(Cpx<T> a, Cpx<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <class T>
__device__ __forceinline__ Cpx<T> csub(Cpx<T> a, Cpx<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
// multiply by -i
template <class T>
__device__ __forceinline__ Cpx<T> cmul_mi(Cpx<T> a) { return mk<T>(a.y, -a.x); }

// One Stockham pass of radix R over nb transforms of length n.
// in/out: [nb][n] (LDS or global memory); ns = product of the radices already applied; tw[j] = exp(-2 pi i j / n).
// NC / RC / NSC: the same three numbers as compile-time constants (0 = take the run-time argument): with them the
// index arithmetic of a butterfly -- two divisions and a remainder by run-time numbers otherwise, as many instructions
// as the butterfly itself -- folds into shifts and multiplications.
template <class T, int NC = 0, int RC = 0, int NSC = 0>
__device__ __forceinline__ void fft_pass(const Cpx<T>* __restrict__ in, Cpx<T>* __restrict__ out, int n_rt, int R_rt, int ns_rt,
                                         int nb, const Cpx<T>* __restrict__ tw, int tid, int nthreads) {
    typedef Cpx<T> C;
    const int n = NC ? NC : n_rt, R = RC ? RC : R_rt, ns = NSC ? NSC : ns_rt;
    const int m = n / R;            // butterflies per transform
    const int twstep = n / (ns * R);
    if (R != 2 && R != 3 && R != 4 && R != 5) {
        // direct DFT of size R on the twiddled inputs (rare sizes only): one output per thread and step, so
        // that a prime length (a single pass with one butterfly) still spreads over the workgroup
        for (int b = 0; b < nb; ++b) {
            const C* ib = in + b * n;
            C* ob = out + b * n;
            for (int idx = tid; idx < m * R; idx += nthreads) {
                const int q = idx / m, j = idx - q * m;
                const int k = j % ns;
                const int o = (j - k) * R + k;
                const int t1 = k * twstep;
                C accv = mk<T>(0, 0);
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
    // the butterflies of all nb transforms handed out together: a pass of m butterflies per transform leaves the
    // workgroup's last round of a transform nearly idle when m is a little more than a multiple of its size
    {
        for (int idx = tid; idx < nb * m; idx += nthreads) {
            const int b = nb == 1 ? 0 : idx / m, j = idx - b * m;
            const C* ib = in + b * n;
            C* ob = out + b * n;
            const int k = j % ns;
            const int o = (j - k) * R + k;
            const int t1 = k * twstep;
            if (R == 4) {
                C v0 = ib[j], v1 = ib[j + m], v2 = ib[j + 2 * m], v3 = ib[j + 3 * m];
                if (k) {
                    v1 = cmul(v1, tw[t1]);
                    v2 = cmul(v2, tw[2 * t1]);
                    v3 = cmul(v3, tw[3 * t1]);
                }
                const C a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), a3 = cmul_mi(csub(v1, v3));
                ob[o] = cadd(a0, a2);
                ob[o + ns] = cadd(a1, a3);
                ob[o + 2 * ns] = csub(a0, a2);
                ob[o + 3 * ns] = csub(a1, a3);
            } else if (R == 2) {
                C v0 = ib[j], v1 = ib[j + m];
                if (k) v1 = cmul(v1, tw[t1]);
                ob[o] = cadd(v0, v1);
                ob[o + ns] = csub(v0, v1);
            } else if (R == 3) {
                C v0 = ib[j], v1 = ib[j + m], v2 = ib[j + 2 * m];
                if (k) {
                    v1 = cmul(v1, tw[t1]);
                    v2 = cmul(v2, tw[2 * t1]);
                }
                const C s = cadd(v1, v2), d = csub(v1, v2);
                const C h = mk<T>(v0.x - (T)0.5 * s.x, v0.y - (T)0.5 * s.y);
                const T c = (T)0.86602540378443864676;
                const C r = mk<T>(c * d.y, -c * d.x);
                ob[o] = cadd(v0, s);
                ob[o + ns] = cadd(h, r);
                ob[o + 2 * ns] = csub(h, r);
            } else if (R == 5) {
                C v0 = ib[j], v1 = ib[j + m], v2 = ib[j + 2 * m], v3 = ib[j + 3 * m], v4 = ib[j + 4 * m];
                if (k) {
                    v1 = cmul(v1, tw[t1]);
                    v2 = cmul(v2, tw[2 * t1]);
                    v3 = cmul(v3, tw[3 * t1]);
                    v4 = cmul(v4, tw[4 * t1]);
                }
                const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;
                const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;
                const C a1 = cadd(v1, v4), a2 = cadd(v2, v3), b1 = csub(v1, v4), b2 = csub(v2, v3);
                const C p1 = mk<T>(v0.x + c1 * a1.x + c2 * a2.x, v0.y + c1 * a1.y + c2 * a2.y);
                const C p2 = mk<T>(v0.x + c2 * a1.x + c1 * a2.x, v0.y + c2 * a1.y + c1 * a2.y);
                const C q1 = cmul_mi(mk<T>(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
                const C q2 = cmul_mi(mk<T>(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
                ob[o] = mk<T>(v0.x + a1.x + a2.x, v0.y + a1.y + a2.y);
                ob[o + ns] = cadd(p1, q1);
                ob[o + 2 * ns] = cadd(p2, q2);
                ob[o + 3 * ns] = csub(p2, q2);
                ob[o + 4 * ns] = csub(p1, q1);
            }
        }
    }
}

// Chirp-z (Bluestein) transform of one signal of any length n, in place, for lengths with a large prime factor
// (a direct DFT pass costs n times that factor).  With c[k] = exp(-i pi k^2 / n),
//     X[k] = c[k] * sum_j (x[j] c[j]) conj(c[k-j]):
// a circular convolution of length m (a power of two, >= 2n-1) with a fixed kernel, done as forward transform,
// product with the kernel's transform, inverse transform.  The forward transform is decimation in frequency
// (natural order in, bit-reversed out), the inverse decimation in time (bit-reversed in, natural out), and the
// kernel's transform is tabulated in bit-reversed order: no reordering pass, one buffer of m elements.
// work[0..n) holds x on entry and X on return (work[n..m) is scratch); LDS or global memory, one workgroup.
// tab: [n] c, then [m/2] exp(-2 pi i j / m), then [m] transform of the kernel / m, bit-reversed.
template <class T>
__device__ void fft_bluestein(Cpx<T>* work, int n, int m, const Cpx<T>* __restrict__ tab, int tid, int nthreads) {
    typedef Cpx<T> C;
    const C* chirp = tab;
    const C* twm = tab + n;
    const C* bhat = twm + m / 2;
    for (int j = tid; j < m; j += nthreads) work[j] = j < n ? cmul(work[j], chirp[j]) : mk<T>(0, 0);
    __syncthreads();
    for (int half = m >> 1, step = 1; half >= 1; half >>= 1, step <<= 1) {
        for (int b = tid; b < m / 2; b += nthreads) {
            const int i = b & (half - 1);
            const int lo = ((b - i) << 1) + i, hi = lo + half;
            const C u = work[lo], v = work[hi];
            work[lo] = cadd(u, v);
            work[hi] = cmul(csub(u, v), twm[i * step]);
        }
        __syncthreads();
    }
    for (int j = tid; j < m; j += nthreads) work[j] = cmul(work[j], bhat[j]);
    __syncthreads();
    for (int half = 1, step = m >> 1; half < m; half <<= 1, step >>= 1) {
        for (int b = tid; b < m / 2; b += nthreads) {
            const int i = b & (half - 1);
            const int lo = ((b - i) << 1) + i, hi = lo + half;
            const C w = twm[i * step];
            const C u = work[lo], v = cmul(work[hi], mk<T>(w.x, -w.y));
            work[lo] = cadd(u, v);
            work[hi] = csub(u, v);
        }
        __syncthreads();
    }
    for (int k = tid; k < n; k += nthreads) work[k] = cmul(work[k], chirp[k]);
    __syncthreads();
}

// The plans make_fft_plan gives the four common frame dimensions, as compile-time constants.
template <class T, int N, int R0, int R1, int R2, int R3, int R4, int R5>
__device__ __forceinline__ Cpx<T>* fft_run_fixed(Cpx<T>* a, Cpx<T>* b, int nb, const Cpx<T>* tw, int tid, int nthreads) {
    static_assert(R0 * R1 * R2 * R3 * R4 * R5 == N, "radices must multiply to N");
    fft_pass<T, N, R0, 1>(a, b, N, R0, 1, nb, tw, tid, nthreads);
    __syncthreads();
    fft_pass<T, N, R1, R0>(b, a, N, R1, R0, nb, tw, tid, nthreads);
    __syncthreads();
    fft_pass<T, N, R2, R0 * R1>(a, b, N, R2, R0 * R1, nb, tw, tid, nthreads);
    __syncthreads();
    fft_pass<T, N, R3, R0 * R1 * R2>(b, a, N, R3, R0 * R1 * R2, nb, tw, tid, nthreads);
    __syncthreads();
    fft_pass<T, N, R4, R0 * R1 * R2 * R3>(a, b, N, R4, R0 * R1 * R2 * R3, nb, tw, tid, nthreads);
    __syncthreads();
    if constexpr (R5 > 1) {
        fft_pass<T, N, R5, R0 * R1 * R2 * R3 * R4>(b, a, N, R5, R0 * R1 * R2 * R3 * R4, nb, tw, tid, nthreads);
        __syncthreads();
        return a;
    }
    return b;
}
// true if plan is the one fft_run_fixed<.., N, R...> runs
inline bool fft_plan_is(const FftPlan& p, int n, int r0, int r1, int r2, int r3, int r4, int r5) {
    const int r[6] = {r0, r1, r2, r3, r4, r5};
    const int np = r5 > 1 ? 6 : 5;
    if (p.n != n || p.bz_m || p.npass != np) return false;
    for (int i = 0; i < np; ++i)
        if (p.radix[i] != r[i]) return false;
    return true;
}

// runs the whole plan; returns the buffer that holds the result
template <class T>
__device__ Cpx<T>* fft_run(Cpx<T>* a, Cpx<T>* b, const FftPlan& plan, int nb, const Cpx<T>* tw, int tid, int nthreads) {
    int ns = 1;
    for (int p = 0; p < plan.npass; ++p) {
        const int R = plan.radix[p];
        fft_pass<T>(a, b, plan.n, R, ns, nb, tw, tid, nthreads);
        __syncthreads();
        Cpx<T>* t = a; a = b; b = t;
        ns *= R;
    }
    return a;
}

}  // namespace stx
