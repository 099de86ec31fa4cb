// Wave-autonomous LDS FFT for the common video / crop sizes.
//
// One wave owns one transform (or, for short transforms, 64/LPT of them): it loads the
// signal into its own LDS slice, runs at most four Stockham passes whose radix-R butterflies
// (R up to 16, built at compile time from radix 2/3/4/5 kernels) live entirely in registers,
// and consumes the result — with no workgroup barrier anywhere, because a wave's LDS
// operations execute in issue order.  Each pass reads all of its inputs into registers before
// it writes anything, so a single LDS buffer per transform suffices.
//
// rows:  two image rows per complex transform, split into the two half spectra, written in a
//        column-blocked layout [kb][y][CBW] so that the column pass reads contiguous memory;
// cols:  a workgroup of CBW waves loads one column block (contiguous), each wave transforms
//        one column and reduces ln(|F|+1) to a (sum, sum of squares) pair.
#include "stx_internal.h"

namespace stx {
namespace fastfft {

// The half spectrum is written once by the row pass and read once by the column pass, and a chunk's worth of it is
// several times the last-level cache.  Measured on 1.2 GB: non-temporal 16-byte loads stream at 7.2 TB/s against 6.4;
// stores of whole 128-byte lines run at 5.9 TB/s either way, stores of half lines at 3.7 (2.0 when marked
// non-temporal: nothing puts the halves together then).
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_line(float4* p, float4 v) { *p = v; }
__device__ __forceinline__ float4 load_stream(const float4* p) {
    const f4v x = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(x.x, x.y, x.z, x.w);
}

// ---------------------------------------------------------------- compile-time twiddles
constexpr double kPi = 3.141592653589793238462643383279502884;

constexpr double ct_sin(double x) {
    double term = x, sum = x;
    for (int i = 1; i < 30; ++i) {
        term *= -x * x / (double)((2 * i) * (2 * i + 1));
        sum += term;
    }
    return sum;
}
constexpr double ct_cos(double x) {
    double term = 1.0, sum = 1.0;
    for (int i = 1; i < 30; ++i) {
        term *= -x * x / (double)((2 * i - 1) * (2 * i));
        sum += term;
    }
    return sum;
}

template <int R>
struct Tw {
    float re[R];
    float im[R];
};
// w^m = exp(-2 pi i m / R)
template <int R>
constexpr Tw<R> make_tw() {
    Tw<R> t{};
    for (int m = 0; m < R; ++m) {
        const int mm = (2 * m > R) ? m - R : m;
        const double ang = -2.0 * kPi * (double)mm / (double)R;
        t.re[m] = (float)ct_cos(ang);
        t.im[m] = (float)ct_sin(ang);
    }
    return t;
}

// Complex product in two packed instructions.  The operand-select / negate modifiers of the
// packed-f32 ops do the shuffling that a straightforward (x*x - y*y, x*y + y*x) needs extra
// moves for:  t = (a.y * -b.y, a.y * b.x);  r = (a.x * b.x + t.x, a.x * b.y + t.y).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    const v2f va = __builtin_bit_cast(v2f, a), vb = __builtin_bit_cast(v2f, b);
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(va), "v"(vb));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(va), "v"(vb), "v"(t));
    return __builtin_bit_cast(float2, r);
}

// Same with a compile-time constant factor: the constant travels in a scalar register pair (one
// scalar operand per instruction is allowed), not in a vector pair that has to be filled with moves.
__device__ __forceinline__ float2 cmulf_const(float2 a, float kre, float kim) {
    const v2f va = __builtin_bit_cast(v2f, a);
    const v2f vb = {kre, kim};
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(va), "s"(vb));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(va), "s"(vb), "v"(t));
    return __builtin_bit_cast(float2, r);
}

// v * exp(-2 pi i m / R) for compile-time-foldable m
template <int R>
__device__ __forceinline__ float2 mul_tw(float2 v, int m, const Tw<R>& tw) {
    if (m == 0) return v;
    if (4 * m == R || 4 * m == 3 * R) {
        // times -i or +i: swap the halves, flip one sign, in a single packed add with zero
        const v2f a = __builtin_bit_cast(v2f, v);
        v2f r;
        if (4 * m == R) asm("v_pk_add_f32 %0, %1, 0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a));
        else asm("v_pk_add_f32 %0, %1, 0 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(r) : "v"(a));
        return __builtin_bit_cast(float2, r);
    }
    if (2 * m == R) return make_float2(-v.x, -v.y);
    return cmulf_const(v, tw.re[m], tw.im[m]);
}

// ---------------------------------------------------------------- register DFTs
template <int R>
struct Split {
    static constexpr int A = (R % 4 == 0 && R > 4) ? 4 : (R % 3 == 0 && R > 3) ? 3 : (R % 2 == 0 && R > 2) ? 2 : (R % 5 == 0 && R > 5) ? 5 : 1;
};

template <int R>
struct Dft {
    // R = A*B Cooley-Tukey in registers: B-point transforms over stride-A subsequences,
    // constant twiddles, then A-point transforms; output in natural order.
    static __device__ __forceinline__ void run(float2 (&v)[R]) {
        constexpr int A = Split<R>::A, B = R / A;
        static_assert(A > 1 && B > 1, "unsupported radix");
        constexpr Tw<R> tw = make_tw<R>();
        float2 y[R];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            float2 t[B];
#pragma unroll
            for (int b = 0; b < B; ++b) t[b] = v[a + A * b];
            Dft<B>::run(t);
#pragma unroll
            for (int kb = 0; kb < B; ++kb) y[a + A * kb] = mul_tw<R>(t[kb], a * kb, tw);
        }
#pragma unroll
        for (int kb = 0; kb < B; ++kb) {
            float2 t[A];
#pragma unroll
            for (int a = 0; a < A; ++a) t[a] = y[a + A * kb];
            Dft<A>::run(t);
#pragma unroll
            for (int ka = 0; ka < A; ++ka) v[kb + B * ka] = t[ka];
        }
    }
};
template <>
struct Dft<1> {
    static __device__ __forceinline__ void run(float2 (&)[1]) {}
};
// Radix kernels on packed pairs.  A multiplication by -i or +i is a swap of the halves with one sign
// flip; the packed-f32 operand selectors do that inside the add that consumes it, so no rotation ever
// costs an instruction of its own.
__device__ __forceinline__ v2f V(float2 a) { return __builtin_bit_cast(v2f, a); }
__device__ __forceinline__ float2 F(v2f a) { return __builtin_bit_cast(float2, a); }
// a + (-i) d = (a.x + d.y, a.y - d.x)
__device__ __forceinline__ v2f add_mi(v2f a, v2f d) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
// a + (+i) d = (a.x - d.y, a.y + d.x)
__device__ __forceinline__ v2f add_pi(v2f a, v2f d) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));
    return r;
}
// a + c (-i) d  and  a - c (-i) d, c a compile-time real constant held in a scalar pair
__device__ __forceinline__ v2f fma_mi(v2f a, float c, v2f d) {
    const v2f k = {c, -c};
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(d), "s"(k), "v"(a));
    return r;
}
__device__ __forceinline__ v2f fma_pi(v2f a, float c, v2f d) {
    const v2f k = {-c, c};
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(d), "s"(k), "v"(a));
    return r;
}

template <>
struct Dft<2> {
    static __device__ __forceinline__ void run(float2 (&v)[2]) {
        const v2f a = V(v[0]), b = V(v[1]);
        v[0] = F(a + b);
        v[1] = F(a - b);
    }
};
template <>
struct Dft<3> {
    static __device__ __forceinline__ void run(float2 (&v)[3]) {
        const float c = 0.86602540378443864676f;
        const v2f v0 = V(v[0]), s = V(v[1]) + V(v[2]), d = V(v[1]) - V(v[2]);
        const v2f h = v0 - 0.5f * s;
        v[0] = F(v0 + s);
        v[1] = F(fma_mi(h, c, d));  // h + c (-i) d
        v[2] = F(fma_pi(h, c, d));  // h - c (-i) d
    }
};
template <>
struct Dft<4> {
    static __device__ __forceinline__ void run(float2 (&v)[4]) {
        const v2f a0 = V(v[0]) + V(v[2]), a1 = V(v[0]) - V(v[2]);
        const v2f a2 = V(v[1]) + V(v[3]), d = V(v[1]) - V(v[3]);
        v[0] = F(a0 + a2);
        v[1] = F(add_mi(a1, d));
        v[2] = F(a0 - a2);
        v[3] = F(add_pi(a1, d));
    }
};
template <>
struct Dft<5> {
    static __device__ __forceinline__ void run(float2 (&v)[5]) {
        const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        const v2f v0 = V(v[0]);
        const v2f a1 = V(v[1]) + V(v[4]), a2 = V(v[2]) + V(v[3]);
        const v2f b1 = V(v[1]) - V(v[4]), b2 = V(v[2]) - V(v[3]);
        const v2f p1 = v0 + c1 * a1 + c2 * a2;
        const v2f p2 = v0 + c2 * a1 + c1 * a2;
        const v2f u1 = s1 * b1 + s2 * b2;
        const v2f u2 = s2 * b1 - s1 * b2;
        v[0] = F(v0 + a1 + a2);
        v[1] = F(add_mi(p1, u1));
        v[2] = F(add_mi(p2, u2));
        v[3] = F(add_pi(p2, u2));
        v[4] = F(add_pi(p1, u1));
    }
};

// ---------------------------------------------------------------- plans
// N = R0*R1*R2*R3 (unused radices are 1); LPT = lanes per transform (16, 32 or 64)
// LDS bank conflicts: reads of a pass are 32 consecutive elements per half wave and never
// conflict; the writes of the FIRST pass are strided by R0 elements.  An odd R0 makes that
// stride hit all banks, so every size with an odd factor starts with one.  Powers of two use an
// XOR swizzle of the low four index bits by the next four instead (it permutes elements inside
// aligned 16-element groups, which keeps the aligned reads conflict-free).
template <int N_, int R0_, int R1_, int R2_, int R3_, int LPT_>
struct Plan {
    static constexpr int N = N_, R0 = R0_, R1 = R1_, R2 = R2_, R3 = R3_, LPT = LPT_;
    static_assert(R0_ * R1_ * R2_ * R3_ == N_, "radices must multiply to N");
    static constexpr int TPW = 64 / LPT_;          // transforms per wave
    static constexpr int S = N_ | 1;               // LDS elements per transform (odd: see cols_kernel)
    static constexpr bool SWZ = (N_ & (N_ - 1)) == 0;
    // Columns per workgroup of the column kernel: as many as fit half of the CU's LDS (two workgroups per CU, so
    // that one loads its block while the other computes), a power of two, at most 16, at least one wave's worth.
    static constexpr int cbw_fit() {
        int c = 16;
        while (c > 1 && (long)c * S * 8 > (159 * 1024) / 2) c >>= 1;
        return c < TPW ? TPW : c;
    }
    static constexpr int CBW = cbw_fit();
    static constexpr int COLS_THREADS = 64 * CBW / TPW;
    static_assert(SWZ || (R0_ & 1), "sizes with an odd factor must start with an odd radix");
};

// Lanes of a wave exchange data through LDS with no barrier instruction: the hardware executes
// a wave's LDS operations in issue order.  The COMPILER, however, reasons per thread and may
// move a lane's load above its own earlier store when it can prove the two addresses differ for
// that lane — which breaks the exchange with other lanes.  This fence (wavefront scope: no
// instruction is emitted) pins the order at every phase boundary.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool SWZ>
__device__ __forceinline__ int pad(int i) {
    return SWZ ? (i ^ ((i >> 4) & 15)) : i;
}

// One Stockham pass of radix R; NS = product of the radices already applied.
// Where a pass puts its outputs: back into the LDS buffer, or (last pass of the column
// transform) straight into a consumer, which saves one LDS write and one LDS read per element.
// A sink that wants the R outputs of a butterfly together (indices o, o + ns, ...) declares it here.
template <class Sink>
struct TakesButterflies { static constexpr bool value = false; };

template <bool SWZ>
struct StoreSink {
    float2* buf;
    __device__ __forceinline__ void operator()(int idx, float2 v) const { buf[pad<SWZ>(idx)] = v; }
};

// Where a pass takes its one table twiddle per butterfly from: the table in global memory, or registers
// that a persistent kernel filled once (the index depends on the lane only, not on the transform; and a
// global load inside the loop would make the wave wait for its own earlier stores to drain, because
// loads and stores retire through one in-order counter).
struct TwGlobal {
    const float2* p;
    __device__ __forceinline__ float2 get(int, int idx) const { return p[idx]; }
};
template <int NSLOT>
struct TwRegs {
    float2 v[NSLOT];
    __device__ __forceinline__ float2 get(int slot, int) const { return v[slot]; }
};

template <int N, int R, int NS, int LPT, bool SWZ, class Sink, class TwS>
__device__ __forceinline__ void pass(float2* __restrict__ buf, const TwS& twg, int t, Sink& sink) {
    if constexpr (R > 1) {
        constexpr int M = N / R;
        constexpr int ITER = (M + LPT - 1) / LPT;
        float2 v[ITER][R];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int j = t + it * LPT;
            if (ITER * LPT == M || j < M) {
#pragma unroll
                for (int r = 0; r < R; ++r) v[it][r] = buf[pad<SWZ>(j + r * M)];
            }
        }
        wave_sync();  // every lane has read its inputs before any lane overwrites them
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int j = t + it * LPT;
            if (ITER * LPT == M || j < M) {
                const int k = (NS > 1) ? (j % NS) : 0;
                if constexpr (NS > 1) {
                    // w^(r k), w = exp(-2 pi i / (NS R)): one table read, the other powers by products
                    float2 w[R];
                    w[1] = twg.get(it, k * (N / (NS * R)));
#pragma unroll
                    for (int r = 2; r < R; ++r) w[r] = cmulf(w[r / 2], w[r - r / 2]);
#pragma unroll
                    for (int r = 1; r < R; ++r) v[it][r] = cmulf(v[it][r], w[r]);
                }
                Dft<R>::run(v[it]);
                const int o = (j - k) * R + k;
                if constexpr (TakesButterflies<Sink>::value) {
                    sink.template butterfly<R>(o, NS, v[it]);
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) sink(o + r * NS, v[it][r]);
                }
            }
        }
        wave_sync();  // the next phase reads what other lanes wrote
    }
}

template <int N, int R, int NS, int LPT, bool SWZ, class Sink>
__device__ __forceinline__ void pass(float2* __restrict__ buf, const float2* __restrict__ twg, int t, Sink& sink) {
    const TwGlobal g{twg};
    pass<N, R, NS, LPT, SWZ>(buf, g, t, sink);
}
// the table twiddles of a pass, loaded once
template <int N, int R, int NS, int LPT = 64>
__device__ __forceinline__ TwRegs<(N / R + LPT - 1) / LPT> load_pass_twiddles(const float2* __restrict__ twg, int t) {
    constexpr int M = N / R, ITER = (M + LPT - 1) / LPT;
    TwRegs<ITER> r;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int j = t + it * LPT;
        r.v[it] = twg[(j < M ? j % NS : 0) * (N / (NS * R))];
    }
    return r;
}

// last: receives the outputs of the final pass (index, value) instead of the LDS buffer
template <class P, class Sink>
__device__ __forceinline__ void transform_to(float2* buf, const float2* twg, int t, Sink& last) {
    constexpr int NP = (P::R3 > 1) ? 4 : (P::R2 > 1) ? 3 : (P::R1 > 1) ? 2 : 1;
    StoreSink<P::SWZ> st{buf};
    wave_sync();  // the signal was stored by other lanes
    if constexpr (NP == 1) pass<P::N, P::R0, 1, P::LPT, P::SWZ>(buf, twg, t, last);
    else pass<P::N, P::R0, 1, P::LPT, P::SWZ>(buf, twg, t, st);
    if constexpr (NP == 2) pass<P::N, P::R1, P::R0, P::LPT, P::SWZ>(buf, twg, t, last);
    else pass<P::N, P::R1, P::R0, P::LPT, P::SWZ>(buf, twg, t, st);
    if constexpr (NP == 3) pass<P::N, P::R2, P::R0 * P::R1, P::LPT, P::SWZ>(buf, twg, t, last);
    else pass<P::N, P::R2, P::R0 * P::R1, P::LPT, P::SWZ>(buf, twg, t, st);
    if constexpr (NP == 4) pass<P::N, P::R3, P::R0 * P::R1 * P::R2, P::LPT, P::SWZ>(buf, twg, t, last);
    else pass<P::N, P::R3, P::R0 * P::R1 * P::R2, P::LPT, P::SWZ>(buf, twg, t, st);
}

// the passes after the first one (whose outputs are already in buf)
template <class P, class Sink>
__device__ __forceinline__ void transform_tail_to(float2* buf, const float2* twg, int t, Sink& last) {
    constexpr int NP = (P::R3 > 1) ? 4 : (P::R2 > 1) ? 3 : (P::R1 > 1) ? 2 : 1;
    static_assert(NP >= 2, "nothing after the first pass");
    StoreSink<P::SWZ> st{buf};
    if constexpr (NP == 2) pass<P::N, P::R1, P::R0, P::LPT, P::SWZ>(buf, twg, t, last);
    else pass<P::N, P::R1, P::R0, P::LPT, P::SWZ>(buf, twg, t, st);
    if constexpr (NP == 3) pass<P::N, P::R2, P::R0 * P::R1, P::LPT, P::SWZ>(buf, twg, t, last);
    else pass<P::N, P::R2, P::R0 * P::R1, P::LPT, P::SWZ>(buf, twg, t, st);
    if constexpr (NP == 4) pass<P::N, P::R3, P::R0 * P::R1 * P::R2, P::LPT, P::SWZ>(buf, twg, t, last);
    else pass<P::N, P::R3, P::R0 * P::R1 * P::R2, P::LPT, P::SWZ>(buf, twg, t, st);
}

template <class P>
__device__ __forceinline__ void transform(float2* buf, const float2* twg, int t) {
    StoreSink<P::SWZ> st{buf};
    transform_to<P>(buf, twg, t, st);
}

// ---------------------------------------------------------------- row transform: fused first and last pass
// First pass straight from global memory.  Butterfly j of a radix-R0 first pass reads x[j + r*M0],
// M0 = N/R0.  When M0 is 64, 128 or 256 a lane that loads B = M0/64 adjacent bytes at byte offset
// B*(t + 64 r), r = 0..R0-1, holds every input of the B butterflies j = B*t .. B*t+B-1, so the signal
// never has to be staged in LDS: R0 narrow loads per image row and lane, each a contiguous
// 64*B-byte segment across the wave.
template <class P>
struct Direct {
    static constexpr int M0 = P::N / P::R0;
    static constexpr int B = M0 / 64;
    static constexpr bool ok = P::LPT == 64 && M0 % 64 == 0 && (B == 1 || B == 2 || B == 4);
};

template <int B>
__device__ __forceinline__ uint32_t load_bytes(const uint8_t* p, int idx) {
    if constexpr (B == 1) return p[idx];
    else if constexpr (B == 2) return reinterpret_cast<const uint16_t*>(p)[idx];
    else return reinterpret_cast<const uint32_t*>(p)[idx];
}

// ra / rb: the loads of the two image rows packed into one complex signal (row a real, row b imaginary)
template <class P>
__device__ __forceinline__ void pass0_direct(float2* __restrict__ buf, const uint32_t (&ra)[P::R0],
                                             const uint32_t (&rb)[P::R0], float m, int t) {
    constexpr int R = P::R0, B = Direct<P>::B;
    float2 out[B * R];
#pragma unroll
    for (int e = 0; e < B; ++e) {
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
            v[r] = make_float2((float)((ra[r] >> (8 * e)) & 255u) - m, (float)((rb[r] >> (8 * e)) & 255u) - m);
        Dft<R>::run(v);
#pragma unroll
        for (int r = 0; r < R; ++r) out[e * R + r] = v[r];
    }
    // butterfly j writes its outputs to j*R0 .. j*R0+R0-1: B*R0 consecutive elements per lane
    const int base = B * t * R;
    if constexpr (!P::SWZ && (B * R) % 2 == 0) {
        float4* d4 = reinterpret_cast<float4*>(buf + base);
#pragma unroll
        for (int q = 0; q < B * R / 2; ++q) d4[q] = make_float4(out[2 * q].x, out[2 * q].y, out[2 * q + 1].x, out[2 * q + 1].y);
    } else {
#pragma unroll
        for (int q = 0; q < B * R; ++q) buf[pad<P::SWZ>(base + q)] = out[q];
    }
    wave_sync();
}

// Last pass fused with the split of the packed transform.  Z[k] and Z[N-k] come out of butterflies j and
// M-j (M = N/R) of the last pass, so a lane that runs both holds every conjugate pair it needs:
//   A[k] = (Z[k] + conj Z[N-k]) / 2,   B[k] = (Z[k] - conj Z[N-k]) / 2i
// go from registers to global memory, and the pass neither writes LDS nor is re-read by a split loop.
// Work item i = 1..(M-1)/2 runs butterflies i and M-i.  Butterflies 0 and (M even) M/2 hold their own
// partners; item 0 runs those two.  emit(k, Z[k], Z[N-k]) is called once for every k in 0..N/2.
template <class P, int R>
struct PairItems {
    static constexpr int M = P::N / R;
    static constexpr bool EVEN = (M % 2) == 0;
    static constexpr int ITEMS = EVEN ? M / 2 : (M + 1) / 2;
    static constexpr int ITER = (ITEMS + 63) / 64;
};
template <class P, int R>
__device__ __forceinline__ TwRegs<2 * PairItems<P, R>::ITER> load_pair_twiddles(const float2* __restrict__ twg, int t) {
    using PI = PairItems<P, R>;
    TwRegs<2 * PI::ITER> r;
#pragma unroll
    for (int it = 0; it < PI::ITER; ++it) {
        int i = t + it * 64;
        if (i >= PI::ITEMS) i = 0;
        const int jp = i ? PI::M - i : (PI::EVEN ? PI::M / 2 : 0);
        r.v[2 * it] = twg[i];
        r.v[2 * it + 1] = twg[jp];
    }
    return r;
}

template <class P, int R, class Emit, class TwS>
__device__ __forceinline__ void last_pass_pairs(const float2* __restrict__ buf, const TwS& twg, int t, Emit& emit) {
    constexpr int N = P::N, M = N / R;
    constexpr bool EVEN = (M % 2) == 0;
    constexpr int ITEMS = EVEN ? M / 2 : (M + 1) / 2;
    constexpr int ITER = (ITEMS + 63) / 64;
    float2 va[ITER][R], vb[ITER][R];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = t + it * 64;
        if (ITER * 64 == ITEMS || i < ITEMS) {
            const int jp = i ? M - i : (EVEN ? M / 2 : 0);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                va[it][r] = buf[pad<P::SWZ>(i + r * M)];
                vb[it][r] = buf[pad<P::SWZ>(jp + r * M)];
            }
        }
    }
    wave_sync();  // every lane holds its inputs: emit may reuse the buffer
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = t + it * 64;
        if (ITER * 64 == ITEMS || i < ITEMS) {
            const int jp = i ? M - i : (EVEN ? M / 2 : 0);
            {
                float2 w[R];
                w[1] = twg.get(2 * it, i);
#pragma unroll
                for (int r = 2; r < R; ++r) w[r] = cmulf(w[r / 2], w[r - r / 2]);
#pragma unroll
                for (int r = 1; r < R; ++r) va[it][r] = cmulf(va[it][r], w[r]);
                w[1] = twg.get(2 * it + 1, jp);
#pragma unroll
                for (int r = 2; r < R; ++r) w[r] = cmulf(w[r / 2], w[r - r / 2]);
#pragma unroll
                for (int r = 1; r < R; ++r) vb[it][r] = cmulf(vb[it][r], w[r]);
            }
            Dft<R>::run(va[it]);
            Dft<R>::run(vb[it]);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // Z[k] = va[r]; Z[N-k] is output R-1-r of butterfly M-i, or output (R-r)%R of butterfly 0 itself
                const int k = i + r * M;
                const float2 zk = va[it][r];
                const float2 zc = (i == 0) ? va[it][(R - r) % R] : vb[it][R - 1 - r];
                const bool low = 2 * k <= N;
                if (i != 0 || low) emit(low ? k : N - k, low ? zk : zc, low ? zc : zk);
            }
            if (EVEN && i == 0) {
                // butterfly M/2: Z[M/2 + r M] pairs with Z[M/2 + (R-1-r) M]
#pragma unroll
                for (int r = 0; 2 * r < R; ++r) emit(M / 2 + r * M, vb[it][r], vb[it][R - 1 - r]);
            }
        }
    }
}

// ---------------------------------------------------------------- kernels
struct RowsArgs {
    const uint8_t* gray;
    int gpitch, grows;
    int foy, fox, fh, fwh;
    int cbw, nkb;  // column-block width (a power of two) and count of the spectrum layout
    int cbw_log2;
    int groups;       // groups of TPW row pairs per frame
    int total_units;  // n_frames * groups
    const float2* tw;
    float2* spec;
};

// Persistent waves: each wave walks over (frame, row pair) units with a stride of the grid size and
// issues the global loads of its NEXT unit before transforming the current one, so the HBM/L2
// latency of the gray rows hides behind the butterflies.
//   DIRECT (one transform per wave, aligned rows, N/R0 in {64,128,256}): the first pass takes its inputs
//   from the prefetch registers, the last pass splits and stores from registers (see above);
//   otherwise the signal is staged in LDS first; sub-wave transforms also split through LDS.
#ifdef STX_ROWS_WAVES  // (experiments: hold the row kernel to the registers of that many waves per SIMD)
#define STX_ROWS_ATTR __attribute__((amdgpu_waves_per_eu(STX_ROWS_WAVES, STX_ROWS_WAVES)))
#else
#define STX_ROWS_ATTR
#endif
template <class P, bool DIRECT, bool STAGED>
__global__ __launch_bounds__(64) STX_ROWS_ATTR void rows_kernel(RowsArgs a) {
    constexpr int N = P::N, LPT = P::LPT;
    constexpr bool PAIRS = (LPT == 64);  // fused last pass
    // Staged output (one transform per wave, column blocks of eight bins): rows a and a+1 of a column block are the two
    // halves of one 128-byte line.  Written as they leave the last pass -- eight bytes per lane, a row at a time --
    // every line is put together in L2 from two instructions' half lines, and the store stream runs at 3.8 TB/s; as
    // whole lines from one instruction it runs at 5.9.  The last pass therefore parks both rows' bins in the wave's
    // buffer (its inputs are in registers by then), row a's at [0, HALF8), row b's at [STAGE_B, STAGE_B + HALF8),
    // and the wave stores them sixteen bytes per lane: lanes 8g .. 8g+3 the first row's half of block g's line,
    // lanes 8g+4 .. 8g+7 the second row's.  STAGE_B is 16 mod 32 elements: the two halves of a 16-lane group read
    // different banks.
    constexpr int HALF8 = (N / 2 + 1 + 7) / 8 * 8;
    constexpr int STAGE_B = HALF8 + ((16 - HALF8 % 32) + 32) % 32;
    constexpr int STAGE = STAGE_B + HALF8;
    static_assert(!STAGED || PAIRS, "the staged output is the fused last pass's");
    constexpr int LDS_ELEMS = (STAGED && STAGE > P::S) ? STAGE : P::TPW * P::S;
    __shared__ __attribute__((aligned(16))) float2 lds[LDS_ELEMS];
    constexpr bool PREFETCH = !DIRECT && (N % 4 == 0);
    constexpr int NL = DIRECT ? P::R0 : (PREFETCH ? (N / 4 + LPT - 1) / LPT : 1);  // loads per lane and row
    constexpr int NP = (P::R3 > 1) ? 4 : (P::R2 > 1) ? 3 : (P::R1 > 1) ? 2 : 1;
    static_assert(!PAIRS || NP >= 2, "a fused first or last pass needs at least two passes");
    const int lane = threadIdx.x;
    // one transform per wave: the row pair, its bounds tests and its addresses are wave-uniform
    const int sub = (P::TPW == 1) ? 0 : lane / LPT, t = (P::TPW == 1) ? lane : lane % LPT;
    float2* buf = lds + sub * P::S;
    const int cbw = a.cbw;
    const bool aligned = DIRECT || (a.fox & 3) == 0;

    uint32_t ra_reg[NL], rb_reg[NL];
    // The transform runs on gray - m, m = the frame's gray level at the origin of the transformed region (only
    // bin (0,0) depends on m; the column pass adds m*H*W back): a constant frame becomes exact zeros.  m travels
    // with the prefetched rows; rows that do not exist are set to m when the unit is consumed.
    uint32_t m_reg = 0;
    auto row_ptr = [&](int unit, int& frame, int& ra) -> const uint8_t* {
        frame = unit / a.groups;
        const int grp = unit - frame * a.groups;
        ra = 2 * (grp * P::TPW + sub);
        return a.gray + (int64_t)frame * a.grows * a.gpitch + (int64_t)(a.foy + ra + GRAY_YOFF) * a.gpitch + a.fox +
               GRAY_XOFF;
    };
    auto origin = [&](int frame) -> const uint8_t* {
        return a.gray + (int64_t)frame * a.grows * a.gpitch + (int64_t)(a.foy + GRAY_YOFF) * a.gpitch + a.fox + GRAY_XOFF;
    };
    auto fetch = [&](int unit) {
        int frame, ra;
        const uint8_t* pa = row_ptr(unit, frame, ra);
        const bool ina = ra < a.fh, inb = ra + 1 < a.fh;
        m_reg = origin(frame)[0];
        if constexpr (DIRECT) {
            constexpr int B = Direct<P>::B;
            if (ina && inb) {  // wave-uniform; every unit but the last row pair of an odd-height frame
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    ra_reg[i] = load_bytes<B>(pa, lane + 64 * i);
                    rb_reg[i] = load_bytes<B>(pa + a.gpitch, lane + 64 * i);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    ra_reg[i] = ina ? load_bytes<B>(pa, lane + 64 * i) : 0u;
                    rb_reg[i] = 0u;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int x4 = t + LPT * i;
                const bool inx = (NL * LPT == N / 4) || x4 < N / 4;
                ra_reg[i] = (ina && inx) ? reinterpret_cast<const uint32_t*>(pa)[x4] : 0u;
                rb_reg[i] = (inb && inx) ? reinterpret_cast<const uint32_t*>(pa + a.gpitch)[x4] : 0u;
            }
        }
    };

    // table twiddles of the passes after the first, once per wave
    constexpr int RL = (NP == 2) ? P::R1 : (NP == 3) ? P::R2 : P::R3;  // radix of the last pass
    [[maybe_unused]] const auto tw_last = load_pair_twiddles<P, PAIRS ? RL : P::N>(a.tw, t);
    [[maybe_unused]] const auto tw_p1 = load_pass_twiddles<P::N, (PAIRS && NP >= 3) ? P::R1 : P::N, P::R0>(a.tw, t);
    [[maybe_unused]] const auto tw_p2 = load_pass_twiddles<P::N, (PAIRS && NP == 4) ? P::R2 : P::N, P::R0 * P::R1>(a.tw, t);
    // sub-wave transforms of two passes: the second pass's table twiddle (one per lane)
    constexpr bool SUB2 = !PAIRS && NP == 2;
    [[maybe_unused]] const auto tw_s1 = load_pass_twiddles<P::N, SUB2 ? P::R1 : P::N, P::R0, LPT>(a.tw, t);

    // Loads and stores retire through one in-order counter, so waiting for the prefetched rows at the
    // top of a unit would also wait for the previous unit's freshly issued stores.  The prefetch registers
    // are therefore touched BEFORE a unit's stores go out (and once before the loop): by then the loads
    // have had the middle passes to arrive, and the only older memory operations are stores of the unit
    // before, long since retired.
    auto settle_prefetch = [&]() {
        if constexpr (DIRECT || PREFETCH) {
#pragma unroll
            for (int i = 0; i < NL; ++i) asm volatile("" : "+v"(ra_reg[i]), "+v"(rb_reg[i]));
            asm volatile("" : "+v"(m_reg));
        }
    };

    int unit = blockIdx.x;
    if ((DIRECT || (PREFETCH && aligned)) && unit < a.total_units) fetch(unit);
    settle_prefetch();
    for (; unit < a.total_units; unit += gridDim.x) {
        int frame, ra;
        const uint8_t* pa = row_ptr(unit, frame, ra);
        const uint8_t* pb = pa + a.gpitch;
        const bool ina = ra < a.fh, inb = ra + 1 < a.fh;
        uint32_t mcur = m_reg;
        if (!(DIRECT || (PREFETCH && aligned))) mcur = origin(frame)[0];
        const float m = (float)mcur;
        if constexpr (DIRECT || PREFETCH) {
            if (!ina || !inb) {  // rare (wave-uniform with one transform per wave): rows past the end of the frame transform as zeros
                const uint32_t m4 = mcur * 0x01010101u;
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    if (!ina) ra_reg[i] = m4;
                    if (!inb) rb_reg[i] = m4;
                }
            }
        }
        if constexpr (DIRECT) {
            pass0_direct<P>(buf, ra_reg, rb_reg, m, t);
            if (unit + (int)gridDim.x < a.total_units) fetch(unit + gridDim.x);
        } else if (PREFETCH && aligned) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int x4 = t + LPT * i;
                if ((NL * LPT == N / 4) || x4 < N / 4) {
                    const uint32_t wa = ra_reg[i], wb = rb_reg[i];
                    float2 e[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        e[q] = make_float2((float)((wa >> (8 * q)) & 255u) - m, (float)((wb >> (8 * q)) & 255u) - m);
                    if constexpr (!P::SWZ) {
                        // 16-byte stores: half the bank conflicts of four strided 8-byte stores
                        float4* d4 = reinterpret_cast<float4*>(buf + 4 * x4);
                        d4[0] = make_float4(e[0].x, e[0].y, e[1].x, e[1].y);
                        d4[1] = make_float4(e[2].x, e[2].y, e[3].x, e[3].y);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) buf[pad<P::SWZ>(4 * x4 + q)] = e[q];
                    }
                }
            }
            if (unit + (int)gridDim.x < a.total_units) fetch(unit + gridDim.x);
        } else if ((N % 4 == 0) && aligned) {
            for (int x4 = t; x4 < N / 4; x4 += LPT) {
                const uint32_t fill = mcur * 0x01010101u;
                const uint32_t wa = ina ? reinterpret_cast<const uint32_t*>(pa)[x4] : fill;
                const uint32_t wb = inb ? reinterpret_cast<const uint32_t*>(pb)[x4] : fill;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    buf[pad<P::SWZ>(4 * x4 + q)] = make_float2((float)((wa >> (8 * q)) & 255u) - m, (float)((wb >> (8 * q)) & 255u) - m);
            }
        } else {
            for (int x = t; x < N; x += LPT)
                buf[pad<P::SWZ>(x)] =
                    make_float2(ina ? (float)pa[x] - m : 0.f, inb ? (float)pb[x] - m : 0.f);
        }
        // One half-spectrum bin of both rows; column-blocked layout [kb][row][cbw].  The row part of the
        // address is wave-uniform (folded into the base pointer); the bin part depends on the lane only,
        // so it stays a 32-bit offset that does not change from unit to unit.
        float2* base = a.spec + (int64_t)frame * a.nkb * a.fh * cbw + ((int64_t)ra << a.cbw_log2);
        auto emit = [&](int k, float2 zk, float2 zc) {
            if constexpr (STAGED) {
                buf[k] = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
                buf[STAGE_B + k] = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
            } else if (ina) {
                const uint32_t off = ((((uint32_t)k >> a.cbw_log2) * (uint32_t)a.fh) << a.cbw_log2) + ((uint32_t)k & (uint32_t)(cbw - 1));
                float2* o = base + off;
                const float2 va = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
                const float2 vb = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
                o[0] = va;
                if (inb) o[cbw] = vb;
            }
        };
        if constexpr (PAIRS) {
            StoreSink<P::SWZ> st{buf};
            if constexpr (!DIRECT) {
                wave_sync();  // the signal was stored by other lanes
                pass<P::N, P::R0, 1, P::LPT, P::SWZ>(buf, a.tw, t, st);
            }
            if constexpr (NP == 2) {
                if constexpr (!STAGED) settle_prefetch();
                last_pass_pairs<P, P::R1>(buf, tw_last, t, emit);
            } else if constexpr (NP == 3) {
                pass<P::N, P::R1, P::R0, P::LPT, P::SWZ>(buf, tw_p1, t, st);
                if constexpr (!STAGED) settle_prefetch();
                last_pass_pairs<P, P::R2>(buf, tw_last, t, emit);
            } else {
                pass<P::N, P::R1, P::R0, P::LPT, P::SWZ>(buf, tw_p1, t, st);
                pass<P::N, P::R2, P::R0 * P::R1, P::LPT, P::SWZ>(buf, tw_p2, t, st);
                if constexpr (!STAGED) settle_prefetch();
                last_pass_pairs<P, P::R3>(buf, tw_last, t, emit);
            }
            if constexpr (STAGED) {
                settle_prefetch();  // (here the stores go out after the last pass: the prefetch has had that long as well)
                wave_sync();  // the bins were parked by other lanes
                constexpr int NBLK = HALF8 / 8;
                const int g = lane >> 3, l = lane & 7;
                const float2* sp = buf + ((l & 4) ? STAGE_B : 0) + g * 8 + 2 * (l & 3);
                float4* dp = reinterpret_cast<float4*>(base + (uint32_t)g * (uint32_t)a.fh * 8u) + l;
                const bool lane_ok = ina && (l < 4 || inb);
#pragma unroll
                for (int q = 0; q < (NBLK + 7) / 8; ++q) {
                    if ((q * 8 + 7 < NBLK || q * 8 + g < NBLK) && lane_ok)
                        store_line(dp + (int64_t)q * a.fh * 32, *reinterpret_cast<const float4*>(sp + q * 64));
                }
            }
        } else {
            if constexpr (SUB2) {
                StoreSink<P::SWZ> st{buf};
                wave_sync();  // the signal was stored by other lanes
                pass<P::N, P::R0, 1, P::LPT, P::SWZ>(buf, a.tw, t, st);
                pass<P::N, P::R1, P::R0, P::LPT, P::SWZ>(buf, tw_s1, t, st);
            } else {
                transform<P>(buf, a.tw, t);
            }
            settle_prefetch();
            // split the packed transform: A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / 2i
            if (cbw == 16) {
                // Column blocks of sixteen bins: a row of a block is one 128-byte line, and the wave's 2*TPW image rows
                // are neighbours in every block.  Eight lanes write one row's line, sixteen bytes (two bins) each, so
                // one instruction per block covers all the wave's rows -- 1 KB in one piece at TPW = 4 -- instead of
                // two instructions of alternate lines; every lane splits the two bins of ITS row (either the a or
                // the b row of a pair) from the pair's transform in LDS.
                const int row = lane >> 3, col2 = 2 * (lane & 7);
                const bool second = (row & 1) != 0;
                const float2* zb = lds + (row >> 1) * P::S;
                const int ra0 = ra - 2 * sub;  // the first image row of the wave's group (wave-uniform)
                const bool row_ok = row < 2 * P::TPW && ra0 + row < a.fh;
                float4* gp = reinterpret_cast<float4*>(a.spec + (int64_t)frame * a.nkb * a.fh * 16 + (int64_t)(ra0 + row) * 16 + col2);
                for (int kb = 0; kb * 16 <= N / 2; ++kb) {
                    const int k0 = 16 * kb + col2;
                    if (row_ok && k0 <= N / 2) {
                        float2 o2[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int k = k0 + j;
                            o2[j] = make_float2(0.f, 0.f);
                            if (k <= N / 2) {
                                const float2 zk = zb[pad<P::SWZ>(k)], zc = zb[pad<P::SWZ>(k ? N - k : 0)];
                                o2[j] = second ? make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x))
                                               : make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
                            }
                        }
                        const f4v v = {o2[0].x, o2[0].y, o2[1].x, o2[1].y};
                        __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(gp + (int64_t)kb * a.fh * 8));
                    }
                }
            } else {
                for (int k = t; k <= N / 2; k += LPT) emit(k, buf[pad<P::SWZ>(k)], buf[pad<P::SWZ>(k ? N - k : 0)]);
            }
        }
        wave_sync();  // the next unit's signal overwrites what other lanes have just read
    }
}

struct ColsArgs {
    const float2* spec;
    int fw, fwh;
    int cbw, nkb;
    int cbw_log2;
    const float2* tw;
    FftPart* part;  // [frame][nkb*cbw]
    float small;    // bins below this magnitude also go into the t1 / t2 sums
    const uint8_t* gray0;  // gray level at the origin of frame 0's transformed region; frames gstride apart
    int64_t gstride;
};

// First column pass straight from global memory.  Butterfly j of the radix-R0 first pass reads rows
// j + r*M0 (M0 = N/R0) of one column.  A 16-byte load holds one row of two adjacent columns, so a thread
// that loads rows j + r*M0, r = 0..R0-1, of one column pair has every input of two butterflies, and
// consecutive threads (column pair fastest, then j) read consecutive 16-byte pieces: the block streams in
// as before, but goes through the first butterflies on its way into LDS instead of being parked there
// first -- one LDS write and one LDS read less per element.  With one column per wave the workgroup has
// 64*cbw threads for M0*cbw/2 such tasks: one round when M0 <= 128.
template <class P>
struct ColsDirect {
    static constexpr int M0 = P::N / P::R0;
    static constexpr bool ok = P::TPW == 1 && P::R1 > 1 && P::R0 > 1 && P::R0 <= 10 && M0 <= 128;
};

// Where the column pass's last butterflies go: ln(|F| + 1) summed and summed squared, per lane.
struct LogMagSink {
    double ls = 0.0, ls2 = 0.0;
    float t1 = 0.f, t2 = 0.f, e2 = 0.f;
    float dc_fix, small;
    bool dc_col;
    __device__ __forceinline__ void operator()(int y, float2 v) {
        float2 one[1] = {v};
        butterfly<1>(y, 0, one);
    }
    // The R bins of one butterfly: magnitudes and logarithms of all of them in one straight run of instructions (the
    // square root and the logarithm are quarter-rate and dependent; side by side they overlap), and one test for
    // the rare near-zero bin per butterfly instead of one per bin.
    template <int R>
    __device__ __forceinline__ void butterfly(int y0, int ns, float2 (&v)[R]) {
        float mag[R], lf[R];
        float lo = 3.0e38f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float p = v[r].x * v[r].x + v[r].y * v[r].y;
            e2 += p;  // the energy of gray - m: what the transform's rounding noise scales with
            if (dc_col && y0 + r * ns == 0) {
                const float x = v[r].x + dc_fix;
                p = x * x + v[r].y * v[r].y;
            }
            // hardware sqrt / log2 (1 ulp each): errors of 1e-7 per bin, far inside the 1e-4 budget of
            // the statistic; the sums stay in fp64 because the variance is a small difference
            mag[r] = __builtin_amdgcn_sqrtf(p);
            lf[r] = __builtin_amdgcn_logf(mag[r] + 1.0f) * 0.69314718055994530942f;
            lo = fminf(lo, mag[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double l = (double)lf[r];
            ls += l;
            ls2 += l * l;
        }
        if (__builtin_expect(__any(lo < small), 0)) {  // rare on camera-like content: the wave skips it as a whole
            // (an empty volatile statement: without it the compiler turns this block into selects and computes the
            // reciprocal, a quarter-rate instruction, for every bin)
            asm volatile("");
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float q = mag[r] < small ? __builtin_amdgcn_rcpf(mag[r] + 1.0f) : 0.f;
                t1 += q;
                t2 += q * lf[r];
            }
        }
    }
};
template <>
struct TakesButterflies<LogMagSink> { static constexpr bool value = true; };

// blockDim.x = 64 * cbw / TPW ; dynamic LDS = cbw * S float2
template <class P>
__global__ __launch_bounds__(P::COLS_THREADS) void cols_kernel(ColsArgs a) {
    extern __shared__ float2 dynlds[];
    constexpr int N = P::N, LPT = P::LPT, S = P::S;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int frame = blockIdx.y, kb = blockIdx.x;
    const int cbw = a.cbw;
    // the block [N][cbw] is contiguous in global memory
    const float2* src = a.spec + ((int64_t)frame * a.nkb + kb) * (int64_t)N * cbw;
    bool first_pass_done = false;
    // Plans of three passes whose first is fused with the load keep the table twiddles of the other two in registers,
    // loaded here, ahead of the block: they arrive with it, and no pass waits on global memory afterwards.
    constexpr bool REGTW = ColsDirect<P>::ok && P::R2 > 1 && P::R3 == 1;
    TwRegs<REGTW ? (P::N / P::R1 + 63) / 64 : 1> tw1;
    TwRegs<REGTW ? (P::N / P::R2 + 63) / 64 : 1> tw2;
    constexpr bool SUB2 = !ColsDirect<P>::ok && P::R1 > 1 && P::R2 == 1;
    [[maybe_unused]] const auto tw_s1 = load_pass_twiddles<P::N, SUB2 ? P::R1 : P::N, P::R0, P::LPT>(a.tw, (tid & 63) % P::LPT);
    if constexpr (REGTW) {
        if (cbw >= 2) {
            tw1 = load_pass_twiddles<P::N, P::R1, P::R0>(a.tw, tid & 63);
            tw2 = load_pass_twiddles<P::N, P::R2, P::R0 * P::R1>(a.tw, tid & 63);
        }
    }
    if constexpr (ColsDirect<P>::ok) {
        if (cbw >= 2) {
            constexpr int R0 = P::R0, M0 = ColsDirect<P>::M0;
            const float4* src4 = reinterpret_cast<const float4*>(src);
            const int half = cbw >> 1;
            const int cp = tid & (half - 1), j = tid >> (a.cbw_log2 - 1);
            if (j < M0) {
                float4 q[R0];
#pragma unroll
                for (int r = 0; r < R0; ++r) q[r] = load_stream(src4 + (j + r * M0) * half + cp);
                float2 v0[R0], v1[R0];
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    v0[r] = make_float2(q[r].x, q[r].y);
                    v1[r] = make_float2(q[r].z, q[r].w);
                }
                Dft<R0>::run(v0);
                Dft<R0>::run(v1);
                float2* b0 = dynlds + (2 * cp) * S;
                float2* b1 = b0 + S;
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    b0[pad<P::SWZ>(j * R0 + r)] = v0[r];
                    b1[pad<P::SWZ>(j * R0 + r)] = v1[r];
                }
            }
            first_pass_done = true;
        }
    }
    if (first_pass_done) {
    } else if (cbw >= 2) {
        // N*cbw/2 16-byte loads over 64*cbw/TPW threads: N*TPW/128 per thread whatever cbw is.  All of a
        // thread's loads are issued before the first LDS write, so the block arrives at memory latency once,
        // not once per load.
        const float4* src4 = reinterpret_cast<const float4*>(src);
        const int half = cbw >> 1;
        constexpr int LOADS = (N * P::TPW + 127) / 128;
        float4 q[LOADS];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int e = tid + i * nthreads;
            q[i] = (e < N * half) ? load_stream(src4 + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int e = tid + i * nthreads;
            if (e < N * half) {
                const int y = e >> (a.cbw_log2 - 1), c = 2 * (e & (half - 1));
                dynlds[c * S + pad<P::SWZ>(y)] = make_float2(q[i].x, q[i].y);
                dynlds[(c + 1) * S + pad<P::SWZ>(y)] = make_float2(q[i].z, q[i].w);
            }
        }
    } else {
        for (int y = tid; y < N; y += nthreads) dynlds[pad<P::SWZ>(y)] = src[y];
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const int sub = lane / LPT, t = lane % LPT;
    const int c = wave * P::TPW + sub;  // column within the block
    float2* buf = dynlds + c * S;
    const int k = kb * cbw + c;
    // the last pass hands its outputs straight to the log-magnitude reduction
    LogMagSink sink;
    sink.small = a.small;
    sink.dc_col = (k == 0);
    sink.dc_fix = sink.dc_col ? (float)a.gray0[(int64_t)frame * a.gstride] * (float)N * (float)a.fw : 0.f;
    if constexpr (REGTW) {
        if (first_pass_done) {
            StoreSink<P::SWZ> st{buf};
            pass<P::N, P::R1, P::R0, P::LPT, P::SWZ>(buf, tw1, t, st);
            pass<P::N, P::R2, P::R0 * P::R1, P::LPT, P::SWZ>(buf, tw2, t, sink);
        } else {
            transform_to<P>(buf, a.tw, t, sink);
        }
    } else if constexpr (ColsDirect<P>::ok) {
        if (first_pass_done) transform_tail_to<P>(buf, a.tw, t, sink);
        else transform_to<P>(buf, a.tw, t, sink);
    } else if constexpr (SUB2) {
        StoreSink<P::SWZ> st{buf};
        wave_sync();  // the signal was stored by other lanes
        pass<P::N, P::R0, 1, P::LPT, P::SWZ>(buf, a.tw, t, st);
        pass<P::N, P::R1, P::R0, P::LPT, P::SWZ>(buf, tw_s1, t, sink);
    } else {
        transform_to<P>(buf, a.tw, t, sink);
    }
    double ls = sink.ls, ls2 = sink.ls2;
    float t1 = sink.t1, t2 = sink.t2, e2 = sink.e2;
    if (k >= a.fwh) { ls = 0.0; ls2 = 0.0; t1 = 0.f; t2 = 0.f; e2 = 0.f; }  // columns past the half spectrum hold no data
    // Sums over the LPT lanes of a transform, left in its last lane: shifts by 1, 2, 4, 8 inside each row of 16 lanes,
    // then lane 15 of a row into the next row, then lane 31 into the upper half -- operands fetched from the other
    // lane inside the vector unit.  (As __shfl_xor butterflies these were 42 trips through the LDS crossbar per
    // transform, half of the kernel's LDS instructions.)
#define STX_LANE_SUMS(CTRL, ROWS)                                                                                   \
    {                                                                                                              \
        const double a_ = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(ls), CTRL, ROWS, 0xf, false),   \
                                           __builtin_amdgcn_update_dpp(0, __double2loint(ls), CTRL, ROWS, 0xf, false));  \
        const double b_ = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(ls2), CTRL, ROWS, 0xf, false),  \
                                           __builtin_amdgcn_update_dpp(0, __double2loint(ls2), CTRL, ROWS, 0xf, false)); \
        const float c_ = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t1), CTRL, ROWS, 0xf, false));     \
        const float d_ = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t2), CTRL, ROWS, 0xf, false));     \
        const float e_ = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e2), CTRL, ROWS, 0xf, false));     \
        ls += a_; ls2 += b_; t1 += c_; t2 += d_; e2 += e_;                                                         \
    }
    static_assert(LPT == 16 || LPT == 32 || LPT == 64, "a transform's lanes are whole rows of sixteen");
    STX_LANE_SUMS(0x111, 0xf)  // row_shr:1
    STX_LANE_SUMS(0x112, 0xf)  // row_shr:2
    STX_LANE_SUMS(0x114, 0xf)  // row_shr:4
    STX_LANE_SUMS(0x118, 0xf)  // row_shr:8
    if (LPT >= 32) STX_LANE_SUMS(0x142, 0xa)  // row_bcast:15
    if (LPT == 64) STX_LANE_SUMS(0x143, 0xc)  // row_bcast:31
#undef STX_LANE_SUMS
    if (t == LPT - 1) {
        // columns 0 and (even W) W/2 are their own mirror images; every other column stands for two
        const double wgt = (k >= a.fwh) ? 0.0 : ((k == 0 || ((a.fw & 1) == 0 && k == a.fw / 2)) ? 1.0 : 2.0);
        a.part[((int64_t)frame * a.nkb + kb) * cbw + c] = FftPart{wgt * ls, wgt * ls2, wgt * (double)t1, wgt * (double)t2, wgt * (double)e2};
    }
}

// ---------------------------------------------------------------- size table
#define STX_FFT_SIZES(X)       \
    X(64, 8, 8, 1, 1, 16)      \
    X(80, 5, 16, 1, 1, 16)     \
    X(96, 3, 8, 4, 1, 32)      \
    X(128, 16, 8, 1, 1, 16)    \
    X(160, 5, 8, 4, 1, 32)     \
    X(240, 15, 16, 1, 1, 16)   \
    X(256, 16, 16, 1, 1, 16)   \
    X(320, 5, 8, 8, 1, 64)     \
    X(360, 9, 8, 5, 1, 64)     \
    X(480, 15, 8, 4, 1, 64)    \
    X(512, 8, 8, 8, 1, 64)     \
    X(540, 9, 10, 6, 1, 64)    \
    X(640, 5, 16, 8, 1, 64)    \
    X(720, 9, 16, 5, 1, 64)    \
    X(768, 3, 16, 16, 1, 64)   \
    X(960, 15, 8, 8, 1, 64)    \
    X(1024, 16, 8, 8, 1, 64)   \
    X(1080, 9, 20, 6, 1, 64)   \
    X(1280, 5, 8, 8, 4, 64)   \
    X(1440, 9, 16, 10, 1, 64)  \
    X(1920, 15, 16, 8, 1, 64)  \
    X(2048, 16, 16, 8, 1, 64)  \
    X(2160, 9, 15, 16, 1, 64)  \
    X(2560, 5, 16, 16, 2, 64)  \
    X(3840, 15, 16, 16, 1, 64) \
    X(4096, 16, 16, 16, 1, 64)

template <class P>
void launch_rows_t(hipStream_t s, RowsArgs a, int n_frames) {
    const int pairs = (a.fh + 1) / 2;
    a.groups = (pairs + P::TPW - 1) / P::TPW;
    a.total_units = n_frames * a.groups;
    // A persistent grid: exactly as many single-wave workgroups as the device holds at once (registers and LDS
    // both limit that; a larger grid would run its surplus as a second, mostly empty round).
    auto launch = [&](auto kernel) {
        static int per_cu = 0;  // per instantiation
        if (per_cu == 0) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 64, 0) != hipSuccess || n < 1) n = 8;
            per_cu = n > 16 ? 16 : n;
        }
        const int grid = a.total_units < 256 * per_cu ? a.total_units : 256 * per_cu;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, s, a);
    };
    // whole-line stores: one transform per wave and column blocks of eight bins (two rows of a block = 128 bytes)
    const bool staged = P::LPT == 64 && a.cbw == 8;
    if constexpr (Direct<P>::ok) {
        // narrow loads need B-byte aligned rows: gray rows start 4-byte aligned, the crop origin decides
        if ((a.fox & 3) == 0 && (a.gpitch & 3) == 0) {
            if constexpr (P::LPT == 64) {
                if (staged) { launch(rows_kernel<P, true, true>); return; }
            }
            launch(rows_kernel<P, true, false>);
            return;
        }
    }
    if constexpr (P::LPT == 64) {
        if (staged) { launch(rows_kernel<P, false, true>); return; }
    }
    launch(rows_kernel<P, false, false>);
}

template <class P>
void launch_cols_t(hipStream_t s, const ColsArgs& a, int n_frames) {
    const int waves = (a.cbw + P::TPW - 1) / P::TPW;
    dim3 grid(a.nkb, n_frames);
    hipLaunchKernelGGL(cols_kernel<P>, grid, dim3(64 * waves), (size_t)a.cbw * P::S * sizeof(float2), s, a);
}

template <class P>
hipError_t set_cols_lds() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(cols_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               159 * 1024);
}

}  // namespace fastfft

using namespace fastfft;

bool fft_fast_available(int n) {
    switch (n) {
#define X(N, A, B, C, D, L) case N:
        STX_FFT_SIZES(X)
#undef X
        return true;
        default:
            return false;
    }
}

// columns per workgroup of the fast column kernel (0 if the size has no fast kernel)
int fft_fast_cbw(int n) {
    switch (n) {
#define X(N, A, B, C, D, L) \
    case N:                 \
        return Plan<N, A, B, C, D, L>::CBW;
        STX_FFT_SIZES(X)
#undef X
        default:
            return 0;
    }
}

hipError_t fft_fast_init() {
    hipError_t e = hipSuccess;
#define X(N, A, B, C, D, L)                          \
    e = set_cols_lds<Plan<N, A, B, C, D, L>>();      \
    if (e != hipSuccess) return e;
    STX_FFT_SIZES(X)
#undef X
    return e;
}

bool launch_fft_rows_fast(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, const FftLayout& lay,
                          const float2* tw, float2* spec) {
    RowsArgs a;
    a.gray = gray; a.gpitch = g.gpitch; a.grows = g.grows;
    a.foy = g.foy; a.fox = g.fox; a.fh = g.fh; a.fwh = g.fwh;
    a.cbw = lay.cbw; a.nkb = lay.nkb; a.cbw_log2 = __builtin_ctz(lay.cbw); a.tw = tw; a.spec = spec;
    switch (g.fw) {
#define X(N, A, B, C, D, L)                                    \
    case N:                                                    \
        launch_rows_t<Plan<N, A, B, C, D, L>>(s, a, n_frames); \
        return true;
        STX_FFT_SIZES(X)
#undef X
        default:
            return false;
    }
}

bool launch_fft_cols_fast(hipStream_t s, const uint8_t* gray0, int64_t gstride, const float2* spec, int n_frames,
                          const Geom& g, const FftLayout& lay, const float2* tw, FftPart* part, float small) {
    ColsArgs a;
    a.gray0 = gray0; a.gstride = gstride;
    a.spec = spec; a.fw = g.fw; a.fwh = g.fwh; a.cbw = lay.cbw; a.nkb = lay.nkb; a.cbw_log2 = __builtin_ctz(lay.cbw);
    a.tw = tw; a.part = part; a.small = small;
    switch (g.fh) {
#define X(N, A, B, C, D, L)                                    \
    case N:                                                    \
        launch_cols_t<Plan<N, A, B, C, D, L>>(s, a, n_frames); \
        return true;
        STX_FFT_SIZES(X)
#undef X
        default:
            return false;
    }
}

}  // namespace stx
