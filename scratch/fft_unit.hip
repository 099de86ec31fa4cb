// unit harness: one wave transforms one length-N complex signal with the production passes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <complex>
#include "../sift_amd/csrc/stx_fft_fast.hip"
using namespace stx::fastfft;

template <class P>
__global__ __launch_bounds__(64) void unit_kernel(const float2* in, float2* out, const float2* tw) {
    __shared__ __attribute__((aligned(16))) float2 lds[P::TPW * P::S];
    const int lane = threadIdx.x, sub = lane / P::LPT, t = lane % P::LPT;
    float2* buf = lds + sub * P::S;
    for (int x = t; x < P::N; x += P::LPT) buf[pad<P::SWZ>(x)] = in[sub * P::N + x];
    transform<P>(buf, tw, t);
    for (int x = t; x < P::N; x += P::LPT) out[sub * P::N + x] = buf[pad<P::SWZ>(x)];
}

template <class P>
void run() {
    const int N = P::N, T = P::TPW;
    std::vector<float2> h(N * T), o(N * T), tw(N);
    for (int i = 0; i < N * T; ++i) h[i] = make_float2((float)((i * 7919) % 251) - 125.f, (float)((i * 104729) % 241) - 120.f);
    for (int k = 0; k < N; ++k) { double a = -2.0 * M_PI * k / N; tw[k] = make_float2((float)cos(a), (float)sin(a)); }
    float2 *di, *dout, *dt;
    hipMalloc(&di, sizeof(float2) * N * T); hipMalloc(&dout, sizeof(float2) * N * T); hipMalloc(&dt, sizeof(float2) * N);
    hipMemcpy(di, h.data(), sizeof(float2) * N * T, hipMemcpyHostToDevice);
    hipMemcpy(dt, tw.data(), sizeof(float2) * N, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(unit_kernel<P>, dim3(1), dim3(64), 0, 0, di, dout, dt);
    hipMemcpy(o.data(), dout, sizeof(float2) * N * T, hipMemcpyDeviceToHost);
    double maxerr = 0, maxmag = 0; int worst = -1;
    for (int s = 0; s < T; ++s)
        for (int k = 0; k < N; ++k) {
            std::complex<double> acc = 0;
            for (int n = 0; n < N; ++n) {
                double a = -2.0 * M_PI * (double)((long)k * n % N) / N;
                acc += std::complex<double>(h[s * N + n].x, h[s * N + n].y) * std::complex<double>(cos(a), sin(a));
            }
            double e = std::abs(acc - std::complex<double>(o[s * N + k].x, o[s * N + k].y));
            if (e > maxerr) { maxerr = e; worst = k; }
            maxmag = std::max(maxmag, std::abs(acc));
        }
    printf("N=%d radices %d %d %d %d LPT %d: max abs err %.3e (max |X| %.3e) at k=%d %s\n", N, P::R0, P::R1, P::R2, P::R3,
           P::LPT, maxerr, maxmag, worst, maxerr > 1e-3 * maxmag ? "BAD" : "");
    hipFree(di); hipFree(dout); hipFree(dt);
}

int main() {
#define X(N, A, B, C, D, L) run<Plan<N, A, B, C, D, L>>();
    STX_FFT_SIZES(X)
#undef X
    return 0;
}
