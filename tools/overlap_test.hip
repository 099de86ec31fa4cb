// Experiments on what the big kernels of the path do to each other and to the memory system, on the library's own
// kernels (links its object files):
//   * the stencil pass, the FFT row pass and the FFT column pass alone, back to back and on two streams (stream
//     priorities with STX_EXP_PRIO=1|2), a run of 24 clips on one stream and with the stencil of clip k+1 beside the
//     FFT pair of clip k;
//   * the stencil pass beside kernels that only store, only load or only issue vector instructions, as ordinary grids
//     of one-wave workgroups and as grids of a few long-lived waves per compute unit;
//   * what a 1.2 GB store or load stream reaches by shape: contiguous, non-temporal, whole 128-byte lines far apart,
//     half lines written by two instructions, hipMemsetAsync.
// (The runs with the stencil held to two or three workgroups per compute unit, and the row pass held to a few waves
// per compute unit, were made with a dynamic-LDS reservation on the stencil's launch and a grid-size override in the
// row pass's launcher, neither of which is in the library.)
//   usage: overlap_test <h> <w> <frames>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../sift_amd/csrc/stx_internal.h"
using namespace stx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
// diagnostic neighbours: one-wave workgroups that only store, only load, or only issue vector instructions
__global__ __launch_bounds__(64) void store_only(float4* out, int per_wg) {
    float4* o = out + (size_t)blockIdx.x * per_wg * 64 + threadIdx.x;
    const float4 v = make_float4(blockIdx.x, threadIdx.x, 1.f, 2.f);
    for (int i = 0; i < per_wg; ++i) o[(size_t)i * 64] = v;
}
__global__ __launch_bounds__(64) void store_nt(float4* out, int per_wg) {
    float4* o = out + (size_t)blockIdx.x * per_wg * 64 + threadIdx.x;
    const float4 v = make_float4(blockIdx.x, threadIdx.x, 1.f, 2.f);
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 vv = {v.x, v.y, v.z, v.w};
    for (int i = 0; i < per_wg; ++i) __builtin_nontemporal_store(vv, reinterpret_cast<f4*>(&o[(size_t)i * 64]));
}
__global__ __launch_bounds__(64) void load_nt(const float4* in, float* sink, int per_wg) {
    const float4* p = in + (size_t)blockIdx.x * per_wg * 64 + threadIdx.x;
    float acc = 0.f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int i = 0; i < per_wg; ++i) { const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(&p[(size_t)i * 64])); acc += v.x + v.y + v.z + v.w; }
    if (acc == 1234.5f) sink[0] = acc;
}
// the row pass's shape: a store instruction covers eight full 128-byte lines, 540 lines apart (one per column block)
template <bool NT>
__global__ __launch_bounds__(64) void store_lines(float4* out, int per_wg, int nwg) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x;
    const size_t blk = blockIdx.x / 540, row = blockIdx.x % 540;   // 540 row pairs of a frame, 8 column blocks per instruction
    f4* o = reinterpret_cast<f4*>(out) + (blk * per_wg * 8 * 540 + row) * 8 + (size_t)(lane >> 3) * 540 * 8 + (lane & 7);
    const f4 v = {(float)blockIdx.x, (float)lane, 1.f, 2.f};
    for (int i = 0; i < per_wg; ++i) {
        if (NT) __builtin_nontemporal_store(v, o + (size_t)i * 8 * 540 * 8); else o[(size_t)i * 8 * 540 * 8] = v;
    }
}
// the same bytes as 8-byte stores: a 64-byte half line per eight lanes, two instructions per line
template <bool NT>
__global__ __launch_bounds__(64) void store_halves(float2* out, int per_wg, int nwg) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x;
    const size_t blk = blockIdx.x / 540, row = blockIdx.x % 540;
    f2* o = reinterpret_cast<f2*>(out) + ((blk * per_wg * 4 * 540 + row) * 8) * 2 + (size_t)(lane >> 3) * 540 * 16 + (lane & 7);
    const f2 v = {(float)blockIdx.x, (float)lane};
    for (int i = 0; i < per_wg; ++i) {
        f2* q = o + (size_t)i * 8 * 540 * 16;
        if (NT) { __builtin_nontemporal_store(v, q); __builtin_nontemporal_store(v, q + 8); } else { q[0] = v; q[8] = v; }
    }
}
__global__ __launch_bounds__(256) void store_wg256(float4* out, int per_wg) {
    float4* o = out + (size_t)blockIdx.x * per_wg * 256 + threadIdx.x;
    const float4 v = make_float4(blockIdx.x, threadIdx.x, 1.f, 2.f);
    for (int i = 0; i < per_wg; ++i) o[(size_t)i * 256] = v;
}
__global__ __launch_bounds__(64) void load_only(const float4* in, float* sink, int per_wg) {
    const float4* p = in + (size_t)blockIdx.x * per_wg * 64 + threadIdx.x;
    float acc = 0.f;
    for (int i = 0; i < per_wg; ++i) { const float4 v = p[(size_t)i * 64]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1234.5f) sink[0] = acc;
}
__global__ __launch_bounds__(64) void valu_only(float* sink, int iters) {
    float a = threadIdx.x, b = blockIdx.x, c = 1.0001f, d = 0.5f;
    for (int i = 0; i < iters; ++i) { a = a * c + d; b = b * c + a; d = d * c + b; c = c * 0.99999f + 1e-6f; }
    if (a + b + c + d == 1234.5f) sink[0] = a;
}
int main(int argc, char** argv) {
    const int h = argc > 1 ? atoi(argv[1]) : 720, w = argc > 2 ? atoi(argv[2]) : 1280, F = argc > 3 ? atoi(argv[3]) : 150;
    Geom g;
    g.H = h; g.W = w; g.gpitch = ((w + 127) / 128) * 128 + 64; g.grows = h + 4;
    g.tiles_x = (w + 127) / 128; g.tiles_y = (h + 127) / 128; g.bm_words = (h + 63) / 64; g.bm_tiles = (w + 63) / 64;
    g.fh = h; g.fw = w; g.foy = 0; g.fox = 0; g.fwh = w / 2 + 1;
    FftPlan pw, ph;
    make_fft_plan(w, &pw); make_fft_plan(h, &ph);
    FftLayout lay = fft_choose_layout(g, ph);
    CK(fft_set_lds_limits());
    uint8_t *bgr, *gray[2], *tie, *sflag; uint32_t* part; unsigned long long *strong, *cand; float2 *spec, *tw;
    const size_t bgr_b = (size_t)F * h * w * 3, gray_b = (size_t)F * g.grows * g.gpitch, map_b = (size_t)F * g.W * g.bm_words * 8;
    CK(hipMalloc(&bgr, bgr_b)); CK(hipMalloc(&gray[0], gray_b)); CK(hipMalloc(&gray[1], gray_b));
    CK(hipMalloc(&part, (size_t)F * g.tiles_x * g.tiles_y * 64)); CK(hipMalloc(&strong, map_b)); CK(hipMalloc(&cand, map_b));
    CK(hipMalloc(&sflag, (size_t)F * g.bm_tiles * g.bm_words + 64));
    CK(hipMalloc(&spec, (size_t)F * lay.nkb * g.fh * lay.cbw * sizeof(float2)));
    LbpOffsets off;
    for (int i = 0; i < 8; ++i) { const double a = 2.0 * M_PI * i / 8; off.rp[i] = rint(-sin(a) * 1e5) / 1e5; off.cp[i] = rint(cos(a) * 1e5) / 1e5; }
    std::vector<uint8_t> blob; int ncls;
    build_tie_tables(off, &blob, &ncls);
    CK(hipMalloc(&tie, blob.size())); CK(hipMemcpy(tie, blob.data(), blob.size(), hipMemcpyHostToDevice));
    std::vector<double> t(2 * w); std::vector<float2> tf(w);
    fft_fill_table(pw, t.data());
    for (int k = 0; k < w; ++k) tf[k] = make_float2((float)t[2 * k], (float)t[2 * k + 1]);
    CK(hipMalloc(&tw, w * sizeof(float2))); CK(hipMemcpy(tw, tf.data(), w * sizeof(float2), hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    if (getenv("STX_EXP_PRIO")) {   // the stencil's stream ahead of the transforms' (1), or behind it (2)
        int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        printf("priorities: least %d, greatest %d\n", lo, hi);
        const bool first = atoi(getenv("STX_EXP_PRIO")) == 1;
        CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, first ? hi : lo)); CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, first ? lo : hi));
    } else { CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); }
    launch_synth(sa, bgr, 1, 0x5EED0002, 0, 0, F, h, w);
    BgrSource src = {bgr, 3 * w, (int64_t)3 * w * h, 15};
    auto stencil = [&](hipStream_t s, int which) {
        launch_stencil(s, gray[which], &src, F, g, true, off, 50, 150, part, strong, cand, nullptr, tie, sflag, true);
    };
    auto rows = [&](hipStream_t s, int which) { launch_fft_rows(s, gray[which], F, g, lay, pw, tw, spec); };
    stencil(sa, 0); stencil(sa, 1); rows(sa, 0);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char* name, auto fn) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, sa));
            fn();
            CK(hipStreamSynchronize(sb));
            CK(hipEventRecord(e1, sa));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-28s %.3f ms\n", name, best);
        return 0;
    };
    FftPart* fpart; float2* twh;
    CK(hipMalloc(&fpart, (size_t)F * lay.parts_per_frame * sizeof(FftPart)));
    {
        std::vector<double> th(2 * h); std::vector<float2> thf(h);
        fft_fill_table(ph, th.data());
        for (int k = 0; k < h; ++k) thf[k] = make_float2((float)th[2 * k], (float)th[2 * k + 1]);
        CK(hipMalloc(&twh, h * sizeof(float2))); CK(hipMemcpy(twh, thf.data(), h * sizeof(float2), hipMemcpyHostToDevice));
    }
    auto cols = [&](hipStream_t s, int which) {
        launch_fft_cols(s, gray[which], spec, F, g, lay, ph, twh, fpart, fft_small_threshold(g.fh, g.fw));
    };
    cols(sa, 0);
    CK(hipDeviceSynchronize());
    // steady state over a run of clips: one stream (stencil, rows, columns back to back), and two streams (the stencil
    // of clip k+1 beside the FFT pair of clip k; two gray planes, ordered by events)
    const int NC = 24;
    hipEvent_t st_done[2], rows_done[2];
    for (int i = 0; i < 2; ++i) { CK(hipEventCreateWithFlags(&st_done[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&rows_done[i], hipEventDisableTiming)); }
    timed("24 clips, one stream", [&] { for (int k = 0; k < NC; ++k) { stencil(sa, k & 1); rows(sa, k & 1); cols(sa, k & 1); } });
    timed("24 clips, two streams", [&] {
        for (int k = 0; k < NC; ++k) {
            const int b = k & 1;
            if (k >= 2) (void)hipStreamWaitEvent(sa, rows_done[b], 0);  // the plane's last reader
            stencil(sa, b);
            (void)hipEventRecord(st_done[b], sa);
            (void)hipStreamWaitEvent(sb, st_done[b], 0);
            rows(sb, b);
            (void)hipEventRecord(rows_done[b], sb);
            cols(sb, b);
        }
    });
    timed("rows then columns", [&] { rows(sa, 0); cols(sa, 0); });
    timed("rows || columns (2 streams)", [&] { rows(sa, 0); cols(sb, 0); });
    timed("columns || rows (2 streams)", [&] { cols(sb, 0); rows(sa, 0); });
    timed("stencil alone", [&] { stencil(sa, 1); });
    timed("rows alone", [&] { rows(sa, 0); });
    timed("columns alone", [&] { cols(sa, 0); });
    timed("stencil then rows", [&] { stencil(sa, 1); rows(sa, 0); });
    timed("stencil || rows (2 streams)", [&] { stencil(sa, 1); rows(sb, 0); });
    timed("rows || stencil (2 streams)", [&] { rows(sb, 0); stencil(sa, 1); });
    {
        const int nwg = 150000, per = 8;   // 150,000 x 8 KB = 1.23 GB
        float* sink; CK(hipMalloc(&sink, 64));
        float4* big = (float4*)spec;        // the spectrum buffer is at least that large at 1080p
        if ((size_t)F * lay.nkb * g.fh * lay.cbw * sizeof(float2) < (size_t)nwg * per * 1024) { printf("spectrum too small for the diagnostics\n"); return 0; }
        auto st = [&](hipStream_t s) { hipLaunchKernelGGL(store_only, dim3(nwg), dim3(64), 0, s, big, per); };
        auto ld = [&](hipStream_t s) { hipLaunchKernelGGL(load_only, dim3(nwg), dim3(64), 0, s, big, sink, per); };
        auto va = [&](hipStream_t s) { hipLaunchKernelGGL(valu_only, dim3(nwg), dim3(64), 0, s, sink, 900); };
        timed("stores alone", [&] { st(sa); });
        {
            float4* fresh; CK(hipMalloc(&fresh, (size_t)nwg * per * 1024));
            timed("stores alone, a buffer nothing else has touched", [&] { hipLaunchKernelGGL(store_only, dim3(nwg), dim3(64), 0, sa, fresh, per); });
            timed("the same on the null stream", [&] { hipLaunchKernelGGL(store_only, dim3(nwg), dim3(64), 0, 0, fresh, per); (void)hipStreamSynchronize(0); });
            CK(hipFree(fresh));
        }
        timed("non-temporal stores alone", [&] { hipLaunchKernelGGL(store_nt, dim3(nwg), dim3(64), 0, sa, big, per); });
        timed("stores, 256-thread groups", [&] { hipLaunchKernelGGL(store_wg256, dim3(nwg / 4), dim3(256), 0, sa, big, per); });
        {
            const int n3 = 540 * (nwg / 540);   // whole groups of 540 row pairs
            timed("full lines 540 apart", [&] { hipLaunchKernelGGL(store_lines<false>, dim3(n3), dim3(64), 0, sa, big, per, n3); });
            timed("full lines 540 apart, nt", [&] { hipLaunchKernelGGL(store_lines<true>, dim3(n3), dim3(64), 0, sa, big, per, n3); });
            timed("half lines twice", [&] { hipLaunchKernelGGL(store_halves<false>, dim3(n3), dim3(64), 0, sa, (float2*)big, 2 * per, n3); });
            timed("half lines twice, nt", [&] { hipLaunchKernelGGL(store_halves<true>, dim3(n3), dim3(64), 0, sa, (float2*)big, 2 * per, n3); });
        }
        timed("hipMemsetAsync", [&] { (void)hipMemsetAsync(big, 1, (size_t)nwg * per * 1024, sa); });
        timed("non-temporal loads alone", [&] { hipLaunchKernelGGL(load_nt, dim3(nwg), dim3(64), 0, sa, big, sink, per); });
        timed("stencil || stores", [&] { stencil(sa, 1); st(sb); });
        timed("loads alone", [&] { ld(sa); });
        timed("stencil || loads", [&] { stencil(sa, 1); ld(sb); });
        timed("valu alone", [&] { va(sa); });
        timed("stencil || valu", [&] { stencil(sa, 1); va(sb); });
        for (int per_cu : {4, 8, 16}) {   // the same neighbours as grids of a few long-lived waves per compute unit
            const int n2 = 256 * per_cu, per2 = (int)((size_t)nwg * per / n2);
            auto st2 = [&](hipStream_t s) { hipLaunchKernelGGL(store_only, dim3(n2), dim3(64), 0, s, big, per2); };
            auto ld2 = [&](hipStream_t s) { hipLaunchKernelGGL(load_only, dim3(n2), dim3(64), 0, s, big, sink, per2); };
            auto va2 = [&](hipStream_t s) { hipLaunchKernelGGL(valu_only, dim3(n2), dim3(64), 0, s, sink, 900 * (nwg / n2) / 4); };
            printf("neighbour: %d long-lived waves per compute unit\n", per_cu);
            timed("  stores alone", [&] { st2(sa); });
            timed("  stores, then stencil queued", [&] { st2(sb); stencil(sa, 1); });
            timed("  loads alone", [&] { ld2(sa); });
            timed("  loads, then stencil queued", [&] { ld2(sb); stencil(sa, 1); });
            timed("  valu alone", [&] { va2(sa); });
            timed("  valu, then stencil queued", [&] { va2(sb); stencil(sa, 1); });
        }
        timed("rows || valu", [&] { rows(sa, 0); va(sb); });
        timed("rows || loads", [&] { rows(sa, 0); ld(sb); });
    }
    return 0;
}
