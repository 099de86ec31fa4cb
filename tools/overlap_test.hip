// Experiment: how much of the FFT row pass (bound by HBM stores) hides beside the stencil pass (bound by vector issue)
// when the two run at the same time on two streams -- with the row kernel as built, and with it held to 128 registers
// so that its waves fit beside the stencil's.  Links the library's object files.
//   usage: overlap_test <h> <w> <frames>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../sift_amd/csrc/stx_internal.h"
using namespace stx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
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
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
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
    timed("stencil alone", [&] { stencil(sa, 1); });
    timed("rows alone", [&] { rows(sa, 0); });
    timed("stencil then rows", [&] { stencil(sa, 1); rows(sa, 0); });
    timed("stencil || rows (2 streams)", [&] { stencil(sa, 1); rows(sb, 0); });
    timed("rows || stencil (2 streams)", [&] { rows(sb, 0); stencil(sa, 1); });
    return 0;
}
