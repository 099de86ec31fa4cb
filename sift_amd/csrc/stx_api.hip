// Host side of libsift_texture.so: context, workspace, the per-chunk kernel schedule and
// the scalar epilogue that turns device reductions into the five metrics of
// TextureNoiseDetector.analyze_frame (/root/reference/app/passive_checker.py:96-131).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "stx_internal.h"

using namespace stx;

namespace {

constexpr int HYST_MAX_PASSES = 8192;
constexpr int HYST_FIRST_PASSES = 8;  // grid-wide hysteresis passes queued with every chunk
// Contours that need more (long thin structures) continue as one workgroup per frame that repeats the pass with
// barriers instead of launches (up to 63 rounds per launch) -- measured on 150 ordinary 1080p frames the five
// last grid passes take 34 us and one such launch 44 us, so it is the continuation, not the default.
constexpr int HYST_GRID_PASSES = HYST_FIRST_PASSES;
// Raw-threshold chunks (see nms_sparse_kernel): the flood of the raw maps keeps its own pass flags behind the
// ordinary ones: four grid-wide passes, two per-frame finishing launches, and in the last word the size of the
// flooded set.
constexpr int RAW_GRID_PASSES = 4, RAW_FINISH = 2, RAW_WORDS = 16, RAW_REACHED = 15;
// The flood after the suppression has a handful of tiles per frame to work on: no grid-wide passes, the per-frame
// kernel from the start, as pass RAW_SECOND_START (even: the flags it reads as those of the pass before are the second
// half, where the suppression marked the tiles that hold a strong pixel; changed[RAW_SECOND_START - 1] is preset).
constexpr int RAW_SECOND_GRID = 0, RAW_SECOND_START = 2, RAW_SECOND_PASSES = 2;
constexpr int HYST_GROUP = 1;         // further launches per host check when the first batch was not enough

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool fixed = false;  // a piece of the static scratch array: never freed, never grown
};

struct TimedSpan {
    int kind;
    int launches;
    hipEvent_t e0, e1;
};

}  // namespace

// Workspace of stx_analyze_batch_device_noalloc: every intermediate of that entry point lives here, so it never
// allocates.  One copy per GPU the library runs on, held until the process ends.  Host code must not use this name
// as an address: stx_create looks the device address up with hipGetSymbolAddress and keeps it per context.
static __device__ __align__(256) unsigned char scratch[STX_NOALLOC_SCRATCH_BYTES];

// Layout of the scratch array: four twiddle tables of STX_MAX_DIM entries, then the buffers of one chunk of frames.
constexpr size_t NA_TW_W = 0;
constexpr size_t NA_TW_H = NA_TW_W + FFT_TABLE_MAX * sizeof(float2);
constexpr size_t NA_TW64_W = NA_TW_H + FFT_TABLE_MAX * sizeof(float2);
constexpr size_t NA_TW64_H = NA_TW64_W + FFT_TABLE_MAX * sizeof(double2);
constexpr size_t NA_BUFFERS = NA_TW64_H + FFT_TABLE_MAX * sizeof(double2);
constexpr int NA_MAX_CHUNK = 64;  // frames per pass over the scratch array (the pinned result staging holds this many)

// What this process knows about each GPU's copy of the scratch array.  The array belongs to the GPU, not to a context:
// the lengths of the twiddle tables it holds are recorded here, where every context of that GPU sees them, and `busy`
// turns a second allocation-free call that would overlap the first into an error instead of a race.
constexpr int NA_MAX_DEVICES = 64;
struct NaDevice {
    std::mutex m;
    bool busy = false;
    int tw_n[4] = {0, 0, 0, 0};  // lengths of the tables currently in the array (0: none)
};
static NaDevice g_na_device[NA_MAX_DEVICES];

// Everything derived from (frame size, parameters) that the kernels of one call need.
struct Prepared {
    stx_params p;
    Geom g;
    FftLayout lay;
    FftPlan plan_w, plan_h;
    float2 *tw_w, *tw_h;
    LbpOffsets off;
    bool ext = false;        // extended statistics requested
    LbpOffsets off2, off3;   // their extra LBP radii: (16, 2) and (24, 3)
};

// One in-flight chunk: its own intermediates and result buffers, so the host can queue chunk
// i+1 while the device still works on chunk i.
struct Slot {
    DevBuf gray, strong, cand, spec, part, tiles, tflag, spart, lpart, qpart, xpart, acc, changed, staging, nmslist, nmscount, cand2;
    FrameAcc* acc_host = nullptr;  // pinned
    size_t acc_host_cap = 0;
    unsigned int* changed_host = nullptr;  // pinned, one word
    hipEvent_t done = nullptr;
    hipEvent_t gray_ready = nullptr, fft_done = nullptr, stencil_done = nullptr;
    bool busy = false;      // an asynchronous submission occupies this slot
    stx_metrics* async_out = nullptr;
    stx_extended* async_ext = nullptr;
    Prepared async_pr;
    int passes = 0;  // hysteresis passes queued so far
    bool raw = false;  // the stencil pass left raw threshold maps: flood, suppression on the flooded set, flood again
    const uint8_t* src_bgr = nullptr;  // the chunk's frames (a raw chunk whose first flood does not settle is queued again)
    int64_t src_row_stride = 0, src_frame_stride = 0;
    uint8_t* src_codes = nullptr;
    int F = 0;
    std::vector<TimedSpan> spans;
    std::vector<uint8_t> pack;  // host rows of strided frames, packed for one linear copy
    std::vector<int> redo;      // frames of the chunk whose transform is repeated in fp64
};
constexpr int N_SLOTS = 2;

struct stx_ctx {
    int device = 0;
    // Three in-order streams shared by all chunks: kernels of one kind from consecutive chunks run
    // one after the other, kernels of different kinds overlap (the gray pass of chunk i+1 is
    // memory-bound, the stencil of chunk i issue-bound, its FFT row pass bound by HBM stores).
    hipStream_t stream = nullptr;      // stencil, hysteresis, reductions, result copies, misc
    hipStream_t stream_pre = nullptr;  // host->device copies and the gray pass
    hipStream_t stream_fft = nullptr;  // FFT rows + columns
    Slot* prev_chunk = nullptr;        // the chunk queued last (its big kernels gate the next chunk's gray pass)
    std::string err;
    Slot slot[N_SLOTS];
    DevBuf codes, plane, taps, prep_out, tie;
    DevBuf spec64, scratch64, part64;  // double-precision transform of single frames
    std::map<int, void*> twiddles64;
    std::vector<double2> part64_host;
    uint64_t fft64_frames = 0;         // frames whose spectral statistic was recomputed in double precision
    // stx_analyze_batch_device_noalloc
    uint8_t* na_base = nullptr;        // device address of this GPU's copy of the scratch array
    Slot na;                           // its buffers point into the scratch array
    // pinned host staging for table uploads: FFT_TABLE_MAX double2, then as many float2, then the 2 * STX_MAX_DIM
    // double2 a chirp-z table is built in
    void* na_stage = nullptr;
    int tie_ncls = 0;
    double tie_key[4] = {0, 0, 0, 0};  // the offsets the tie tables were built for
    std::map<int, float2*> twiddles;
    int chunk_override = 0;
    bool concurrent_fft = true;
    bool streaming = false;  // stx_synth_device_async has been used: chunks record the end of their stencil pass for it
    // the stencil pass converts BGR itself where it can; SIFT_TEXTURE_SEPARATE_GRAY=1 in the environment keeps the
    // separate gray pass for every call (measurements, and tests that run both paths on one input)
    bool fuse_gray = true;
    // where Canny's non-maximum suppression runs (stx_set_canny_mode; SIFT_TEXTURE_CANNY=dense|sparse|auto)
    int canny_mode = STX_CANNY_AUTO;
    bool raw_next = true;   // auto: what the next chunk does
    int dense_run = 0;      // auto: dense chunks since the last sparse one
    int raw_probe = 32;     // auto: dense chunks before the next sparse try
    uint64_t raw_requeued = 0;
    bool timing = false;
    uint32_t timing_kinds = ~0u;       // bit k: launches of kernel family k are bracketed by events while timing is on
    stx_timing tacc;
    std::vector<hipEvent_t> event_pool;
    std::vector<TimedSpan> spans_kept;
    std::vector<TimedSpan> synth_spans;  // timing spans of stx_synth_device_async launches not collected yet
};

namespace {

int fail(stx_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess)                                                                       \
            return fail((c), STX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));       \
    } while (0)

int ensure(stx_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return STX_OK;
    if (b.fixed) return fail(c, STX_ERR_NOMEM, "internal: a buffer carved from the static scratch array is too small");
    if (b.p) {
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(c, STX_ERR_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
    }
    b.cap = want;
    return STX_OK;
}

hipEvent_t get_event(stx_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    // timing events: a release at device scope, which keeps the event's own cost out of the durations it brackets
    if (hipEventCreateWithFlags(&e, hipEventReleaseToDevice) != hipSuccess) return nullptr;
    return e;
}

// begin/end bracket kernel launches of one family on the context stream
struct Span {
    stx_ctx* c;
    std::vector<TimedSpan>* dst;
    TimedSpan t;
    bool on;
    hipStream_t st;
    Span(stx_ctx* c_, std::vector<TimedSpan>* dst_, int kind, int launches, hipStream_t stream = nullptr)
        : c(c_), dst(dst_), on(c_->timing && ((c_->timing_kinds >> kind) & 1u)), st(stream ? stream : c_->stream) {
        if (!on) return;
        t.kind = kind;
        t.launches = launches;
        t.e0 = get_event(c);
        t.e1 = get_event(c);
        if (!t.e0 || !t.e1) { on = false; return; }
        (void)hipEventRecord(t.e0, st);
    }
    void end() {
        if (!on) return;
        (void)hipEventRecord(t.e1, st);
        dst->push_back(t);
    }
};

// call only after the stream has passed the spans' end events
void collect_spans(stx_ctx* c, std::vector<TimedSpan>& spans) {
    for (auto& t : spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) {
            c->tacc.ms[t.kind] += ms;
            c->tacc.launches[t.kind] += (uint64_t)t.launches;
        }
        c->event_pool.push_back(t.e0);
        c->event_pool.push_back(t.e1);
    }
    spans.clear();
}

// The table a plan's kernels read (twiddles, or the chirp-z tables of a length with a large prime factor), one per
// length and precision, kept for the life of the context.
int get_twiddles(stx_ctx* c, const FftPlan& plan, float2** out) {
    auto it = c->twiddles.find(plan.n);
    if (it != c->twiddles.end()) { *out = it->second; return STX_OK; }
    const size_t entries = fft_table_entries(plan);
    std::vector<double> t;
    std::vector<float2> h;
    try {
        t.resize(2 * entries);
        h.resize(entries);
    } catch (const std::bad_alloc&) {
        return fail(c, STX_ERR_NOMEM, "out of host memory");
    }
    fft_fill_table(plan, t.data());
    for (size_t k = 0; k < entries; ++k) h[k] = make_float2((float)t[2 * k], (float)t[2 * k + 1]);
    float2* d = nullptr;
    HIPCHK(c, hipMalloc(&d, sizeof(float2) * entries));
    HIPCHK(c, hipMemcpy(d, h.data(), sizeof(float2) * entries, hipMemcpyHostToDevice));
    c->twiddles[plan.n] = d;
    *out = d;
    return STX_OK;
}

int get_twiddles64(stx_ctx* c, const FftPlan& plan, void** out) {
    auto it = c->twiddles64.find(plan.n);
    if (it != c->twiddles64.end()) { *out = it->second; return STX_OK; }
    const size_t entries = fft_table_entries(plan);
    std::vector<double> t;
    try {
        t.resize(2 * entries);
    } catch (const std::bad_alloc&) {
        return fail(c, STX_ERR_NOMEM, "out of host memory");
    }
    fft_fill_table(plan, t.data());
    void* d = nullptr;
    HIPCHK(c, hipMalloc(&d, sizeof(double2) * entries));
    HIPCHK(c, hipMemcpy(d, t.data(), sizeof(double2) * entries, hipMemcpyHostToDevice));
    c->twiddles64[plan.n] = d;
    *out = d;
    return STX_OK;
}

// numpy.round(x, 5) = rint(x * 1e5) / 1e5; the sample offsets of skimage's local_binary_pattern
int lbp_offsets(int P, double R, LbpOffsets* o) {
    if (P < 1 || P > 24) return -1;
    for (int i = 0; i < P; ++i) {
        const double ang = 2.0 * M_PI * (double)i / (double)P;
        o->rp[i] = rint(-R * sin(ang) * 100000.0) / 100000.0;
        o->cp[i] = rint(R * cos(ang) * 100000.0) / 100000.0;
    }
    return 0;
}

// The (P=8, R=1) tables for flat neighbourhoods: built once per context (stx_create does it for the default
// offsets, so the per-call paths find them in place).
int ensure_tie_tables(stx_ctx* c, const LbpOffsets& off) {
    const double key[4] = {off.rp[1], off.rp[5], off.cp[1], off.cp[3]};
    if (c->tie.p && memcmp(key, c->tie_key, sizeof(key)) == 0) return STX_OK;
    std::vector<uint8_t> blob;
    build_tie_tables(off, &blob, &c->tie_ncls);
    int rc;
    if ((rc = ensure(c, c->tie, blob.size())) != STX_OK) return rc;
    HIPCHK(c, hipMemcpy(c->tie.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
    memcpy(c->tie_key, key, sizeof(key));
    return STX_OK;
}

double dmin(double a, double b) { return a < b ? a : b; }

// passive_checker.py:145-151
double texture_score(const stx_metrics& m) {
    return dmin(m.lbp_variance / 50.0, 1.0) * 0.3 + dmin(m.edge_density * 10, 1.0) * 0.2 +
           (1.0 - dmin(m.compression_artifacts / 10.0, 1.0)) * 0.2 + dmin(m.noise_level / 20.0, 1.0) * 0.15 +
           (1.0 - dmin(m.frequency_anomaly / 100.0, 1.0)) * 0.15;
}

// Population statistics (ddof = 0) from the device reductions: np.var :104, np.std :108/:123,
// np.mean :113, count/size :117.
void finalize(const FrameAcc& a, const Geom& g, int P, stx_metrics* m) {
    memset(m, 0, sizeof(*m));
    const uint64_t n = (uint64_t)g.H * (uint64_t)g.W;
    const double dn = (double)n;
    m->n_pixels = n;
    for (int k = 0; k < STX_LBP_MAX_BINS; ++k) m->lbp_hist[k] = a.lbp_hist[k];
    m->edge_count = a.edge_count;
    m->flags = a.pad0 ? STX_FLAG_FFT_F64 : 0u;
    m->absdiff_sum = a.absdiff_sum;
    m->hp_sum = a.hp_sum;
    m->hp_sumsq = a.hp_sumsq;
    m->gray_sum = a.gray_sum;
    m->fft_sum = a.fft_sum;
    m->fft_sumsq = a.fft_sumsq;
    double s1 = 0.0;
    for (int k = 0; k < P + 2; ++k) s1 += (double)k * (double)a.lbp_hist[k];
    const double mean = s1 / dn;
    double acc = 0.0;
    for (int k = 0; k < P + 2; ++k) acc += (double)a.lbp_hist[k] * ((double)k - mean) * ((double)k - mean);
    m->lbp_variance = acc / dn;
    const unsigned __int128 num =
        (unsigned __int128)n * a.hp_sumsq - (unsigned __int128)a.hp_sum * (unsigned __int128)a.hp_sum;
    m->noise_level = sqrt((double)num / (dn * dn));
    m->compression_artifacts = (double)a.absdiff_sum / dn;
    m->edge_density = (double)a.edge_count / dn;
    const double nf = (double)g.fh * (double)g.fw;
    const double fm = a.fft_sum / nf;
    const double fv = a.fft_sumsq / nf - fm * fm;
    m->frequency_anomaly = fv > 0.0 ? sqrt(fv) : 0.0;
    m->texture_score = texture_score(*m);
}

// Whether the fp32 transform's rounding noise can move the spectral statistic of this frame by more than 2e-5 of
// its value (the parity budget is 1e-4).  Bins well above the noise are perturbed by a relative, zero-mean amount
// that cancels in the sums; a bin near zero is not: its ln(|F|+1) rises by up to err/(|F|+1).  With l = ln(|F|+1)
// and mean mu, the variance then changes by at most 2 err sum_small |mu - l| / (|F|+1) / N to first order, which
// the column pass's t1 = sum 1/(|F|+1) and t2 = sum l/(|F|+1) over the small bins give.  err scales with the
// root-mean-square of the transform's input (gray minus the frame's first gray level), which Parseval reads off the
// spectrum: sum |F|^2 = fh fw sum (gray - m)^2.  The small bins were picked at launch for the largest possible
// input; counting more bins than a quiet frame's noise reaches only makes the bound more careful.  A constant frame
// has no energy and no noise: its spectrum is exactly zero, as in the reference.
bool fft_needs_f64(const FrameAcc& a, const Geom& g) {
    if (!(a.fft_t1 > 0.0)) return false;
    const double nf = (double)g.fh * (double)g.fw;
    const double rms = sqrt((a.fft_e2 > 0.0 ? a.fft_e2 : 0.0) / (nf * nf));
    const double mu = a.fft_sum / nf;
    const double var = a.fft_sumsq / nf - mu * mu;
    const double lmax = log((double)fft_small_threshold(g.fh, g.fw) + 1.0);
    const double dev = mu >= lmax ? mu * a.fft_t1 - a.fft_t2 : lmax * a.fft_t1;
    const double dvar = 2.0 * fft_f32_bin_error_rms(g.fh, g.fw, rms < 255.0 ? rms : 255.0) * dev / nf;
    return dvar > 4e-5 * (var > 0.0 ? var : 0.0);  // d(std)/std = dvar / (2 var)
}

// Population variances of the Laplacian image and of the gray image (main.py:209-210) from exact integer sums.
void finalize_quality(const FrameAcc& a, const Geom& g, stx_metrics* m) {
    const uint64_t n = (uint64_t)g.H * (uint64_t)g.W;
    const double dn = (double)n;
    m->lap_sum = a.lap_sum;
    m->lap_sumsq = a.lap_sumsq;
    m->gray_sumsq = a.gray_sumsq;
    const __int128 ls = (__int128)a.lap_sum;
    const unsigned __int128 lnum = (unsigned __int128)n * a.lap_sumsq - (unsigned __int128)(ls * ls);
    m->laplacian_variance = (double)lnum / (dn * dn);
    const unsigned __int128 gnum =
        (unsigned __int128)n * a.gray_sumsq - (unsigned __int128)a.gray_sum * (unsigned __int128)a.gray_sum;
    m->brightness_variance = (double)gnum / (dn * dn);
}

int make_geom(stx_ctx* c, int h, int w, const stx_params& p, Geom* g) {
    if (h < 1 || w < 1) return fail(c, STX_ERR_INVALID, "frame height and width must be >= 1");
    if (h > STX_MAX_DIM || w > STX_MAX_DIM) return fail(c, STX_ERR_UNSUPPORTED, "frame larger than 8192 in one dimension");
    g->H = h; g->W = w;
    g->gpitch = ((w + 127) / 128) * 128 + 64;
    g->grows = h + 4;
    g->tiles_x = (w + 127) / 128;
    g->tiles_y = (h + 127) / 128;
    g->bm_words = (h + 63) / 64;
    g->bm_tiles = (w + 63) / 64;
    g->fh = p.fft_h > 0 ? p.fft_h : h;
    g->fw = p.fft_w > 0 ? p.fft_w : w;
    if (g->fh > h || g->fw > w) return fail(c, STX_ERR_INVALID, "fft crop larger than the frame");
    g->foy = (h - g->fh) / 2;
    g->fox = (w - g->fw) / 2;
    g->fwh = g->fw / 2 + 1;
    return STX_OK;
}

int check_params(stx_ctx* c, const stx_params& p) {
    if (p.lbp_points < 1 || p.lbp_points > 24) return fail(c, STX_ERR_UNSUPPORTED, "lbp_points must be in 1..24");
    if (!(p.lbp_radius > 0.0) || p.lbp_radius > 16.0) return fail(c, STX_ERR_UNSUPPORTED, "lbp_radius must be in (0,16]");
    if (p.gray_shift != 14 && p.gray_shift != 15) return fail(c, STX_ERR_INVALID, "gray_shift must be 14 or 15");
    if (p.canny_low < 0 || p.canny_high < 0) return fail(c, STX_ERR_INVALID, "canny thresholds must be >= 0");
    if (p.fft_h < 0 || p.fft_w < 0) return fail(c, STX_ERR_INVALID, "fft crop must be >= 0");
    if (p.fft_exact < -1 || p.fft_exact > 1) return fail(c, STX_ERR_INVALID, "fft_exact must be -1, 0 or 1");
    return STX_OK;
}

size_t per_frame_workspace(const Geom& g) {
    return (size_t)g.grows * g.gpitch + 2 * (size_t)g.W * g.bm_words * 8 + (size_t)g.fh * (g.fwh + 15) * sizeof(float2);
}

int pick_chunk(const stx_ctx* c, const Geom& g, int n_frames) {
    int f;
    if (c->chunk_override > 0) {
        f = c->chunk_override;
    } else {
        // Large chunks: every kernel then launches thousands of workgroups and the tail of a
        // launch is negligible (measured at 1080p: 15-frame chunks 5.1 ms per 150 frames, one
        // 150-frame chunk 4.3 ms; the kernels are compute-bound, so keeping the intermediates
        // inside the Infinity Cache buys nothing).  2 GiB of intermediates per in-flight chunk.
        const size_t budget = (size_t)2 << 30;
        f = (int)std::max<size_t>(1, budget / per_frame_workspace(g));
        f = std::min(f, 4096);
    }
    return std::min(f, n_frames);
}

// One host frame into device staging rows of dpitch bytes.  Frames whose rows are already dpitch apart (every
// width that is a multiple of four, from a contiguous array) go as one linear copy: a 2-D copy out of pageable
// memory is served row by row by the runtime (measured 2.5 ms for a 640x480 frame against 0.1 ms linear).
hipError_t upload_frame(uint8_t* dst, int64_t dpitch, const uint8_t* src, int64_t row_stride, int w, int h, hipStream_t st) {
    if (row_stride == dpitch) return hipMemcpyAsync(dst, src, (size_t)(dpitch * (h - 1) + 3 * (int64_t)w), hipMemcpyHostToDevice, st);
    return hipMemcpy2DAsync(dst, (size_t)dpitch, src, (size_t)row_stride, (size_t)3 * w, (size_t)h, hipMemcpyHostToDevice, st);
}

struct ChunkOpts {
    uint8_t* codes = nullptr;  // device, optional [F][H][W]
    bool force_dense = false;
};
constexpr int RAW_PROBE = 32, RAW_PROBE_MAX = 1024;  // auto: dense chunks between two sparse tries, doubling while the tries fail
constexpr int64_t RAW_MIN_PIXELS = (int64_t)64 << 20;  // auto: chunks smaller than this stay dense
constexpr double RAW_MAX_DENSITY = 0.02;  // flooded pixels per pixel above which the suppression goes back into the stencil pass


int prepare(stx_ctx* c, int h, int w, const stx_params* params, Prepared* pr, bool own_twiddles = true) {
    if (!c) return STX_ERR_INVALID;
    if (params) pr->p = *params; else stx_default_params(&pr->p);
    int rc;
    if ((rc = check_params(c, pr->p)) != STX_OK) return rc;
    if (pr->p.canny_low > pr->p.canny_high) std::swap(pr->p.canny_low, pr->p.canny_high);  // cv2.Canny swaps them
    if ((rc = make_geom(c, h, w, pr->p, &pr->g)) != STX_OK) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (!make_fft_plan(pr->g.fw, &pr->plan_w) || !make_fft_plan(pr->g.fh, &pr->plan_h))
        return fail(c, STX_ERR_UNSUPPORTED, "FFT size has too many prime factors");
    pr->lay = fft_choose_layout(pr->g, pr->plan_h);
    pr->tw_w = pr->tw_h = nullptr;
    if (own_twiddles) {
        if ((rc = get_twiddles(c, pr->plan_w, &pr->tw_w)) != STX_OK) return rc;
        if ((rc = get_twiddles(c, pr->plan_h, &pr->tw_h)) != STX_OK) return rc;
    }
    if (lbp_offsets(pr->p.lbp_points, pr->p.lbp_radius, &pr->off) != 0) return fail(c, STX_ERR_INVALID, "bad lbp_points");
    return STX_OK;
}

bool any_busy(const stx_ctx* c) {
    for (const Slot& sl : c->slot)
        if (sl.busy) return true;
    return false;
}

void sync_all(stx_ctx* c) {
    if (c->stream_pre) (void)hipStreamSynchronize(c->stream_pre);
    if (c->stream_fft) (void)hipStreamSynchronize(c->stream_fft);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
}

int hyst_passes(stx_ctx* c, Slot& sl, int n, const Geom& g, hipStream_t st = nullptr) {
    if (sl.passes + n > HYST_MAX_PASSES) return fail(c, STX_ERR_UNSUPPORTED, "Canny hysteresis did not converge");
    if (!st) st = c->stream;
    // (a raw chunk's ordinary passes run on the map the suppression filled)
    const unsigned long long* cand = (const unsigned long long*)(sl.raw ? sl.cand2.p : sl.cand.p);
    Span sp(c, &sl.spans, STX_K_HYST, n, st);
    for (int i = 0; i < n; ++i, ++sl.passes) {
        // (after the suppression on a flooded set few tiles are left to grow: two grid-wide passes, then per-frame ones)
        if (sl.passes < (sl.raw ? RAW_SECOND_GRID : HYST_GRID_PASSES))
            launch_hyst_pass(st, (unsigned long long*)sl.strong.p, cand, sl.F, g,
                             (unsigned int*)sl.changed.p, sl.passes, (uint32_t*)sl.tiles.p, (uint8_t*)sl.tflag.p, sl.raw);
        else
            launch_hyst_finish(st, (unsigned long long*)sl.strong.p, cand, sl.F, g,
                               (unsigned int*)sl.changed.p, sl.passes, (uint32_t*)sl.tiles.p, (uint8_t*)sl.tflag.p, sl.raw);
    }
    sp.end();
    return STX_OK;
}

int queue_results(stx_ctx* c, Slot& sl, const Prepared& pr, hipStream_t s = nullptr) {
    if (!s) s = c->stream;
    const bool lbp81 = (pr.p.lbp_points == 8 && pr.p.lbp_radius == 1.0);
    {
        Span sp(c, &sl.spans, STX_K_REDUCE, 1, s);
        launch_frame_reduce(s, (const uint32_t*)sl.spart.p, lbp81, (const uint32_t*)sl.tiles.p, (const FftPart*)sl.part.p,
                            pr.lay.parts_per_frame, sl.F, pr.g, (FrameAcc*)sl.acc.p);
        sp.end();
    }
    HIPCHK(c, hipMemcpyAsync(sl.changed_host, (unsigned int*)sl.changed.p + (sl.passes - 1), sizeof(unsigned int),
                             hipMemcpyDeviceToHost, s));
    if (sl.raw)  // "the first flood was still growing" and the size of the flooded set, behind the word above
        HIPCHK(c, hipMemcpyAsync(sl.changed_host + 1, (unsigned int*)sl.changed.p + HYST_MAX_PASSES + RAW_GRID_PASSES + RAW_FINISH - 1,
                                 (size_t)(RAW_WORDS - (RAW_GRID_PASSES + RAW_FINISH - 1)) * sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(sl.acc_host, sl.acc.p, (size_t)sl.F * sizeof(FrameAcc), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipEventRecord(sl.done, s));
    return STX_OK;
}

// Bytes of every device buffer a chunk of F frames needs, in the order of ChunkBuf.
enum ChunkBuf { CB_GRAY, CB_STRONG, CB_CAND, CB_SPEC, CB_PART, CB_TILES, CB_TFLAG, CB_SPART, CB_ACC, CB_CHANGED, CB_LPART,
                CB_QPART, CB_XPART, CB_NMSLIST, CB_NMSCOUNT, CB_CAND2, CB_COUNT };
void chunk_sizes(const Prepared& pr, int F, size_t (&n)[CB_COUNT], bool raw = false) {
    const Geom& g = pr.g;
    const bool lbp81 = (pr.p.lbp_points == 8 && pr.p.lbp_radius == 1.0);
    n[CB_GRAY] = (size_t)F * g.grows * g.gpitch;
    n[CB_STRONG] = n[CB_CAND] = (size_t)F * g.W * g.bm_words * sizeof(unsigned long long);
    n[CB_SPEC] = (size_t)F * pr.lay.nkb * g.fh * pr.lay.cbw * sizeof(float2);
    n[CB_PART] = (size_t)F * pr.lay.parts_per_frame * sizeof(FftPart);
    n[CB_TILES] = (size_t)F * g.bm_tiles * g.bm_words * sizeof(uint32_t);
    // two flag halves (pass parities), a third part for the validity bytes of a raw chunk's second candidate map;
    // four spare bytes: the second half is cleared in whole words
    n[CB_TFLAG] = (size_t)3 * F * g.bm_tiles * g.bm_words + 4;
    n[CB_SPART] = (size_t)F * g.tiles_x * g.tiles_y * 16 * sizeof(uint32_t);
    n[CB_ACC] = (size_t)F * sizeof(FrameAcc);
    n[CB_CHANGED] = (size_t)(HYST_MAX_PASSES + RAW_WORDS) * sizeof(unsigned int);
    n[CB_LPART] = (!lbp81 || pr.ext) ? lbp_generic_part_bytes(g, F) : 0;
    n[CB_QPART] = pr.p.quality_stats ? quality_part_bytes(g, F) : 0;
    n[CB_XPART] = pr.ext ? extended_part_bytes(g, F) : 0;
    n[CB_NMSLIST] = raw ? (size_t)F * nms_list_cap(g) * sizeof(uint32_t) : 0;
    n[CB_NMSCOUNT] = raw ? (size_t)F * sizeof(unsigned int) : 0;
    n[CB_CAND2] = raw ? n[CB_CAND] : 0;
}
DevBuf* chunk_buf(Slot& sl, int i) {
    DevBuf* b[CB_COUNT] = {&sl.gray, &sl.strong, &sl.cand, &sl.spec, &sl.part, &sl.tiles, &sl.tflag, &sl.spart, &sl.acc,
                           &sl.changed, &sl.lpart, &sl.qpart, &sl.xpart, &sl.nmslist, &sl.nmscount, &sl.cand2};
    return b[i];
}

// Queues the whole path for F device-resident frames on the context stream; returns without
// waiting.  finish_chunk() waits and leaves the reductions in sl.acc_host[0..F).
int enqueue_chunk(stx_ctx* c, Slot& sl, const uint8_t* bgr_dev, int64_t row_stride, int64_t frame_stride, int F,
                  const Prepared& pr, const ChunkOpts& opts) {
    hipStream_t s = c->stream;
    hipStream_t spre = c->concurrent_fft ? c->stream_pre : s;
    const Geom& g = pr.g;
    const stx_params& p = pr.p;
    const FftLayout& lay = pr.lay;
    int rc;
    sl.src_bgr = bgr_dev; sl.src_row_stride = row_stride; sl.src_frame_stride = frame_stride; sl.src_codes = opts.codes;
    {
        const bool eligible = p.lbp_points == 8 && p.lbp_radius == 1.0 && !opts.codes && !opts.force_dense && !sl.gray.fixed;
        // auto: chunks below RAW_MIN_PIXELS stay dense and do not count -- the sparse path saves a tenth of a millisecond
        // per 150 frames of 1080p in the stencil pass and costs two more launches and a clear, which is what a call on
        // one frame or ten is made of (analyze_frame at 1080p: 0.32 ms dense, 0.38 ms sparse)
        const bool big = (int64_t)F * g.H * g.W >= RAW_MIN_PIXELS;
        sl.raw = eligible && (c->canny_mode == STX_CANNY_SPARSE || (c->canny_mode == STX_CANNY_AUTO && c->raw_next && big));
        if (eligible && big && c->canny_mode == STX_CANNY_AUTO && !sl.raw && ++c->dense_run >= c->raw_probe) {
            c->raw_next = true;
            c->dense_run = 0;
        }
    }
    size_t need[CB_COUNT];
    chunk_sizes(pr, F, need, sl.raw);
    for (int i = 0; i < CB_COUNT; ++i)
        if (need[i] && (rc = ensure(c, *chunk_buf(sl, i), need[i])) != STX_OK) return rc;
    if ((size_t)F > sl.acc_host_cap) {
        if (sl.gray.fixed) return fail(c, STX_ERR_NOMEM, "internal: result staging of the allocation-free path is too small");
        if (sl.acc_host) HIPCHK(c, hipHostFree(sl.acc_host));
        sl.acc_host = nullptr;
        sl.acc_host_cap = 0;
        HIPCHK(c, hipHostMalloc((void**)&sl.acc_host, sizeof(FrameAcc) * (size_t)F * 2, hipHostMallocDefault));
        sl.acc_host_cap = (size_t)F * 2;
    }
    sl.F = F;
    sl.passes = 0;
    FrameAcc* acc = (FrameAcc*)sl.acc.p;
    // Chunks queue up on the device but do not run side by side: this chunk's gray pass starts when the stencil and
    // FFT kernels of the chunk queued before it are done (under its hysteresis passes, which leave the device mostly
    // idle).  The big kernels of two chunks in each other's way cost more than the gaps they fill: 1.640 ms per
    // 150-frame 1080p clip with two clips in flight unchained, 1.602 ms chained, 1.65 ms one blocking call at a time.
    // The stencil pass converts the frames itself where it can (a separate gray pass is a fifth of a millisecond of
    // HBM streaming per 150-frame 1080p clip, the conversion inside the stencil's tile loader a quarter of that);
    // every kernel after it reads the plane it leaves behind.
    const BgrSource bsrc = {bgr_dev, row_stride, frame_stride, p.gray_shift};
    const bool fused_gray = c->fuse_gray && stencil_reads_bgr(g, bsrc);
    if (!fused_gray) {
        if (c->prev_chunk && spre != s) {
            HIPCHK(c, hipStreamWaitEvent(spre, c->prev_chunk->stencil_done, 0));
            HIPCHK(c, hipStreamWaitEvent(spre, c->prev_chunk->fft_done, 0));
        }
        Span sp(c, &sl.spans, STX_K_GRAY, 1, spre);
        launch_gray(spre, bgr_dev, row_stride, frame_stride, F, g, p.gray_shift, (uint8_t*)sl.gray.p);
        sp.end();
    }
    // Clears.  The second half of the tile flags doubles as "this tile holds a strong pixel" between the stencil and
    // the first hysteresis pass (the second pass is the first to write it as change flags).
    // With a separate gray pass they sit between the gray pass and the stencil on purpose.  They do not depend on the
    // gray plane, but queued ahead of the wait the stencil starts together with the FFT row pass, holds the wave
    // slots, and the row pass (a persistent grid that needs half a SIMD's registers per wave) trickles in beside it:
    // 1.79 ms per 150-frame 1080p clip against 1.75 ms when the row pass gets the 30 us head start the clears give
    // it, runs by itself at its HBM-store bound, and the stencil then shares the device with the column pass.
    // Without one they run on the otherwise idle first stream as soon as they are queued, under the chunk before.
    hipStream_t side = c->concurrent_fft ? c->stream_fft : s;
    const bool fused_flow = fused_gray && side != s;
    hipStream_t sclr = fused_flow ? spre : s;
    if (!fused_flow) {
        HIPCHK(c, hipEventRecord(sl.gray_ready, spre));
        if (spre != s) HIPCHK(c, hipStreamWaitEvent(s, sl.gray_ready, 0));
    }
    const bool lbp81 = (p.lbp_points == 8 && p.lbp_radius == 1.0);
    if (lbp81 && (rc = ensure_tie_tables(c, pr.off)) != STX_OK) return rc;
    {
        unsigned int* ch = (unsigned int*)sl.changed.p;
        const size_t half = (size_t)F * g.bm_tiles * g.bm_words;
        // (the second flag half may start at an odd byte: cleared from the word that holds its first byte -- the bytes
        // of the first half in that word are flags no kernel has written yet -- to the word that holds its last)
        uint8_t* sf = (uint8_t*)sl.tflag.p + (half & ~(size_t)3);
        launch_chunk_init(sclr, acc, F, ch, HYST_FIRST_PASSES, sl.raw ? ch + HYST_MAX_PASSES : nullptr, RAW_WORDS,
                          sl.raw ? ch + RAW_SECOND_START - 1 : nullptr, sl.raw ? (unsigned int*)sl.nmscount.p : nullptr,
                          sf, half + (half & 3));
    }
    if (fused_flow) {
        HIPCHK(c, hipEventRecord(sl.gray_ready, spre));
        HIPCHK(c, hipStreamWaitEvent(s, sl.gray_ready, 0));
    }
    {
        Span sp(c, &sl.spans, STX_K_STENCIL, 1, s);
        launch_stencil(s, (uint8_t*)sl.gray.p, fused_gray ? &bsrc : nullptr, F, g, lbp81, pr.off, p.canny_low, p.canny_high, (uint32_t*)sl.spart.p,
                       (unsigned long long*)sl.strong.p, (unsigned long long*)sl.cand.p, lbp81 ? opts.codes : nullptr,
                       (const uint8_t*)c->tie.p, (uint8_t*)sl.tflag.p + (size_t)F * g.bm_tiles * g.bm_words, sl.raw);
        sp.end();
    }
    HIPCHK(c, hipEventRecord(sl.stencil_done, s));
    if (!lbp81) {
        Span sp(c, &sl.spans, STX_K_LBP_GENERIC, 2, s);
        const bool ok = launch_lbp_generic(s, (const uint8_t*)sl.gray.p, F, g, p.lbp_points, pr.off, (uint32_t*)sl.lpart.p, acc,
                                           opts.codes);
        sp.end();
        if (!ok) return fail(c, STX_ERR_UNSUPPORTED, "lbp_radius too large for one LDS tile");
    }
    if (p.quality_stats) {
        Span sp(c, &sl.spans, STX_K_QUALITY, 2, s);
        launch_quality(s, (const uint8_t*)sl.gray.p, F, g, (unsigned long long*)sl.qpart.p, acc);
        sp.end();
    }
    if (pr.ext) {
        {
            Span sp(c, &sl.spans, STX_K_EXTENDED, 2, s);
            launch_extended(s, (const uint8_t*)sl.gray.p, F, g, p.canny_high, (unsigned long long*)sl.xpart.p, acc);
            sp.end();
        }
        Span sp(c, &sl.spans, STX_K_LBP_GENERIC, 3, s);
        const bool ok = launch_lbp_rings(s, (const uint8_t*)sl.gray.p, F, g, pr.off2, pr.off3, (uint32_t*)sl.lpart.p, acc);
        sp.end();
        if (!ok) return fail(c, STX_ERR_UNSUPPORTED, "lbp_radius too large for one LDS tile");
    }
    // A fixed first batch of hysteresis passes (a pass exits at once when the one before it changed nothing);
    // finish_chunk() checks the last counter and only then adds more.
    // With a separate gray pass the FFT pair only needs the gray plane: it runs on a second stream, beside the
    // stencil and hysteresis kernels, and the first stream collects the results.  When the stencil pass makes the
    // gray plane, the FFT pair follows it on the same stream, so that the big kernels of this chunk and the next run
    // back to back with no event between them; the hysteresis passes, which leave the device mostly idle, run on the
    // second stream beside the row pass, and so does the collection of the results, beside the next chunk's stencil
    // pass.
    // A raw chunk's floods and suppression start there too.  (While the row pass was bound by its half-line stores they
    // waited for it to finish: beside it the first flood took 0.3 ms and cost it 0.04 ms.  Storing whole lines, the row
    // pass gives up 0.012 ms for them and the column pass gets 0.034 ms back: 1.231 -> 1.216 ms per clip.)
    auto queue_side = [&]() -> int {
        hipStream_t hs = fused_flow ? side : s;
        if (fused_flow) HIPCHK(c, hipStreamWaitEvent(side, sl.stencil_done, 0));
        if (sl.raw) {
            // flood of the raw maps (its own pass flags), suppression on what it reached; the ordinary passes then
            // start from pass 0 on the suppressed maps
            unsigned int* ch1 = (unsigned int*)sl.changed.p + HYST_MAX_PASSES;
            Span sp(c, &sl.spans, STX_K_HYST, RAW_GRID_PASSES + RAW_FINISH + 2, hs);
            for (int k = 0; k < RAW_GRID_PASSES; ++k)
                launch_hyst_pass(hs, (unsigned long long*)sl.strong.p, (const unsigned long long*)sl.cand.p, F, g, ch1, k,
                                 (uint32_t*)sl.tiles.p, (uint8_t*)sl.tflag.p);
            for (int k = RAW_GRID_PASSES; k < RAW_GRID_PASSES + RAW_FINISH; ++k)
                launch_hyst_finish(hs, (unsigned long long*)sl.strong.p, (const unsigned long long*)sl.cand.p, F, g, ch1, k,
                                   (uint32_t*)sl.tiles.p, (uint8_t*)sl.tflag.p);
            launch_nms_sparse(hs, (unsigned long long*)sl.strong.p, (unsigned long long*)sl.cand2.p, (const uint8_t*)sl.gray.p, F, g,
                              p.canny_low, p.canny_high, (uint32_t*)sl.tiles.p, (uint8_t*)sl.tflag.p, (uint32_t*)sl.nmslist.p,
                              (unsigned int*)sl.nmscount.p,
                              ch1 + RAW_GRID_PASSES + RAW_FINISH - 1, ch1 + RAW_REACHED);
            sp.end();
            sl.passes = RAW_SECOND_START;
        }
        if ((rc = hyst_passes(c, sl, sl.raw ? RAW_SECOND_PASSES : HYST_FIRST_PASSES, g, hs)) != STX_OK) return rc;
        return STX_OK;
    };
    if ((rc = queue_side()) != STX_OK) return rc;
    {
        hipStream_t sf = fused_gray ? s : side;
        if (sf != s) HIPCHK(c, hipStreamWaitEvent(sf, sl.gray_ready, 0));
        {
            Span sp(c, &sl.spans, STX_K_FFT_ROWS, 1, sf);
            launch_fft_rows(sf, (const uint8_t*)sl.gray.p, F, g, lay, pr.plan_w, pr.tw_w, (float2*)sl.spec.p);
            sp.end();
        }
        {
            Span sp(c, &sl.spans, STX_K_FFT_COLS, 1, sf);
            launch_fft_cols(sf, (const uint8_t*)sl.gray.p, (const float2*)sl.spec.p, F, g, lay, pr.plan_h, pr.tw_h,
                            (FftPart*)sl.part.p, fft_small_threshold(g.fh, g.fw));
            sp.end();
        }
        if (side != s) {
            HIPCHK(c, hipEventRecord(sl.fft_done, sf));
            HIPCHK(c, hipStreamWaitEvent(fused_flow ? side : s, sl.fft_done, 0));
        }
    }
    c->prev_chunk = &sl;
    return queue_results(c, sl, pr, fused_flow ? side : s);
}

// Means, deviations and ratios of the extended statistics from their exact integer sums.
void finalize_extended(const FrameAcc& a, const Geom& g, stx_extended* x) {
    memset(x, 0, sizeof(*x));
    const uint64_t n = (uint64_t)g.H * (uint64_t)g.W;
    const double dn = (double)n;
    x->sobel_sum = a.sobel_sum;
    x->sobel_sumsq = a.sobel_sumsq;
    x->sobel_strong = a.sobel_strong;
    for (int k = 0; k < STX_LBP_MAX_BINS; ++k) {
        x->lbp_hist_p16r2[k] = a.lbp_hist2[k];
        x->lbp_hist_p24r3[k] = a.lbp_hist3[k];
    }
    const uint64_t hb = (uint64_t)((g.W - 1) / 8), vb = (uint64_t)((g.H - 1) / 8);
    x->block_counts[0] = (uint64_t)g.H * hb;
    x->block_counts[1] = (uint64_t)g.H * (uint64_t)(g.W - 1) - x->block_counts[0];
    x->block_counts[2] = (uint64_t)g.W * vb;
    x->block_counts[3] = (uint64_t)g.W * (uint64_t)(g.H - 1) - x->block_counts[2];
    for (int k = 0; k < 4; ++k) x->block_sums[k] = a.blk[k];
    x->sobel_mean = (double)a.sobel_sum / dn;
    const unsigned __int128 num =
        (unsigned __int128)n * a.sobel_sumsq - (unsigned __int128)a.sobel_sum * (unsigned __int128)a.sobel_sum;
    x->sobel_std = sqrt((double)num / (dn * dn));
    x->strong_gradient_fraction = (double)a.sobel_strong / dn;
    x->edge_consistency = a.sobel_strong ? (double)a.edge_count / (double)a.sobel_strong : 0.0;
    const uint64_t ne = x->block_counts[0] + x->block_counts[2], ni = x->block_counts[1] + x->block_counts[3];
    const double me = ne ? (double)(a.blk[0] + a.blk[2]) / (double)ne : 0.0;
    const double mi = ni ? (double)(a.blk[1] + a.blk[3]) / (double)ni : 0.0;
    x->blockiness = me / (mi + 1e-10);
}

// Recomputes fft_sum / fft_sumsq of the listed frames of a finished chunk in double precision (the chunk's gray
// planes are still in the slot).  One frame at a time: this is the path of rare, synthetic-looking frames.
int na_twiddles(stx_ctx* c, int which, const FftPlan& plan);

int refine_fft64(stx_ctx* c, Slot& sl, const Prepared& pr, const int* frames, int n_redo) {
    const Geom& g = pr.g;
    Fft64Work w;
    fft64_sizes(g, pr.plan_w, pr.plan_h, &w.spec_bytes, &w.scratch_bytes, &w.part_bytes);
    int rc;
    void *tw_w = nullptr, *tw_h = nullptr;
    double2* part_host;
    int batch;  // frames per launch: as many as the workspace holds, at most FFT64_BATCH
    if (sl.gray.fixed) {
        // allocation-free path: the workspace was carved from the scratch array with the chunk's buffers (one
        // frame's worth), the tables have their own places in it, the column sums come back through the pinned staging
        w.spec = c->spec64.p; w.scratch = c->scratch64.p; w.part = c->part64.p;
        if (c->spec64.cap < w.spec_bytes || c->scratch64.cap < w.scratch_bytes || c->part64.cap < w.part_bytes)
            return fail(c, STX_ERR_NOMEM, "internal: fp64 workspace of the allocation-free path is too small");
        if ((rc = na_twiddles(c, 2, pr.plan_w)) != STX_OK || (rc = na_twiddles(c, 3, pr.plan_h)) != STX_OK) return rc;
        tw_w = c->na_base + NA_TW64_W;
        tw_h = c->na_base + NA_TW64_H;
        part_host = (double2*)c->na_stage;
        batch = 1;
    } else {
        const size_t per_frame = w.spec_bytes + w.scratch_bytes + w.part_bytes;
        batch = (int)std::min<size_t>((size_t)std::min(n_redo, FFT64_BATCH), std::max<size_t>(1, ((size_t)1 << 30) / per_frame));
        if ((rc = ensure(c, c->spec64, w.spec_bytes * batch)) != STX_OK) return rc;
        if (w.scratch_bytes && (rc = ensure(c, c->scratch64, w.scratch_bytes * batch)) != STX_OK) return rc;
        if ((rc = ensure(c, c->part64, w.part_bytes * batch)) != STX_OK) return rc;
        w.spec = c->spec64.p; w.scratch = c->scratch64.p; w.part = c->part64.p;
        if ((rc = get_twiddles64(c, pr.plan_w, &tw_w)) != STX_OK) return rc;
        if ((rc = get_twiddles64(c, pr.plan_h, &tw_h)) != STX_OK) return rc;
        try {
            c->part64_host.resize((size_t)g.fwh * n_redo);
        } catch (const std::bad_alloc&) {
            return fail(c, STX_ERR_NOMEM, "out of host memory");
        }
        part_host = c->part64_host.data();
    }
    // Every batch is queued before the host waits once (the batches reuse the device buffers in stream order, and each
    // one's column sums are copied to their own place on the host); the allocation-free path has staging for one
    // frame's sums and waits per frame.
    const bool per_batch_wait = sl.gray.fixed;
    auto sum_batch = [&](int i0, int nb, const double2* host) {
        for (int b = 0; b < nb; ++b) {
            const int f = frames[i0 + b];
            const double2* ph = host + (size_t)b * g.fwh;
            double s = 0.0, s2 = 0.0;  // fixed order: reproducible
            for (int k = 0; k < g.fwh; ++k) { s += ph[k].x; s2 += ph[k].y; }
            sl.acc_host[f].fft_sum = s;
            sl.acc_host[f].fft_sumsq = s2;
            sl.acc_host[f].pad0 = 1u;
        }
    };
    for (int i0 = 0; i0 < n_redo; i0 += batch) {
        const int nb = std::min(batch, n_redo - i0);
        launch_fft64(c->stream, (const uint8_t*)sl.gray.p, frames + i0, nb, g, pr.plan_w, pr.plan_h, tw_w, tw_h, w);
        double2* dst = per_batch_wait ? part_host : part_host + (size_t)i0 * g.fwh;
        HIPCHK(c, hipMemcpyAsync(dst, w.part, w.part_bytes * nb, hipMemcpyDeviceToHost, c->stream));  // part_bytes = fwh double2
        if (per_batch_wait) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipGetLastError());
            sum_batch(i0, nb, part_host);
        }
    }
    if (!per_batch_wait) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        for (int i0 = 0; i0 < n_redo; i0 += batch) sum_batch(i0, std::min(batch, n_redo - i0), part_host + (size_t)i0 * g.fwh);
    }
    c->fft64_frames += (uint64_t)n_redo;
    return STX_OK;
}

// Waits for an event by polling first, for a bounded time: a call on a few frames is over in a fraction of a
// millisecond, and the wake-up of a sleeping wait costs tens of microseconds that the next call's first kernel then
// starts late by.  Longer waits (a clip takes a millisecond or more) go to sleep, so that a rank that waits for its
// GPU leaves its core to the process that decodes the next frames.  SIFT_TEXTURE_SPIN_US sets the polling time.
int g_spin_us = -1;
hipError_t wait_event(hipEvent_t ev) {
    if (g_spin_us < 0) {
        const char* e = getenv("SIFT_TEXTURE_SPIN_US");
        const long v = e ? strtol(e, nullptr, 10) : 150;
        g_spin_us = (int)std::min<long>(std::max<long>(v, 0), 1000000);
    }
    if (g_spin_us > 0) {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(g_spin_us);
        do {
            for (int i = 0; i < 16; ++i) {
                const hipError_t e = hipEventQuery(ev);
                if (e != hipErrorNotReady) return e;
            }
        } while (std::chrono::steady_clock::now() < until);
    }
    return hipEventSynchronize(ev);
}

int finish_chunk(stx_ctx* c, Slot& sl, const Prepared& pr, stx_metrics* out, stx_extended* ext = nullptr) {
    HIPCHK(c, wait_event(sl.done));
    if (sl.raw) {
        if (sl.changed_host[1] != 0) {
            // rare: the flood of the raw maps was still growing when the suppression ran on it (contours hundreds of
            // tiles long): the chunk goes through again with the suppression in the stencil pass
            ChunkOpts o;
            o.codes = sl.src_codes;
            o.force_dense = true;
            ++c->raw_requeued;
            if (c->canny_mode == STX_CANNY_AUTO) {  // and the next chunks are dense; the next try comes twice as late
                c->raw_next = false;
                c->dense_run = 0;
                c->raw_probe = std::min(c->raw_probe * 2, RAW_PROBE_MAX);
            }
            sl.spans.swap(c->spans_kept);  // the first attempt's timing spans stay counted
            const int rc = enqueue_chunk(c, sl, sl.src_bgr, sl.src_row_stride, sl.src_frame_stride, sl.F, pr, o);
            if (rc != STX_OK) return rc;
            sl.spans.insert(sl.spans.end(), c->spans_kept.begin(), c->spans_kept.end());
            c->spans_kept.clear();
            HIPCHK(c, wait_event(sl.done));
        } else if (c->canny_mode == STX_CANNY_AUTO) {
            const double px = (double)sl.F * (double)pr.g.H * (double)pr.g.W;
            if ((double)sl.changed_host[1 + RAW_REACHED - (RAW_GRID_PASSES + RAW_FINISH - 1)] > RAW_MAX_DENSITY * px) {
                c->raw_next = false;
                c->dense_run = 0;
                c->raw_probe = std::min(c->raw_probe * 2, RAW_PROBE_MAX);
            } else {
                c->raw_probe = RAW_PROBE;
            }
        }
    }
    while (*sl.changed_host != 0) {
        // rare: long contours still growing after the first batch of passes
        int rc;
        const int more = HYST_GROUP;
        if (sl.passes + more > HYST_MAX_PASSES) return fail(c, STX_ERR_UNSUPPORTED, "Canny hysteresis did not converge");
        HIPCHK(c, hipMemsetAsync((unsigned int*)sl.changed.p + sl.passes, 0, (size_t)more * sizeof(unsigned int), c->stream));
        if ((rc = hyst_passes(c, sl, more, pr.g)) != STX_OK) return rc;
        if ((rc = queue_results(c, sl, pr)) != STX_OK) return rc;
        HIPCHK(c, wait_event(sl.done));
    }
    HIPCHK(c, hipGetLastError());
    // spans recorded at enqueue time are collected even if timing has been switched off since (their events go
    // back to the pool)
    if (!sl.spans.empty()) {
        collect_spans(c, sl.spans);
        c->tacc.frames += (uint64_t)sl.F;
    }
    if (out && pr.p.fft_exact >= 0) {
        sl.redo.clear();  // (the allocation-free path's list is reserved at stx_create)
        for (int i = 0; i < sl.F; ++i)
            if (pr.p.fft_exact > 0 || fft_needs_f64(sl.acc_host[i], pr.g)) sl.redo.push_back(i);
        if (!sl.redo.empty()) {
            const int rc = refine_fft64(c, sl, pr, sl.redo.data(), (int)sl.redo.size());
            if (rc != STX_OK) return rc;
        }
    }
    if (out)
        for (int i = 0; i < sl.F; ++i) {
            finalize(sl.acc_host[i], pr.g, pr.p.lbp_points, out + i);
            if (pr.p.quality_stats) finalize_quality(sl.acc_host[i], pr.g, out + i);
            if (ext) finalize_extended(sl.acc_host[i], pr.g, ext + i);
        }
    return STX_OK;
}

int analyze_device_impl(stx_ctx* c, const uint8_t* bgr_dev, int n_frames, int64_t row_stride, int64_t frame_stride,
                        const Prepared& pr, stx_metrics* out, stx_extended* ext = nullptr) {
    const int chunk = pick_chunk(c, pr.g, n_frames);
    int prev_slot = -1, prev_f0 = 0, k = 0;
    for (int f0 = 0; f0 < n_frames; f0 += chunk, ++k) {
        const int F = std::min(chunk, n_frames - f0);
        Slot& sl = c->slot[k % N_SLOTS];
        ChunkOpts o;
        int rc = enqueue_chunk(c, sl, bgr_dev + (int64_t)f0 * frame_stride, row_stride, frame_stride, F, pr, o);
        if (rc == STX_OK && prev_slot >= 0) rc = finish_chunk(c, c->slot[prev_slot], pr, out + prev_f0, ext ? ext + prev_f0 : nullptr);
        if (rc != STX_OK) {
            sync_all(c);
            return rc;
        }
        prev_slot = k % N_SLOTS;
        prev_f0 = f0;
    }
    if (prev_slot >= 0) return finish_chunk(c, c->slot[prev_slot], pr, out + prev_f0, ext ? ext + prev_f0 : nullptr);
    return STX_OK;
}

int prepare_extended(stx_ctx* c, Prepared* pr) {
    pr->ext = true;
    if (lbp_offsets(16, 2.0, &pr->off2) != 0 || lbp_offsets(24, 3.0, &pr->off3) != 0) return fail(c, STX_ERR_INVALID, "bad lbp_points");
    return STX_OK;
}

}  // namespace

extern "C" {

const char* stx_version(void) { return "sift-texture-gfx950 0.1"; }

void stx_default_params(stx_params* p) {
    if (!p) return;
    p->lbp_points = 8;
    p->lbp_radius = 1.0;
    p->canny_low = 50;
    p->canny_high = 150;
    p->gray_shift = 15;
    p->fft_h = 0;
    p->fft_w = 0;
    p->quality_stats = 0;
    p->fft_exact = 0;
}

static thread_local std::string g_create_err = "null context";

// events that only order this context's own streams on the device: no timing, and a release at device scope (the
// data they guard is read by kernels, never by the host)
static const unsigned INNER_EVENT = hipEventDisableTiming | hipEventReleaseToDevice;

int stx_create(int device_id, stx_ctx** out) {
    if (!out) return STX_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) {
        g_create_err = std::string("hipGetDeviceCount: ") + (e != hipSuccess ? hipGetErrorString(e) : "no devices");
        return STX_ERR_HIP;
    }
    if (device_id < 0 || device_id >= ndev) {
        g_create_err = "device id out of range";
        return STX_ERR_INVALID;
    }
    stx_ctx* c = new (std::nothrow) stx_ctx();
    if (!c) return STX_ERR_NOMEM;
    c->device = device_id;
    memset(&c->tacc, 0, sizeof(c->tacc));
    {
        const char* sep = getenv("SIFT_TEXTURE_SEPARATE_GRAY");
        c->fuse_gray = !(sep && sep[0] && sep[0] != '0');
        const char* cm = getenv("SIFT_TEXTURE_CANNY");
        if (cm && !strcmp(cm, "dense")) c->canny_mode = STX_CANNY_DENSE;
        else if (cm && !strcmp(cm, "sparse")) c->canny_mode = STX_CANNY_SPARSE;
    }
    const char* step = "hipSetDevice";
    e = hipSetDevice(device_id);
    if (e == hipSuccess) { step = "hipStreamCreateWithFlags"; e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream_pre, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream_fft, hipStreamNonBlocking);
    for (int i = 0; i < N_SLOTS && e == hipSuccess; ++i) {
        step = "hipHostMalloc";
        e = hipHostMalloc((void**)&c->slot[i].changed_host, sizeof(unsigned int) * 16, hipHostMallocDefault);
        if (e == hipSuccess) {
            *c->slot[i].changed_host = 0;
            step = "hipEventCreate";
            e = hipEventCreateWithFlags(&c->slot[i].done, hipEventDisableTiming);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->slot[i].gray_ready, INNER_EVENT);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->slot[i].fft_done, INNER_EVENT);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->slot[i].stencil_done, INNER_EVENT);
    }
    if (e == hipSuccess) { step = "hipFuncSetAttribute(MaxDynamicSharedMemorySize)"; e = fft_set_lds_limits(); }
    // the allocation-free entry point: this GPU's copy of the scratch array, result and twiddle staging, events
    if (e == hipSuccess) {
        step = "hipGetSymbolAddress(scratch)";
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        c->na_base = (uint8_t*)p;
    }
    if (e == hipSuccess) {
        step = "hipHostMalloc";
        e = hipHostMalloc((void**)&c->na.acc_host, sizeof(FrameAcc) * (size_t)NA_MAX_CHUNK, hipHostMallocDefault);
        if (e == hipSuccess) c->na.acc_host_cap = NA_MAX_CHUNK;
    }
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->na.changed_host, sizeof(unsigned int) * 16, hipHostMallocDefault);
    if (e == hipSuccess) {
        *c->na.changed_host = 0;
        e = hipHostMalloc(&c->na_stage, (sizeof(double2) + sizeof(float2)) * FFT_TABLE_MAX + sizeof(double2) * 2 * STX_MAX_DIM,
                          hipHostMallocDefault);
    }
    if (e == hipSuccess) { step = "hipEventCreate"; e = hipEventCreateWithFlags(&c->na.done, hipEventDisableTiming); }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->na.gray_ready, INNER_EVENT);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->na.fft_done, INNER_EVENT);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->na.stencil_done, INNER_EVENT);
    if (e == hipSuccess) {
        try {
            c->na.redo.reserve(NA_MAX_CHUNK);
        } catch (const std::bad_alloc&) {
            stx_destroy(c);
            return STX_ERR_NOMEM;
        }
        // the flat-neighbourhood tables of the default LBP, so that no later call has to build them
        LbpOffsets off;
        if (lbp_offsets(8, 1.0, &off) != 0 || ensure_tie_tables(c, off) != STX_OK) {
            g_create_err = c->err;
            stx_destroy(c);
            return STX_ERR_HIP;
        }
    }
    if (e != hipSuccess) {
        g_create_err = std::string(step) + ": " + hipGetErrorString(e);
        stx_destroy(c);
        return STX_ERR_HIP;
    }
    *out = c;
    return STX_OK;
}

int stx_destroy(stx_ctx* c) {
    if (!c) return STX_ERR_INVALID;
    (void)hipSetDevice(c->device);
    sync_all(c);
    for (Slot& sl : c->slot) {
        DevBuf* bufs[] = {&sl.gray, &sl.strong, &sl.cand, &sl.spec, &sl.part, &sl.tiles, &sl.tflag, &sl.spart, &sl.lpart, &sl.qpart, &sl.xpart, &sl.acc, &sl.changed, &sl.staging, &sl.nmslist, &sl.nmscount, &sl.cand2};
        for (DevBuf* b : bufs)
            if (b->p) (void)hipFree(b->p);
        if (sl.acc_host) (void)hipHostFree(sl.acc_host);
        if (sl.changed_host) (void)hipHostFree(sl.changed_host);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.gray_ready) (void)hipEventDestroy(sl.gray_ready);
        if (sl.fft_done) (void)hipEventDestroy(sl.fft_done);
        if (sl.stencil_done) (void)hipEventDestroy(sl.stencil_done);
        for (auto& t : sl.spans) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
    }
    if (c->na.acc_host) (void)hipHostFree(c->na.acc_host);
    if (c->na.changed_host) (void)hipHostFree(c->na.changed_host);
    if (c->na_stage) (void)hipHostFree(c->na_stage);
    if (c->na.done) (void)hipEventDestroy(c->na.done);
    if (c->na.gray_ready) (void)hipEventDestroy(c->na.gray_ready);
    if (c->na.fft_done) (void)hipEventDestroy(c->na.fft_done);
    if (c->na.stencil_done) (void)hipEventDestroy(c->na.stencil_done);
    for (auto& t : c->na.spans) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
    // (the buffers of c->na and, after an allocation-free call, spec64 / scratch64 / part64 point into the scratch array)
    if (c->spec64.fixed) c->spec64 = DevBuf();
    if (c->scratch64.fixed) c->scratch64 = DevBuf();
    if (c->part64.fixed) c->part64 = DevBuf();
    if (c->codes.p) (void)hipFree(c->codes.p);
    if (c->plane.p) (void)hipFree(c->plane.p);
    if (c->taps.p) (void)hipFree(c->taps.p);
    if (c->tie.p) (void)hipFree(c->tie.p);
    if (c->spec64.p) (void)hipFree(c->spec64.p);
    if (c->scratch64.p) (void)hipFree(c->scratch64.p);
    if (c->part64.p) (void)hipFree(c->part64.p);
    for (auto& kv : c->twiddles64) (void)hipFree(kv.second);
    if (c->prep_out.p) (void)hipFree(c->prep_out.p);
    for (auto& kv : c->twiddles) (void)hipFree(kv.second);
    for (auto& t : c->synth_spans) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream_pre) (void)hipStreamDestroy(c->stream_pre);
    if (c->stream_fft) (void)hipStreamDestroy(c->stream_fft);
    delete c;
    return STX_OK;
}

const char* stx_last_error(const stx_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int stx_backend(const stx_ctx* c) { return c ? 1 : 0; }

int stx_analyze_batch_device(stx_ctx* c, const uint8_t* bgr_dev, int n_frames, int h, int w, int64_t row_stride,
                             int64_t frame_stride, const stx_params* params, stx_metrics* out) {
    if (!c) return STX_ERR_INVALID;
    if (n_frames < 0) return fail(c, STX_ERR_INVALID, "n_frames < 0");
    if (n_frames == 0) return STX_OK;
    if (!bgr_dev || !out) return fail(c, STX_ERR_INVALID, "null pointer");
    if (row_stride < (int64_t)3 * w || frame_stride < row_stride * (int64_t)(h - 1) + 3 * (int64_t)w)
        return fail(c, STX_ERR_INVALID, "strides smaller than the frame");
    if (any_busy(c)) return fail(c, STX_ERR_INVALID, "asynchronous submissions are outstanding: call stx_wait first");
    Prepared pr;
    int rc = prepare(c, h, w, params, &pr);
    if (rc != STX_OK) return rc;
    return analyze_device_impl(c, bgr_dev, n_frames, row_stride, frame_stride, pr, out);
}

}  // extern "C"

namespace {

// Twiddle table `which` (0: fp32 rows, 1: fp32 columns, 2: fp64 rows, 3: fp64 columns) of length n in its place in
// the scratch array; uploaded through the pinned staging only when the length changes.
int na_twiddles(stx_ctx* c, int which, const FftPlan& plan) {
    const int n = plan.n;
    // (the caller holds this GPU's busy flag, so nobody else reads or writes the record meanwhile)
    int* tw_n = g_na_device[c->device].tw_n;
    if (tw_n[which] == n) return STX_OK;
    static const size_t off[4] = {NA_TW_W, NA_TW_H, NA_TW64_W, NA_TW64_H};
    const size_t entries = fft_table_entries(plan);
    size_t bytes;
    const void* src = c->na_stage;
    // filled in double precision in the first part of the staging; the fp32 tables are narrowed into the second; a
    // chirp-z table is built in the third
    fft_fill_table(plan, (double*)c->na_stage,
                   (double*)((char*)c->na_stage + (sizeof(double2) + sizeof(float2)) * FFT_TABLE_MAX));
    if (which < 2) {
        const double* t = (const double*)c->na_stage;
        float2* h = (float2*)((char*)c->na_stage + sizeof(double2) * FFT_TABLE_MAX);
        for (size_t k = 0; k < entries; ++k) h[k] = make_float2((float)t[2 * k], (float)t[2 * k + 1]);
        bytes = sizeof(float2) * entries;
        src = h;
    } else {
        bytes = sizeof(double2) * entries;
    }
    tw_n[which] = 0;
    HIPCHK(c, hipMemcpyAsync(c->na_base + off[which], src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the staging is reused by the next table
    tw_n[which] = n;
    return STX_OK;
}

// Points the buffers of the allocation-free slot (and the fp64 workspace) at pieces of the scratch array for a chunk
// of F frames; returns the bytes used, or 0 if they do not fit.
size_t na_carve(stx_ctx* c, const Prepared& pr, int F) {
    size_t need[CB_COUNT];
    chunk_sizes(pr, F, need);
    size_t at = NA_BUFFERS;
    auto take = [&](DevBuf& b, size_t bytes) {
        b.p = bytes ? c->na_base + at : nullptr;
        b.cap = bytes;
        b.fixed = true;
        at += (bytes + 255) & ~(size_t)255;
    };
    for (int i = 0; i < CB_COUNT; ++i) take(*chunk_buf(c->na, i), need[i]);
    if (pr.p.fft_exact >= 0) {
        size_t sb, cb, pb;
        fft64_sizes(pr.g, pr.plan_w, pr.plan_h, &sb, &cb, &pb);
        take(c->spec64, sb);
        take(c->scratch64, cb);
        take(c->part64, pb);
    }
    return at <= (size_t)STX_NOALLOC_SCRATCH_BYTES ? at : 0;
}

}  // namespace

extern "C" {

int stx_analyze_batch_device_noalloc(stx_ctx* c, const uint8_t* bgr_dev, int n_frames, int h, int w, int64_t row_stride,
                                     int64_t frame_stride, const stx_params* params, stx_metrics* out) {
    if (!c) return STX_ERR_INVALID;
    if (n_frames < 0) return fail(c, STX_ERR_INVALID, "n_frames < 0");
    if (n_frames == 0) return STX_OK;
    if (!bgr_dev || !out) return fail(c, STX_ERR_INVALID, "null pointer");
    if (row_stride < (int64_t)3 * w || frame_stride < row_stride * (int64_t)(h - 1) + 3 * (int64_t)w)
        return fail(c, STX_ERR_INVALID, "strides smaller than the frame");
    if (any_busy(c)) return fail(c, STX_ERR_INVALID, "asynchronous submissions are outstanding: call stx_wait first");
    if (c->timing) return fail(c, STX_ERR_INVALID, "switch kernel timing off for the allocation-free entry point");
    if (params && params->quality_stats) return fail(c, STX_ERR_UNSUPPORTED, "quality_stats is not available on the allocation-free entry point");
    Prepared pr;
    int rc = prepare(c, h, w, params, &pr, false);
    if (rc != STX_OK) return rc;
    if (!(pr.p.lbp_points == 8 && pr.p.lbp_radius == 1.0))
        return fail(c, STX_ERR_UNSUPPORTED, "the allocation-free entry point computes LBP at (P=8, R=1) only");
    if (c->device >= NA_MAX_DEVICES) return fail(c, STX_ERR_UNSUPPORTED, "device id beyond the allocation-free entry point's table");
    // one call at a time per GPU: every context of this GPU works in the same scratch array
    NaDevice& nd = g_na_device[c->device];
    {
        std::lock_guard<std::mutex> lk(nd.m);
        if (nd.busy)
            return fail(c, STX_ERR_INVALID, "another allocation-free call is running on this GPU (its contexts share one scratch array)");
        nd.busy = true;
    }
    // the fp64 workspace of spec64 / scratch64 / part64 may belong to the allocating entry points: park it
    DevBuf keep[3] = {c->spec64, c->scratch64, c->part64};
    auto restore = [&]() {
        if (!keep[0].fixed) c->spec64 = keep[0]; else c->spec64 = DevBuf();
        if (!keep[1].fixed) c->scratch64 = keep[1]; else c->scratch64 = DevBuf();
        if (!keep[2].fixed) c->part64 = keep[2]; else c->part64 = DevBuf();
        std::lock_guard<std::mutex> lk(nd.m);
        nd.busy = false;
    };
    int chunk = std::min(n_frames, NA_MAX_CHUNK);
    while (chunk >= 1 && na_carve(c, pr, chunk) == 0) chunk = chunk > 8 ? chunk - chunk / 8 : chunk - 1;
    if (chunk < 1) {
        restore();
        return fail(c, STX_ERR_UNSUPPORTED, "frame too large for the static scratch array of the allocation-free entry point "
                                          "(gray plane, edge maps, half spectrum and, unless fft_exact is -1, one fp64 "
                                          "spectrum must fit 128 MiB)");
    }
    if ((rc = na_twiddles(c, 0, pr.plan_w)) == STX_OK) rc = na_twiddles(c, 1, pr.plan_h);
    pr.tw_w = (float2*)(c->na_base + NA_TW_W);
    pr.tw_h = (float2*)(c->na_base + NA_TW_H);
    for (int f0 = 0; f0 < n_frames && rc == STX_OK; f0 += chunk) {
        const int F = std::min(chunk, n_frames - f0);
        ChunkOpts o;
        rc = enqueue_chunk(c, c->na, bgr_dev + (int64_t)f0 * frame_stride, row_stride, frame_stride, F, pr, o);
        if (rc == STX_OK) rc = finish_chunk(c, c->na, pr, out + f0);
    }
    if (rc != STX_OK) sync_all(c);
    restore();
    return rc;
}

int stx_analyze_extended_batch_device(stx_ctx* c, const uint8_t* bgr_dev, int n_frames, int h, int w, int64_t row_stride,
                                      int64_t frame_stride, const stx_params* params, stx_metrics* out, stx_extended* ext) {
    if (!c) return STX_ERR_INVALID;
    if (n_frames < 0) return fail(c, STX_ERR_INVALID, "n_frames < 0");
    if (n_frames == 0) return STX_OK;
    if (!bgr_dev || !out || !ext) return fail(c, STX_ERR_INVALID, "null pointer");
    if (row_stride < (int64_t)3 * w || frame_stride < row_stride * (int64_t)(h - 1) + 3 * (int64_t)w)
        return fail(c, STX_ERR_INVALID, "strides smaller than the frame");
    if (any_busy(c)) return fail(c, STX_ERR_INVALID, "asynchronous submissions are outstanding: call stx_wait first");
    Prepared pr;
    int rc = prepare(c, h, w, params, &pr);
    if (rc != STX_OK) return rc;
    if ((rc = prepare_extended(c, &pr)) != STX_OK) return rc;
    return analyze_device_impl(c, bgr_dev, n_frames, row_stride, frame_stride, pr, out, ext);
}

int stx_max_async_frames(stx_ctx* c, int h, int w) {
    if (!c) return STX_ERR_INVALID;
    stx_params p;
    stx_default_params(&p);
    Geom g;
    if (make_geom(c, h, w, p, &g) != STX_OK) return 0;
    return pick_chunk(c, g, 1 << 30);
}

static int submit_common(stx_ctx* c, const uint8_t* bgr_dev, int n_frames, int h, int w, int64_t row_stride,
                         int64_t frame_stride, const stx_params* params, stx_metrics* out, stx_extended* ext, bool want_ext,
                         int* ticket) {
    if (!c) return STX_ERR_INVALID;
    if (!ticket || !bgr_dev || !out || n_frames < 1 || (want_ext && !ext)) return fail(c, STX_ERR_INVALID, "bad argument");
    if (row_stride < (int64_t)3 * w || frame_stride < row_stride * (int64_t)(h - 1) + 3 * (int64_t)w)
        return fail(c, STX_ERR_INVALID, "strides smaller than the frame");
    int si = -1;
    for (int i = 0; i < N_SLOTS; ++i)
        if (!c->slot[i].busy) { si = i; break; }
    if (si < 0) return fail(c, STX_ERR_INVALID, "too many submissions in flight: call stx_wait first");
    Slot& sl = c->slot[si];
    sl.async_pr = Prepared();
    int rc = prepare(c, h, w, params, &sl.async_pr);
    if (rc != STX_OK) return rc;
    if (want_ext && (rc = prepare_extended(c, &sl.async_pr)) != STX_OK) return rc;
    if (n_frames > pick_chunk(c, sl.async_pr.g, 1 << 30))
        return fail(c, STX_ERR_UNSUPPORTED, "n_frames exceeds stx_max_async_frames for this frame size");
    ChunkOpts o;
    rc = enqueue_chunk(c, sl, bgr_dev, row_stride, frame_stride, n_frames, sl.async_pr, o);
    if (rc != STX_OK) { sync_all(c); return rc; }
    sl.busy = true;
    sl.async_out = out;
    sl.async_ext = want_ext ? ext : nullptr;
    *ticket = si;
    return STX_OK;
}

int stx_submit_batch_device(stx_ctx* c, const uint8_t* bgr_dev, int n_frames, int h, int w, int64_t row_stride,
                            int64_t frame_stride, const stx_params* params, stx_metrics* out, int* ticket) {
    return submit_common(c, bgr_dev, n_frames, h, w, row_stride, frame_stride, params, out, nullptr, false, ticket);
}

int stx_submit_extended_batch_device(stx_ctx* c, const uint8_t* bgr_dev, int n_frames, int h, int w, int64_t row_stride,
                                     int64_t frame_stride, const stx_params* params, stx_metrics* out, stx_extended* ext,
                                     int* ticket) {
    return submit_common(c, bgr_dev, n_frames, h, w, row_stride, frame_stride, params, out, ext, true, ticket);
}

int stx_wait(stx_ctx* c, int ticket) {
    if (!c) return STX_ERR_INVALID;
    if (ticket < 0 || ticket >= N_SLOTS || !c->slot[ticket].busy) return fail(c, STX_ERR_INVALID, "unknown ticket");
    Slot& sl = c->slot[ticket];
    const int rc = finish_chunk(c, sl, sl.async_pr, sl.async_out, sl.async_ext);
    sl.busy = false;
    sl.async_out = nullptr;
    sl.async_ext = nullptr;
    if (rc != STX_OK) sync_all(c);
    return rc;
}

static int analyze_host_common(stx_ctx* c, const uint8_t* base, const uint8_t* const* list, int n_frames, int h, int w,
                               int64_t row_stride, int64_t frame_stride, const stx_params* params, stx_metrics* out,
                               stx_extended* ext = nullptr) {
    if (!c) return STX_ERR_INVALID;
    if (n_frames < 0) return fail(c, STX_ERR_INVALID, "n_frames < 0");
    if (n_frames == 0) return STX_OK;
    if ((!base && !list) || !out) return fail(c, STX_ERR_INVALID, "null pointer");
    if (any_busy(c)) return fail(c, STX_ERR_INVALID, "asynchronous submissions are outstanding: call stx_wait first");
    if (row_stride < (int64_t)3 * w) return fail(c, STX_ERR_INVALID, "row stride smaller than 3*w");
    Prepared pr;
    int rc = prepare(c, h, w, params, &pr);
    if (rc != STX_OK) return rc;
    if (ext && (rc = prepare_extended(c, &pr)) != STX_OK) return rc;
    // device staging: rows padded to a multiple of 4 bytes so the gray kernel's dword path applies
    const int64_t dpitch = ((int64_t)3 * w + 3) & ~(int64_t)3;
    const int64_t dframe = dpitch * h;
    const int chunk = pick_chunk(c, pr.g, n_frames);
    const bool dense = !list && row_stride == (int64_t)3 * w && dpitch == row_stride && frame_stride == dframe;
    int prev_slot = -1, prev_f0 = 0, k = 0;
    for (int f0 = 0; f0 < n_frames; f0 += chunk, ++k) {
        const int F = std::min(chunk, n_frames - f0);
        Slot& sl = c->slot[k % N_SLOTS];
        if ((rc = ensure(c, sl.staging, (size_t)chunk * dframe)) != STX_OK) break;
        if (dense) {
            rc = STX_OK;
            hipError_t e = hipMemcpyAsync(sl.staging.p, base + (int64_t)f0 * frame_stride, (size_t)F * dframe,
                                          hipMemcpyHostToDevice, c->concurrent_fft ? c->stream_pre : c->stream);
            if (e != hipSuccess) rc = fail(c, STX_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
        } else if (row_stride != dpitch && (size_t)F * dframe <= ((size_t)512 << 20)) {
            // strided rows (a width that is not a multiple of four, or a view): pack them on the host, then one
            // linear copy -- the runtime's 2-D copy out of pageable memory goes row by row
            rc = STX_OK;
            try {
                sl.pack.resize((size_t)F * dframe);
            } catch (const std::bad_alloc&) {
                rc = fail(c, STX_ERR_NOMEM, "out of host memory");
            }
            for (int i = 0; i < F && rc == STX_OK; ++i) {
                const uint8_t* src = list ? list[f0 + i] : base + (int64_t)(f0 + i) * frame_stride;
                if (!src) { rc = fail(c, STX_ERR_INVALID, "null frame pointer"); break; }
                uint8_t* dst = sl.pack.data() + (size_t)i * dframe;
                for (int y = 0; y < h; ++y) memcpy(dst + (size_t)y * dpitch, src + (int64_t)y * row_stride, (size_t)3 * w);
            }
            if (rc == STX_OK) {
                hipError_t e = hipMemcpyAsync(sl.staging.p, sl.pack.data(), (size_t)F * dframe, hipMemcpyHostToDevice,
                                              c->concurrent_fft ? c->stream_pre : c->stream);
                if (e != hipSuccess) rc = fail(c, STX_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
            }
        } else {
            for (int i = 0; i < F && rc == STX_OK; ++i) {
                const uint8_t* src = list ? list[f0 + i] : base + (int64_t)(f0 + i) * frame_stride;
                if (!src) { rc = fail(c, STX_ERR_INVALID, "null frame pointer"); break; }
                hipError_t e = upload_frame((uint8_t*)sl.staging.p + (int64_t)i * dframe, dpitch, src, row_stride, w, h,
                                            c->concurrent_fft ? c->stream_pre : c->stream);
                if (e != hipSuccess) rc = fail(c, STX_ERR_HIP, std::string("host-to-device frame copy: ") + hipGetErrorString(e));
            }
        }
        if (rc != STX_OK) break;
        ChunkOpts o;
        rc = enqueue_chunk(c, sl, (const uint8_t*)sl.staging.p, dpitch, dframe, F, pr, o);
        if (rc == STX_OK && prev_slot >= 0) rc = finish_chunk(c, c->slot[prev_slot], pr, out + prev_f0, ext ? ext + prev_f0 : nullptr);
        if (rc != STX_OK) break;
        prev_slot = k % N_SLOTS;
        prev_f0 = f0;
    }
    if (rc != STX_OK) {
        sync_all(c);
        return rc;
    }
    if (prev_slot >= 0) return finish_chunk(c, c->slot[prev_slot], pr, out + prev_f0, ext ? ext + prev_f0 : nullptr);
    return STX_OK;
}

int stx_analyze_extended_frames(stx_ctx* c, const uint8_t* const* frames, int n_frames, int h, int w, int64_t row_stride,
                                const stx_params* params, stx_metrics* out, stx_extended* ext) {
    if (c && n_frames > 0 && !ext) return fail(c, STX_ERR_INVALID, "null pointer");
    return analyze_host_common(c, nullptr, frames, n_frames, h, w, row_stride, 0, params, out, ext);
}

int stx_analyze_batch(stx_ctx* c, const uint8_t* bgr, int n_frames, int h, int w, int64_t row_stride,
                      int64_t frame_stride, const stx_params* params, stx_metrics* out) {
    if (c && n_frames > 1 && frame_stride < row_stride * (int64_t)(h - 1) + 3 * (int64_t)w)
        return fail(c, STX_ERR_INVALID, "frame stride smaller than the frame");
    return analyze_host_common(c, bgr, nullptr, n_frames, h, w, row_stride, frame_stride, params, out);
}

int stx_analyze_frames(stx_ctx* c, const uint8_t* const* frames, int n_frames, int h, int w, int64_t row_stride,
                       const stx_params* params, stx_metrics* out) {
    return analyze_host_common(c, nullptr, frames, n_frames, h, w, row_stride, 0, params, out);
}

int stx_score_summary(const stx_metrics* m, int n, double threshold, double* average, int* passed) {
    if (n < 0 || (n > 0 && !m) || !average || !passed) return STX_ERR_INVALID;
    // numpy.mean: pairwise summation as numpy's add.reduce does it (eight running partial
    // sums below 128 elements, halving above), so the average matches np.mean(scores) at
    // passive_checker.py:155 bit for bit; 0.0 for an empty list
    std::vector<double> v((size_t)n);
    for (int i = 0; i < n; ++i) v[i] = m[i].texture_score;
    struct PW {
        static double sum(const double* a, int n) {
            if (n < 8) {
                double r = 0.0;
                for (int i = 0; i < n; ++i) r += a[i];
                return r;
            }
            if (n <= 128) {
                double r[8];
                for (int j = 0; j < 8; ++j) r[j] = a[j];
                int i = 8;
                for (; i < n - (n % 8); i += 8)
                    for (int j = 0; j < 8; ++j) r[j] += a[i + j];
                double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (; i < n; ++i) res += a[i];
                return res;
            }
            int n2 = n / 2;
            n2 -= n2 % 8;
            return sum(a, n2) + sum(a + n2, n - n2);
        }
    };
    const double s = PW::sum(v.data(), n);
    *average = n ? s / (double)n : 0.0;
    *passed = (*average >= threshold) ? 1 : 0;
    return STX_OK;
}

int stx_debug_planes(stx_ctx* c, const uint8_t* bgr, int h, int w, int64_t row_stride, const stx_params* params,
                     uint8_t* gray_out, uint8_t* lbp_codes_out, uint8_t* edges_out) {
    if (!c) return STX_ERR_INVALID;
    if (!bgr) return fail(c, STX_ERR_INVALID, "null pointer");
    if (row_stride < (int64_t)3 * w) return fail(c, STX_ERR_INVALID, "row stride smaller than 3*w");
    if (any_busy(c)) return fail(c, STX_ERR_INVALID, "asynchronous submissions are outstanding: call stx_wait first");
    Prepared pr;
    int rc = prepare(c, h, w, params, &pr);
    if (rc != STX_OK) return rc;
    const int64_t dpitch = ((int64_t)3 * w + 3) & ~(int64_t)3;
    Slot& sl = c->slot[0];
    if ((rc = ensure(c, sl.staging, (size_t)dpitch * h)) != STX_OK) return rc;
    if ((rc = ensure(c, c->codes, (size_t)h * w)) != STX_OK) return rc;
    if ((rc = ensure(c, c->plane, (size_t)h * w)) != STX_OK) return rc;
    HIPCHK(c, upload_frame((uint8_t*)sl.staging.p, dpitch, bgr, row_stride, w, h, c->concurrent_fft ? c->stream_pre : c->stream));
    ChunkOpts o;
    o.codes = (uint8_t*)c->codes.p;
    if ((rc = enqueue_chunk(c, sl, (const uint8_t*)sl.staging.p, dpitch, dpitch * h, 1, pr, o)) != STX_OK) return rc;
    if ((rc = finish_chunk(c, sl, pr, nullptr)) != STX_OK) return rc;
    if (gray_out)
        HIPCHK(c, hipMemcpy2DAsync(gray_out, (size_t)w, (const uint8_t*)sl.gray.p + (size_t)GRAY_YOFF * pr.g.gpitch + GRAY_XOFF,
                                   (size_t)pr.g.gpitch, (size_t)w, (size_t)h, hipMemcpyDeviceToHost, c->stream));
    if (lbp_codes_out)
        HIPCHK(c, hipMemcpyAsync(lbp_codes_out, c->codes.p, (size_t)h * w, hipMemcpyDeviceToHost, c->stream));
    if (edges_out) {
        launch_edges_expand(c->stream, (const unsigned long long*)sl.strong.p, pr.g, (uint8_t*)c->plane.p);
        HIPCHK(c, hipMemcpyAsync(edges_out, c->plane.p, (size_t)h * w, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return STX_OK;
}

// ---- frame preparation for the CNN branch (passive_checker.py:393-418)
static int preprocess_common(stx_ctx* c, const uint8_t* bgr_dev, int n_frames, int h, int w, int64_t row_stride,
                             int64_t frame_stride, int out_h, int out_w, float* out_dev) {
    std::vector<uint8_t> host(preprocess_tap_bytes(out_w) + preprocess_tap_bytes(out_h));
    preprocess_fill_taps(h, w, out_h, out_w, host.data(), host.data() + preprocess_tap_bytes(out_w));
    int rc;
    if ((rc = ensure(c, c->taps, host.size())) != STX_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(c->taps.p, host.data(), host.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // host is a local
    launch_preprocess(c->stream, bgr_dev, row_stride, frame_stride, n_frames, out_h, out_w, c->taps.p,
                      (const uint8_t*)c->taps.p + preprocess_tap_bytes(out_w), preprocess_is_area2(h, w, out_h, out_w),
                      out_dev);
    HIPCHK(c, hipGetLastError());
    return STX_OK;
}

static int preprocess_check(stx_ctx* c, int n_frames, int h, int w, int64_t row_stride, int out_h, int out_w) {
    if (n_frames < 0 || n_frames > 65535) return fail(c, STX_ERR_INVALID, "n_frames out of range");
    if (h < 1 || w < 1 || h > 8192 || w > 8192) return fail(c, STX_ERR_INVALID, "frame size out of range (1..8192)");
    if (out_h < 1 || out_w < 1 || out_h > 4096 || out_w > 4096) return fail(c, STX_ERR_INVALID, "output size out of range (1..4096)");
    if (row_stride < (int64_t)3 * w) return fail(c, STX_ERR_INVALID, "row stride smaller than 3*w");
    if (any_busy(c)) return fail(c, STX_ERR_INVALID, "asynchronous submissions are outstanding: call stx_wait first");
    return STX_OK;
}

int stx_preprocess_batch_device(stx_ctx* c, const uint8_t* bgr_dev, int n_frames, int h, int w, int64_t row_stride,
                                int64_t frame_stride, int out_h, int out_w, float* out_dev) {
    if (!c) return STX_ERR_INVALID;
    int rc = preprocess_check(c, n_frames, h, w, row_stride, out_h, out_w);
    if (rc != STX_OK) return rc;
    if (n_frames == 0) return STX_OK;
    if (!bgr_dev || !out_dev) return fail(c, STX_ERR_INVALID, "null pointer");
    if (n_frames > 1 && frame_stride < row_stride * (int64_t)(h - 1) + 3 * (int64_t)w)
        return fail(c, STX_ERR_INVALID, "frame stride smaller than the frame");
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = preprocess_common(c, bgr_dev, n_frames, h, w, row_stride, frame_stride, out_h, out_w, out_dev)) != STX_OK)
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return STX_OK;
}

int stx_preprocess_frames(stx_ctx* c, const uint8_t* const* frames, int n_frames, int h, int w, int64_t row_stride,
                          int out_h, int out_w, float* out_host) {
    if (!c) return STX_ERR_INVALID;
    int rc = preprocess_check(c, n_frames, h, w, row_stride, out_h, out_w);
    if (rc != STX_OK) return rc;
    if (n_frames == 0) return STX_OK;
    if (!frames || !out_host) return fail(c, STX_ERR_INVALID, "null pointer");
    for (int f = 0; f < n_frames; ++f)
        if (!frames[f]) return fail(c, STX_ERR_INVALID, "null frame pointer");
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t dpitch = ((int64_t)3 * w + 3) & ~(int64_t)3;
    const size_t plane = (size_t)3 * out_h * out_w;
    // frames go through the staging buffer a bounded number at a time
    const int per = std::max<int64_t>(1, std::min<int64_t>(n_frames, ((int64_t)256 << 20) / (dpitch * h)));
    Slot& sl = c->slot[0];
    if ((rc = ensure(c, sl.staging, (size_t)dpitch * h * per)) != STX_OK) return rc;
    if ((rc = ensure(c, c->prep_out, plane * per * sizeof(float))) != STX_OK) return rc;
    for (int f0 = 0; f0 < n_frames; f0 += per) {
        const int F = std::min(per, n_frames - f0);
        for (int f = 0; f < F; ++f)
            HIPCHK(c, upload_frame((uint8_t*)sl.staging.p + (size_t)f * dpitch * h, dpitch, frames[f0 + f], row_stride, w, h, c->stream));
        if ((rc = preprocess_common(c, (const uint8_t*)sl.staging.p, F, h, w, dpitch, dpitch * h, out_h, out_w,
                                    (float*)c->prep_out.p)) != STX_OK)
            return rc;
        HIPCHK(c, hipMemcpyAsync(out_host + (size_t)f0 * plane, c->prep_out.p, plane * F * sizeof(float),
                                 hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return STX_OK;
}

int stx_synth_device(stx_ctx* c, uint8_t* bgr_dev, int kind, uint64_t seed, uint32_t video, uint32_t frame0, int n_frames,
                     int h, int w) {
    if (!c) return STX_ERR_INVALID;
    if (!bgr_dev || n_frames < 0 || h < 1 || w < 1 || h > 8192 || w > 8192 || kind < 0 || kind > 5 ||
        (uint64_t)frame0 + (uint64_t)n_frames > 1024 || video >= (1u << 28) || n_frames > 65535)
        return fail(c, STX_ERR_INVALID, "synthetic generator limits: h,w<=8192, frame<1024, video<2^28");
    if (n_frames == 0) return STX_OK;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<TimedSpan> spans;
    Span sp(c, &spans, STX_K_SYNTH, 1);
    launch_synth(c->stream, bgr_dev, kind, seed, video, frame0, n_frames, h, w);
    sp.end();
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    collect_spans(c, spans);
    return STX_OK;
}

int stx_synth_device_async(stx_ctx* c, uint8_t* bgr_dev, int kind, uint64_t seed, uint32_t video, uint32_t frame0,
                           int n_frames, int h, int w) {
    if (!c) return STX_ERR_INVALID;
    if (!bgr_dev || n_frames < 0 || h < 1 || w < 1 || h > 8192 || w > 8192 || kind < 0 || kind > 5 ||
        (uint64_t)frame0 + (uint64_t)n_frames > 1024 || video >= (1u << 28) || n_frames > 65535)
        return fail(c, STX_ERR_INVALID, "synthetic generator limits: h,w<=8192, frame<1024, video<2^28");
    if (n_frames == 0) return STX_OK;
    HIPCHK(c, hipSetDevice(c->device));
    // the stream every analysis call queues its first work on (copies, clears, a separate gray pass), and whose
    // progress the stencil pass waits for: a call that follows reads finished frames, and until then the generator
    // runs beside whatever the device is busy with
    hipStream_t st = c->concurrent_fft ? c->stream_pre : c->stream;
    // The generator is bound by vector issue and by HBM stores; of the chunk in flight, the stencil pass is bound by
    // vector issue too, the FFT row pass by stores alone: the generator starts when that chunk's stencil pass ends
    // and runs beside its row pass.
    if (c->streaming && c->prev_chunk && st != c->stream)
        HIPCHK(c, hipStreamWaitEvent(st, c->prev_chunk->stencil_done, 0));
    c->streaming = true;
    Span sp(c, &c->synth_spans, STX_K_SYNTH, 1, st);
    launch_synth(st, bgr_dev, kind, seed, video, frame0, n_frames, h, w);
    sp.end();
    HIPCHK(c, hipGetLastError());
    return STX_OK;
}

int stx_device_alloc(stx_ctx* c, uint64_t bytes, void** out) {
    if (!c || !out) return STX_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipError_t e = hipMalloc(out, (size_t)bytes);
    if (e != hipSuccess) return fail(c, STX_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return STX_OK;
}
int stx_device_free(stx_ctx* c, void* p) {
    if (!c) return STX_ERR_INVALID;
    HIPCHK(c, hipFree(p));
    return STX_OK;
}
int stx_pinned_alloc(stx_ctx* c, uint64_t bytes, void** out) {
    if (!c || !out) return STX_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipError_t e = hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(c, STX_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return STX_OK;
}
int stx_pinned_free(stx_ctx* c, void* p) {
    if (!c) return STX_ERR_INVALID;
    HIPCHK(c, hipHostFree(p));
    return STX_OK;
}
int stx_memcpy_d2h(stx_ctx* c, void* dst, const void* src, uint64_t bytes) {
    if (!c) return STX_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return STX_OK;
}
int stx_memcpy_h2d(stx_ctx* c, void* dst, const void* src, uint64_t bytes) {
    if (!c) return STX_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return STX_OK;
}

int stx_set_chunk_frames(stx_ctx* c, int frames) {
    if (!c || frames < 0) return STX_ERR_INVALID;
    c->chunk_override = frames;
    return STX_OK;
}
int stx_set_concurrency(stx_ctx* c, int enabled) {
    if (!c) return STX_ERR_INVALID;
    c->concurrent_fft = enabled != 0;
    return STX_OK;
}
int stx_set_canny_mode(stx_ctx* c, int mode) {
    if (!c || mode < STX_CANNY_DENSE || mode > STX_CANNY_AUTO) return STX_ERR_INVALID;
    c->canny_mode = mode;
    c->raw_next = true;
    c->dense_run = 0;
    c->raw_probe = RAW_PROBE;
    return STX_OK;
}
int stx_set_timing(stx_ctx* c, int enabled) {
    if (!c) return STX_ERR_INVALID;
    c->timing = enabled != 0;
    c->timing_kinds = (enabled & 1) ? ~0u : ((uint32_t)enabled >> 1);
    return STX_OK;
}
int stx_get_timing(stx_ctx* c, stx_timing* out, int reset) {
    if (!c || !out) return STX_ERR_INVALID;
    if (!c->synth_spans.empty()) {
        HIPCHK(c, hipEventSynchronize(c->synth_spans.back().e1));
        collect_spans(c, c->synth_spans);
    }
    *out = c->tacc;
    out->fft64_frames = c->fft64_frames;
    if (reset) {
        memset(&c->tacc, 0, sizeof(c->tacc));
        c->fft64_frames = 0;
    }
    return STX_OK;
}

}  // extern "C"
