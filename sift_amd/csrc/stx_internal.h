// Internal declarations shared by the HIP translation units of libsift_texture.so.
#pragma once

#include <hip/hip_runtime.h>

#include <vector>
#include <stdint.h>

#include "../../include/sift_texture.h"

namespace stx {

// Per-frame accumulators filled on the device, copied to the host once per chunk.
struct FrameAcc {
    uint32_t lbp_hist[STX_LBP_MAX_BINS];
    uint32_t edge_count;
    uint32_t pad0;
    unsigned long long absdiff_sum;
    unsigned long long hp_sum;
    unsigned long long hp_sumsq;
    unsigned long long gray_sum;
    double fft_sum;
    double fft_sumsq;
    double fft_t1, fft_t2;  // sums over the near-zero bins of the spectrum (FftPart)
    double fft_e2;          // sum of |F|^2 over the spectrum of gray - m (the transform's input energy times fh*fw)
    // frame-quality sums (only filled when stx_params.quality_stats is set)
    long long lap_sum;
    unsigned long long lap_sumsq;
    unsigned long long gray_sumsq;
    // extended statistics (only filled by the extended entry points)
    uint32_t lbp_hist2[STX_LBP_MAX_BINS];  // second and third LBP radius
    uint32_t lbp_hist3[STX_LBP_MAX_BINS];
    unsigned long long sobel_sum, sobel_sumsq;
    unsigned long long blk[4];  // |differences| across 8-pixel boundaries / inside blocks, horizontal then vertical
    uint32_t sobel_strong;
    uint32_t pad1;
};

// Geometry of one chunk of equally sized frames in the workspace.
struct Geom {
    int H, W;        // frame size
    int gpitch;      // bytes per row of the padded gray plane: roundup(W,128)+64
    int grows;       // rows of the padded gray plane: H+4 (two reflected rows above and below)
    int tiles_x;     // ceil(W/128) stencil tiles
    int tiles_y;     // ceil(H/128)
    // Canny bitmaps run down the columns: one 64-bit word = 64 vertically adjacent pixels of one column,
    // stored [frame][bm_words][W].  A hysteresis tile is 64 columns x 64 rows.
    int bm_words;    // ceil(H/64) words per column
    int bm_tiles;    // ceil(W/64) hysteresis tiles across
    int fh, fw;      // FFT extent
    int foy, fox;    // FFT crop origin in the frame
    int fwh;         // fw/2+1 half-spectrum columns
};

// The padded gray plane stores pixel (y,x) at [(y+2)*gpitch + x+4]; rows -2..H+1 and
// columns -2..W+1 hold BORDER_REFLECT_101 copies.
constexpr int GRAY_XOFF = 4;
constexpr int GRAY_YOFF = 2;

struct LbpOffsets {
    double rp[24];
    double cp[24];
};

constexpr int FFT_MAX_PASSES = 16;
struct FftPlan {
    int n;
    int npass;
    int radix[FFT_MAX_PASSES];
    // 0, or the convolution length of the chirp-z transform that lengths with a prime factor above
    // FFT_BLUESTEIN_MIN_PRIME take instead of the passes (their table is fft_table_entries(plan) long, see
    // fft_fill_table)
    int bz_m;
};
constexpr int FFT_BLUESTEIN_MIN_PRIME = 256;
// Entries of the table a plan's kernels read: n twiddles exp(-2 pi i k / n), or the three chirp-z tables.
inline size_t fft_table_entries(const FftPlan& p) { return p.bz_m ? (size_t)p.n + (size_t)p.bz_m / 2 + (size_t)p.bz_m : (size_t)p.n; }
constexpr size_t FFT_TABLE_MAX = (size_t)STX_MAX_DIM + (size_t)STX_MAX_DIM + 2 * (size_t)STX_MAX_DIM;  // n <= STX_MAX_DIM, m <= 2 STX_MAX_DIM
// Fills a plan's table in double precision (re, im interleaved).  work: room for 2 * bz_m doubles where a chirp-z
// table is built, or nullptr to take it from the heap.
void fft_fill_table(const FftPlan& p, double* out, double* work = nullptr);

// Partial sums of one spectrum column (weighted twice where the column stands for its mirror image too).
// l = ln(|F|+1).  t1, t2 cover only the bins with |F| < the `small` threshold of the launch: those are the bins
// whose value the fp32 rounding noise of the transform can change to first order, and
// mean(l) * t1 - t2 = sum (mean(l) - l) / (|F|+1) bounds what that does to the variance of l.
struct FftPart {
    double s, s2;   // sum l, sum l^2
    double t1, t2;  // sum 1/(|F|+1), sum l/(|F|+1) over the small bins
    double e2;      // sum |F|^2, before the gray level m goes back into bin (0,0)
};

// ---- launchers (each file owns its kernels) ----
void launch_gray(hipStream_t s, const uint8_t* bgr, int64_t row_stride, int64_t frame_stride, int n_frames,
                 const Geom& g, int shift, uint8_t* gray);

// Where the stencil pass converts the frames itself instead of reading a finished gray plane.
struct BgrSource {
    const uint8_t* bgr;  // device, packed BGR
    int64_t row_stride, frame_stride;
    int shift;           // 14 or 15: fixed-point precision of the conversion
};
// True if the stencil pass can read these frames directly: its loader takes four pixels (twelve bytes) per lane and
// mirrors the border with one reflection.
bool stencil_reads_bgr(const Geom& g, const BgrSource& src);
// gray: the padded plane.  With src == nullptr it is the pass's input; otherwise the pass converts the BGR frames
// while it loads its tiles and writes the plane, reflected frame included, for the kernels after it.
void launch_stencil(hipStream_t s, uint8_t* gray, const BgrSource* src, int n_frames, const Geom& g, bool do_lbp81,
                    const LbpOffsets& off, int low, int high, uint32_t* part, unsigned long long* strong,
                    unsigned long long* cand, uint8_t* codes_out, const uint8_t* tie_blob, uint8_t* sflag,
                    bool raw_canny = false);
// Tables the (P=8,R=1) LBP uses for flat neighbourhoods (built on the host once per context and copied to the
// device as one blob); TIE_COORDS entries per coordinate table cover every row / column a tile can touch.
constexpr int TIE_COORDS = STX_MAX_DIM + 256;
void build_tie_tables(const LbpOffsets& off, std::vector<uint8_t>* blob, int* n_cls_c);

// Any (P <= 24, R): per-tile histograms into part (lbp_generic_part_bytes), then summed into acc[].lbp_hist.
// Returns false when the radius needs a halo that does not fit an LDS tile.
size_t lbp_generic_part_bytes(const Geom& g, int n_frames);
bool launch_lbp_generic(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, int P, const LbpOffsets& off,
                        uint32_t* part, FrameAcc* acc, uint8_t* codes_out, int hist_slot = 0);
// (P=16, R=2) and (P=24, R=3) in one launch from one tile, into acc[].lbp_hist2 / lbp_hist3 (two launches of the kernel
// above where the offsets are not those of the compiled cell tables).
bool launch_lbp_rings(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, const LbpOffsets& off16,
                      const LbpOffsets& off24, uint32_t* part, FrameAcc* acc);
// Sobel-magnitude and 8x8 blockiness sums per frame into acc[] (stx_extended.hip).
size_t extended_part_bytes(const Geom& g, int n_frames);
void launch_extended(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, int high, unsigned long long* part,
                     FrameAcc* acc);

// One hysteresis pass over every 64x64 tile of every frame; pass p exits at once when pass p-1
// changed nothing.  changed[] must be zeroed before pass 0.  tile_count: [frame][bm_tiles][bm_words].
// cand_sparse: `cand` is the map launch_nms_sparse filled -- only the tiles its validity bytes (the third part of
// tflag) name were written, every other tile counts as empty.
void launch_hyst_pass(hipStream_t s, unsigned long long* strong, const unsigned long long* cand, int n_frames,
                      const Geom& g, unsigned int* changed, int pass, uint32_t* tile_count, uint8_t* tflag,
                      bool cand_sparse = false);
// Pass number `pass` >= 1 as one workgroup per frame that repeats the pass until the frame is done (or a
// round limit is reached: then changed[pass] is set and the caller queues it again).
void launch_hyst_finish(hipStream_t s, unsigned long long* strong, const unsigned long long* cand, int n_frames,
                      const Geom& g, unsigned int* changed, int pass, uint32_t* tile_count, uint8_t* tflag,
                      bool cand_sparse = false);
// One launch for all the clears a chunk needs before its first kernel (see chunk_init_kernel).  raw_changed / preset_one /
// nmscount may be null together (a chunk without the flood of raw maps); sflag_bytes is rounded up to whole words.
void launch_chunk_init(hipStream_t s, FrameAcc* acc, int n_frames, unsigned int* changed, int n_changed,
                       unsigned int* raw_changed, int n_raw_changed, unsigned int* preset_one, unsigned int* nmscount,
                       uint8_t* sflag, size_t sflag_bytes);
// After a flood of RAW threshold maps (`strong` = the flooded set, tile_count = its size per tile as the flood left it):
// non-maximum suppression of the pixels in the set from the gray plane.  Leaves the suppressed candidates in `cand2`
// (written only in the tiles the flood reached; one byte per tile in the third part of tflag, which must have room for
// it, says which those are), "tile holds a strong pixel" in the second half of tflag (what hysteresis pass 0 reads, and
// what launch_hyst_finish reads as the flags of an odd pass before it), the suppressed strong pixels in `strong`, zeros
// in tile_count, and adds the size of the set to *reached_total.
// list: [n_frames][nms_list_cap(g)] entries, count: [n_frames], zeroed by the caller; *unsettled is set when a frame's
// set does not fit its list (the maps are then unusable and the caller runs the chunk the ordinary way).
int nms_list_cap(const Geom& g);
void launch_nms_sparse(hipStream_t s, unsigned long long* strong, unsigned long long* cand2, const uint8_t* gray, int n_frames,
                       const Geom& g, int low, int high, uint32_t* tile_count, uint8_t* tflag, uint32_t* list,
                       unsigned int* count, unsigned int* unsettled, unsigned int* reached_total);
void launch_frame_reduce(hipStream_t s, const uint32_t* spart, bool lbp_from_tiles, const uint32_t* tile_count,
                         const FftPart* fpart, int nfparts, int n_frames, const Geom& g, FrameAcc* acc);
void launch_edges_expand(hipStream_t s, const unsigned long long* strong, const Geom& g, uint8_t* edges);

// Real 2-D FFT of the (cropped) gray plane minus its first gray level, log-magnitude statistics.
// spec: [n_frames][nkb][fh][cbw] float2 half spectrum in column blocks.
// part: [n_frames][parts_per_frame] FftPart.
struct FftLayout {
    int cbw, nkb;
    int parts_per_frame;
    bool fast_rows, fast_cols;
};
FftLayout fft_choose_layout(const Geom& g, const FftPlan& plan_h);
void launch_fft_rows(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, const FftLayout& lay,
                     const FftPlan& plan, const float2* tw, float2* spec);
void launch_fft_cols(hipStream_t s, const uint8_t* gray, const float2* spec, int n_frames, const Geom& g,
                     const FftLayout& lay, const FftPlan& plan, const float2* tw, FftPart* part, float small);
// Bound on the absolute error of one bin of the fp32 transform of an fh x fw region of 8-bit pixels, and the
// magnitude below which a bin counts as small.
double fft_f32_bin_error(int fh, int fw);
// The same bound for a signal of known energy: rms = sqrt(sum (gray - m)^2 / (fh*fw)) in place of 255.
double fft_f32_bin_error_rms(int fh, int fw, double rms);
inline float fft_small_threshold(int fh, int fw) { return (float)(64.0 * fft_f32_bin_error(fh, fw)); }

// The same transform and statistic in double precision for single frames (stx_fft64.hip): used for the frames whose
// spectrum has enough near-zero bins that fp32 rounding noise would show in the statistic.
struct Fft64Work {
    void* spec = nullptr;     // half spectrum as double2, in blocks of four columns
    void* scratch = nullptr;  // second pass buffer of the workgroups when it does not fit LDS
    void* part = nullptr;     // [fwh] double2 (sum, sum of squares) per column, weighted
    size_t spec_bytes = 0, scratch_bytes = 0, part_bytes = 0;
};
void fft64_sizes(const Geom& g, const FftPlan& plan_w, const FftPlan& plan_h, size_t* spec_bytes, size_t* scratch_bytes,
                 size_t* part_bytes);
// gray_chunk: the padded gray planes of the chunk, frames[0..n_batch) the ones to transform; tw_w / tw_h: the plans' double2 tables
// spec / scratch / part hold one slot of spec_bytes / scratch_bytes / part_bytes per frame of the batch
constexpr int FFT64_BATCH = 32;
void launch_fft64(hipStream_t s, const uint8_t* gray_chunk, const int* frames, int n_batch, const Geom& g,
                  const FftPlan& plan_w, const FftPlan& plan_h, const void* tw_w, const void* tw_h, const Fft64Work& w);
hipError_t fft64_set_lds_limits();
// compile-time-planned kernels for the common sizes (stx_fft_fast.hip)
bool fft_fast_available(int n);
int fft_fast_cbw(int n);
hipError_t fft_fast_init();
bool launch_fft_rows_fast(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, const FftLayout& lay,
                          const float2* tw, float2* spec);
bool launch_fft_cols_fast(hipStream_t s, const uint8_t* gray0, int64_t gstride, const float2* spec, int n_frames,
                          const Geom& g, const FftLayout& lay, const float2* tw, FftPart* part, float small);

// Laplacian / gray-square sums per frame into acc[] (stx_quality.hip); part holds quality_part_bytes.
size_t quality_part_bytes(const Geom& g, int n_frames);
void launch_quality(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, unsigned long long* part, FrameAcc* acc);

// Frame preparation of the CNN branch (stx_preprocess.hip): bilinear resize in cv::resize's fixed point,
// BGR -> RGB planes, / 255.  The tap tables are filled on the host and copied to the device by the caller.
size_t preprocess_tap_bytes(int n);
void preprocess_fill_taps(int h, int w, int out_h, int out_w, void* xt_host, void* yt_host);
bool preprocess_is_area2(int h, int w, int out_h, int out_w);
void launch_preprocess(hipStream_t s, const uint8_t* bgr, int64_t row_stride, int64_t frame_stride, int n_frames,
                       int out_h, int out_w, const void* xt, const void* yt, bool area2, float* out);

void launch_synth(hipStream_t s, uint8_t* bgr, int kind, uint64_t seed, uint32_t video, uint32_t frame0, int n_frames,
                  int h, int w);

bool make_fft_plan(int n, FftPlan* plan);
hipError_t fft_set_lds_limits();

}  // namespace stx
