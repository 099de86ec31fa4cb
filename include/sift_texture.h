/*
 * sift_texture.h — C ABI of libsift_texture.so, the MI355X (gfx950) implementation of
 * the texture/noise hot path of y-mitsuyoshi/sift.
 *
 * The reference has no FFI for this path: the boundary is the Python class
 * TextureNoiseDetector (/root/reference/app/passive_checker.py:88-164).  These entry
 * points are what a ctypes binding of that class calls; sift_amd/texture.py is such a
 * binding.  Every function returns STX_OK (0) or a negative error code, never throws,
 * never aborts; stx_last_error() gives the message for the last failure on a context.
 *
 * Frame layout everywhere: uint8, BGR interleaved, [H][W][3], the layout
 * extract_frames() produces (/root/reference/app/video_processor.py:18-49).
 */
#ifndef SIFT_TEXTURE_H
#define SIFT_TEXTURE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STX_OK 0
#define STX_ERR_INVALID (-1)     /* bad argument */
#define STX_ERR_HIP (-2)         /* a HIP runtime call failed */
#define STX_ERR_UNSUPPORTED (-3) /* size or parameter outside what the kernels handle */
#define STX_ERR_NOMEM (-4)

#define STX_LBP_MAX_BINS 26 /* P + 2 bins, P <= 24 */
#define STX_FLAG_FFT_F64 1u /* stx_metrics.flags: fft_sum / fft_sumsq come from the fp64 transform */
#define STX_MAX_DIM 8192    /* largest frame height or width */

typedef struct stx_ctx stx_ctx;

/* Parameters of analyze_frame; the defaults (stx_default_params) are the literals in
 * passive_checker.py:100-123. */
typedef struct stx_params {
    int lbp_points;    /* P, default 8      (passive_checker.py:103) */
    double lbp_radius; /* R, default 1.0    (passive_checker.py:103) */
    int canny_low;     /* default 50        (passive_checker.py:116) */
    int canny_high;    /* default 150       (passive_checker.py:116) */
    int gray_shift;    /* 15 (default: OpenCV 4.x's RGB2Gray<uchar>) or 14 (OpenCV 3.x): fixed-point width of the
                          BGR->gray weights (:100) */
    int fft_h, fft_w;  /* 0,0 (default) = transform the whole frame (:120); else a centred crop */
    int quality_stats; /* 0 (default); 1 = also fill the frame-quality fields below, which the callers of the
                          texture path compute per frame (/root/reference/app/main.py:120-123, :209-216) */
    int fft_exact;     /* precision of the transform behind frequency_anomaly (:120-123; numpy's is complex128):
                          0 (default) = fp32, and any frame whose spectrum has enough near-zero bins for fp32 rounding
                          noise to show in the statistic is transformed again in fp64; 1 = every frame in fp64;
                          -1 = fp32 only */
} stx_params;

/* Per-frame result: the five metrics of passive_checker.py:125-131, the per-frame score
 * of :145-151, and the integer reductions they are computed from (bit-exact quantities). */
typedef struct stx_metrics {
    double lbp_variance;          /* np.var(lbp)            :104 */
    double noise_level;           /* np.std(high_freq)      :108 */
    double compression_artifacts; /* np.mean(absdiff)       :113 */
    double edge_density;          /* count(edges>0)/size    :117 */
    double frequency_anomaly;     /* np.std(log|F|+1)       :123 */
    double texture_score;         /* weighted sum           :145-151 */
    uint32_t lbp_hist[STX_LBP_MAX_BINS]; /* histogram of uniform LBP codes 0..P+1 */
    uint32_t edge_count;          /* Canny edge pixels */
    uint32_t flags;               /* STX_FLAG_* */
    uint64_t absdiff_sum;         /* sum |gray - GaussianBlur5x5(gray)| */
    uint64_t hp_sum;              /* sum of the saturated 3x3 high-pass image */
    uint64_t hp_sumsq;            /* sum of its squares */
    uint64_t gray_sum;            /* sum of the gray image */
    uint64_t n_pixels;            /* H*W */
    double fft_sum, fft_sumsq;    /* sum, sum of squares of ln(|F|+1) over all FFT bins */
    /* frame quality (zero unless stx_params.quality_stats): what the fallback checkers score a frame from */
    double laplacian_variance;    /* cv2.Laplacian(gray, CV_64F).var()   main.py:120, :209 */
    double brightness_variance;   /* gray.var()                          main.py:123, :210 */
    int64_t lap_sum;              /* sum of the 3x3 Laplacian (aperture 1, BORDER_REFLECT_101) */
    uint64_t lap_sumsq;           /* sum of its squares */
    uint64_t gray_sumsq;          /* sum of squared gray levels */
} stx_metrics;

/* Accumulated device time per kernel family, measured with HIP events on the stream
 * the kernels run on (only while stx_set_timing is in effect, and only for the families it names). */
#define STX_N_KERNELS 10
typedef struct stx_timing {
    double ms[STX_N_KERNELS];       /* total elapsed milliseconds */
    uint64_t launches[STX_N_KERNELS]; /* kernel launches counted into ms[] */
    uint64_t frames;                /* frames analysed while timing was on */
    uint64_t fft64_frames;          /* frames since the last reset whose transform was repeated in fp64 (stx_params.fft_exact) */
} stx_timing;
/* indices into stx_timing.ms / .launches */
#define STX_K_GRAY 0
#define STX_K_STENCIL 1
#define STX_K_LBP_GENERIC 2
#define STX_K_HYST 3
#define STX_K_FFT_ROWS 4
#define STX_K_FFT_COLS 5
#define STX_K_REDUCE 6
#define STX_K_SYNTH 7
#define STX_K_QUALITY 8  /* the optional frame-quality pass (stx_params.quality_stats) */
#define STX_K_EXTENDED 9 /* Sobel / blockiness sums of the extended entry points */

const char* stx_version(void);

/* Replaces TextureNoiseDetector.__init__ (passive_checker.py:93-94): binds a context to
 * one GPU, creates its stream and workspace.  Safe to create several per process. */
int stx_create(int device_id, stx_ctx** out);
int stx_destroy(stx_ctx* ctx);
const char* stx_last_error(const stx_ctx* ctx);
/* 1 = HIP on gfx950.  There is no CPU backend: without a usable GPU stx_create fails. */
int stx_backend(const stx_ctx* ctx);
void stx_default_params(stx_params* p);

/* Replaces the body of analyze_frame (passive_checker.py:96-131) for n_frames frames of one
 * size held in HOST memory (pageable or pinned).  row_stride_bytes >= 3*w; frames are
 * frame_stride_bytes apart.  params == NULL means the reference literals.  out receives
 * n_frames records.  The call returns after the results are on the host. */
int stx_analyze_batch(stx_ctx* ctx, const uint8_t* bgr, int n_frames, int h, int w, int64_t row_stride_bytes,
                      int64_t frame_stride_bytes, const stx_params* params, stx_metrics* out);

/* Same, for a Python list of separately allocated frames (the loop at passive_checker.py:140-141). */
int stx_analyze_frames(stx_ctx* ctx, const uint8_t* const* frames, int n_frames, int h, int w,
                       int64_t row_stride_bytes, const stx_params* params, stx_metrics* out);

/* Same, for frames already resident in DEVICE memory of the context's GPU (the benchmark path:
 * no PCIe transfer in the timed region).  out is host memory. */
int stx_analyze_batch_device(stx_ctx* ctx, const uint8_t* bgr_dev, int n_frames, int h, int w,
                             int64_t row_stride_bytes, int64_t frame_stride_bytes, const stx_params* params,
                             stx_metrics* out);

/* stx_analyze_batch_device for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc,
 * processes in which another thread is capturing a HIP graph); it has no counterpart in the reference.  The call runs
 * eagerly and synchronises with the context's streams like stx_analyze_batch_device (it cannot itself be captured
 * into a graph), but allocates nothing, on the device or the host: every intermediate (gray planes, edge maps, spectra, partial sums, twiddle tables, the
 * fp64 workspace of stx_params.fft_exact) lives in a static device array of STX_NOALLOC_SCRATCH_BYTES that each GPU
 * the library runs on holds until the process ends; streams, events and pinned staging are created by stx_create.
 * Frames go through the array at most 64 at a time.  Limits: LBP at (P=8, R=1) only, no quality_stats, and one
 * frame's workspace must fit the array (1080p and 4K frames do; STX_ERR_UNSUPPORTED with a message otherwise).
 * Results equal stx_analyze_batch_device's.  All contexts of one GPU share the array, one call at a time per GPU: a
 * call that starts while another allocation-free call is running on the same GPU (from another thread, through
 * another context) returns STX_ERR_INVALID instead of racing with it. */
#define STX_NOALLOC_SCRATCH_BYTES (128u << 20)
int stx_analyze_batch_device_noalloc(stx_ctx* ctx, const uint8_t* bgr_dev, int n_frames, int h, int w,
                                     int64_t row_stride_bytes, int64_t frame_stride_bytes, const stx_params* params,
                                     stx_metrics* out);

/* Asynchronous form of stx_analyze_batch_device (no counterpart in the reference): queues the work
 * on the GPU and returns a ticket; stx_wait(ticket) blocks until out[0..n_frames) is filled.  Up to
 * two submissions may be in flight, so the host's part of a clip (queueing it, collecting the one
 * before) runs while the device works, and the device goes from one clip's kernels to the next
 * without idling.  n_frames must not exceed
 * stx_max_async_frames(ctx, h, w); bgr_dev, params and out must stay valid until stx_wait returns.
 * The synchronous entry points refuse to run while a ticket is outstanding. */
int stx_submit_batch_device(stx_ctx* ctx, const uint8_t* bgr_dev, int n_frames, int h, int w,
                            int64_t row_stride_bytes, int64_t frame_stride_bytes, const stx_params* params,
                            stx_metrics* out, int* ticket);
int stx_wait(stx_ctx* ctx, int ticket);
int stx_max_async_frames(stx_ctx* ctx, int h, int w);

/* compute_texture_score's reduction (passive_checker.py:145-156) over n records: mean of the
 * per-frame scores (0.0 when n == 0) and the >= threshold verdict. */
int stx_score_summary(const stx_metrics* m, int n, double threshold, double* average, int* passed);

/* Intermediate planes for parity tests: gray (h*w), uniform LBP codes (h*w), Canny edge map
 * (h*w, 0/255) of ONE host frame; any output pointer may be NULL. */
int stx_debug_planes(stx_ctx* ctx, const uint8_t* bgr, int h, int w, int64_t row_stride_bytes,
                     const stx_params* params, uint8_t* gray_out, uint8_t* lbp_codes_out, uint8_t* edges_out);

/* Deterministic synthetic frames written straight into device memory (benchmark input; same
 * integer arithmetic as sift_amd/synth.py).  kind: 0 noise, 1 natural, 2 ramp, 3 flat, 4 tie, 5 step. */
int stx_synth_device(stx_ctx* ctx, uint8_t* bgr_dev, int kind, uint64_t seed, uint32_t video, uint32_t frame0,
                     int n_frames, int h, int w);

/* The same generator without the wait (no counterpart in the reference): the frames are written by the time any
 * analysis call made on this context afterwards reads them -- the generator is queued on the stream those calls
 * start on -- and until then it runs beside the context's other work, so a job can make clip i+1 while clip i is
 * analysed and never hold more than the clips in flight.  bgr_dev must not be read by the host, or freed, before a
 * later call on the context has returned (or stx_get_timing, which waits for the generator's timing events). */
int stx_synth_device_async(stx_ctx* ctx, uint8_t* bgr_dev, int kind, uint64_t seed, uint32_t video, uint32_t frame0,
                           int n_frames, int h, int w);

/* Device / pinned-host buffers for callers without their own allocator. */
int stx_device_alloc(stx_ctx* ctx, uint64_t bytes, void** out);
int stx_device_free(stx_ctx* ctx, void* p);
int stx_pinned_alloc(stx_ctx* ctx, uint64_t bytes, void** out);
int stx_pinned_free(stx_ctx* ctx, void* p);
int stx_memcpy_d2h(stx_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);
int stx_memcpy_h2d(stx_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);

/* Extended per-frame statistics.  No counterpart in the reference: they are the multi-scale additions
 * (LBP at further radii, Sobel gradient statistics on the images cv2.Canny builds at
 * app/passive_checker.py:116, 8x8 blockiness) and are only reported here, never in the reference's dict.
 * Integer fields are exact; the floats derive from them on the host.
 *   mag = |gx| + |gy| (Sobel 3x3, replicated borders): sobel_sum, sobel_sumsq, sobel_strong = count(mag > canny_high)
 *   block_sums[0..3] = sum |g(y,x)-g(y,x-1)| over x%8==0 (x>=8) / over the other x>=1, then the same for rows;
 *   block_counts[] = how many differences each sum holds
 *   blockiness = mean difference across 8-pixel boundaries / (mean difference inside blocks + 1e-10)
 *   edge_consistency = Canny edge pixels / sobel_strong (0 when there is none) */
typedef struct stx_extended {
    double sobel_mean;
    double sobel_std;
    double strong_gradient_fraction;
    double edge_consistency;
    double blockiness;
    uint64_t sobel_sum;
    uint64_t sobel_sumsq;
    uint64_t block_sums[4];
    uint64_t block_counts[4];
    uint32_t sobel_strong;
    uint32_t reserved;
    uint32_t lbp_hist_p16r2[STX_LBP_MAX_BINS]; /* uniform LBP histogram at P=16, R=2 (18 bins used) */
    uint32_t lbp_hist_p24r3[STX_LBP_MAX_BINS]; /* ... at P=24, R=3 (26 bins) */
} stx_extended;

/* stx_analyze_frames / stx_analyze_batch_device plus one stx_extended record per frame. */
int stx_analyze_extended_frames(stx_ctx* ctx, const uint8_t* const* frames, int n_frames, int h, int w, int64_t row_stride,
                                const stx_params* params, stx_metrics* out, stx_extended* ext);
int stx_analyze_extended_batch_device(stx_ctx* ctx, const uint8_t* bgr_dev, int n_frames, int h, int w, int64_t row_stride,
                                      int64_t frame_stride, const stx_params* params, stx_metrics* out, stx_extended* ext);

/* Asynchronous form of stx_analyze_extended_batch_device: a ticket for stx_wait, as stx_submit_batch_device; ext must
 * stay valid until stx_wait returns, like out. */
int stx_submit_extended_batch_device(stx_ctx* ctx, const uint8_t* bgr_dev, int n_frames, int h, int w, int64_t row_stride,
                                     int64_t frame_stride, const stx_params* params, stx_metrics* out, stx_extended* ext,
                                     int* ticket);

/* Frame preparation of the CNN branch: replaces PassiveChecker._preprocess_frame
 * (app/passive_checker.py:393-418), called per frame at :432 from the loop at :473 -- BGR -> RGB, cv2.resize to
 * (out_w, out_h) with the default bilinear interpolation (OpenCV's 8-bit fixed-point arithmetic, bit for bit),
 * float32 / 255, planes first.  out holds n_frames x 3 x out_h x out_w floats (R, G, B planes per frame).
 * The reference's input size is (80, 80) (passive_checker.py:355).  The _device form takes frames and output
 * in device memory. */
int stx_preprocess_frames(stx_ctx* ctx, const uint8_t* const* frames, int n_frames, int h, int w, int64_t row_stride,
                          int out_h, int out_w, float* out_host);
int stx_preprocess_batch_device(stx_ctx* ctx, const uint8_t* bgr_dev, int n_frames, int h, int w, int64_t row_stride,
                                int64_t frame_stride, int out_h, int out_w, float* out_dev);

/* Frames per internal chunk (0 = choose from the frame size: up to 2 GiB of intermediates per
 * in-flight chunk, two chunks in flight). */
int stx_set_chunk_frames(stx_ctx* ctx, int frames);
/* 1 (default): the FFT kernels run on a second stream beside the stencil / hysteresis kernels;
 * 0: every kernel runs back to back on one stream (clean per-kernel timings). */
int stx_set_concurrency(stx_ctx* ctx, int enabled);
/* enabled: 0 = off; 1 = every kernel family; otherwise a set of families, bit k+1 for family STX_K_<k> (an event
 * pair around a kernel costs a few microseconds of device time between it and its neighbours on the stream: a caller
 * that wants one family's durations without slowing the rest of the path names just that one). */
int stx_set_timing(stx_ctx* ctx, int enabled);
int stx_get_timing(stx_ctx* ctx, stx_timing* out, int reset);
/* Where the non-maximum suppression of cv2.Canny (passive_checker.py:116) runs; the edge map is the same either way.
 * STX_CANNY_DENSE: in the stencil pass, on every pixel.  STX_CANNY_SPARSE: the stencil pass leaves the raw double
 * threshold, a flood of those maps finds the pixels an edge can reach, the suppression runs on those only (a fraction
 * of a percent of a camera frame), the ordinary hysteresis follows; a chunk whose flood does not settle in its passes,
 * or reaches more than 4 % of a frame, goes through again the dense way.  STX_CANNY_AUTO (default): sparse for chunks
 * of 64 Mpixel and more (a call on a few frames is made of launches, and the sparse path has more of them) while the
 * flooded set stays below 2 % of the pixels, dense otherwise, with one chunk in 32 tried sparse again (one in 64,
 * 128 ... 1024 while the tries keep failing).  Calls that ask
 * for code planes, LBP radii other than (8,1) and the allocation-free entry point are always dense.  The environment
 * variable SIFT_TEXTURE_CANNY=dense|sparse|auto sets the mode a new context starts with. */
#define STX_CANNY_DENSE 0
#define STX_CANNY_SPARSE 1
#define STX_CANNY_AUTO 2
int stx_set_canny_mode(stx_ctx* ctx, int mode);


#ifdef __cplusplus
}
#endif
#endif
