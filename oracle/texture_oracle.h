/*
 * CPU oracle for the texture/noise hot path — TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of what TextureNoiseDetector.analyze_frame
 * (/root/reference/app/passive_checker.py:96-131) and compute_texture_score
 * (:133-164) compute, including the third-party routines they call
 * (OpenCV cvtColor/filter2D/GaussianBlur/absdiff/Canny, scikit-image
 * local_binary_pattern, numpy fft2/std/var/mean).  Nothing in the product
 * library links or loads this file; only tests/, the smoke check and the
 * benchmark's cpu_baseline leg do.
 *
 * Pinning status:
 *   - LBP (orc_lbp_uniform): pinned bit-exact against scikit-image 0.18.3
 *     code images committed under tests/golden/.
 *   - FFT statistic (orc_fft_logmag_stats): pinned against numpy 2.2.6
 *     (tests/golden/, 1e-12 relative).
 *   - gray / high-pass / Gaussian / Canny: PARITY UNPINNED — OpenCV is not
 *     available offline and the reference holds no fixtures; these follow
 *     OpenCV's published algorithm and are cross-checked against an
 *     independent numpy/scipy restatement (oracle/texture_oracle.py).
 */
#ifndef TEXTURE_ORACLE_H
#define TEXTURE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_LBP_MAX_BINS 26 /* P + 2 for P <= 24 */

typedef struct {
    int lbp_points;    /* P; reference literal 8   (passive_checker.py:103) */
    double lbp_radius; /* R; reference literal 1.0 (passive_checker.py:103) */
    int canny_low;     /* 50  (passive_checker.py:116) */
    int canny_high;    /* 150 (passive_checker.py:116) */
    int gray_shift;    /* 15 (default): (3735 B + 19235 G + 9798 R + 16384) >> 15; 14: (1868 B + 9617 G + 4899 R + 8192) >> 14 */
    int fft_h, fft_w;  /* 0,0 = whole frame; else centred crop of that size */
} orc_params;

typedef struct {
    /* the five metrics of passive_checker.py:125-131 and the per-frame score of :145-151 */
    double lbp_variance, noise_level, compression_artifacts, edge_density, frequency_anomaly;
    double texture_score;
    /* integer reductions they are derived from */
    uint32_t lbp_hist[ORC_LBP_MAX_BINS];
    uint32_t edge_count;
    uint32_t reserved;
    uint64_t absdiff_sum; /* sum |gray - blur| */
    uint64_t hp_sum;      /* sum of saturated high-pass pixels */
    uint64_t hp_sumsq;    /* sum of their squares */
    uint64_t gray_sum;
    uint64_t n_pixels;
    double fft_sum, fft_sumsq; /* sum / sum of squares of ln(|F|+1) over all fft_h*fft_w bins */
} orc_metrics;

void orc_default_params(orc_params* p);

/* stage functions; all images are dense row-major h x w */
void orc_bgr2gray(const uint8_t* bgr, int h, int w, int64_t row_stride, int shift, uint8_t* gray);
int orc_lbp_offsets(int P, double R, double* rp, double* cp);
int orc_lbp_uniform(const uint8_t* gray, int h, int w, int P, double R, uint8_t* codes /*may be NULL*/,
                    uint32_t* hist /*P+2 bins*/);
void orc_highpass3(const uint8_t* gray, int h, int w, uint8_t* out);
void orc_gauss5(const uint8_t* gray, int h, int w, uint8_t* out);
/* Laplacian (aperture 1, reflect-101) sum / sum of squares and gray sum of squares: main.py:209-210 */
void orc_quality_sums(const uint8_t* gray, int h, int w, int64_t* lap_sum, uint64_t* lap_sumsq, uint64_t* gray_sumsq);
int orc_canny(const uint8_t* gray, int h, int w, int low, int high, uint8_t* edges /*0/255*/,
              uint8_t* nms_map /*may be NULL: 0 none,1 weak,2 strong before hysteresis*/);
int orc_fft_logmag_stats(const uint8_t* gray, int h, int w, int64_t stride, double* sum, double* sumsq,
                         double* spectrum /*may be NULL: h*w*2 doubles re,im*/);

/* whole path */
int orc_analyze_frame(const uint8_t* bgr, int h, int w, int64_t row_stride, const orc_params* p, orc_metrics* out);
void orc_finalize_metrics(orc_metrics* m, int lbp_points, uint64_t n_fft_bins);
double orc_texture_score(const orc_metrics* m);

/* deterministic synthetic frames (same arithmetic as sift_amd/synth.py) */
int orc_synth_frame(int kind, uint64_t seed, uint32_t video, uint32_t frame, int h, int w, uint8_t* bgr);

#ifdef __cplusplus
}
#endif
#endif
