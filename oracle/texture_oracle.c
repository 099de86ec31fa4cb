/*
 * CPU oracle for the texture/noise hot path — TEST INFRASTRUCTURE ONLY
 * (see texture_oracle.h for the pinning status of each stage).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  The LBP
 * stage depends on fp64 operations being rounded one at a time, so FMA
 * contraction must stay off.
 */
#include "texture_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_default_params(orc_params* p) {
    p->lbp_points = 8;
    p->lbp_radius = 1.0;
    p->canny_low = 50;
    p->canny_high = 150;
    p->gray_shift = 15;
    p->fft_h = 0;
    p->fft_w = 0;
}

/* ------------------------------------------------------------------ gray
 * cv2.cvtColor(frame, COLOR_BGR2GRAY) at passive_checker.py:100.  OpenCV's
 * 8-bit path is fixed point with round-to-nearest.  OpenCV 4.x (what an
 * unpinned `opencv-python-headless` installs, requirements.txt:5) uses the
 * 15-bit coefficients 3735/19235/9798 in RGB2Gray<uchar>; OpenCV 3.x used the
 * 14-bit set 1868/9617/4899 (shift == 14, kept as a knob). */
void orc_bgr2gray(const uint8_t* bgr, int h, int w, int64_t row_stride, int shift, uint8_t* gray) {
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = bgr + (int64_t)y * row_stride;
        uint8_t* d = gray + (int64_t)y * w;
        for (int x = 0; x < w; ++x) {
            int b = s[3 * x], g = s[3 * x + 1], r = s[3 * x + 2];
            if (shift == 15)
                d[x] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15);
            else
                d[x] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
        }
    }
}

/* OpenCV borderInterpolate for BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba) */
static int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * len - 2 - p;
    }
    return p;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------ LBP
 * skimage.feature.local_binary_pattern(gray, P, R, 'uniform') at
 * passive_checker.py:103.  Sample offsets are rounded to 5 decimals the way
 * numpy.round does (rint(x*1e5)/1e5); each sample is a bilinear interpolation
 * in fp64 with pixels outside the image read as 0; bit i is
 * (sample_i - centre >= 0); a pattern with at most two 0/1 changes among
 * consecutive bits is coded by its number of ones, any other pattern by P+1. */
int orc_lbp_offsets(int P, double R, double* rp, double* cp) {
    if (P < 1 || P > 24) return -1;
    for (int i = 0; i < P; ++i) {
        double ang = 2.0 * M_PI * (double)i / (double)P;
        double rr = -R * sin(ang);
        double cc = R * cos(ang);
        rp[i] = rint(rr * 100000.0) / 100000.0;
        cp[i] = rint(cc * 100000.0) / 100000.0;
    }
    return 0;
}

static inline double px0(const uint8_t* g, int h, int w, long r, long c) {
    if (r < 0 || r >= h || c < 0 || c >= w) return 0.0;
    return (double)g[r * (long)w + c];
}

int orc_lbp_uniform(const uint8_t* gray, int h, int w, int P, double R, uint8_t* codes, uint32_t* hist) {
    double rp[24], cp[24];
    if (orc_lbp_offsets(P, R, rp, cp) != 0) return -1;
    for (int i = 0; i < P + 2; ++i) hist[i] = 0;
    for (int r = 0; r < h; ++r) {
        for (int c = 0; c < w; ++c) {
            double centre = (double)gray[(long)r * w + c];
            int bits[24];
            for (int i = 0; i < P; ++i) {
                double rf = (double)r + rp[i];
                double cf = (double)c + cp[i];
                long minr = (long)floor(rf), minc = (long)floor(cf);
                long maxr = (long)ceil(rf), maxc = (long)ceil(cf);
                double dr = rf - (double)minr;
                double dc = cf - (double)minc;
                double tl = px0(gray, h, w, minr, minc), tr = px0(gray, h, w, minr, maxc);
                double bl = px0(gray, h, w, maxr, minc), br = px0(gray, h, w, maxr, maxc);
                double top = (1 - dc) * tl + dc * tr;
                double bottom = (1 - dc) * bl + dc * br;
                double t = (1 - dr) * top + dr * bottom;
                bits[i] = (t - centre >= 0) ? 1 : 0;
            }
            int changes = 0, ones = 0;
            for (int i = 0; i < P - 1; ++i) changes += (bits[i] != bits[i + 1]);
            for (int i = 0; i < P; ++i) ones += bits[i];
            int code = (changes <= 2) ? ones : P + 1;
            if (codes) codes[(long)r * w + c] = (uint8_t)code;
            hist[code]++;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ high-pass
 * cv2.filter2D(gray, -1, [[-1,-1,-1],[-1,8,-1],[-1,-1,-1]]) at
 * passive_checker.py:107: correlation, BORDER_REFLECT_101, result saturated
 * to uint8 (negative -> 0). */
void orc_highpass3(const uint8_t* gray, int h, int w, uint8_t* out) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int v = gray[(long)reflect101(y + dy, h) * w + reflect101(x + dx, w)];
                    acc += (dy == 0 && dx == 0) ? 8 * v : -v;
                }
            out[(long)y * w + x] = (uint8_t)clampi(acc, 0, 255);
        }
}

/* ------------------------------------------------------------------ frame quality statistics
 * The fallback checker's per-frame inputs (/root/reference/app/main.py:209-210):
 *   cv2.Laplacian(gray, cv2.CV_64F).var()  -- aperture 1: kernel [0 1 0; 1 -4 1; 0 1 0],
 *                                             BORDER_REFLECT_101, exact integers as doubles
 *   gray.var()                             -- population variance of the gray image
 * Both variances follow from integer sums. */
void orc_quality_sums(const uint8_t* gray, int h, int w, int64_t* lap_sum, uint64_t* lap_sumsq, uint64_t* gray_sumsq) {
    int64_t s = 0;
    uint64_t q = 0, g2 = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int c = gray[(long)y * w + x];
            const int l = gray[(long)y * w + reflect101(x - 1, w)], r = gray[(long)y * w + reflect101(x + 1, w)];
            const int u = gray[(long)reflect101(y - 1, h) * w + x], d = gray[(long)reflect101(y + 1, h) * w + x];
            const int lap = l + r + u + d - 4 * c;
            s += lap;
            q += (uint64_t)((int64_t)lap * lap);
            g2 += (uint64_t)(c * c);
        }
    *lap_sum = s;
    *lap_sumsq = q;
    *gray_sumsq = g2;
}

/* ------------------------------------------------------------------ Gaussian
 * cv2.GaussianBlur(gray, (5,5), 0) at passive_checker.py:111: sigma<=0 with
 * ksize 5 selects the fixed kernel [1 4 6 4 1]/16; the 8-bit path is fixed
 * point with a single rounding at the end: (S + 128) >> 8, S the 2-D weighted
 * sum with integer weights w_i*w_j; BORDER_REFLECT_101. */
void orc_gauss5(const uint8_t* gray, int h, int w, uint8_t* out) {
    static const int k[5] = {1, 4, 6, 4, 1};
    int* hrow = (int*)malloc(sizeof(int) * (size_t)h * (size_t)w);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int i = -2; i <= 2; ++i) s += k[i + 2] * gray[(long)y * w + reflect101(x + i, w)];
            hrow[(long)y * w + x] = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int i = -2; i <= 2; ++i) s += k[i + 2] * hrow[(long)reflect101(y + i, h) * w + x];
            out[(long)y * w + x] = (uint8_t)((s + 128) >> 8);
        }
    free(hrow);
}

/* ------------------------------------------------------------------ Canny
 * cv2.Canny(gray, 50, 150) at passive_checker.py:116: 3x3 Sobel into int16
 * with BORDER_REPLICATE, L1 magnitude (zero outside the image), non-maximum
 * suppression along the quantised gradient direction (tan 22.5 deg as
 * 13573/2^15), double threshold, 8-connected hysteresis. */
int orc_canny(const uint8_t* gray, int h, int w, int low, int high, uint8_t* edges, uint8_t* nms_map) {
    size_t n = (size_t)h * (size_t)w;
    if (low > high) { int t = low; low = high; high = t; } /* cv::Canny swaps reversed thresholds */
    int16_t* dx = (int16_t*)malloc(n * sizeof(int16_t));
    int16_t* dy = (int16_t*)malloc(n * sizeof(int16_t));
    /* magnitude with a one-pixel zero frame */
    int mw = w + 2;
    int* mag = (int*)calloc((size_t)(h + 2) * (size_t)mw, sizeof(int));
    uint8_t* map = (uint8_t*)malloc(n); /* 0 none, 1 weak, 2 strong */
    long* stack = (long*)malloc(n * sizeof(long));
    if (!dx || !dy || !mag || !map || !stack) {
        free(dx); free(dy); free(mag); free(map); free(stack);
        return -1;
    }
#define G(yy, xx) ((int)gray[(long)clampi((yy), 0, h - 1) * w + clampi((xx), 0, w - 1)])
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int gx = (G(y - 1, x + 1) - G(y - 1, x - 1)) + 2 * (G(y, x + 1) - G(y, x - 1)) +
                     (G(y + 1, x + 1) - G(y + 1, x - 1));
            int gy = (G(y + 1, x - 1) - G(y - 1, x - 1)) + 2 * (G(y + 1, x) - G(y - 1, x)) +
                     (G(y + 1, x + 1) - G(y - 1, x + 1));
            dx[(long)y * w + x] = (int16_t)gx;
            dy[(long)y * w + x] = (int16_t)gy;
            mag[(long)(y + 1) * mw + (x + 1)] = abs(gx) + abs(gy);
        }
#undef G
    long sp = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int* ma = mag + (long)(y + 1) * mw + (x + 1);
            const int* mp = ma - mw;
            const int* mn = ma + mw;
            int m = ma[0];
            uint8_t cls = 0;
            if (m > low) {
                int xs = dx[(long)y * w + x], ys = dy[(long)y * w + x];
                int ax = abs(xs), ay = abs(ys) << 15;
                int tg22x = ax * 13573;
                int keep;
                if (ay < tg22x) {
                    keep = (m > ma[-1] && m >= ma[1]);
                } else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) {
                        keep = (m > mp[0] && m >= mn[0]);
                    } else {
                        int s = ((xs ^ ys) < 0) ? -1 : 1;
                        keep = (m > mp[-s] && m > mn[s]);
                    }
                }
                if (keep) cls = (m > high) ? 2 : 1;
            }
            map[(long)y * w + x] = cls;
            if (cls == 2) stack[sp++] = (long)y * w + x;
        }
    if (nms_map) memcpy(nms_map, map, n);
    while (sp > 0) {
        long p = stack[--sp];
        int y = (int)(p / w), x = (int)(p % w);
        for (int ddy = -1; ddy <= 1; ++ddy)
            for (int ddx = -1; ddx <= 1; ++ddx) {
                int yy = y + ddy, xx = x + ddx;
                if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                long q = (long)yy * w + xx;
                if (map[q] == 1) {
                    map[q] = 2;
                    stack[sp++] = q;
                }
            }
    }
    int count = 0;
    for (size_t i = 0; i < n; ++i) {
        int e = (map[i] == 2);
        if (edges) edges[i] = e ? 255 : 0;
        count += e;
    }
    free(dx); free(dy); free(mag); free(map); free(stack);
    return count;
}

/* ------------------------------------------------------------------ FFT
 * np.fft.fft2(gray) -> fftshift -> log(abs+1) -> np.std at
 * passive_checker.py:120-123.  numpy transforms in complex128; the shift is a
 * permutation and does not change the statistic.  Mixed-radix recursive
 * decimation in time with a generic prime butterfly, so any size works. */
typedef struct { double re, im; } cplx;

static int smallest_factor(int n) {
    if (n % 4 == 0) return 4;
    if (n % 2 == 0) return 2;
    for (int p = 3; (long)p * p <= n; p += 2)
        if (n % p == 0) return p;
    return n;
}

/* out[0..n) = DFT of in[0], in[stride], ...; tw = exp(-2 pi i k / N) table with step tws for this n */
static void fft_rec(const cplx* in, cplx* out, int n, int stride, const cplx* tw, int tws, int N, cplx* tmp) {
    if (n == 1) { out[0] = in[0]; return; }
    int p = smallest_factor(n);
    int m = n / p;
    for (int r = 0; r < p; ++r) fft_rec(in + (long)r * stride, out + (long)r * m, m, stride * p, tw, tws * p, N, tmp);
    /* combine: X[k + q m] = sum_r sub_r[k] w_n^{r k} w_p^{r q} */
    cplx* t = tmp; /* p entries */
    for (int k = 0; k < m; ++k) {
        for (int r = 0; r < p; ++r) {
            cplx a = out[(long)r * m + k];
            const cplx wv = tw[((long)r * k * tws) % N];
            t[r].re = a.re * wv.re - a.im * wv.im;
            t[r].im = a.re * wv.im + a.im * wv.re;
        }
        for (int q = 0; q < p; ++q) {
            double sr = 0.0, si = 0.0;
            for (int r = 0; r < p; ++r) {
                const cplx wv = tw[(((long)r * q) % p) * (long)m * tws % N];
                sr += t[r].re * wv.re - t[r].im * wv.im;
                si += t[r].re * wv.im + t[r].im * wv.re;
            }
            tmp[p + q].re = sr;
            tmp[p + q].im = si;
        }
        for (int q = 0; q < p; ++q) out[(long)q * m + k] = tmp[p + q];
    }
}

static cplx* make_twiddles(int n) {
    cplx* tw = (cplx*)malloc(sizeof(cplx) * (size_t)n);
    for (int k = 0; k < n; ++k) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        tw[k].re = cos(a);
        tw[k].im = sin(a);
    }
    return tw;
}

int orc_fft_logmag_stats(const uint8_t* gray, int h, int w, int64_t stride, double* sum, double* sumsq,
                         double* spectrum) {
    size_t n = (size_t)h * (size_t)w;
    cplx* a = (cplx*)malloc(sizeof(cplx) * n);
    cplx* b = (cplx*)malloc(sizeof(cplx) * n);
    int mx = h > w ? h : w;
    cplx* line = (cplx*)malloc(sizeof(cplx) * (size_t)mx * 2);
    cplx* tmp = (cplx*)malloc(sizeof(cplx) * (size_t)mx * 2 + 64);
    cplx* tww = make_twiddles(w);
    cplx* twh = make_twiddles(h);
    if (!a || !b || !line || !tmp || !tww || !twh) return -1;
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            line[x].re = (double)gray[(int64_t)y * stride + x];
            line[x].im = 0.0;
        }
        fft_rec(line, a + (size_t)y * w, w, 1, tww, 1, w, tmp);
    }
    for (int x = 0; x < w; ++x) {
        for (int y = 0; y < h; ++y) line[y] = a[(size_t)y * w + x];
        fft_rec(line, line + h, h, 1, twh, 1, h, tmp);
        for (int y = 0; y < h; ++y) b[(size_t)y * w + x] = line[h + y];
    }
    double s = 0.0, s2 = 0.0;
    for (size_t i = 0; i < n; ++i) {
        double v = log(hypot(b[i].re, b[i].im) + 1.0);
        s += v;
        s2 += v * v;
        if (spectrum) { spectrum[2 * i] = b[i].re; spectrum[2 * i + 1] = b[i].im; }
    }
    *sum = s;
    *sumsq = s2;
    free(a); free(b); free(line); free(tmp); free(tww); free(twh);
    return 0;
}

/* ------------------------------------------------------------------ reductions -> metrics
 * np.var (passive_checker.py:104), np.std (:108, :123), np.mean (:113),
 * count/size (:117); all population statistics (ddof = 0). */
void orc_finalize_metrics(orc_metrics* m, int lbp_points, uint64_t n_fft_bins) {
    double n = (double)m->n_pixels;
    double s1 = 0.0;
    for (int k = 0; k < lbp_points + 2; ++k) s1 += (double)k * (double)m->lbp_hist[k];
    double mean = s1 / n, acc = 0.0;
    for (int k = 0; k < lbp_points + 2; ++k) acc += (double)m->lbp_hist[k] * ((double)k - mean) * ((double)k - mean);
    m->lbp_variance = acc / n;
    unsigned __int128 num = (unsigned __int128)m->n_pixels * m->hp_sumsq - (unsigned __int128)m->hp_sum * m->hp_sum;
    m->noise_level = sqrt((double)num / (n * n));
    m->compression_artifacts = (double)m->absdiff_sum / n;
    m->edge_density = (double)m->edge_count / n;
    double nf = (double)n_fft_bins;
    double fm = m->fft_sum / nf;
    double fv = m->fft_sumsq / nf - fm * fm;
    m->frequency_anomaly = fv > 0 ? sqrt(fv) : 0.0;
    m->texture_score = orc_texture_score(m);
}

static double dmin(double a, double b) { return a < b ? a : b; }

/* passive_checker.py:145-151 */
double orc_texture_score(const orc_metrics* m) {
    return dmin(m->lbp_variance / 50.0, 1.0) * 0.3 + dmin(m->edge_density * 10, 1.0) * 0.2 +
           (1.0 - dmin(m->compression_artifacts / 10.0, 1.0)) * 0.2 + dmin(m->noise_level / 20.0, 1.0) * 0.15 +
           (1.0 - dmin(m->frequency_anomaly / 100.0, 1.0)) * 0.15;
}

int orc_analyze_frame(const uint8_t* bgr, int h, int w, int64_t row_stride, const orc_params* pp, orc_metrics* out) {
    orc_params dp;
    if (!pp) { orc_default_params(&dp); pp = &dp; }
    if (h < 1 || w < 1 || pp->lbp_points < 1 || pp->lbp_points > 24) return -1;
    size_t n = (size_t)h * (size_t)w;
    uint8_t* gray = (uint8_t*)malloc(n);
    uint8_t* tmp = (uint8_t*)malloc(n);
    if (!gray || !tmp) { free(gray); free(tmp); return -1; }
    memset(out, 0, sizeof(*out));
    out->n_pixels = n;
    orc_bgr2gray(bgr, h, w, row_stride, pp->gray_shift, gray);
    for (size_t i = 0; i < n; ++i) out->gray_sum += gray[i];
    orc_lbp_uniform(gray, h, w, pp->lbp_points, pp->lbp_radius, NULL, out->lbp_hist);
    orc_highpass3(gray, h, w, tmp);
    for (size_t i = 0; i < n; ++i) { out->hp_sum += tmp[i]; out->hp_sumsq += (uint64_t)tmp[i] * tmp[i]; }
    orc_gauss5(gray, h, w, tmp);
    for (size_t i = 0; i < n; ++i) out->absdiff_sum += (uint64_t)abs((int)gray[i] - (int)tmp[i]);
    int ec = orc_canny(gray, h, w, pp->canny_low, pp->canny_high, NULL, NULL);
    if (ec < 0) { free(gray); free(tmp); return -1; }
    out->edge_count = (uint32_t)ec;
    int fh = pp->fft_h > 0 ? pp->fft_h : h, fw = pp->fft_w > 0 ? pp->fft_w : w;
    if (fh > h || fw > w) { free(gray); free(tmp); return -1; }
    int oy = (h - fh) / 2, ox = (w - fw) / 2;
    if (orc_fft_logmag_stats(gray + (size_t)oy * w + ox, fh, fw, w, &out->fft_sum, &out->fft_sumsq, NULL) != 0) {
        free(gray); free(tmp);
        return -1;
    }
    orc_finalize_metrics(out, pp->lbp_points, (uint64_t)fh * (uint64_t)fw);
    free(gray); free(tmp);
    return 0;
}

/* ------------------------------------------------------------------ synthetic frames */
static uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int64_t isin16(int64_t p) {
    p &= 0xFFFF;
    int64_t q = p & 0x7FFF;
    int64_t v = (q * (32768 - q)) >> 14;
    return (p & 0x8000) ? -v : v;
}

static int64_t floor_shr(int64_t v, int s) { return v >= 0 ? (v >> s) : -(((-v) + (((int64_t)1) << s) - 1) >> s); }

int orc_synth_frame(int kind, uint64_t seed, uint32_t video, uint32_t frame, int h, int w, uint8_t* bgr) {
    if (h < 1 || w < 1 || h > 8192 || w > 8192 || frame >= 1024 || video >= (1u << 28) || kind < 0 || kind > 5) return -1;
    int64_t f = frame;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint64_t idx = ((uint64_t)video << 36) | ((uint64_t)frame << 26) | ((uint64_t)y << 13) | (uint64_t)x;
            uint64_t hs = splitmix64(seed * 0xD1342543DE82EF95ull + idx);
            uint8_t* o = bgr + ((size_t)y * w + x) * 3;
            switch (kind) {
                case 0:
                    o[0] = (uint8_t)hs; o[1] = (uint8_t)(hs >> 8); o[2] = (uint8_t)(hs >> 16);
                    break;
                case 1: {
                    int64_t sx = isin16(((int64_t)x * 65536) / 97 + f * 1043);
                    int64_t sy = isin16(((int64_t)y * 65536) / 71 + 16384);
                    int64_t base = 128 + floor_shr(60 * sx * sy, 28);
                    int64_t cx = w / 2 + 3 * f, cy = h / 2 + 2 * f, rad = h / 5;
                    if ((x - cx) * (x - cx) + (y - cy) * (y - cy) < rad * rad) base += 40;
                    for (int ch = 0; ch < 3; ++ch) {
                        int64_t s = 0;
                        for (int k = 0; k < 4; ++k) s += (int64_t)((hs >> (20 * ch + 5 * k)) & 31);
                        int64_t v = base + (((s * 55 + 128) >> 8) - 13) + 10 * (ch - 1);
                        o[ch] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                    }
                    break;
                }
                case 2:
                    o[0] = o[1] = o[2] = (uint8_t)((x + 2 * y + (int)frame) & 255);
                    break;
                case 3:
                    o[0] = 77; o[1] = 150; o[2] = 201;
                    break;
                case 4:
                    o[0] = o[1] = o[2] = (uint8_t)(100 + (hs & 1));
                    break;
                default: {
                    int64_t v = 60 + 90 * ((((x + (int)frame) >> 4) + (y >> 4)) & 1) + ((int64_t)(hs & 3) - 1);
                    o[0] = o[1] = o[2] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                }
            }
        }
    return 0;
}
