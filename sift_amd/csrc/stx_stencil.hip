// Fused stencil pass over the padded gray plane: one LDS tile feeds
//   * uniform LBP (P=8, R=1) codes -> 10-bin histogram      (passive_checker.py:103-104)
//   * 3x3 high-pass, saturated -> sum and sum of squares     (:107-108)
//   * 5x5 Gaussian residual |gray - blur| -> sum             (:111-113)
//   * Sobel + L1 magnitude + non-maximum suppression + double threshold
//     -> two bitmaps (strong, candidate), one bit per pixel  (:116, first half of Canny)
// A 256-thread workgroup owns a 64x128 pixel tile; each wave owns 32 rows, each lane one
// column, and marches down its column with the 5x5 neighbourhood held in registers.  The
// wave's 64 lanes are 64 consecutive pixels, so __ballot() of a per-pixel predicate IS the
// 64-bit bitmap word of that row.
//
// LBP bits are those of the fp64 arithmetic scikit-image performs (every operation rounded
// separately, in its order); an fp32 evaluation with a proven error margin settles the clear
// cases and only near-ties run the fp64 sequence.  This file must be compiled with
// -ffp-contract=off.
#include "stx_internal.h"

namespace stx {

#pragma clang fp contract(off)

constexpr int TILE_W = 64;
constexpr int WAVE_ROWS = 32;       // rows marched by one wave
constexpr int BLOCK_ROWS = 8;       // rows per unrolled block
constexpr int TILE_H = 4 * WAVE_ROWS;
constexpr int LROWS = TILE_H + 4;  // rows y0-2 .. y0+TILE_H+1
constexpr int LPITCH = 72;         // bytes: columns x0-4 .. x0+67
constexpr int LDW = LPITCH / 4;

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64s(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One bilinear sample exactly as skimage's bilinear_interpolation: top/bottom rows first,
// then the vertical blend; returns the LBP bit (sample - centre >= 0).
__device__ __forceinline__ uint32_t lbp_bit(double tl, double tr, double bl, double br, double dr, double dc,
                                            double centre) {
    const double omc = 1.0 - dc;
    const double omr = 1.0 - dr;
    const double top = omc * tl + dc * tr;
    const double bottom = omc * bl + dc * br;
    const double t = omr * top + dr * bottom;
    return (t - centre >= 0.0) ? 1u : 0u;
}

struct StencilArgs {
    const uint8_t* gray;
    int H, W, gpitch, grows, wpr;
    int low, high;
    double rp_neg, rp_pos, cp_pos, cp_neg;  // the four non-integer sample offsets of (P=8,R=1)
    uint32_t* part;  // [frame][tiles][16]: 10 LBP bins, hp sum, hp sum of squares, |gray-blur| sum, gray sum
    unsigned long long* strong;
    unsigned long long* cand;
    uint8_t* codes;  // optional per-pixel LBP codes [frame][H][W]
};

// sign bit of v appended below acc: (acc << 1) | (v >> 31)
__device__ __forceinline__ uint32_t push_sign(uint32_t acc, uint32_t v) { return __builtin_amdgcn_alignbit(acc, v, 31); }

template <bool BORDER, bool DO_LBP, bool CODES>
__device__ __forceinline__ void march(const StencilArgs& a, const uint8_t* tile, int frame, int x0, int y0, int lane,
                                      int wv, uint32_t* red /*[16] per wave*/) {
    const int H = a.H, W = a.W;
    const int x = x0 + lane;
    const int ys = y0 + wv * WAVE_ROWS;
    const int lc = lane + GRAY_XOFF;  // LDS column of pixel x
    const bool xin = !BORDER || x < W;

    // LDS column indices of x-1, x, x+1 in the Canny (replicated border) view
    int cc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) cc[j] = BORDER ? (iclamp(x + j - 1, 0, W - 1) - x0 + GRAY_XOFF) : (lc + j - 1);

    // per-column LBP weights
    double dcP = 0, dcN = 0;
    if (DO_LBP) {
        const double cfp = (double)x + a.cp_pos;
        dcP = cfp - floor(cfp);
        const double cfn = (double)x + a.cp_neg;
        dcN = cfn - floor(cfn);
    }
    const float wcP = (float)dcP, wcN = (float)dcN;

    // Sobel magnitude of the Canny view at any pixel of the strip's neighbourhood (dynamic LDS
    // indices; used for the two halo columns only)
    auto mag_at = [&](int yy, int xx) -> int {
        if (BORDER && (yy < 0 || yy >= H || xx < 0 || xx >= W)) return 0;
        int v[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ry = BORDER ? iclamp(yy + r - 1, 0, H - 1) : yy + r - 1;
                const int rx = BORDER ? iclamp(xx + q - 1, 0, W - 1) : xx + q - 1;
                v[r][q] = tile[(ry - y0 + 2) * LPITCH + (rx - x0 + GRAY_XOFF)];
            }
        const int gx = (v[0][2] - v[0][0]) + 2 * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0]);
        const int gy = (v[2][0] - v[0][0]) + 2 * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2]);
        return abs(gx) + abs(gy);
    };
    // The magnitudes left of lane 0 and right of lane 63 belong to the neighbouring strips: lane l
    // computes them once for row ys-1+l, the marching loop fetches them with a lane read.
    int haloL = 0, haloR = 0;
    if (lane < WAVE_ROWS + 2) {
        haloL = mag_at(ys - 1 + lane, x0 - 1);
        haloR = mag_at(ys - 1 + lane, x0 + TILE_W);
    }

    // fp32 row weights of the LBP samples above / below row ys+lane (lanes 0..31)
    int rowWN = 0, rowWP = 0;
    if (DO_LBP) {
        const double rfn = (double)(ys + lane) + a.rp_neg;
        rowWN = __float_as_int((float)(rfn - floor(rfn)));
        const double rfp = (double)(ys + lane) + a.rp_pos;
        rowWP = __float_as_int((float)(rfp - floor(rfp)));
    }

    // sliding state
    int g[5][5];   // reflected view, rows y-2..y+2, cols x-2..x+2
    int hg[5];     // horizontal [1 4 6 4 1] of those rows
    int rs[5];     // g[r][1]+g[r][2]+g[r][3]
    int sd[3];     // horizontal Sobel difference at x of rows y, y+1, y+2 (Canny view)
    int ss[3];     // horizontal Sobel smoothing at x of the same rows
    int mag[3][3]; // L1 magnitude rows y-1..y+1, cols x-1..x+1 (0 outside the image)
    int dxc[2], dyc[2];  // gradient at (y,x) and (y+1,x)

    // rel: row index relative to ys-2 (0 .. WAVE_ROWS+3); off: 32-bit LDS byte offset of (that
    // row, column x-2) so that inside an unrolled block the row addresses differ by constants only
    const uint32_t wave_off = (uint32_t)((wv * WAVE_ROWS) * LPITCH + lc - 2);
    auto load_row = [&](int rel, uint32_t off, int (&gr)[5], int& hsum, int& rsum, int& d, int& sm) {
#pragma unroll
        for (int j = 0; j < 5; ++j) gr[j] = tile[off + j];
        rsum = gr[1] + gr[2] + gr[3];
        hsum = gr[0] + gr[4] + 4 * (gr[1] + gr[3]) + 6 * gr[2];
        int c3[3];
        if (BORDER) {
            const int r = ys - 2 + rel;
            const uint8_t* rowc = tile + (iclamp(r, 0, H - 1) - y0 + 2) * LPITCH;
#pragma unroll
            for (int j = 0; j < 3; ++j) c3[j] = rowc[cc[j]];
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) c3[j] = gr[j + 1];
        }
        d = c3[2] - c3[0];
        sm = c3[0] + 2 * c3[1] + c3[2];
    };
    // gradient and magnitudes of row r (rows r-1, r, r+1 are d0/s0, d1, d2/s2); the neighbours'
    // magnitudes come from the adjacent lanes
    auto sobel_row = [&](int r, int d0, int d1, int d2, int s0, int s2, int (&m)[3], int& gx, int& gy) {
        gx = d0 + 2 * d1 + d2;
        gy = s2 - s0;
        // |gy| + |gx|: s0, s2 are non-negative, so v_sad_u32 gives |s2 - s0| and adds |gx| for free
        int mm = (int)__usad((uint32_t)s2, (uint32_t)s0, (uint32_t)abs(gx));
        if (BORDER && (r < 0 || r >= H || x >= W)) mm = 0;
        const int rel = r - (ys - 1);
        const int hl = __builtin_amdgcn_readlane(haloL, rel);
        const int hr = __builtin_amdgcn_readlane(haloR, rel);
        m[0] = __builtin_amdgcn_update_dpp(hl, mm, 0x138, 0xf, 0xf, false);  // wave_shr:1, lane 0 keeps hl
        m[1] = mm;
        m[2] = __builtin_amdgcn_update_dpp(hr, mm, 0x130, 0xf, 0xf, false);  // wave_shl:1, lane 63 keeps hr
    };

    // prologue: rows ys-2 .. ys+1
    int dm2, sm2, dm1, sm1;
    load_row(0, wave_off, g[0], hg[0], rs[0], dm2, sm2);
    load_row(1, wave_off + LPITCH, g[1], hg[1], rs[1], dm1, sm1);
    load_row(2, wave_off + 2 * LPITCH, g[2], hg[2], rs[2], sd[0], ss[0]);
    load_row(3, wave_off + 3 * LPITCH, g[3], hg[3], rs[3], sd[1], ss[1]);
    {
        int gx, gy;
        sobel_row(ys - 1, dm2, dm1, sd[0], sm2, ss[0], mag[0], gx, gy);
        sobel_row(ys, dm1, sd[0], sd[1], sm1, ss[1], mag[1], dxc[0], dyc[0]);
    }

    // LBP row state, sliding with the window: the zero-padded 3 neighbours of rows y-1..y+1, the centre
    // column as float, and the row-wise interpolations of the left cell (columns x-1,x; weight dcN)
    // and the right cell (columns x,x+1; weight dcP).  Each image row is converted and interpolated once
    // and serves the three centre rows that touch it.
    int nm[3][3];
    float fcs[3], lLs[3], lRs[3];
    auto lbp_row = [&](int yy, const int (&gr)[5], int (&nrow)[3], float& fc, float& l_left, float& l_right) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            int v = gr[q + 1];
            if (BORDER) {
                const int xx = x + q - 1;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) v = 0;
            }
            nrow[q] = v;
        }
        const float f0 = (float)nrow[0], f1 = (float)nrow[1], f2 = (float)nrow[2];
        fc = f1;
        l_left = __builtin_fmaf(wcN, f1 - f0, f0);
        l_right = __builtin_fmaf(wcP, f2 - f1, f1);
    };
    if (DO_LBP) {
        lbp_row(ys - 1, g[1], nm[0], fcs[0], lLs[0], lRs[0]);
        lbp_row(ys, g[2], nm[1], fcs[1], lLs[1], lRs[1]);
    }

    unsigned long long hist6 = 0;  // ten 6-bit counters (at most 32 pixels per lane)
    uint32_t hp_s = 0, hp_q = 0, ad_s = 0, gr_s = 0;

#pragma unroll 1
    for (int blk = 0; blk < WAVE_ROWS / BLOCK_ROWS; ++blk) {
        const int yb = ys + blk * BLOCK_ROWS;
        if (BORDER && yb >= H) break;  // wave-uniform: nothing left in the image
        const uint32_t blk_off = wave_off + (uint32_t)(blk * BLOCK_ROWS * LPITCH);
        // LBP state of this block: per row one byte of SIGN bits (1 = sample below the centre), pushed
        // in sample order 0..7 (so sample s sits at bit 7-s), and four "needs fp64" flags
        // (diagonal sample 2k+1 at bit 3-k).
        unsigned long long sign64 = 0;
        uint32_t pend = 0;
        unsigned long long bm_s = 0, bm_c = 0;
#pragma unroll
        for (int i = 0; i < BLOCK_ROWS; ++i) {
            const int y = yb + i;
            load_row(blk * BLOCK_ROWS + i + 4, blk_off + (i + 4) * LPITCH, g[4], hg[4], rs[4], sd[2], ss[2]);
            sobel_row(y + 1, sd[0], sd[1], sd[2], ss[0], ss[2], mag[2], dxc[1], dyc[1]);
            const bool valid = !BORDER || (xin && (y < H));
            const int c = g[2][2];

            // ---- high-pass: 8c - sum(neighbours) = 9c - sum(3x3), saturated to u8
            {
                const int v = iclamp(9 * c - (rs[1] + rs[2] + rs[3]), 0, 255);
                if (valid) { hp_s += v; hp_q += v * v; gr_s += c; }
            }
            // ---- Gaussian residual
            {
                const int S = hg[0] + hg[4] + 4 * (hg[1] + hg[3]) + 6 * hg[2];
                const int blur = (S + 128) >> 8;
                if (valid) ad_s += abs(c - blur);
            }
            // ---- LBP (P=8, R=1), first stage
            if (DO_LBP) {
                lbp_row(y + 1, g[3], nm[2], fcs[2], lLs[2], lRs[2]);  // rows y-1 and y slid in earlier
                // the row weights are wave-uniform: lane l computed those of row ys+l before the loop
                const float wrN = __int_as_float(__builtin_amdgcn_readlane(rowWN, blk * BLOCK_ROWS + i));
                const float wrP = __int_as_float(__builtin_amdgcn_readlane(rowWP, blk * BLOCK_ROWS + i));
                const int (&n)[3][3] = nm;
                const float (&lL)[3] = lLs;
                const float (&lR)[3] = lRs;
                const int ce = n[1][1];
                // Diagonal samples in fp32: the value is within 2e-4 of the exact bilinear value and the
                // fp64 sequence within 1e-12, so |value - centre| >= 1e-3 settles the sign; anything
                // closer is queued for the order-exact fp64 evaluation after this block.
                const float fc = fcs[1];
                const float d1 = __builtin_fmaf(wrN, lR[1] - lR[0], lR[0]) - fc;  // sample 1 (-.7,+.7)
                const float d3 = __builtin_fmaf(wrN, lL[1] - lL[0], lL[0]) - fc;  // sample 3 (-.7,-.7)
                const float d5 = __builtin_fmaf(wrP, lL[2] - lL[1], lL[1]) - fc;  // sample 5 (+.7,-.7)
                const float d7 = __builtin_fmaf(wrP, lR[2] - lR[1], lR[1]) - fc;  // sample 7 (+.7,+.7)
                // The axis samples have integer coordinates: the interpolation returns the pixel itself.
                uint32_t sg = 0;
                sg = push_sign(sg, (uint32_t)(n[1][2] - ce));      // sample 0 (0,+1)
                sg = push_sign(sg, __float_as_uint(d1));
                sg = push_sign(sg, (uint32_t)(n[0][1] - ce));      // sample 2 (-1,0)
                sg = push_sign(sg, __float_as_uint(d3));
                sg = push_sign(sg, (uint32_t)(n[1][0] - ce));      // sample 4 (0,-1)
                sg = push_sign(sg, __float_as_uint(d5));
                sg = push_sign(sg, (uint32_t)(n[2][1] - ce));      // sample 6 (+1,0)
                sg = push_sign(sg, __float_as_uint(d7));
                uint32_t un = 0;
                un = push_sign(un, __float_as_uint(__builtin_fabsf(d1) - 1e-3f));
                un = push_sign(un, __float_as_uint(__builtin_fabsf(d3) - 1e-3f));
                un = push_sign(un, __float_as_uint(__builtin_fabsf(d5) - 1e-3f));
                un = push_sign(un, __float_as_uint(__builtin_fabsf(d7) - 1e-3f));
                if (!valid) un = 0;
                sign64 |= (unsigned long long)sg << (8 * i);
                pend |= un << (4 * i);
            }
            // ---- Canny non-maximum suppression + double threshold, branch-free: pick the two
            // neighbours along the quantised gradient direction, then two compares.  The second
            // neighbour is compared with >= for horizontal / vertical gradients and with > for
            // diagonal ones; for integers m >= b is m > b-1.
            {
                const int m = mag[1][1];
                unsigned long long sb = 0, cb = 0;
                // most rows of a smooth image have no pixel above the low threshold: one wave-uniform
                // test skips the whole selection for them
                if (__any(m > a.low && valid)) {
                    const int xs = dxc[0], ysg = dyc[0];
                    const int ax = abs(xs), ay = abs(ysg) << 15;
                    const int tg22x = ax * 13573;
                    const int tg67x = tg22x + (ax << 16);
                    const bool horiz = ay < tg22x;
                    const bool vert = ay > tg67x;
                    const bool opp = (xs ^ ysg) < 0;
                    int na = opp ? mag[0][2] : mag[0][0];
                    int nb = opp ? mag[2][0] : mag[2][2];
                    na = vert ? mag[0][1] : na;
                    nb = vert ? mag[2][1] - 1 : nb;
                    na = horiz ? mag[1][0] : na;
                    nb = horiz ? mag[1][2] - 1 : nb;
                    const bool keep = (m > na) && (m > nb) && valid;
                    sb = __ballot(keep && m > a.high);
                    cb = __ballot(keep && m > a.low);
                }
                // park the two bitmap words of row i in lane i; lanes 0..7 store them after the block
                if (lane == i) { bm_s = sb; bm_c = cb; }
            }
            // ---- slide down one row
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int j = 0; j < 5; ++j) g[r][j] = g[r + 1][j];
                hg[r] = hg[r + 1];
                rs[r] = rs[r + 1];
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                sd[r] = sd[r + 1];
                ss[r] = ss[r + 1];
#pragma unroll
                for (int j = 0; j < 3; ++j) mag[r][j] = mag[r + 1][j];
            }
            dxc[0] = dxc[1];
            dyc[0] = dyc[1];
            if (DO_LBP) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) nm[r][q] = nm[r + 1][q];
                    fcs[r] = fcs[r + 1];
                    lLs[r] = lLs[r + 1];
                    lRs[r] = lRs[r + 1];
                }
            }
        }
        if (lane < BLOCK_ROWS && (!BORDER || yb + lane < H)) {
            const int64_t wi = ((int64_t)frame * a.wpr + (x0 >> 6)) * H + yb + lane;  // [frame][word column][row]
            a.strong[wi] = bm_s;
            a.cand[wi] = bm_c;
        }
        if (DO_LBP) {
            // ---- LBP second stage: near-ties, one (row, sample) per lane and iteration, in the fp64
            // arithmetic of skimage's bilinear_interpolation (operands re-read from the LDS tile)
            auto lbp_px = [&](int yy, int xx) -> double {
                if (BORDER && (yy < 0 || yy >= H || xx < 0 || xx >= W)) return 0.0;
                return (double)tile[(yy - y0 + 2) * LPITCH + (xx - x0 + GRAY_XOFF)];
            };
            while (__any(pend != 0)) {
                if (pend != 0) {
                    const int b = __ffs(pend) - 1;
                    pend &= pend - 1;
                    const int i = b >> 2, k = 3 - (b & 3);  // diagonal sample 2k+1 of row i
                    const int y = yb + i;
                    const int r0 = (k >= 2) ? y : y - 1;             // upper row of the 2x2 cell
                    const int c0 = (k == 1 || k == 2) ? x - 1 : x;  // left column of the cell
                    const double rf = (double)y + ((k >= 2) ? a.rp_pos : a.rp_neg);
                    const double dr = rf - floor(rf);
                    const double dc = (k == 1 || k == 2) ? dcN : dcP;
                    const uint32_t bit = lbp_bit(lbp_px(r0, c0), lbp_px(r0, c0 + 1), lbp_px(r0 + 1, c0),
                                                 lbp_px(r0 + 1, c0 + 1), dr, dc, lbp_px(y, x));
                    const int pos = 8 * i + 7 - (2 * k + 1);
                    sign64 = (sign64 & ~(1ull << pos)) | ((unsigned long long)(bit ^ 1u) << pos);
                }
            }
            // ---- codes and histogram of this block (bit order is reversed, which changes neither
            // the number of 0/1 transitions between consecutive samples nor the number of ones)
#pragma unroll
            for (int i = 0; i < BLOCK_ROWS; ++i) {
                const int y = yb + i;
                const uint32_t bits = ~(uint32_t)(sign64 >> (8 * i)) & 0xFFu;
                const int changes = __popc((bits ^ (bits >> 1)) & 0x7Fu);
                const int code = (changes <= 2) ? __popc(bits) : 9;
                if (!BORDER || (xin && y < H)) {
                    hist6 += 1ull << (6 * code);
                    if (CODES) a.codes[((int64_t)frame * H + y) * W + x] = (uint8_t)code;
                }
            }
        }
    }

    // ---- wave reduction: 10 histogram bins (two words of five 12-bit fields), four sums
    unsigned long long lo = 0, hi = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        lo |= ((hist6 >> (6 * b)) & 63ull) << (12 * b);
        hi |= ((hist6 >> (6 * (b + 5))) & 63ull) << (12 * b);
    }
    lo = wave_sum_u64s(lo);
    hi = wave_sum_u64s(hi);
    hp_s = wave_sum_u32(hp_s);
    hp_q = wave_sum_u32(hp_q);
    ad_s = wave_sum_u32(ad_s);
    gr_s = wave_sum_u32(gr_s);
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            red[b] = (uint32_t)((lo >> (12 * b)) & 0xFFFull);
            red[b + 5] = (uint32_t)((hi >> (12 * b)) & 0xFFFull);
        }
        red[10] = hp_s;
        red[11] = hp_q;
        red[12] = ad_s;
        red[13] = gr_s;
    }
}

template <bool DO_LBP, bool CODES>
__global__ __launch_bounds__(256) void stencil_kernel(StencilArgs a) {
    __shared__ uint32_t tile[LROWS * LDW];
    __shared__ uint32_t red[4][16];
    const int frame = blockIdx.z;
    const int x0 = blockIdx.x * TILE_W;
    const int y0 = blockIdx.y * TILE_H;
    const int tid = threadIdx.x;

    const uint8_t* gp = a.gray + (int64_t)frame * a.grows * a.gpitch;
    for (int i = tid; i < LROWS * LDW; i += 256) {
        const int row = i / LDW;
        const int col = i - row * LDW;
        int gy = y0 + row;  // padded-plane row of global row y0-2+row
        gy = gy < a.grows ? gy : a.grows - 1;
        tile[i] = *reinterpret_cast<const uint32_t*>(gp + (int64_t)gy * a.gpitch + x0 + col * 4);
    }
    if (tid < 64) red[tid >> 4][tid & 15] = 0;
    __syncthreads();

    const int lane = tid & 63, wv = tid >> 6;
    const int ys = y0 + wv * WAVE_ROWS;
    if (ys < a.H) {
        // interior: the wave's strip and its two-pixel halo lie inside the image
        const bool interior = (x0 >= 2) && (ys >= 2) && (x0 + TILE_W + 2 <= a.W) && (ys + WAVE_ROWS + 2 <= a.H);
        if (interior)
            march<false, DO_LBP, CODES>(a, reinterpret_cast<const uint8_t*>(tile), frame, x0, y0, lane, wv, red[wv]);
        else
            march<true, DO_LBP, CODES>(a, reinterpret_cast<const uint8_t*>(tile), frame, x0, y0, lane, wv, red[wv]);
    }
    __syncthreads();
    // per-tile partial sums, one slot per tile: no atomics, summed in a fixed order later
    if (tid < 16) {
        const uint32_t v = tid < 14 ? red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] : 0u;
        const int64_t tileid = ((int64_t)frame * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.part[tileid * 16 + tid] = (tid < 10 && !DO_LBP) ? 0u : v;
    }
}

void launch_stencil(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, bool do_lbp81,
                    const LbpOffsets& off, int low, int high, uint32_t* part, unsigned long long* strong,
                    unsigned long long* cand, uint8_t* codes_out) {
    StencilArgs a;
    a.gray = gray;
    a.H = g.H; a.W = g.W; a.gpitch = g.gpitch; a.grows = g.grows; a.wpr = g.wpr;
    a.low = low; a.high = high;
    // (P=8,R=1): samples 1,3 sit above the centre (rp<0), 5,7 below; 1,7 right (cp>0), 3,5 left
    a.rp_neg = off.rp[1]; a.rp_pos = off.rp[5]; a.cp_pos = off.cp[1]; a.cp_neg = off.cp[3];
    a.part = part; a.strong = strong; a.cand = cand; a.codes = codes_out;
    dim3 grid(g.tiles_x, g.tiles_y, n_frames);
    if (!do_lbp81)
        hipLaunchKernelGGL((stencil_kernel<false, false>), grid, dim3(256), 0, s, a);
    else if (codes_out)
        hipLaunchKernelGGL((stencil_kernel<true, true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((stencil_kernel<true, false>), grid, dim3(256), 0, s, a);
}

}  // namespace stx
