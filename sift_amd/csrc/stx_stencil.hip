// Fused stencil pass: one LDS tile of gray levels -- read from the padded gray plane, or converted from the BGR frame
// on the way in (cv2.cvtColor(BGR2GRAY), passive_checker.py:100; the gray plane is then this pass's by-product for the
// FFT) -- feeds
//   * uniform LBP (P=8, R=1) codes -> 10-bin histogram      (passive_checker.py:103-104)
//   * 3x3 high-pass, saturated -> sum and sum of squares     (:107-108)
//   * 5x5 Gaussian residual |gray - blur| -> sum             (:111-113)
//   * Sobel + L1 magnitude + non-maximum suppression + double threshold
//     -> two bitmaps (strong, candidate), one bit per pixel  (:116, first half of Canny)
// A 256-thread workgroup owns a 128x128 pixel tile; each wave owns 32 rows and marches down
// them.  A lane owns TWO adjacent pixel columns and keeps them as packed 16-bit pairs, so the
// integer stencils run on v_pk_*_u16 / v_sad_u16 / v_dot2_u32_u16 at two pixels per
// instruction.  Every intermediate fits 16 bits: the 5x5 binomial sum is at most 255*256,
// Sobel components at most 1020, the L1 magnitude at most 2040.
//
// The Canny bits of a pixel column are collected down the strip (one 32-bit word per column
// and strip), which is the layout the hysteresis kernel works on: [frame][64-row word][column].
//
// LBP: the four axis samples are the neighbouring pixels themselves; the four diagonal samples are bilinear.
// skimage rounds the sample offsets to five decimals, so a diagonal sample sits at (+-0.70711, +-0.70711) and, the
// bilinear weights adding up to one,
//     sample - centre = 0.70711e-5 * ( 70711 * (D - c) + 29289 * ((A - c) + (B - c)) )
// with D the diagonal neighbour and A, B the two axis neighbours of that quadrant: an integer N decides the sign, and
// N = 0 needs D = c and A + B = 2c (70711 and 29289 are coprime).  29/70 is a continued-fraction convergent of
// 29289/70711: N' = 70 (D - c) + 29 (A + B - 2c) fits 16 bits and is off from 70 N / 70711 by at most
// 0.00536 * 510 < 2.74, so |N'| >= 3 has the sign of N -- which is at least 263 in magnitude then, a sample 1.9e-3
// gray levels from the centre, far outside anything fp64 rounding does.  N' in [-2, 2] (exact ties, mostly: about 3 %
// of the pixels of a low-contrast image) goes through a wave-wide queue in LDS and is re-evaluated in the order-exact
// fp64 sequence by whichever lane is free, so the slow path costs ceil(queued / 64) rounds per block of rows instead
// of the worst lane's count.  This file must be compiled with -ffp-contract=off.
//
// Instruction choice: on gfx950 a plain 32-bit add / sub / and / or / xor / right shift whose operands are vector
// registers or literals issues in about half the time of any packed, three-operand, permute or compare instruction.
// Both pixels of a lane live in the 16-bit halves of one register, so wherever a half cannot carry or borrow into
// the other (sums of small non-negative values; a - b with bit 15 of a set and b < 2^15) the code uses the plain
// instruction on the whole word.  A comparison a >= b is bit 15 of (a | 0x8000) - b.
#include <utility>
#include <vector>

#include "stx_internal.h"

namespace stx {

#pragma clang fp contract(off)

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));

// waves per SIMD the stencil kernel is compiled for: 128 registers each (3 waves measured 10 % slower, 5 spill)
#define STX_STENCIL_WAVES 4
constexpr int TILE_W = 128;        // 64 lanes x 2 pixels
constexpr int WAVE_ROWS = 32;      // rows marched by one wave
constexpr int BLOCK_ROWS = 8;      // rows per unrolled block
constexpr int TILE_H = 4 * WAVE_ROWS;
constexpr int LROWS = TILE_H + 4;  // rows y0-2 .. y0+TILE_H+1
constexpr int LPITCH = TILE_W + 8; // bytes: columns x0-4 .. x0+131
constexpr int LDW = LPITCH / 4;
constexpr int LUT_INVALID = 256;       // histogram table slots 256..511 add nothing
constexpr int LUT_ENTRIES = 512;
constexpr int DENSE_MIN = BLOCK_ROWS * TILE_W / 8;  // a block with more undecided pixels is re-evaluated lane by lane
constexpr int QUEUE_ITEMS = (WAVE_ROWS / BLOCK_ROWS) * DENSE_MIN;  // what the sparse blocks of a strip can queue

// ---- packed 16-bit helpers (operands travel as plain 32-bit words) ----
__device__ __forceinline__ us2 asU(uint32_t v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ ss2 asS(uint32_t v) { return __builtin_bit_cast(ss2, v); }
__device__ __forceinline__ uint32_t word(us2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t word(ss2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return word(asU(a) + asU(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return word(asS(a) - asS(b)); }
template <int K>
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t c) {  // a*K + c in one v_pk_mad_u16 (K is an inline constant)
    static_assert(K >= 0 && K <= 64, "inline constant");
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "n"(K), "v"(c));
    return d;
}
// a*k + c per half with the multiplier in a register (constants above 64 cannot be inline operands)
__device__ __forceinline__ uint32_t pk_mad_r(uint32_t a, uint32_t k, uint32_t c) { return word(asU(a) * asU(k) + asU(c)); }
// the same as one instruction whatever the multiplier is (left to itself the compiler turns a multiplication by a
// power of two, or by minus one, into a shift and a subtraction: two instructions of the same cost each)
__device__ __forceinline__ uint32_t pk_mad_op(uint32_t a, uint32_t k, uint32_t c) {
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(k), "v"(c));
    return d;
}
template <int K>
__device__ __forceinline__ uint32_t pk_mul(uint32_t a) {
    const us2 k = {(unsigned short)K, (unsigned short)K};
    return word(asU(a) * k);
}
template <int N>
__device__ __forceinline__ uint32_t pk_shr(uint32_t a) {
    const us2 n = {(unsigned short)N, (unsigned short)N};
    return word(asU(a) >> n);
}
__device__ __forceinline__ uint32_t pk_sign_mask(uint32_t a) {  // 0xFFFF where the element is negative
    // opaque to the optimiser on purpose: seen through, the mask selects below turn into per-half compares
    uint32_t d;
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ uint32_t pk_abs(uint32_t a) { return word(__builtin_elementwise_max(asS(a), -asS(a))); }
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b) {
    return word(__builtin_elementwise_sub_sat(asU(a), asU(b)));
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return word(__builtin_elementwise_min(asU(a), asU(b))); }
__device__ __forceinline__ uint32_t pk_min_i(uint32_t a, uint32_t b) { return word(__builtin_elementwise_min(asS(a), asS(b))); }
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (mask & a) | (~mask & b); }
// v_bfi_b32 with a constant mask held in a scalar register, as written (the compiler prefers a shift, an and with a
// literal and a share of a three-way or: two and a third instructions for one)
template <uint32_t MASK>
__device__ __forceinline__ uint32_t bfi_lit(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(MASK), "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// |a.lo - b.lo| + |a.hi - b.hi| + acc on unsigned 16-bit halves
__device__ __forceinline__ uint32_t sad_u16(uint32_t a, uint32_t b, uint32_t acc) {
    uint32_t d;
    asm("v_sad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}
// a.lo + a.hi + acc (the sum of absolute differences against an inline zero: no register holds the zero)
__device__ __forceinline__ uint32_t sum_u16(uint32_t a, uint32_t acc) {
    uint32_t d;
    asm("v_sad_u16 %0, %1, 0, %2" : "=v"(d) : "v"(a), "v"(acc));
    return d;
}
__device__ __forceinline__ uint32_t dot2(uint32_t a, uint32_t b, uint32_t acc) {
    return __builtin_amdgcn_udot2(asU(a), asU(b), acc, false);
}
// sign bit of v appended below acc: (acc << 1) | (v >> 31)
__device__ __forceinline__ uint32_t push_sign(uint32_t acc, uint32_t v) { return __builtin_amdgcn_alignbit(acc, v, 31); }

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sums of nine words over the wave, valid in lane 63: shifts by 1, 2, 4, 8 inside each row of 16 lanes, then lane 15
// of a row into the next row, then lane 31 into the upper half (a lane without a source adds zero, a row that is
// masked out keeps its value).  Written out as one block of v_add_u32_dpp: the compiler makes a move and an add of
// each step, and inside the block consecutive uses of a register are nine instructions apart, which covers the wait
// a data-parallel read needs after a write of the same register (the s_nop covers whatever wrote them last).
__device__ __forceinline__ void wave_total9(uint32_t (&w)[9]) {
#define STX_DPP_STEP(ctrl)                                                                                          \
    "v_add_u32_dpp %0, %0, %0 " ctrl "\n v_add_u32_dpp %1, %1, %1 " ctrl "\n v_add_u32_dpp %2, %2, %2 " ctrl "\n"    \
    "v_add_u32_dpp %3, %3, %3 " ctrl "\n v_add_u32_dpp %4, %4, %4 " ctrl "\n v_add_u32_dpp %5, %5, %5 " ctrl "\n"    \
    "v_add_u32_dpp %6, %6, %6 " ctrl "\n v_add_u32_dpp %7, %7, %7 " ctrl "\n v_add_u32_dpp %8, %8, %8 " ctrl "\n"
    asm("s_nop 1\n"
        STX_DPP_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        STX_DPP_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        STX_DPP_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        STX_DPP_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        STX_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
        STX_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]));
#undef STX_DPP_STEP
}

// Inclusive prefix sum over the wave (lane 63 ends up with the total), the same six steps on one word; the s_nop
// between two steps is the wait a data-parallel read needs after a write of the register it reads.
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {
    asm("s_nop 1\n"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 0\n"
        : "+v"(v));
    return v;
}

__device__ __forceinline__ int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Lanes of one wave exchange data through LDS without a barrier instruction (a wave's LDS operations
// execute in issue order); this wavefront-scope fence only keeps the COMPILER from moving a lane's load
// above its own earlier store.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One bilinear sample exactly as skimage's bilinear_interpolation: top/bottom rows first,
// then the vertical blend; returns the LBP bit (sample - centre >= 0).
__host__ __device__ __forceinline__ uint32_t lbp_bit(double tl, double tr, double bl, double br, double dr, double dc,
                                            double centre) {
    const double omc = 1.0 - dc;
    const double omr = 1.0 - dr;
    const double top = omc * tl + dc * tr;
    const double bottom = omc * bl + dc * br;
    const double t = omr * top + dr * bottom;
    return (t - centre >= 0.0) ? 1u : 0u;
}

// LBP code of a table entry: the position of its one-hot counter
__device__ __forceinline__ uint8_t code_of(uint2 e) { return (uint8_t)(e.x ? (__ffs(e.x) - 1) / 6 : 5 + (__ffs(e.y) - 1) / 6); }

struct StencilArgs {
    const uint8_t* gray;
    // FROM_BGR: the frames themselves, the conversion weights scaled to a 16-bit shift and split into high and low
    // bytes (B, G, R in bytes 0..2), and the plane the converted pixels are written to
    const uint8_t* bgr;
    int64_t row_stride, frame_stride;
    uint32_t w_hi, w_lo;
    uint8_t* gray_w;
    int H, W, gpitch, grows, bm_words, tiles_x, tiles_y;
    uint32_t low2, high2;                   // thresholds, clamped to [-1, 4095], in both halves
    double rp_neg, rp_pos, cp_pos, cp_neg;  // the four non-integer sample offsets of (P=8,R=1)
    uint32_t* part;    // [frame][tiles][16]: 10 LBP bins, hp sum, hp sum of squares, |gray-blur| sum, gray sum
    uint32_t* strong;  // [frame][bm_words][W] 64-bit words, written as 32-bit halves (one per 32-row strip)
    uint32_t* cand;
    uint8_t* codes;    // optional per-pixel LBP codes [frame][H][W]
    // Outcome of the four diagonal samples when a pixel and its eight neighbours are all equal: the fp64
    // blend of four equal values is not always that value, and what it rounds to depends only on the value
    // and on the fractional sample offsets, which take a handful of distinct values over all coordinates.
    // One blob (one pointer: kernel arguments live in scalar registers for the whole kernel):
    //   [0, TIE_COORDS)              row y    -> class of the row's two fractional offsets
    //   [TIE_COORDS, 2*TIE_COORDS)   column x -> class of the column's two fractional offsets
    //   [2*TIE_COORDS]               number of column classes
    //   [TIE_LUT_OFF ...)            [row class][column class][gray level] -> LBP bits of samples 1,3,5,7 (bit 3 .. 0)
    const uint8_t* tie;
    // [frame][ceil(W/64)][bm_words] bytes, zeroed by the caller: set to 1 for every 64x64 hysteresis tile that
    // holds a strong pixel (the first hysteresis pass skips tiles with none in or around them unseen)
    uint8_t* sflag;
    int bm_tiles;
};
constexpr int TIE_LUT_OFF = 2 * TIE_COORDS + 256;
constexpr uint32_t HH = 0x80008000u;  // bit 15 of both halves
// A pixel's eight LBP bits (1 = sample >= centre) sit at bits 10..3 of its half, which makes the half the byte
// offset of its entry in the histogram table: samples 1,3,5,7 at bits 10..7, samples 0,2,4,6 at bits 6..3.
constexpr uint32_t TIE_BAND = 0x00050005u;  // N' + 2 in [0, 5) is undecided
// an undecided pixel gets bit 11 of its half as well: table offset 2048 and up, the entries that add nothing
constexpr int UNDECIDED_SHIFT = 4;

// byte selectors for v_perm_b32 over {hi source, lo source}: 0..3 = lo bytes, 4..7 = hi bytes, 0x0c = zero
constexpr uint32_t SEL_PAIR = 0x0c010c00;   // (b0, b1) of the low source
constexpr uint32_t SEL_CROSS = 0x0c040c01;  // (b1 of the low source, b0 of the high source)

// Everything derived from one image row for this lane's two columns x, x+1.
struct RowD {
    uint32_t hg;            // horizontal [1 4 6 4 1] (reflected border)
    uint32_t rs;            // horizontal 3-sum (reflected border)
    uint32_t z0;            // the pixel pair (x,x+1), zero outside the image (LBP view)
    uint32_t zm, zp;        // LBP view of the pairs (x-1,x) and (x+1,x+2)
    // linear parts of the diagonal samples: with rows r-1, r, r+1 and the pairs m = (x-1,x), p = (x+1,x+2),
    //   rp = 70 p + 29 z0, rm = 70 m + 29 z0      (contribution of a row above / below the centre row)
    //   gp = 29 p - 128 z0 + 0x8002, gm likewise    (contribution of the centre row itself)
    // so that rp[r-1] + gp[r] = 0x8000 + 2 + N' of sample 1 and so on; every half stays inside [0x82, 0xff82].
    uint32_t rp, rm, gp, gm;
    uint32_t d, sm;         // horizontal Sobel difference and smoothing (replicated border, Canny view)
};

// RAW: the Canny bits are the raw double threshold (magnitude above low / above high) without non-maximum
// suppression; the suppression then runs on the few pixels a flood of these maps reaches (nms_sparse_kernel).
template <bool BORDER, bool DO_LBP, bool CODES, bool RAW>
__device__ __forceinline__ void march(const StencilArgs& a, const uint8_t* tile, const uint2* lut,
                                      uint16_t* wq /*[QUEUE_ITEMS] per wave*/, uint32_t* wpend /*[blocks][64] per wave*/,
                                      int frame, int x0, int y0, int lane, int wv, uint32_t* red /*[16] per wave*/) {
    const int H = a.H, W = a.W;
    const int x = x0 + 2 * lane;  // this lane's pixels: x (low half) and x+1 (high half)
    const int ys = y0 + wv * WAVE_ROWS;
    const uint32_t vm = BORDER ? ((x < W ? 0xFFFFu : 0u) | (x + 1 < W ? 0xFFFF0000u : 0u)) : 0xFFFFFFFFu;

    // The tile holds BORDER_REFLECT_101 copies outside the image; border strips make the other two views from the pixel
    // pairs (x-1,x) (x,x+1) (x+1,x+2) every row builds anyway: the LBP view (zeros outside) by one AND each with a
    // per-lane mask, the Canny view (the border pixel repeated) by one byte permute each over two neighbouring pairs,
    // with a per-lane selector that is the identity away from the image's left and right edge.
    uint32_t selC[3] = {0, 0, 0}, maskZ[3] = {0, 0, 0};
    if (BORDER) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int e0 = x + k - 1, e1 = e0 + 1;
            // sources of pair k: k = 0: {low = (x-1,x), high = (x,x+1)}, k = 1, 2: {low = (x,x+1), high = (x+1,x+2)};
            // a pixel `off` columns right of the low pair's first one sits in byte 0, 2 or 6 of the two registers
            const int base = k == 0 ? x - 1 : x;
            auto byte_of = [&](int e) {
                const int off = iclamp(e, 0, W - 1) - base;
                return off == 1 ? 2 : (off == 2 ? 6 : 0);  // (out of reach only for lanes wholly outside the image)
            };
            selC[k] = (uint32_t)byte_of(e0) | 0x0c00u | ((uint32_t)byte_of(e1) << 16) | 0x0c000000u;
            maskZ[k] = ((e0 >= 0 && e0 < W) ? 0xFFFFu : 0u) | ((e1 >= 0 && e1 < W) ? 0xFFFF0000u : 0u);
        }
    }

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
    // The magnitudes left of lane 0 and right of lane 63 belong to the neighbouring tiles: lanes compute them once,
    // the left ones in lane l for row ys-1+l, the right ones in lane 63-l.  Rows are visited in order, and each
    // visit rotates the two registers by one lane, so that lane 0 / lane 63 hold the current row's value where the
    // lane shift of the magnitudes leaves its gap.
    int haloL16 = 0, haloR = 0;
    if (!RAW && lane < WAVE_ROWS + 2) haloL16 = mag_at(ys - 1 + lane, x0 - 1) << 16;  // high half: the pair (x0-2, x0-1)
    if (!RAW && lane > 63 - (WAVE_ROWS + 2)) haloR = mag_at(ys - 1 + 63 - lane, x0 + TILE_W);  // low half of (x0+128, x0+129)

    // rel: row index relative to ys-2 (0 .. WAVE_ROWS+3); off: LDS byte offset of (that row, column x-2)
    const uint32_t wave_off = (uint32_t)((wv * WAVE_ROWS) * LPITCH + 2 * lane + 2);
    auto load_row = [&](int rel, uint32_t off, RowD& o) {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(tile + off);
        const uint32_t A = p[0], B = p[1], C = p[2];  // bytes (x-2,x-1) (x,x+1) (x+2,x+3)
        const uint32_t tm2 = perm(0, A, SEL_PAIR), tm1 = perm(B, A, SEL_CROSS), t0 = perm(0, B, SEL_PAIR),
                       tp1 = perm(C, B, SEL_CROSS), tp2 = perm(0, C, SEL_PAIR);
        const uint32_t s1 = tm1 + tp1;  // halves of at most 510: plain adds
        o.hg = pk_mad<6>(t0, pk_mad<4>(s1, tm2 + tp2));
        o.rs = s1 + t0;
        uint32_t zm1, zp1;
        if (!BORDER) {
            zm1 = tm1; o.z0 = t0; zp1 = tp1;
            o.d = pk_sub(tp1, tm1);
            o.sm = pk_mad<2>(t0, s1);
        } else {
            const int ry = ys - 2 + rel;
            uint32_t um1 = tm1, u0 = t0, up1 = tp1;  // the pairs the Canny view is cut from
            if (ry >= 0 && ry < H) {  // wave-uniform
                zm1 = tm1 & maskZ[0]; o.z0 = t0 & maskZ[1]; zp1 = tp1 & maskZ[2];
            } else {
                // a row above or below the image: zeros for LBP, the first / last image row again for Canny
                zm1 = 0u; o.z0 = 0u; zp1 = 0u;
                const uint16_t* q = reinterpret_cast<const uint16_t*>(tile + (iclamp(ry, 0, H - 1) - y0 + 2) * LPITCH +
                                                                      2 * lane + 2);
                const uint32_t A2 = q[0], B2 = q[1], C2 = q[2];
                um1 = perm(B2, A2, SEL_CROSS); u0 = perm(0, B2, SEL_PAIR); up1 = perm(C2, B2, SEL_CROSS);
            }
            const uint32_t cm1 = perm(u0, um1, selC[0]), c0 = perm(up1, u0, selC[1]), cp1 = perm(up1, u0, selC[2]);
            o.d = pk_sub(cp1, cm1);
            o.sm = pk_mad<2>(c0, cm1 + cp1);
        }
        if (DO_LBP) {
            const uint32_t q0 = pk_mul<29>(o.z0);
            o.rp = pk_mad_r(zp1, 0x00460046u, q0);  // 70
            o.rm = pk_mad_r(zm1, 0x00460046u, q0);
            const uint32_t e = pk_mad_op(o.z0, 0xFF80FF80u, 0x80028002u);  // 0x8002 - 128 c per half
            o.gp = pk_mad<29>(zp1, e);
            o.gm = pk_mad<29>(zm1, e);
            o.zm = zm1;
            o.zp = zp1;
        }
    };

    // Gradient and L1 magnitude of row r from the Sobel partials of rows r-1, r, r+1, and the same
    // magnitudes seen from the left / right neighbour columns: ML = (x-1, x), MR = (x+1, x+2).
    struct MagRow { uint32_t ML, M, MR, ax, ay, sxy; };
    auto mag_row = [&](int r, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t s0, uint32_t s2, MagRow& o) {
        const uint32_t gx = pk_mad<2>(d1, pk_add(d0, d2));  // 2 d1 + d0 + d2 (the low 16 bits do not care about signs)
        const uint32_t gy = pk_sub(s2, s0);
        o.ax = pk_abs(gx);
        o.ay = pk_abs(gy);
        o.sxy = gx ^ gy;
        uint32_t m = o.ax + o.ay;  // at most 2040 per half
        if (BORDER) m &= (r >= 0 && r < H) ? vm : 0u;
        o.M = m;
        if (!RAW) {
            // The rotations first: every lane of a rotation has a source, so no register has to hold a value for
            // lanes without one, and the shifts after them can then write into the old halo registers, which are dead
            // by then (the other way round each shift costs a register copy).
            const int nextL = __builtin_amdgcn_mov_dpp(haloL16, 0x134, 0xf, 0xf, false);  // wave_rol:1, lane i takes lane i+1
            const int nextR = __builtin_amdgcn_mov_dpp(haloR, 0x13C, 0xf, 0xf, false);    // wave_ror:1, lane i takes lane i-1
            const uint32_t mp = (uint32_t)__builtin_amdgcn_update_dpp(haloL16, (int)m, 0x138, 0xf, 0xf, false);  // wave_shr:1
            const uint32_t mn = (uint32_t)__builtin_amdgcn_update_dpp(haloR, (int)m, 0x130, 0xf, 0xf, false);    // wave_shl:1
            haloL16 = nextL;
            haloR = nextR;
            o.ML = __builtin_amdgcn_alignbit(m, mp, 16);
            o.MR = __builtin_amdgcn_alignbit(mn, m, 16);
        }
    };

    // sliding state: rows y-2..y+2 of the binomial row sums, y-1..y+2 of the rest
    RowD rw[5];   // rw[j] = row y-2+j
    MagRow mg[3]; // rows y-1, y, y+1
    {
        load_row(0, wave_off, rw[0]);
        load_row(1, wave_off + LPITCH, rw[1]);
        load_row(2, wave_off + 2 * LPITCH, rw[2]);
        load_row(3, wave_off + 3 * LPITCH, rw[3]);
        mag_row(ys - 1, rw[0].d, rw[1].d, rw[2].d, rw[0].sm, rw[2].sm, mg[0]);
        mag_row(ys, rw[1].d, rw[2].d, rw[3].d, rw[1].sm, rw[3].sm, mg[1]);
    }

    // Ten 6-bit counters for both pixel columns together: bins 0..4 in one word, 5..9 in the other.  A counter holds
    // at most 63; 16 rows add at most 32 (whichever lane counts a pixel in a dense block, it is one of its own) and
    // the second stage at most QUEUE_ITEMS / 64 = 8 more: the first half of the strip is set aside after its 16 rows.
    uint32_t hist_a = 0, hist_b = 0, hist_a0 = 0, hist_b0 = 0;
    uint32_t hp_pk = 0, hp_q = 0, ad_s = 0, gr_pk = 0;
    static_assert(WAVE_ROWS * 255 < 65536, "a strip's sums of 8-bit values fit the halves of a word");
    uint32_t accS = 0, accC = 0;      // Canny bits of up to 16 rows, first row highest, per half
    uint32_t accS0 = 0, accC0 = 0;    // rows 0..15 of the strip

    auto count_pixel = [&](uint32_t idx, int y, int xx) {
        const uint2 e = lut[idx];
        hist_a += e.x;
        hist_b += e.y;
        if (CODES) a.codes[((int64_t)frame * H + y) * W + xx] = code_of(e);
    };
    // table index of one pixel with every sample evaluated exactly
    auto exact_index = [&](int y, int xx) -> uint32_t {
        double n[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int yy = y - 1 + r, xq = xx - 1 + q;
                n[r][q] = (BORDER && (yy < 0 || yy >= H || xq < 0 || xq >= W))
                              ? 0.0
                              : (double)tile[(yy - y0 + 2) * LPITCH + (xq - x0 + GRAY_XOFF)];
            }
        const double rfn = (double)y + a.rp_neg, rfp = (double)y + a.rp_pos;
        const double drN = rfn - floor(rfn), drP = rfp - floor(rfp);
        const double cfp = (double)xx + a.cp_pos, cfn = (double)xx + a.cp_neg;
        const double dcP = cfp - floor(cfp), dcN = cfn - floor(cfn);
        const double ce = n[1][1];
        const uint32_t b1 = lbp_bit(n[0][1], n[0][2], n[1][1], n[1][2], drN, dcP, ce);
        const uint32_t b3 = lbp_bit(n[0][0], n[0][1], n[1][0], n[1][1], drN, dcN, ce);
        const uint32_t b5 = lbp_bit(n[1][0], n[1][1], n[2][0], n[2][1], drP, dcN, ce);
        const uint32_t b7 = lbp_bit(n[1][1], n[1][2], n[2][1], n[2][2], drP, dcP, ce);
        const uint32_t b0 = n[1][2] >= ce, b2 = n[0][1] >= ce, b4 = n[1][0] >= ce, b6 = n[2][1] >= ce;
        return (b1 << 7) | (b3 << 6) | (b5 << 5) | (b7 << 4) | (b0 << 3) | (b2 << 2) | (b4 << 1) | b6;
    };

#pragma unroll 1
    for (int blk = 0; blk < WAVE_ROWS / BLOCK_ROWS; ++blk) {
        const int yb = ys + blk * BLOCK_ROWS;
        if (!BORDER || yb < H) {  // wave-uniform: below the image nothing is left to do, the remaining bits are zero
            const uint32_t blk_off = wave_off + (uint32_t)(blk * BLOCK_ROWS * LPITCH);
            // pend holds the undecided flags of this block, row i at bit 15-i (low pixel) and 31-i (high pixel)
            uint32_t pend = 0;
            // room for this block's Canny bits: row i goes to bit 7-i of each half, the block before moves up (a half
            // holds at most eight bits here, so nothing crosses into the other half)
            static_assert(BLOCK_ROWS == 8, "Canny bits are collected a byte per block");
            accC <<= 8;
            accS <<= 8;
#pragma unroll
            for (int i = 0; i < BLOCK_ROWS; ++i) {
                const int y = yb + i;
                load_row(blk * BLOCK_ROWS + i + 4, blk_off + (i + 4) * LPITCH, rw[4]);
                mag_row(y + 1, rw[2].d, rw[3].d, rw[4].d, rw[2].sm, rw[4].sm, mg[2]);
                const uint32_t vmy = BORDER ? ((y < H) ? vm : 0u) : 0xFFFFFFFFu;
                const uint32_t c2 = rw[2].z0;  // the two centre pixels (zero where invalid)

                // ---- high-pass: 8c - sum(neighbours) = 9c - sum(3x3), saturated to u8
                {
                    const uint32_t sum3 = rw[1].rs + rw[2].rs + rw[3].rs;  // at most 2295 per half
                    const uint32_t v = pk_min(pk_sub_sat(pk_mul<9>(c2), sum3), 0x00FF00FFu);
                    // (halves of at most 255, 32 rows: plain adds on the whole word, split at the end of the strip)
                    hp_pk += v;
                    hp_q = dot2(v, v, hp_q);
                    gr_pk += c2;
                }
                // ---- Gaussian residual
                {
                    const uint32_t S = pk_mad<6>(rw[2].hg, pk_mad<4>(rw[1].hg + rw[3].hg, rw[0].hg + rw[4].hg));
                    uint32_t blur = pk_shr<8>(S + 0x00800080u);  // S is at most 255 * 256 per half
                    if (BORDER) blur &= vmy;
                    ad_s = sad_u16(c2, blur, ad_s);
                }
                // ---- LBP (P=8, R=1), first stage: bit 15 of each half of a difference is "sample >= centre"
                if (DO_LBP) {
                    // (neighbour | 0x8000) - centre per half, as one plain subtraction of the centre word less bit 15
                    // of each half: a half's result is at least 0x8000 - 255, so nothing borrows from the other half
                    const uint32_t cm = c2 - HH;
                    const uint32_t a0 = rw[2].zp - cm;  // sample 0 (0,+1)
                    const uint32_t a2 = rw[1].z0 - cm;  // sample 2 (-1,0)
                    const uint32_t a4 = rw[2].zm - cm;  // sample 4 (0,-1)
                    const uint32_t a6 = rw[3].z0 - cm;  // sample 6 (+1,0)
                    const uint32_t f1 = rw[1].rp + rw[2].gp;  // (-.7,+.7): 0x8002 + N'
                    const uint32_t f3 = rw[1].rm + rw[2].gm;  // (-.7,-.7)
                    const uint32_t f5 = rw[3].rm + rw[2].gm;  // (+.7,-.7)
                    const uint32_t f7 = rw[3].rp + rw[2].gp;  // (+.7,+.7)
                    uint32_t bits = (f1 >> 5) & 0x04000400u;
                    bits = bfi_lit<0x02000200u>(f3 >> 6, bits);
                    bits = bfi_lit<0x01000100u>(f5 >> 7, bits);
                    bits = bfi_lit<0x00800080u>(f7 >> 8, bits);
                    bits = bfi_lit<0x00400040u>(a0 >> 9, bits);
                    bits = bfi_lit<0x00200020u>(a2 >> 10, bits);
                    bits = bfi_lit<0x00100010u>(a4 >> 11, bits);
                    bits = bfi_lit<0x00080008u>(a6 >> 12, bits);
                    // undecided: 0x8000 <= f < 0x8005, i.e. subtracting 5 clears bit 15 (f is at least 0x82)
                    // (read as signed halves the band is the bottom of the range, so the smallest of the four is in
                    // it whenever any of them is: three packed minima and one test instead of four tests)
                    const uint32_t fmin = pk_min_i(pk_min_i(f1, f3), pk_min_i(f5, f7));
                    uint32_t und = fmin ^ (fmin - TIE_BAND);
                    und &= BORDER ? (vmy & HH) : HH;
                    pend |= und >> i;
                    // histogram (and codes): a pixel's half of the bit word is the byte offset of its entry in the
                    // table of one-hot 6-bit counters; undecided pixels add nothing here, the second stage adds them
                    // (pixels outside the image get bit 11 as well: ~vmy has it, and more, exactly in their halves)
                    const uint32_t g = bits | (und >> UNDECIDED_SHIFT) | (BORDER ? (~vmy & (HH >> UNDECIDED_SHIFT)) : 0u);
                    const uint32_t il = g & 0xFFFFu, ih = g >> 16;
                    const uint2 el = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(lut) + il);
                    const uint2 eh = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(lut) + ih);
                    hist_a += el.x + eh.x;
                    hist_b += el.y + eh.y;
                    if (CODES) {
                        if (el.x | el.y) a.codes[((int64_t)frame * H + y) * W + x] = code_of(el);
                        if (eh.x | eh.y) a.codes[((int64_t)frame * H + y) * W + x + 1] = code_of(eh);
                    }
                }
                // ---- Canny non-maximum suppression + double threshold on both pixels at once.  The sign
                // bit of each 16-bit difference is the predicate; masks select the two neighbours along the
                // quantised gradient direction.  tan(22.5 deg) = 13573/2^15 and |gx| <= 1020, so
                // floor(|gx|*13573/2^15) = (|gx|*53 + (|gx|*5 >> 8)) >> 7 stays inside 16 bits.
                {
                    const uint32_t m = mg[1].M;
                    const uint32_t nz = pk_sub(a.low2, m);  // negative where m > low
                    if (RAW) {
                        if (__any((nz & 0x80008000u) != 0)) {
                            const uint32_t ks = nz & pk_sub(a.high2, m);
                            accC |= (nz >> (8 + i)) & (0x00800080u >> i);
                            accS |= (ks >> (8 + i)) & (0x00800080u >> i);
                        }
                    } else if (__any((nz & 0x80008000u) != 0)) {
                        const uint32_t ax = mg[1].ax, ay = mg[1].ay;
                        const uint32_t q22 = pk_shr<7>(pk_mad<53>(ax, pk_shr<8>(pk_mul<5>(ax))));
                        const uint32_t nothoriz = pk_sign_mask(pk_sub(q22, ay));              // |gy| > tan22 |gx|
                        const uint32_t vert = pk_sign_mask(pk_sub(pk_mad<2>(ax, q22), ay));   // |gy| > tan67 |gx|
                        const uint32_t opp = pk_sign_mask(mg[1].sxy);
                        uint32_t na = bfi(opp, mg[0].MR, mg[0].ML);
                        uint32_t nb = bfi(opp, mg[2].ML, mg[2].MR);
                        na = bfi(vert, mg[0].M, na);
                        nb = bfi(vert, mg[2].M, nb);
                        na = bfi(nothoriz, na, mg[1].ML);
                        nb = bfi(nothoriz, nb, mg[1].MR);
                        // the second neighbour is compared with >= for horizontal / vertical gradients:
                        // m >= nb is m > nb - 1
                        nb = pk_add(nb, bfi(nothoriz, vert, 0xFFFFFFFFu));
                        const uint32_t kc = pk_sub(na, m) & pk_sub(nb, m) & nz;
                        const uint32_t ks = kc & pk_sub(a.high2, m);
                        // the sign bits go straight to the row's place in the words (plain shifts and masks; a row
                        // without candidates leaves the words alone)
                        accC |= (kc >> (8 + i)) & (0x00800080u >> i);
                        accS |= (ks >> (8 + i)) & (0x00800080u >> i);
                    }
                }
                // ---- slide down one row
#pragma unroll
                for (int r = 0; r < 4; ++r) rw[r] = rw[r + 1];
                mg[0] = mg[1];
                mg[1] = mg[2];
            }
            if (DO_LBP) {
                // ---- undecided pixels of this block.  Normally few: their flags wait in LDS for the end of the strip.
                // Flat regions flag (nearly) every pixel: then the lanes are evenly loaded as they are and each
                // re-evaluates its own pixels here.
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_u32(__popc(pend)), 63);
                if (tot > DENSE_MIN) {
                    {
                        // Flat neighbourhoods first: where the four columns x-1 .. x+2 hold one value over three
                        // consecutive rows, both pixels of the middle row have all axis bits set and their diagonal
                        // bits come from the table instead of trips through the fp64 sequence.  Rolled loops: this
                        // runs for flat content only and must not cost the common path registers.  The queue is idle
                        // here: BLOCK_ROWS of its bytes per lane hold the rows' centre values.
                        const uint8_t* tp = tile + (yb - 1 - y0 + 2) * LPITCH + (x - x0 + GRAY_XOFF) - 1;
                        uint8_t* cvb = reinterpret_cast<uint8_t*>(wq) + lane * BLOCK_ROWS;
                        uint32_t f0 = 0, f1 = 0, v0 = 0, v1 = 0, elig = 0;
                        // (on a border strip the tile holds mirrored pixels where LBP sees zeros: a row outside the
                        // image, or a lane whose four columns are not all inside it, is never flat)
                        const bool cols_in = !BORDER || (x >= 1 && x + 2 < W);
#pragma unroll 1
                        for (int r = 0; r < BLOCK_ROWS + 2; ++r) {
                            const uint32_t b0 = tp[r * LPITCH], b1 = tp[r * LPITCH + 1], b2 = tp[r * LPITCH + 2],
                                           b3 = tp[r * LPITCH + 3];
                            uint32_t f = (((b0 ^ b1) | (b1 ^ b2) | (b2 ^ b3)) == 0) ? 1u : 0u;
                            if (BORDER && !(cols_in && yb - 1 + r >= 0 && yb - 1 + r < H)) f = 0u;
                            if (r >= 2 && (f & f1 & f0) != 0 && b0 == v1 && v1 == v0) elig |= 1u << (r - 2);
                            if (r >= 1 && r <= BLOCK_ROWS) cvb[r - 1] = (uint8_t)b1;
                            f0 = f1; v0 = v1; f1 = f; v1 = b0;
                        }
                        uint32_t want = 0;
#pragma unroll
                        for (int i = 0; i < BLOCK_ROWS; ++i) want |= ((elig >> i) & 1u) * (HH >> i);
                        want &= pend;
                        if (want != 0) {
                            const uint8_t* tie = a.tie;
                            const uint32_t cc = *reinterpret_cast<const uint16_t*>(tie + TIE_COORDS + x);
                            const uint32_t ncc = tie[2 * TIE_COORDS];
                            const uint8_t* tie_lut = tie + TIE_LUT_OFF;
#pragma unroll 1
                            for (int i = 0; i < BLOCK_ROWS; i += 2) {
                                uint32_t l[4];
#pragma unroll
                                for (int k = 0; k < 2; ++k) {
                                    const uint32_t base = (uint32_t)tie[yb + i + k] * ncc;
                                    const uint32_t cv = cvb[i + k];
                                    l[2 * k] = tie_lut[((base + (cc & 0xFFu)) << 8) + cv];
                                    l[2 * k + 1] = tie_lut[((base + (cc >> 8)) << 8) + cv];
                                }
#pragma unroll
                                for (int k = 0; k < 2; ++k) {
                                    if ((want >> (15 - (i + k))) & 1u) count_pixel((l[2 * k] << 4) | 0xFu, yb + i + k, x);
                                    if ((want >> (31 - (i + k))) & 1u) count_pixel((l[2 * k + 1] << 4) | 0xFu, yb + i + k, x + 1);
                                }
                            }
                            pend &= ~want;
                        }
                    }
                    while (__any(pend != 0)) {
                        if (pend != 0) {
                            const uint32_t p = (uint32_t)__ffs(pend) - 1u;
                            pend &= pend - 1;
                            const int y = yb + 15 - (int)(p & 15u), xx = x + (int)(p >> 4);
                            count_pixel(exact_index(y, xx), y, xx);
                        }
                    }
                }
                wpend[blk * 64 + lane] = pend;
            }
        } else {
            if (DO_LBP) wpend[blk * 64 + lane] = 0;
            if (BLOCK_ROWS < 16) {
                accS = pk_mul<(1 << (BLOCK_ROWS & 15))>(accS);
                accC = pk_mul<(1 << (BLOCK_ROWS & 15))>(accC);
            }
        }
        if (blk == WAVE_ROWS / BLOCK_ROWS / 2 - 1) {
            accS0 = accS; accC0 = accC;
            accS = 0; accC = 0;
            hist_a0 = hist_a; hist_b0 = hist_b;
            hist_a = 0; hist_b = 0;
        }
    }

    // ---- LBP second stage.  Every lane appends its undecided pixels to the wave's queue (positions from a prefix sum
    // of the counts); then lane l takes queue entry l, l+64, ...: the pixel's four axis bits again and all four
    // diagonal samples in the fp64 arithmetic of skimage's bilinear_interpolation (operands re-read from the tile),
    // and counts the pixel in its own histogram counters -- which lane counts a pixel does not matter.  One queue
    // per strip: about 2 % of the pixels of camera-like content are undecided, eighty-odd per strip, two rounds
    // with most lanes busy (per block of eight rows it was a round with a third of them busy, four times).
    if (DO_LBP) {
        wave_sync();
        uint32_t pq[WAVE_ROWS / BLOCK_ROWS];
        uint32_t cnt = 0;
#pragma unroll
        for (int b = 0; b < WAVE_ROWS / BLOCK_ROWS; ++b) {
            pq[b] = wpend[b * 64 + lane];
            cnt += __popc(pq[b]);
        }
        if (__any(cnt != 0)) {
            // inclusive prefix sum over the wave with data-parallel-primitive adds (no trip through the LDS
            // crossbar): shifts by 1, 2, 4, 8 inside each row of 16 lanes, then lane 15 of a row into the
            // next row, then lane 31 into the upper half
            const uint32_t inc = wave_scan_u32(cnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            uint32_t qpos = inc - cnt;
            // item: (owner lane << 6) | (row of the strip << 1) | pixel, from the position of a flag.  Two blocks'
            // flags share a word (the second block's byte pair moves below the first's: row r of the sixteen at bit
            // 15-r and 31-r), so the wave runs two loops of the worst lane's count instead of four.
            static_assert(WAVE_ROWS / BLOCK_ROWS == 4 && BLOCK_ROWS == 8, "flag words are merged in pairs of blocks");
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                uint32_t flags = pq[2 * b] | (pq[2 * b + 1] >> 8);
                while (flags != 0) {
                    const uint32_t p = (uint32_t)__ffs(flags) - 1u;
                    flags &= flags - 1;
                    wq[qpos++] = (uint16_t)(((uint32_t)lane << 6) | ((uint32_t)(b * 16 + 15 - (int)(p & 15u)) << 1) | (p >> 4));
                }
            }
            wave_sync();
            for (uint32_t qb = 0; qb < total; qb += 64) {
                if (qb + lane < total) {
                    const uint32_t item = wq[qb + lane];
                    const int y = ys + (int)((item & 63u) >> 1), xx = x0 + 2 * (int)(item >> 6) + (int)(item & 1u);
                    count_pixel(exact_index(y, xx), y, xx);
                }
            }
        }
    }

    // ---- the strip's Canny bits: row r of the strip at bit r of a 32-bit word per pixel column
    {
        const int64_t wbase = ((int64_t)frame * a.bm_words + (ys >> 6)) * W;
        const int half = (ys >> 5) & 1;
        if (x < W) {
            const int64_t wi = (wbase + x) * 2 + half;
            a.strong[wi] = __brev((accS0 << 16) | (accS & 0xFFFFu));
            a.cand[wi] = __brev((accC0 << 16) | (accC & 0xFFFFu));
        }
        if (x + 1 < W) {
            const int64_t wi = (wbase + x + 1) * 2 + half;
            a.strong[wi] = __brev((accS0 & 0xFFFF0000u) | (accS >> 16));
            a.cand[wi] = __brev((accC0 & 0xFFFF0000u) | (accC >> 16));
        }
        // lanes 0..31 hold the columns of one hysteresis tile, lanes 32..63 those of the next
        const unsigned long long sb = __ballot(x < W && (accS | accS0) != 0);
        if (sb != 0 && lane == 0) {
            uint8_t* f = a.sflag + ((int64_t)frame * a.bm_tiles + (x0 >> 6)) * a.bm_words + (ys >> 6);
            if ((uint32_t)sb != 0) f[0] = 1;
            if ((sb >> 32) != 0) f[a.bm_words] = 1;
        }
    }

    // ---- wave reduction.  Nine 32-bit words per lane: the ten histogram bins as 16-bit fields (a strip holds 4096
    // pixels, a lane 64 of them), the sum of squares (at most 64 * 255^2 per lane), three plain sums (at most 64 * 255 per
    // lane).  Each word is summed over the wave by six adds whose second operand comes from another lane inside the
    // instruction; the total arrives in lane 63.
    {
        auto bin2 = [&](uint32_t h, uint32_t h0, int f) -> uint32_t {  // fields f and f+1 of a counter word, 16 bits apart
            const uint32_t lo = ((h >> (6 * f)) & 63u) + ((h0 >> (6 * f)) & 63u);
            const uint32_t hi = ((h >> (6 * f + 6)) & 63u) + ((h0 >> (6 * f + 6)) & 63u);
            return lo | (hi << 16);
        };
        uint32_t w[9];
        w[0] = bin2(hist_a, hist_a0, 0);                                                    // bins 0, 1
        w[1] = bin2(hist_a, hist_a0, 2);                                                    // bins 2, 3
        w[2] = (((hist_a >> 24) & 63u) + ((hist_a0 >> 24) & 63u)) | ((((hist_b) & 63u) + ((hist_b0) & 63u)) << 16);  // 4, 5
        w[3] = bin2(hist_b, hist_b0, 1);                                                    // bins 6, 7
        w[4] = bin2(hist_b, hist_b0, 3);                                                    // bins 8, 9
        w[5] = (hp_pk & 0xFFFFu) + (hp_pk >> 16); w[6] = hp_q; w[7] = ad_s; w[8] = (gr_pk & 0xFFFFu) + (gr_pk >> 16);
        wave_total9(w);
        if (lane == 63) {
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                red[2 * b] = w[b] & 0xFFFFu;
                red[2 * b + 1] = w[b] >> 16;
            }
            red[10] = w[5];
            red[11] = w[6];
            red[12] = w[7];
            red[13] = w[8];
        }
    }
}

struct __attribute__((aligned(4))) Bgr4 {
    uint32_t a, b, c;  // 12 bytes = four BGR pixels
};

// Four BGR pixels to four gray bytes, as cv2.cvtColor's fixed-point path: (B wB + G wG + R wR + half) >> shift.  The
// weights come scaled so that the shift is 16 and split into bytes, w = 256 hi + lo, which makes a pixel two four-byte
// dot products: a pixel's three bytes sit in one word after at most one byte alignment, and the fourth byte meets a
// zero weight.  floor((256 H + L) / 65536) = floor((H + floor(L / 256)) / 256): the low product, shifted, is the
// accumulator of the high one, and the gray level is byte 1 of the result.
__device__ __forceinline__ uint32_t gray4(Bgr4 v, uint32_t w_hi, uint32_t w_lo) {
    // bytes: a = B0 G0 R0 B1 | b = G1 R1 B2 G2 | c = R2 B3 G3 R3
    const uint32_t p1 = __builtin_amdgcn_alignbyte(v.b, v.a, 3), p2 = __builtin_amdgcn_alignbyte(v.c, v.b, 2);
    auto px = [](uint32_t p, uint32_t hi, uint32_t lo) {
        return __builtin_amdgcn_udot4(p, hi, __builtin_amdgcn_udot4(p, lo, 32768u, false) >> 8, false);
    };
    const uint32_t s0 = px(v.a, w_hi, w_lo), s1 = px(p1, w_hi, w_lo), s2 = px(p2, w_hi, w_lo),
                   s3 = px(v.c, w_hi << 8, w_lo << 8);
    return perm(s1, s0, 0x0c0c0501u) | perm(s3, s2, 0x05010c0cu);
}

// The stencil tile straight from the BGR frame: LROWS x LDW words of four pixels, converted on the way in, each word
// from one 12-byte load.  A thread keeps one word column and walks down the rows, LD_ROWS apart (no division;
// everything that depends on the column is worked out once).  Rows above and below the image and the word left of
// column 0 / right of column W-1 hold BORDER_REFLECT_101 copies: a mirrored row is another row's load; a mirrored word
// is its neighbour's load with the bytes picked in reverse after the conversion ((-2,-1) = (2,1), (W,W+1) =
// (W-2,W-3); further out nothing reads).  The words go to the gray plane as well, after the tile is complete: a load
// issued behind a store waits for it.
// Loads and stores are buffer instructions: the address is a scalar base (the frame), a scalar offset (the pass's
// first row) and one per-thread offset that does not change from pass to pass, so an access costs no address
// arithmetic, and an offset past the end of the buffer turns a store into nothing (how a thread whose word is not
// part of the plane "skips" it).
// ROWS_INSIDE: every tile row is an image row (otherwise a row's offset is mirrored per lane).
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
constexpr int BUFFER_WORD3 = 0x00020000;       // raw buffer of 32-bit elements
constexpr uint32_t OFF_NOWHERE = 0x7FFF0000u;  // past the end of any buffer here (their sizes stay below 2^31)
template <bool ROWS_INSIDE>
__device__ __forceinline__ void load_tile_bgr(const StencilArgs& a, uint32_t* tile, int frame, int x0, int y0, int tid) {
    constexpr int LD_ROWS = 256 / LDW;                       // rows per pass (7)
    constexpr int PASSES = (LROWS + LD_ROWS - 1) / LD_ROWS;  // 19
    constexpr int BATCH = ROWS_INSIDE ? PASSES : 10;
    const int r0 = tid / LDW, d = tid - r0 * LDW;
    if (r0 >= LD_ROWS) return;
    const uint32_t rs = (uint32_t)a.row_stride;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(a.bgr) + (int64_t)frame * a.frame_stride, 0, (int)((uint32_t)(a.H - 1) * rs + 3u * (uint32_t)a.W), BUFFER_WORD3);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
        a.gray_w + (int64_t)frame * a.grows * a.gpitch, 0, a.grows * a.gpitch, BUFFER_WORD3);
    const int xs = x0 - 4 + 4 * d;
    const uint32_t xbyte = 3u * (uint32_t)(xs < 0 ? 0 : (xs >= a.W ? a.W - 4 : xs));
    const uint32_t bsel = xs < 0 ? 0x01020000u : (xs >= a.W ? 0x00000102u : 0x03020100u);
    const int ytop = y0 - 2 + r0, ymir = 2 * a.H - 2;
    const uint32_t voff = __umul24((uint32_t)r0, rs) + xbyte;
    uint32_t gq[PASSES];
#pragma unroll
    for (int kb = 0; kb < PASSES; kb += BATCH) {
        u32x3 v[BATCH];  // the loads of a batch are issued before its first conversion
#pragma unroll
        for (int k = kb; k < kb + BATCH && k < PASSES; ++k) {
            if (ROWS_INSIDE) {
                // (the last pass runs past the tile for some threads: an image row or past the buffer, both harmless)
                v[k - kb] = __builtin_amdgcn_raw_buffer_load_b96(src, (int)voff, (y0 - 2 + LD_ROWS * k) * (int)rs, 0);
            } else {
                int yy = ytop + LD_ROWS * k;
                yy = yy < 0 ? -yy : yy;
                const int ym = ymir - yy;
                yy = yy < ym ? yy : ym;
                yy = yy < 0 ? 0 : yy;  // rows far below the image: nothing reads them
                v[k - kb] = __builtin_amdgcn_raw_buffer_load_b96(src, (int)(__umul24((uint32_t)yy, rs) + xbyte), 0, 0);
            }
        }
#pragma unroll
        for (int k = kb; k < kb + BATCH && k < PASSES; ++k) {
            const Bgr4 q = {v[k - kb].x, v[k - kb].y, v[k - kb].z};
            gq[k] = perm(0, gray4(q, a.w_hi, a.w_lo), bsel);
            if (LD_ROWS * k + LD_ROWS <= LROWS || r0 + LD_ROWS * k < LROWS) tile[tid + LD_ROWS * LDW * k] = gq[k];
        }
    }
    // plane row of tile row r is y0 + r (the tile starts two rows above y0, the plane two rows above the image)
    // Which words of a tile row go to the plane: words 0..31, i.e. the pixels x0-4 .. x0+123, which is one aligned
    // 128-byte line of the plane (pixel x lives at byte x + 4) -- the tile's own 128 pixels would straddle two lines,
    // and partly written lines cost the memory system read-modify-write cycles.  The neighbour to the right writes
    // the pixels x0+124 .. x0+127 as its word 0; the last tile of a row has no such neighbour and writes the rest of
    // its words up to the one that holds the mirrored columns W, W+1.  Word 0 of the first tile holds the mirrored
    // columns -2, -1; the first and the last tile row write their mirrored rows as well: the plane comes out with the
    // same BORDER_REFLECT_101 frame the separate gray pass gives it.
    const bool col_in = xs <= a.W && (d < TILE_W / 4 || x0 + TILE_W >= a.W);
    const uint32_t woff = col_in ? __umul24((uint32_t)r0, (uint32_t)a.gpitch) + (uint32_t)(xs + GRAY_XOFF) : OFF_NOWHERE;
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
        const int row = r0 + LD_ROWS * k;
        bool ok;
        if (ROWS_INSIDE) {
            ok = (LD_ROWS * k >= 2 || row >= 2) && (LD_ROWS * k + LD_ROWS <= LROWS - 2 || row < LROWS - 2);
        } else {
            const int yy = ytop + LD_ROWS * k;  // image row; H and H + 1 are the mirrored rows below the image
            ok = (row >= 2 || y0 == 0) && row < LROWS && (yy < a.H ? row < LROWS - 2 : yy < a.H + 2);
        }
        __builtin_amdgcn_raw_buffer_store_b32(gq[k], dst, (int)(ok ? woff : OFF_NOWHERE), (y0 + LD_ROWS * k) * a.gpitch, 0);
    }
}

template <bool DO_LBP, bool CODES, bool FROM_BGR, bool RAW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(STX_STENCIL_WAVES, STX_STENCIL_WAVES))) void stencil_kernel(StencilArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[LROWS * LDW];
    __shared__ uint2 lut[LUT_ENTRIES];
    __shared__ uint16_t wq[4][DO_LBP ? QUEUE_ITEMS : 2];
    __shared__ uint32_t wpend[4][DO_LBP ? (WAVE_ROWS / BLOCK_ROWS) * 64 : 2];
    __shared__ uint32_t red[4][16];
    // Workgroups go to the eight accelerator dies round-robin by their linear number, and each die has its own L2.
    // Tiles are numbered row by row, frame by frame; workgroup L takes tile (L % 8) * N/8 + L / 8, so that a die
    // works through one contiguous run of tiles: the halo rows and columns a tile shares with its neighbours, and
    // the cache lines its row segments share with them, are fetched into one L2 once instead of into two.
    uint32_t tileid;
    {
        const uint32_t L = blockIdx.x, N = gridDim.x, q = N >> 3, r = N & 7u, die = L & 7u;
        tileid = die * q + (die < r ? die : r) + (L >> 3);
    }
    const uint32_t tiles_per_frame = (uint32_t)(a.tiles_x * a.tiles_y);
    const int frame = (int)(tileid / tiles_per_frame);
    const uint32_t tin = tileid - (uint32_t)frame * tiles_per_frame;
    const int tyi = (int)(tin / (uint32_t)a.tiles_x), txi = (int)(tin - (uint32_t)tyi * (uint32_t)a.tiles_x);
    const int x0 = txi * TILE_W;
    const int y0 = tyi * TILE_H;
    const int tid = threadIdx.x;

    // Tile rows start 64-byte aligned in the padded plane (x0 is a multiple of 128, the pitch of 64) and are
    // LPITCH = 8 * 16 + 8 bytes long: sixteen lanes per row, nine of them busy (eight 16-byte pieces and the tail),
    // sixteen rows per pass.  The index arithmetic is a shift and a mask; a flat loop over 32-bit words cost a
    // division per word and a tenth of the kernel's time.
    if (FROM_BGR) {
        // block-uniform: every tile row is an image row (no mirrored rows, one address step from row to row)
        if (y0 >= 2 && y0 + TILE_H + 2 <= a.H)
            load_tile_bgr<true>(a, tile, frame, x0, y0, tid);
        else
            load_tile_bgr<false>(a, tile, frame, x0, y0, tid);
    } else {
        const uint8_t* gp = a.gray + (int64_t)frame * a.grows * a.gpitch + x0;
        const int lc = tid & 15;
        if (lc < 9) {
            constexpr int PASSES = (LROWS + 15) / 16;
            uint4 v[PASSES];  // every load is issued before the first store: one memory latency per tile, not nine
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                int gy = y0 + (tid >> 4) + 16 * k;  // padded-plane row of global row y0-2+row
                gy = gy < a.grows ? gy : a.grows - 1;
                // the tail lane reads 16 bytes too (the plane's pitch has 64 spare bytes) and keeps 8
                v[k] = *reinterpret_cast<const uint4*>(gp + (int64_t)gy * a.gpitch + lc * 16);
            }
            uint8_t* tb = reinterpret_cast<uint8_t*>(tile) + lc * 16;
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                const int row = (tid >> 4) + 16 * k;
                if (row < LROWS) {
                    uint2* dst = reinterpret_cast<uint2*>(tb + row * LPITCH);
                    dst[0] = make_uint2(v[k].x, v[k].y);
                    if (lc < 8) dst[1] = make_uint2(v[k].z, v[k].w);
                }
            }
        }
    }
    if (DO_LBP) {
        // Histogram table.  Index: bits 3..0 = LBP bits of samples 0,2,4,6, bits 7..4 = those of samples 1,3,5,7
        // (1 = sample >= centre).  Uniform patterns (at most two 0/1 changes along samples 0..7)
        // get their number of ones, everything else P+1 = 9.
        uint32_t pat = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t sa = ((uint32_t)tid >> (3 - k)) & 1u, sd = ((uint32_t)tid >> (7 - k)) & 1u;
            pat |= (sa << (2 * k)) | (sd << (2 * k + 1));  // LBP bit of sample s at bit s
        }
        const int changes = __popc((pat ^ (pat >> 1)) & 0x7Fu);
        const int code = (changes <= 2) ? __popc(pat) : 9;
        lut[tid] = code < 5 ? make_uint2(1u << (6 * code), 0u) : make_uint2(0u, 1u << (6 * (code - 5)));
        lut[LUT_INVALID + tid] = make_uint2(0u, 0u);
    }
    if (tid < 64) red[tid >> 4][tid & 15] = 0;
    __syncthreads();

    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the per-wave LDS bases in scalar registers
    const int ys = y0 + wv * WAVE_ROWS;
    if (ys < a.H) {
        // interior: the wave's strip and its two-pixel halo lie inside the image
        const bool interior = (x0 >= 2) && (ys >= 2) && (x0 + TILE_W + 2 <= a.W) && (ys + WAVE_ROWS + 2 <= a.H);
        if (interior)
            march<false, DO_LBP, CODES, RAW>(a, reinterpret_cast<const uint8_t*>(tile), lut, wq[wv], wpend[wv], frame, x0, y0, lane, wv, red[wv]);
        else
            march<true, DO_LBP, CODES, RAW>(a, reinterpret_cast<const uint8_t*>(tile), lut, wq[wv], wpend[wv], frame, x0, y0, lane, wv, red[wv]);
    } else if ((ys >> 6) < a.bm_words) {
        // a strip below the image that shares its 64-row bitmap word with image rows: its half is empty
        const int64_t wbase = ((int64_t)frame * a.bm_words + (ys >> 6)) * a.W;
        const int half = (ys >> 5) & 1;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int xx = x0 + 2 * lane + e;
            if (xx < a.W) {
                a.strong[(wbase + xx) * 2 + half] = 0;
                a.cand[(wbase + xx) * 2 + half] = 0;
            }
        }
    }
    __syncthreads();
    // per-tile partial sums, one slot per tile: no atomics, summed in a fixed order later
    if (tid < 16) {
        const uint32_t v = tid < 14 ? red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] : 0u;
        a.part[(int64_t)tileid * 16 + tid] = (tid < 10 && !DO_LBP) ? 0u : v;
    }
}

// Tables for flat neighbourhoods, in the layout StencilArgs::tie describes.
void build_tie_tables(const LbpOffsets& off, std::vector<uint8_t>* blob, int* n_cls_c) {
    const double rp_neg = off.rp[1], rp_pos = off.rp[5], cp_pos = off.cp[1], cp_neg = off.cp[3];
    std::vector<std::pair<double, double>> rcls, ccls;
    blob->assign((size_t)TIE_LUT_OFF, 0);
    auto classify = [&](std::vector<std::pair<double, double>>& cls, double oa, double ob, uint8_t* dst) {
        for (int v = 0; v < TIE_COORDS; ++v) {
            const double fa = (double)v + oa, fb = (double)v + ob;
            const std::pair<double, double> d(fa - floor(fa), fb - floor(fb));
            size_t k = 0;
            while (k < cls.size() && cls[k] != d) ++k;
            if (k == cls.size()) cls.push_back(d);
            dst[v] = (uint8_t)k;
        }
    };
    classify(rcls, rp_neg, rp_pos, blob->data());
    classify(ccls, cp_pos, cp_neg, blob->data() + TIE_COORDS);
    *n_cls_c = (int)ccls.size();
    (*blob)[2 * (size_t)TIE_COORDS] = (uint8_t)ccls.size();
    blob->resize((size_t)TIE_LUT_OFF + rcls.size() * ccls.size() * 256);
    uint8_t* lut = blob->data() + TIE_LUT_OFF;
    for (size_t r = 0; r < rcls.size(); ++r)
        for (size_t k = 0; k < ccls.size(); ++k)
            for (int v = 0; v < 256; ++v) {
                const double drN = rcls[r].first, drP = rcls[r].second, dcP = ccls[k].first, dcN = ccls[k].second;
                const double d = (double)v;
                const uint32_t b1 = lbp_bit(d, d, d, d, drN, dcP, d), b3 = lbp_bit(d, d, d, d, drN, dcN, d);
                const uint32_t b5 = lbp_bit(d, d, d, d, drP, dcN, d), b7 = lbp_bit(d, d, d, d, drP, dcP, d);
                lut[(r * ccls.size() + k) * 256 + v] = (uint8_t)((b1 << 3) | (b3 << 2) | (b5 << 1) | b7);
            }
}

bool stencil_reads_bgr(const Geom& g, const BgrSource& src) {
    const bool aligned4 = ((reinterpret_cast<uintptr_t>(src.bgr) | (uintptr_t)src.row_stride | (uintptr_t)src.frame_stride) & 3u) == 0;
    // byte offsets inside a frame are computed in 32 bits
    return aligned4 && (g.W & 3) == 0 && g.H >= 3 && src.row_stride > 0 && src.row_stride * (int64_t)g.H < ((int64_t)1 << 31);
}

void launch_stencil(hipStream_t s, uint8_t* gray, const BgrSource* src, int n_frames, const Geom& g, bool do_lbp81,
                    const LbpOffsets& off, int low, int high, uint32_t* part, unsigned long long* strong,
                    unsigned long long* cand, uint8_t* codes_out, const uint8_t* tie_blob, uint8_t* sflag, bool raw_canny) {
    StencilArgs a;
    a.bgr = nullptr;
    a.row_stride = a.frame_stride = 0;
    a.w_hi = a.w_lo = 0;
    a.gray_w = gray;
    if (src) {
        if (!stencil_reads_bgr(g, *src)) abort();
        a.bgr = src->bgr;
        a.row_stride = src->row_stride;
        a.frame_stride = src->frame_stride;
        // cv2's weights for B, G, R at 14 and at 15 fractional bits, scaled to 16
        const uint32_t w[3] = {src->shift == 15 ? 3735u << 1 : 1868u << 2, src->shift == 15 ? 19235u << 1 : 9617u << 2,
                               src->shift == 15 ? 9798u << 1 : 4899u << 2};
        for (int k = 0; k < 3; ++k) {
            a.w_hi |= (w[k] >> 8) << (8 * k);
            a.w_lo |= (w[k] & 255u) << (8 * k);
        }
    }
    a.sflag = sflag;
    a.bm_tiles = g.bm_tiles;
    a.tie = tie_blob;
    a.gray = gray;
    a.H = g.H; a.W = g.W; a.gpitch = g.gpitch; a.grows = g.grows; a.bm_words = g.bm_words;
    a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y;
    // magnitudes are at most 2040: thresholds outside [-1, 4095] behave like the nearest bound
    const uint32_t lo = (uint32_t)(low < -1 ? -1 : (low > 4095 ? 4095 : low)) & 0xFFFFu;
    const uint32_t hi = (uint32_t)(high < -1 ? -1 : (high > 4095 ? 4095 : high)) & 0xFFFFu;
    a.low2 = lo | (lo << 16);
    a.high2 = hi | (hi << 16);
    // (P=8,R=1): samples 1,3 sit above the centre (rp<0), 5,7 below; 1,7 right (cp>0), 3,5 left
    a.rp_neg = off.rp[1]; a.rp_pos = off.rp[5]; a.cp_pos = off.cp[1]; a.cp_neg = off.cp[3];
    // the integer pre-decision of the diagonal samples is written for the offsets skimage uses at (P=8, R=1)
    if (do_lbp81 && (llround(a.rp_neg * 1e5) != -70711 || llround(a.rp_pos * 1e5) != 70711 ||
                     llround(a.cp_pos * 1e5) != 70711 || llround(a.cp_neg * 1e5) != -70711))
        abort();
    a.part = part;
    a.strong = reinterpret_cast<uint32_t*>(strong);
    a.cand = reinterpret_cast<uint32_t*>(cand);
    a.codes = codes_out;
    if ((int64_t)g.tiles_x * g.tiles_y * n_frames > 0x7FFFFFFF) abort();
    dim3 grid((unsigned)(g.tiles_x * g.tiles_y * n_frames), 1, 1);
    // the raw double threshold is compiled for the (8,1) kernels without a code plane only
    if (raw_canny && (!do_lbp81 || codes_out)) abort();
    if (src) {
        if (!do_lbp81)
            hipLaunchKernelGGL((stencil_kernel<false, false, true, false>), grid, dim3(256), 0, s, a);
        else if (codes_out)
            hipLaunchKernelGGL((stencil_kernel<true, true, true, false>), grid, dim3(256), 0, s, a);
        else if (raw_canny)
            hipLaunchKernelGGL((stencil_kernel<true, false, true, true>), grid, dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((stencil_kernel<true, false, true, false>), grid, dim3(256), 0, s, a);
    } else if (!do_lbp81)
        hipLaunchKernelGGL((stencil_kernel<false, false, false, false>), grid, dim3(256), 0, s, a);
    else if (codes_out)
        hipLaunchKernelGGL((stencil_kernel<true, true, false, false>), grid, dim3(256), 0, s, a);
    else if (raw_canny)
        hipLaunchKernelGGL((stencil_kernel<true, false, false, true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((stencil_kernel<true, false, false, false>), grid, dim3(256), 0, s, a);
}

}  // namespace stx
