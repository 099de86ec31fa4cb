// Uniform LBP for any (P <= 24, R): the scikit-image algorithm
// (skimage.feature.local_binary_pattern(..., 'uniform'), called with P=8, R=1 at
// /root/reference/app/passive_checker.py:103) with no specialisation of the sample
// positions.  The fused stencil kernel covers the reference's (8,1) case; this kernel serves
// the multi-radius configurations.
//
// A 256-thread workgroup owns a 64x64 pixel tile held in LDS with a zero-filled halo (pixels
// outside the image are 0.0 for scikit-image).  Every sample is a bilinear blend of a 2x2
// cell whose position relative to the centre and whose four weights are the same for every
// pixel, so they are tabulated once per launch: the blend runs in 16.16 fixed point (two
// v_dot2_u32_u16 over packed pixel pairs) and decides the sign whenever the result is further from the centre
// than the rounding of the weights can explain; the remaining samples (ties, mostly) are
// replayed in scikit-image's fp64 operation order, one rounding per operation: compile with
// -ffp-contract=off.  Histograms leave the kernel as one partial per tile (no global atomics).
#include "stx_internal.h"

namespace stx {

#pragma clang fp contract(off)

constexpr int GL_TILE = 64;        // tile width and height
constexpr int GL_ROWS = GL_TILE / 4;  // rows per wave
constexpr int GL_BINS = 32;        // partial histogram slots per tile (P + 2 <= 26 used)
constexpr int GL_COPIES = 16;      // copies of the workgroup's histogram in LDS
constexpr int GL_HSTRIDE = 33;     // words between copies

struct LbpGenArgs {
    const uint8_t* gray;
    int H, W, gpitch, grows, P;
    int halo, lpitch;     // zero-filled border of the LDS tile, entries per tile row
    uint32_t margin;      // bound on |fixed-point sample - exact sample| in 1/65536 gray levels, plus a guard
    int otop[24];         // LDS byte offset of the cell's top-left pixel relative to the centre pixel
    int obot[24];         // ... of its bottom-left pixel (same as otop for a cell of one row)
    uint32_t wtop[24];    // 16.16 weights (left | right << 16) of the upper cell row
    uint32_t wbot[24];    // ... of the lower cell row (zero for a cell of one row)
    uint32_t lim[24];     // 2*margin + 1, or 0 for samples that sit exactly on a pixel (always decided)
    LbpOffsets off;       // the exact sample offsets for the fp64 path
    uint32_t* part;       // [frame][tiles][GL_BINS]
    uint8_t* codes;       // optional per-pixel codes
};

// The LDS tile holds, for every pixel position, the pixel and its right neighbour as one packed pair
// (low half = the pixel, high half = the next one): a cell row is then one aligned 32-bit read and one
// v_dot2_u32_u16 with the row's two weights.  PT > 0 unrolls the sample loop for that P (the per-sample
// constants become scalar registers loaded in batches); PT == 0 is the loop for any other P.
//
// RING: the cell positions of (P=16, R=2) and (P=24, R=3) -- the two extra radii of the extended analysis -- are
// compile-time tables, so a cell row's LDS address is the pixel's base register plus an immediate offset (two
// address additions less per sample, a fifth of the instructions of the sample loop).
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dot2(uint32_t a, uint32_t b, uint32_t acc) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), acc, false);
}

// floor of the row / column offset of sample p, and whether the cell has a second row / column
constexpr int RING16_FR[16] = {0, -1, -2, -2, -2, -2, -2, -1, 0, 0, 1, 1, 2, 1, 1, 0};
constexpr int RING16_FC[16] = {2, 1, 1, 0, 0, -1, -2, -2, -2, -2, -2, -1, 0, 0, 1, 1};
constexpr int RING16_D1[16] = {0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1};  // rows and columns alike
constexpr int RING24_FR[24] = {0, -1, -2, -3, -3, -3, -3, -3, -3, -3, -2, -1, 0, 0, 1, 2, 2, 2, 3, 2, 2, 2, 1, 0};
constexpr int RING24_FC[24] = {3, 2, 2, 2, 1, 0, 0, -1, -2, -3, -3, -3, -3, -3, -3, -3, -2, -1, 0, 0, 1, 2, 2, 2};
constexpr int RING24_D1[24] = {0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1};
template <int PT>
struct Ring {
    static constexpr int HALO = PT == 16 ? 2 : 3;
    static constexpr int LPITCH = (GL_TILE + 2 * HALO) | 1;
    static constexpr int fr(int p) { return PT == 16 ? RING16_FR[p] : RING24_FR[p]; }
    static constexpr int fc(int p) { return PT == 16 ? RING16_FC[p] : RING24_FC[p]; }
    static constexpr int d1(int p) { return PT == 16 ? RING16_D1[p] : RING24_D1[p]; }
    // entry offsets relative to the entry HALO rows and HALO columns up and left of the centre (all >= 0)
    static constexpr int top(int p) { return (fr(p) + HALO) * LPITCH + fc(p) + HALO; }
    static constexpr int bot(int p) { return top(p) + d1(p) * LPITCH; }
};

constexpr int GL_QUEUE = 256;  // undecided pixels a wave can queue per strip of GL_ROWS rows

// Fills the LDS tile of packed pairs (pixel, right neighbour) for the GL_TILE x GL_TILE pixels at (x0, y0) and `halo`
// pixels around them; pixels outside the image are zero.  gp: the frame's padded gray plane.
__device__ __forceinline__ void load_pair_tile(uint32_t* gl_tile, const uint8_t* gp, int gpitch, int H, int W, int x0, int y0,
                                               int halo, int lpitch, int tid) {
    const int trows = GL_TILE + 2 * halo;
    // Tiles that lie inside the image with their halo (and one more column for the last pair) need no zero fill:
    // a thread takes eight entries of a row from twelve bytes (three unaligned 32-bit loads), every load of the
    // tile is issued before the first LDS store, and there is no division.  (The general loop below waits for
    // two dependent byte loads per entry, nineteen times in a row per thread.)
    const bool inside = x0 - halo >= 0 && y0 - halo >= 0 && x0 + GL_TILE + halo + 1 <= W && y0 + GL_TILE + halo <= H;
    if (inside) {
        constexpr int SEG = 8;
        const int segs = (GL_TILE + 2 * halo + SEG - 1) / SEG;  // per row; the last one may run past the row's entries
        const int tasks = trows * segs;
        constexpr int ROUNDS = 4;  // tasks in flight per thread: 1024 per pass of the outer loop (a halo of 3 has 630)
        uint32_t w[ROUNDS][3];
        int rr[ROUNDS], ss[ROUNDS];
        for (int t0 = 0; t0 < tasks; t0 += 256 * ROUNDS) {
#pragma unroll
            for (int k = 0; k < ROUNDS; ++k) {
                const int t = t0 + tid + 256 * k;
                rr[k] = t / segs;
                ss[k] = t - rr[k] * segs;
                if (t < tasks) {
                    const uint8_t* src = gp + (int64_t)(y0 - halo + rr[k] + GRAY_YOFF) * gpitch + GRAY_XOFF + x0 - halo + SEG * ss[k];
                    __builtin_memcpy(&w[k][0], src, 4);
                    __builtin_memcpy(&w[k][1], src + 4, 4);
                    __builtin_memcpy(&w[k][2], src + 8, 4);
                }
            }
#pragma unroll
            for (int k = 0; k < ROUNDS; ++k) {
                const int t = t0 + tid + 256 * k;
                if (t < tasks) {
                    uint32_t* dst = gl_tile + rr[k] * lpitch + SEG * ss[k];
                    const int left = lpitch - SEG * ss[k];  // entries of the row from here on
#pragma unroll
                    for (int j = 0; j < SEG; ++j) {
                        // bytes j and j+1 of the twelve, one per 16-bit half
                        const uint32_t lo = w[k][j >> 2], hi = w[k][(j + 1) >> 2];
                        const uint32_t e = ((lo >> (8 * (j & 3))) & 0xFFu) | (((hi >> (8 * ((j + 1) & 3))) & 0xFFu) << 16);
                        if (j < left) dst[j] = e;
                    }
                }
            }
        }
    } else
    for (int i = tid; i < trows * lpitch; i += 256) {
        const int r = i / lpitch, cidx = i - r * lpitch;
        const int gy = y0 - halo + r, gx = x0 - halo + cidx;
        uint32_t v = 0;
        if (gy >= 0 && gy < H) {
            const uint8_t* rowp = gp + (int64_t)(gy + GRAY_YOFF) * gpitch + GRAY_XOFF;
            if (gx >= 0 && gx < W) v = rowp[gx];
            if (gx + 1 >= 0 && gx + 1 < W) v |= (uint32_t)rowp[gx + 1] << 16;
        }
        gl_tile[i] = v;
    }
}

template <int PT, bool RING>
__global__ __launch_bounds__(256) void lbp_generic_kernel(LbpGenArgs a) {
    extern __shared__ uint32_t gl_tile[];  // (GL_TILE + 2*halo) rows of lpitch pairs
    // Sixteen copies of the histogram, 33 words apart (a copy per bank offset): neighbouring pixels mostly
    // fall into the same bin, and 64 lanes adding to one LDS word would be served one after the other.
    __shared__ uint32_t hist[GL_COPIES * GL_HSTRIDE];
    __shared__ double offs[2][24];
    // undecided pixels of each wave's strip: .x = sign bits | row << 24, .y = undecided samples | lane << 24
    __shared__ uint2 queue[4][GL_QUEUE];
    const int frame = blockIdx.z;
    const int x0 = blockIdx.x * GL_TILE, y0 = blockIdx.y * GL_TILE;
    const int tid = threadIdx.x;
    const int halo = RING ? Ring<PT>::HALO : a.halo, lpitch = RING ? Ring<PT>::LPITCH : a.lpitch;
    load_pair_tile(gl_tile, a.gray + (int64_t)frame * a.grows * a.gpitch, a.gpitch, a.H, a.W, x0, y0, halo, lpitch, tid);
    for (int i = tid; i < GL_COPIES * GL_HSTRIDE; i += 256) hist[i] = 0;
    if (tid < 24) { offs[0][tid] = a.off.rp[tid]; offs[1][tid] = a.off.cp[tid]; }
    __syncthreads();

    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = x0 + lane;
    const int P = PT > 0 ? PT : a.P;
    const uint32_t pmask = (P >= 32) ? 0xFFFFFFFFu : ((1u << P) - 1u);
    // 0/1 changes between consecutive samples (not circular), as scikit-image counts them; the reversed bit order
    // of the sign words changes neither that count nor the number of ones
    const uint32_t pairmask = (P >= 2) ? ((1u << (P - 1)) - 1u) : 0u;
    auto count_pixel = [&](uint32_t bits, int y, int xx) {
        const int changes = __popc((bits ^ (bits >> 1)) & pairmask);
        const int code = (changes <= 2) ? __popc(bits) : P + 1;
        atomicAdd(&hist[(lane & (GL_COPIES - 1)) * GL_HSTRIDE + code], 1u);
        if (a.codes) a.codes[((int64_t)frame * a.H + y) * a.W + xx] = (uint8_t)code;
    };
    // the undecided samples (bits of near) of pixel (y, xx) in skimage's bilinear_interpolation, operation by operation
    auto settle = [&](uint32_t bits, uint32_t near, int y, int xx) -> uint32_t {
        const double c = (double)(gl_tile[(y - y0 + halo) * lpitch + (xx - x0 + halo)] & 0xFFFFu);
        if constexpr (RING) {
            // which of the pixel's blended samples are too close to call: the row loop's arithmetic once more
            const uint8_t* pr = reinterpret_cast<const uint8_t*>(gl_tile) + 4 * ((y - y0) * lpitch + (xx - x0));
            const uint32_t init = a.margin - ((uint32_t)c << 16);
            near = 0;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                near += near;
                if (Ring<PT>::d1(p) != 0) {
                    const uint32_t top = *reinterpret_cast<const uint32_t*>(pr + 4 * Ring<PT>::top(p));
                    const uint32_t bot = *reinterpret_cast<const uint32_t*>(pr + 4 * Ring<PT>::bot(p));
                    const uint32_t t = dot2(top, a.wtop[p], dot2(bot, a.wbot[p], init));
                    near |= t < 2u * a.margin + 1u ? 1u : 0u;
                }
            }
        }
        while (near != 0) {
            const int b = __ffs(near) - 1;
            near &= near - 1;
            const int p = P - 1 - b;
            const double rf = (double)y + offs[0][p], cf = (double)xx + offs[1][p];
            const double fr = floor(rf), fc = floor(cf);
            const int minr = (int)fr, minc = (int)fc;
            const int maxr = (int)ceil(rf), maxc = (int)ceil(cf);
            const double dr = rf - fr, dc = cf - fc;
            auto px = [&](int yy, int xq) -> double {
                return (double)(gl_tile[(yy - y0 + halo) * lpitch + (xq - x0 + halo)] & 0xFFFFu);
            };
            const double tl = px(minr, minc), tr = px(minr, maxc), bl = px(maxr, minc), br = px(maxr, maxc);
            const double top = (1.0 - dc) * tl + dc * tr;
            const double bottom = (1.0 - dc) * bl + dc * br;
            const double t = (1.0 - dr) * top + dr * bottom;
            const uint32_t bit = (t - c >= 0.0) ? 1u : 0u;
            bits = (bits & ~(1u << b)) | (bit << b);
        }
        return bits;
    };
    uint32_t qn = 0;  // entries in this wave's queue (wave-uniform)
    for (int i = 0; i < GL_ROWS; ++i) {
        const int y = y0 + wv * GL_ROWS + i;
        if (y >= a.H) break;
        const int pa = (halo + wv * GL_ROWS + i) * lpitch + halo + lane;
        const uint8_t* pb = reinterpret_cast<const uint8_t*>(gl_tile) + 4 * pa;
        // RING: every cell offset is non-negative from here, i.e. fits the immediate offset of an LDS read.  The byte
        // offset passes through an empty asm statement so that the compiler keeps this base and does not re-base the
        // reads on another register, which costs an address addition for every read that then lies below it.
        uint32_t pro = 4u * (uint32_t)((wv * GL_ROWS + i) * lpitch + lane);
        asm volatile("" : "+v"(pro));
        const uint8_t* pr = reinterpret_cast<const uint8_t*>(gl_tile) + pro;
        const uint32_t c = gl_tile[pa] & 0xFFFFu;
        const uint32_t init = a.margin - (c << 16);
        uint32_t signs = 0, near = 0;  // sample p at bit P-1-p
        // RING: the smallest t of the row's blended samples instead of a flag per sample (one v_min3 per two samples
        // against a compare, a select and an or per sample); which samples of an undecided pixel are the undecided
        // ones is worked out again when the pixel is settled
        uint32_t tmin = 0xFFFFFFFFu;
        auto sample = [&](int p) {
            // t = (sample - centre) + margin in 1/65536 gray levels
            uint32_t top, bot;
            if constexpr (RING) {
                top = *reinterpret_cast<const uint32_t*>(pr + 4 * Ring<PT>::top(p));
                if (Ring<PT>::d1(p) == 0) {
                    // the sample is a pixel: one read, one shift-and-add, never undecided
                    signs = __builtin_amdgcn_alignbit(signs, (top << 16) + init, 31);
                    return;
                }
                bot = *reinterpret_cast<const uint32_t*>(pr + 4 * Ring<PT>::bot(p));
                const uint32_t t = dot2(top, a.wtop[p], dot2(bot, a.wbot[p], init));
                signs = __builtin_amdgcn_alignbit(signs, t, 31);
                tmin = tmin < t ? tmin : t;
                return;
            } else {
                top = *reinterpret_cast<const uint32_t*>(pb + a.otop[p]);
                bot = *reinterpret_cast<const uint32_t*>(pb + a.obot[p]);
            }
            const uint32_t t = dot2(top, a.wtop[p], dot2(bot, a.wbot[p], init));
            signs = __builtin_amdgcn_alignbit(signs, t, 31);
            // 0 <= t <= 2*margin: too close to call (t < lim as unsigned; lim = 0 never flags).  Written as an
            // add so that the comparison's carry feeds the doubling of near directly.
            near = near + near + (t < a.lim[p] ? 1u : 0u);
        };
        if constexpr (PT > 0) {
#pragma unroll
            for (int p = 0; p < PT; ++p) sample(p);
        } else {
            for (int p = 0; p < P; ++p) sample(p);
        }
        const uint32_t bits = ~signs & pmask;  // LBP bit = sample >= centre
        const bool valid = x < a.W;
        // 0 <= t <= 2*margin: too close to call
        const bool und = valid && (RING ? tmin < 2u * a.margin + 1u : near != 0);
        if (valid && !und) count_pixel(bits, y, x);
        // Undecided pixels (ties, mostly) wait in the wave's queue for the end of the strip, where any lane takes
        // any entry: the fp64 replay then costs ceil(queued / 64) rounds instead of the worst lane's count per row.
        const unsigned long long um = __ballot(und);
        if (um != 0) {
            const uint32_t cnt = (uint32_t)__popcll(um);
            if (qn + cnt <= GL_QUEUE) {
                if (und) {
                    const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(um >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)um, 0u));
                    queue[wv][pos] = make_uint2(bits | ((uint32_t)i << 24), near | ((uint32_t)lane << 24));
                }
                qn += cnt;
            } else if (und) {
                count_pixel(settle(bits, near, y, x), y, x);  // a full queue: flat content, every lane is busy anyway
            }
        }
    }
    for (uint32_t qb = 0; qb < qn; qb += 64) {
        if (qb + lane < qn) {
            const uint2 e = queue[wv][qb + lane];
            const int y = y0 + wv * GL_ROWS + (int)(e.x >> 24), xx = x0 + (int)(e.y >> 24);
            count_pixel(settle(e.x & 0xFFFFFFu, e.y & 0xFFFFFFu, y, xx), y, xx);
        }
    }
    __syncthreads();
    if (tid < GL_BINS) {
        const int64_t tileid = ((int64_t)frame * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < GL_COPIES; ++k) v += hist[k * GL_HSTRIDE + tid];
        a.part[tileid * GL_BINS + tid] = v;
    }
}


// ---- (P=16, R=2) and (P=24, R=3) together: the two extra radii of the extended analysis from one tile.
// One LDS tile with the wider halo serves both rings.  A lane owns a pixel column and marches down its strip: the
// lower cell row of a sample at image row y is the upper cell row of the same sample at row y+1, so each blended
// sample keeps the word it read last in a register and reads one new word per row instead of two (41 LDS reads per
// pixel for the 32 blended and 8 on-pixel samples of the two rings and the centre, against 74 as two launches; the
// rows are unrolled in pairs so that the two sets of registers swap roles without copies).  Decisions, margins and
// the fp64 replay of undecided pixels are those of lbp_generic_kernel<PT, true>.
struct LbpRingsArgs {
    const uint8_t* gray;
    int H, W, gpitch, grows;
    uint32_t margin16, margin24;
    uint32_t wtop16[16], wbot16[16], wtop24[24], wbot24[24];
    LbpOffsets off16, off24;
    uint32_t* part16;  // [frame][tiles][GL_BINS]
    uint32_t* part24;
};

// cell offsets of ring PT in the tile of halo 3 (entries, relative to the entry 3 rows and 3 columns up and left of the centre)
template <int PT>
struct Ring3 {
    static constexpr int HALO = 3;
    static constexpr int LPITCH = (GL_TILE + 2 * HALO) | 1;
    static constexpr int top(int p) { return (Ring<PT>::fr(p) + HALO) * LPITCH + Ring<PT>::fc(p) + HALO; }
    static constexpr int bot(int p) { return top(p) + Ring<PT>::d1(p) * LPITCH; }
};

// Which blended samples of ring PT at one pixel are too close to call (sample p at bit PT-1-p), both cell rows read
// again: t < lim as unsigned numbers is the borrow of t - lim, except where t itself has its top bit set (a sample
// decidedly below the centre), which the pixel's sign word -- `below`, one bit per sample -- tells.
template <int PT>
__device__ __forceinline__ uint32_t ring_near(const uint8_t* pr, const uint32_t* wtop, const uint32_t* wbot, uint32_t init,
                                              uint32_t lim, uint32_t below) {
    uint32_t near = 0;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        if (Ring<PT>::d1(p) != 0) {
            const uint32_t top = *reinterpret_cast<const uint32_t*>(pr + 4 * Ring3<PT>::top(p));
            const uint32_t bot = *reinterpret_cast<const uint32_t*>(pr + 4 * Ring3<PT>::bot(p));
            const uint32_t t = dot2(top, wtop[p], dot2(bot, wbot[p], init));
            near = __builtin_amdgcn_alignbit(near, t - lim, 31);
        } else {
            near += near;
        }
    }
    return near & ~below;
}

// the samples of `near` (sample p at bit P-1-p) of pixel (y, xx) in skimage's bilinear_interpolation, operation by operation
__device__ __forceinline__ uint32_t ring_settle(const uint32_t* tile, int lpitch, int halo, int x0, int y0, int P,
                                                const double* rp, const double* cp, uint32_t bits, uint32_t near, int y, int xx) {
    const double c = (double)(tile[(y - y0 + halo) * lpitch + (xx - x0 + halo)] & 0xFFFFu);
    while (near != 0) {
        const int b = __ffs(near) - 1;
        near &= near - 1;
        const int p = P - 1 - b;
        const double rf = (double)y + rp[p], cf = (double)xx + cp[p];
        const double fr = floor(rf), fc = floor(cf);
        const int minr = (int)fr, minc = (int)fc;
        const int maxr = (int)ceil(rf), maxc = (int)ceil(cf);
        const double dr = rf - fr, dc = cf - fc;
        auto px = [&](int yy, int xq) -> double {
            return (double)(tile[(yy - y0 + halo) * lpitch + (xq - x0 + halo)] & 0xFFFFu);
        };
        const double tl = px(minr, minc), tr = px(minr, maxc), bl = px(maxr, minc), br = px(maxr, maxc);
        const double top = (1.0 - dc) * tl + dc * tr;
        const double bottom = (1.0 - dc) * bl + dc * br;
        const double t = (1.0 - dr) * top + dr * bottom;
        const uint32_t bit = (t - c >= 0.0) ? 1u : 0u;
        bits = (bits & ~(1u << b)) | (bit << b);
    }
    return bits;
}

__global__ __launch_bounds__(256) void lbp_rings_kernel(LbpRingsArgs a) {
    constexpr int HALO = Ring3<24>::HALO, LP = Ring3<24>::LPITCH, TROWS = GL_TILE + 2 * HALO;
    __shared__ uint32_t tile[TROWS * LP];
    __shared__ uint32_t hist16[GL_COPIES * GL_HSTRIDE], hist24[GL_COPIES * GL_HSTRIDE];
    __shared__ double offs[4][24];  // rp16, cp16, rp24, cp24
    // undecided pixels of each wave's strip: .x = bits of ring 16 | row << 16 | undecided in ring 16 << 30 | in ring 24
    // << 31, .y = bits of ring 24 | lane << 24
    // (128 entries per 1024-pixel strip: with the smaller queue a fifth workgroup fits a compute unit's LDS)
    constexpr int RQ = 128;
    __shared__ uint2 queue[4][RQ];
    const int frame = blockIdx.z;
    const int x0 = blockIdx.x * GL_TILE, y0 = blockIdx.y * GL_TILE;
    const int tid = threadIdx.x;
    load_pair_tile(tile, a.gray + (int64_t)frame * a.grows * a.gpitch, a.gpitch, a.H, a.W, x0, y0, HALO, LP, tid);
    for (int i = tid; i < GL_COPIES * GL_HSTRIDE; i += 256) { hist16[i] = 0; hist24[i] = 0; }
    if (tid < 24) {
        offs[0][tid] = a.off16.rp[tid]; offs[1][tid] = a.off16.cp[tid];
        offs[2][tid] = a.off24.rp[tid]; offs[3][tid] = a.off24.cp[tid];
    }
    __syncthreads();

    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = x0 + lane;
    const bool valid = x < a.W;
    const uint32_t lim16 = 2u * a.margin16 + 1u, lim24 = 2u * a.margin24 + 1u;
    auto count = [&](uint32_t* hist, int P, uint32_t bits) {
        const int changes = __popc((bits ^ (bits >> 1)) & ((1u << (P - 1)) - 1u));
        const int code = (changes <= 2) ? __popc(bits) : P + 1;
        atomicAdd(&hist[(lane & (GL_COPIES - 1)) * GL_HSTRIDE + code], 1u);
    };
    // the words the blended samples read last: X holds the upper cell rows of the row about to be processed
    uint32_t X16[16], Y16[16], X24[24], Y24[24];
    const int ys = y0 + wv * GL_ROWS;
    uint32_t pro = 4u * (uint32_t)(wv * GL_ROWS * LP + lane);
    asm volatile("" : "+v"(pro));
    const uint8_t* pr0 = reinterpret_cast<const uint8_t*>(tile) + pro;
#pragma unroll
    for (int p = 0; p < 16; ++p)
        if (Ring<16>::d1(p)) X16[p] = *reinterpret_cast<const uint32_t*>(pr0 + 4 * Ring3<16>::top(p));
#pragma unroll
    for (int p = 0; p < 24; ++p)
        if (Ring<24>::d1(p)) X24[p] = *reinterpret_cast<const uint32_t*>(pr0 + 4 * Ring3<24>::top(p));
    uint32_t qn = 0;  // entries in this wave's queue (wave-uniform)
    // one image row: T = upper cell rows (read a row ago), B = lower cell rows (read now, next row's upper ones)
    auto row = [&](int i, uint32_t (&T16)[16], uint32_t (&B16)[16], uint32_t (&T24)[24], uint32_t (&B24)[24]) {
        const uint8_t* pr = pr0 + 4 * i * LP;
        const uint32_t c = *reinterpret_cast<const uint32_t*>(pr + 4 * (HALO * LP + HALO)) & 0xFFFFu;
        const uint32_t init16 = a.margin16 - (c << 16), init24 = a.margin24 - (c << 16);
        uint32_t s16 = 0, s24 = 0, tmin16 = 0xFFFFFFFFu, tmin24 = 0xFFFFFFFFu;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            if (Ring<16>::d1(p) == 0) {
                const uint32_t v = *reinterpret_cast<const uint32_t*>(pr + 4 * Ring3<16>::top(p));
                s16 = __builtin_amdgcn_alignbit(s16, (v << 16) + init16, 31);
            } else {
                B16[p] = *reinterpret_cast<const uint32_t*>(pr + 4 * Ring3<16>::bot(p));
                const uint32_t t = dot2(T16[p], a.wtop16[p], dot2(B16[p], a.wbot16[p], init16));
                s16 = __builtin_amdgcn_alignbit(s16, t, 31);
                tmin16 = tmin16 < t ? tmin16 : t;
            }
        }
#pragma unroll
        for (int p = 0; p < 24; ++p) {
            if (Ring<24>::d1(p) == 0) {
                const uint32_t v = *reinterpret_cast<const uint32_t*>(pr + 4 * Ring3<24>::top(p));
                s24 = __builtin_amdgcn_alignbit(s24, (v << 16) + init24, 31);
            } else {
                B24[p] = *reinterpret_cast<const uint32_t*>(pr + 4 * Ring3<24>::bot(p));
                const uint32_t t = dot2(T24[p], a.wtop24[p], dot2(B24[p], a.wbot24[p], init24));
                s24 = __builtin_amdgcn_alignbit(s24, t, 31);
                tmin24 = tmin24 < t ? tmin24 : t;
            }
        }
        const uint32_t bits16 = ~s16 & 0xFFFFu, bits24 = ~s24 & 0xFFFFFFu;  // LBP bit = sample >= centre
        const bool und16 = valid && tmin16 < lim16, und24 = valid && tmin24 < lim24;
        if (valid && !und16) count(hist16, 16, bits16);
        if (valid && !und24) count(hist24, 24, bits24);
        const bool und = und16 || und24;
        const unsigned long long um = __ballot(und);
        if (um != 0) {
            const uint32_t cnt = (uint32_t)__popcll(um);
            if (qn + cnt <= RQ) {
                if (und) {
                    const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(um >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)um, 0u));
                    queue[wv][pos] = make_uint2(bits16 | ((uint32_t)i << 16) | (und16 ? 1u << 30 : 0u) | (und24 ? 1u << 31 : 0u),
                                                bits24 | ((uint32_t)lane << 24));
                }
                qn += cnt;
            } else if (und) {  // a full queue: flat content, every lane is busy anyway
                const int y = ys + i;
                if (und16)
                    count(hist16, 16, ring_settle(tile, LP, HALO, x0, y0, 16, offs[0], offs[1], bits16,
                                                  ring_near<16>(pr, a.wtop16, a.wbot16, init16, lim16, ~bits16), y, x));
                if (und24)
                    count(hist24, 24, ring_settle(tile, LP, HALO, x0, y0, 24, offs[2], offs[3], bits24,
                                                  ring_near<24>(pr, a.wtop24, a.wbot24, init24, lim24, ~bits24), y, x));
            }
        }
    };
    for (int i = 0; i < GL_ROWS; i += 2) {
        if (ys + i >= a.H) break;
        row(i, X16, Y16, X24, Y24);
        if (ys + i + 1 >= a.H) break;
        row(i + 1, Y16, X16, Y24, X24);
    }
    for (uint32_t qb = 0; qb < qn; qb += 64) {
        if (qb + lane < qn) {
            const uint2 e = queue[wv][qb + lane];
            const int i = (int)((e.x >> 16) & 0xFFu), ql = (int)(e.y >> 24);
            const int y = ys + i, xx = x0 + ql;
            const uint8_t* pr = reinterpret_cast<const uint8_t*>(tile) + 4 * ((wv * GL_ROWS + i) * LP + ql);
            const uint32_t c = tile[(wv * GL_ROWS + i + HALO) * LP + ql + HALO] & 0xFFFFu;
            if (e.x & (1u << 30))
                count(hist16, 16, ring_settle(tile, LP, HALO, x0, y0, 16, offs[0], offs[1], e.x & 0xFFFFu,
                                              ring_near<16>(pr, a.wtop16, a.wbot16, a.margin16 - (c << 16), lim16, ~e.x), y, xx));
            if (e.x & (1u << 31))
                count(hist24, 24, ring_settle(tile, LP, HALO, x0, y0, 24, offs[2], offs[3], e.y & 0xFFFFFFu,
                                              ring_near<24>(pr, a.wtop24, a.wbot24, a.margin24 - (c << 16), lim24, ~e.y), y, xx));
        }
    }
    __syncthreads();
    if (tid < 2 * GL_BINS) {
        const int64_t tileid = ((int64_t)frame * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const int b = tid & (GL_BINS - 1);
        const uint32_t* hist = tid < GL_BINS ? hist16 : hist24;
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < GL_COPIES; ++k) v += hist[k * GL_HSTRIDE + b];
        (tid < GL_BINS ? a.part16 : a.part24)[tileid * GL_BINS + b] = v;
    }
}

// Per-frame sum of the tile histograms, in a fixed order.
__global__ __launch_bounds__(256) void lbp_hist_reduce_kernel(const uint32_t* __restrict__ part, int tiles, int bins,
                                                              FrameAcc* __restrict__ acc, int hist_slot) {
    __shared__ uint32_t red[8][GL_BINS];
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int q = tid & (GL_BINS - 1), grp = tid >> 5;
    uint32_t s = 0;
    for (int t = grp; t < tiles; t += 8) s += part[((int64_t)frame * tiles + t) * GL_BINS + q];
    red[grp][q] = s;
    __syncthreads();
    if (tid < bins) {
        uint32_t v = 0;
        for (int g = 0; g < 8; ++g) v += red[g][tid];
        uint32_t* h = hist_slot == 0 ? acc[frame].lbp_hist : (hist_slot == 1 ? acc[frame].lbp_hist2 : acc[frame].lbp_hist3);
        h[tid] = v;
    }
}

int lbp_generic_tiles(const Geom& g) { return ((g.W + GL_TILE - 1) / GL_TILE) * ((g.H + GL_TILE - 1) / GL_TILE); }
size_t lbp_generic_part_bytes(const Geom& g, int n_frames) {
    // (two sets: the fused kernel for the two extra radii leaves both histograms at once)
    return (size_t)2 * n_frames * lbp_generic_tiles(g) * GL_BINS * sizeof(uint32_t);
}

// Cell geometry, 16.16 weights and error margin of every sample of (P, off); *ring says whether the geometry is that
// of the compile-time tables for this P (P = 16 or 24).  Returns false when the radius needs more LDS than a tile has.
static bool build_gen_args(const Geom& g, int P, const LbpOffsets& off, LbpGenArgs* ap, size_t* lds_out, bool* ring) {
    LbpGenArgs& a = *ap;
    a.H = g.H; a.W = g.W; a.gpitch = g.gpitch; a.grows = g.grows; a.P = P;
    a.off = off;
    // cell geometry and weights from the nominal offsets (pixel (0,0)): the per-pixel fp64 weights differ from
    // these by rounding in the 13th digit, which the guard in the margin covers
    int halo = 1;
    double worst = 0.0;
    int fr[24], fc[24], dr1[24], dc1[24];
    double wr[24], wc[24];
    for (int p = 0; p < P; ++p) {
        const double rp = off.rp[p], cp = off.cp[p];
        const double flr = floor(rp), flc = floor(cp);
        fr[p] = (int)flr; fc[p] = (int)flc;
        wr[p] = rp - flr; wc[p] = cp - flc;
        dr1[p] = wr[p] > 0.0 ? 1 : 0;
        dc1[p] = wc[p] > 0.0 ? 1 : 0;
        const int need_r = fr[p] < 0 ? -fr[p] : fr[p] + dr1[p];
        const int need_c = fc[p] < 0 ? -fc[p] : fc[p] + dc1[p];
        if (need_r > halo) halo = need_r;
        if (need_c > halo) halo = need_c;
    }
    // odd pitch in 32-bit entries: consecutive tile rows start in different LDS banks
    const int lpitch = (GL_TILE + 2 * halo) | 1;
    const size_t lds = (size_t)(GL_TILE + 2 * halo) * lpitch * sizeof(uint32_t);
    *lds_out = lds;
    if (lds > 52 * 1024) return false;  // radius beyond what one tile with its halo can hold (static LDS takes 11 of the 64 KB)
    a.halo = halo; a.lpitch = lpitch;
    for (int p = 0; p < 24; ++p) { a.otop[p] = 0; a.obot[p] = 0; a.wtop[p] = 0; a.wbot[p] = 0; a.lim[p] = 0; }
    for (int p = 0; p < P; ++p) {
        a.otop[p] = 4 * (fr[p] * lpitch + fc[p]);
        a.obot[p] = a.otop[p] + 4 * dr1[p] * lpitch;
        const double wex[4] = {(1.0 - wr[p]) * (1.0 - wc[p]), (1.0 - wr[p]) * wc[p], wr[p] * (1.0 - wc[p]), wr[p] * wc[p]};
        double ep = 0.0, en = 0.0;
        uint32_t q4[4];
        for (int k = 0; k < 4; ++k) {
            const double exact = 65536.0 * wex[k];
            const long long q = llround(exact);
            const double e = (double)q - exact;
            if (e > 0) ep += e; else en -= e;
            q4[k] = (uint32_t)q;
        }
        // a weight of exactly 65536 (a sample on a pixel) does not fit 16 bits: split it between the two
        // rows of the "cell", which are the same row then
        if (dr1[p] == 0 && dc1[p] == 0) {
            a.wtop[p] = 32768u; a.wbot[p] = 32768u;
        } else if (dr1[p] == 0) {
            // one row, two columns: the lower row's weights are zero; split the upper row over both reads
            a.wtop[p] = (q4[0] >> 1) | ((q4[1] >> 1) << 16);
            a.wbot[p] = (q4[0] - (q4[0] >> 1)) | ((q4[1] - (q4[1] >> 1)) << 16);
        } else if (dc1[p] == 0) {
            // one column, two rows
            a.wtop[p] = q4[0] >= 65536u ? 65535u : q4[0];
            a.wbot[p] = q4[2] >= 65536u ? 65535u : q4[2];
        } else {
            a.wtop[p] = q4[0] | (q4[1] << 16);
            a.wbot[p] = q4[2] | (q4[3] << 16);
        }
        worst = fmax(worst, 255.0 * fmax(ep, en));
    }
    a.margin = (uint32_t)ceil(worst) + 2u;
    for (int p = 0; p < P; ++p) a.lim[p] = (dr1[p] == 0 && dc1[p] == 0) ? 0u : 2u * a.margin + 1u;
    // the compile-time cell tables apply when this launch's geometry is theirs (it is for R = 2 at P = 16 and R = 3 at
    // P = 24, whatever rounding of the offsets the caller's platform produced is checked here, not assumed)
    auto ring_ok = [&](auto rg) {
        typedef decltype(rg) G;
        if (halo != G::HALO || lpitch != G::LPITCH) return false;
        for (int p = 0; p < P; ++p)
            if (fr[p] != G::fr(p) || fc[p] != G::fc(p) || dr1[p] != G::d1(p) || dc1[p] != G::d1(p)) return false;
        return true;
    };
    *ring = (P == 16 && ring_ok(Ring<16>())) || (P == 24 && ring_ok(Ring<24>()));
    return true;
}

bool launch_lbp_generic(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, int P, const LbpOffsets& off,
                        uint32_t* part, FrameAcc* acc, uint8_t* codes_out, int hist_slot) {
    LbpGenArgs a;
    size_t lds = 0;
    bool ring = false;
    if (!build_gen_args(g, P, off, &a, &lds, &ring)) return false;
    a.gray = gray; a.part = part; a.codes = codes_out;
    dim3 grid((g.W + GL_TILE - 1) / GL_TILE, (g.H + GL_TILE - 1) / GL_TILE, n_frames);
    if (P == 16 && ring)
        hipLaunchKernelGGL((lbp_generic_kernel<16, true>), grid, dim3(256), lds, s, a);
    else if (P == 24 && ring)
        hipLaunchKernelGGL((lbp_generic_kernel<24, true>), grid, dim3(256), lds, s, a);
    else
        switch (P) {
            case 8: hipLaunchKernelGGL((lbp_generic_kernel<8, false>), grid, dim3(256), lds, s, a); break;
            case 16: hipLaunchKernelGGL((lbp_generic_kernel<16, false>), grid, dim3(256), lds, s, a); break;
            case 24: hipLaunchKernelGGL((lbp_generic_kernel<24, false>), grid, dim3(256), lds, s, a); break;
            default: hipLaunchKernelGGL((lbp_generic_kernel<0, false>), grid, dim3(256), lds, s, a); break;
        }
    hipLaunchKernelGGL(lbp_hist_reduce_kernel, dim3(n_frames), dim3(256), 0, s, part, lbp_generic_tiles(g), P + 2, acc, hist_slot);
    return true;
}

bool launch_lbp_rings(hipStream_t s, const uint8_t* gray, int n_frames, const Geom& g, const LbpOffsets& off16,
                      const LbpOffsets& off24, uint32_t* part, FrameAcc* acc) {
    LbpGenArgs a16, a24;
    size_t lds = 0;
    bool ring16 = false, ring24 = false;
    // the fused kernel is written for the cell geometry of the compile-time tables; anything else takes two launches
    if (!build_gen_args(g, 16, off16, &a16, &lds, &ring16) || !build_gen_args(g, 24, off24, &a24, &lds, &ring24) ||
        !ring16 || !ring24)
        return launch_lbp_generic(s, gray, n_frames, g, 16, off16, part, acc, nullptr, 1) &&
               launch_lbp_generic(s, gray, n_frames, g, 24, off24, part, acc, nullptr, 2);
    LbpRingsArgs a;
    a.gray = gray; a.H = g.H; a.W = g.W; a.gpitch = g.gpitch; a.grows = g.grows;
    a.margin16 = a16.margin; a.margin24 = a24.margin;
    for (int p = 0; p < 16; ++p) { a.wtop16[p] = a16.wtop[p]; a.wbot16[p] = a16.wbot[p]; }
    for (int p = 0; p < 24; ++p) { a.wtop24[p] = a24.wtop[p]; a.wbot24[p] = a24.wbot[p]; }
    a.off16 = off16; a.off24 = off24;
    const int tiles = lbp_generic_tiles(g);
    a.part16 = part;
    a.part24 = part + (size_t)n_frames * tiles * GL_BINS;
    dim3 grid((g.W + GL_TILE - 1) / GL_TILE, (g.H + GL_TILE - 1) / GL_TILE, n_frames);
    hipLaunchKernelGGL(lbp_rings_kernel, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(lbp_hist_reduce_kernel, dim3(n_frames), dim3(256), 0, s, a.part16, tiles, 18, acc, 1);
    hipLaunchKernelGGL(lbp_hist_reduce_kernel, dim3(n_frames), dim3(256), 0, s, a.part24, tiles, 26, acc, 2);
    return true;
}

}  // namespace stx
