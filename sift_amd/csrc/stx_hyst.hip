// Canny hysteresis (second half of cv2.Canny at /root/reference/app/passive_checker.py:116)
// on one-bit-per-pixel maps: "strong" grows into every "candidate" pixel that is
// 8-connected to it, transitively.
//
// The maps run down the image columns: a "line" is one pixel column, a 64-bit word holds 64
// vertically adjacent pixels of it, and the maps are stored [frame][word][line], so the 64
// words of a 64x64 tile are contiguous.  (8-connectivity does not care which axis the words
// run along; the stencil kernel produces its bits column by column.)  One wave owns a tile:
// lane = line.  Neighbours across lines come from the adjacent lanes, neighbours along a
// line from shifts, and a strong bit floods a whole run of candidates along its line in O(1)
// with a carry chain: adding the seed bits to the run mask ripples a carry to the end of
// each run.  The tile
// iterates to its local fix-point, then the host launches further passes until a pass
// changes nothing anywhere (stale reads of a neighbouring tile's words are harmless: the
// map only ever grows, and the last pass re-checks every tile against final neighbours).
#include <algorithm>

#include "stx_internal.h"

namespace stx {

__device__ __forceinline__ unsigned long long brev64(unsigned long long v) {
    return ((unsigned long long)__brev((uint32_t)v) << 32) | (unsigned long long)__brev((uint32_t)(v >> 32));
}

// A word from the neighbouring lane without a trip through the LDS crossbar: lane i takes lane i-1's v (lane 0 takes
// `edge`), or lane i+1's (lane 63 takes `edge`) -- the tile's halo lines slot in where the shift has no source.
__device__ __forceinline__ unsigned long long from_lane_below(unsigned long long v, unsigned long long edge) {
    const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)edge, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);  // wave_shr:1
    const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(edge >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
    return (unsigned long long)(uint32_t)lo | ((unsigned long long)(uint32_t)hi << 32);
}
__device__ __forceinline__ unsigned long long from_lane_above(unsigned long long v, unsigned long long edge) {
    const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)edge, (int)(uint32_t)v, 0x130, 0xf, 0xf, false);  // wave_shl:1
    const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(edge >> 32), (int)(uint32_t)(v >> 32), 0x130, 0xf, 0xf, false);
    return (unsigned long long)(uint32_t)lo | ((unsigned long long)(uint32_t)hi << 32);
}

// all bits of the candidate runs in c that contain a bit of x (x subset of c)
__device__ __forceinline__ unsigned long long flood_runs(unsigned long long c, unsigned long long x) {
    const unsigned long long up = (c ^ (c + x)) & c;
    const unsigned long long cr = brev64(c), xr = brev64(x);
    const unsigned long long dn = brev64((cr ^ (cr + xr)) & cr);
    return up | dn | x;
}

// tflag: [2][frames][ntiles][nwords] bytes, one per tile and pass parity: "this tile changed in
// that pass".  From pass 1 on a tile is revisited only if it or one of its eight neighbours
// changed in the previous pass.  nlines = image width, nwords = words per line, ntiles =
// 64-line tiles across the image.
// returns (wave-uniform) whether the tile changed; changed may be null
// cvalid: null, or one byte per tile: 0 = the tile's words of `cand` were never written and count as empty
__device__ __forceinline__ bool hyst_tile(unsigned long long* __restrict__ strong, const unsigned long long* __restrict__ cand,
                                          int nlines, int nwords, int ntiles, int tile_id, unsigned int* changed, int pass,
                                          uint32_t* __restrict__ tile_count, uint8_t* tflag, int64_t tflag_half,
                                          const uint8_t* __restrict__ cvalid);

// Four waves per workgroup, each working on tiles of its own.
constexpr int HYST_WAVES = 4;
constexpr int HYST_MAX_GRID = 4096;  // workgroups of the first two passes; each walks over the tiles with this stride
constexpr int HYST_LATER_GRID = 1024;  // the passes after them visit few tiles: fewer workgroups to dispatch
__global__ __launch_bounds__(64 * HYST_WAVES) void hyst_pass_kernel(unsigned long long* __restrict__ strong,
                                                                    const unsigned long long* __restrict__ cand,
                                                                    int nlines, int nwords, int ntiles, int total_tiles,
                                                                    unsigned int* changed, int pass,
                                                                    uint32_t* __restrict__ tile_count, uint8_t* tflag,
                                                                    int64_t tflag_half, const uint8_t* __restrict__ cvalid) {
    if (pass > 0 && changed[pass - 1] == 0) return;
    const int wave = blockIdx.x * HYST_WAVES + (threadIdx.x >> 6), nwaves = gridDim.x * HYST_WAVES;
    if (pass == 0) {
        // The stencil left, in the flag half that pass 1 will overwrite, which tiles hold a strong pixel.  Nothing
        // can grow in a tile with no strong pixel in it or in its eight neighbours: a wave tests 64 tiles at once
        // (one per lane) and enters only the others.
        const int lane0 = threadIdx.x & 63;
        const int tpf0 = ntiles * nwords;
        const uint8_t* sflag = tflag + tflag_half;
        for (int j0 = 0; j0 * nwaves + wave < total_tiles; j0 += 64) {
            const int tile_id = (j0 + lane0) * nwaves + wave;
            int f = 0;
            if (tile_id < total_tiles) {
                const int frame = tile_id / tpf0;
                const int rem = tile_id - frame * tpf0;
                const int tl = rem / nwords, wx = rem - tl * nwords;
                const uint8_t* sf = sflag + (int64_t)frame * tpf0;
#pragma unroll
                for (int dl = -1; dl <= 1; ++dl)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int nl = tl + dl, nx = wx + dx;
                        if (nl >= 0 && nl < ntiles && nx >= 0 && nx < nwords) f |= sf[nl * nwords + nx];
                    }
                if (!f) {
                    tflag[tile_id] = 0;
                    tile_count[tile_id] = 0;
                }
            }
            unsigned long long todo = __ballot(f != 0);
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                hyst_tile(strong, cand, nlines, nwords, ntiles, (j0 + l) * nwaves + wave, changed, pass, tile_count, tflag, tflag_half, cvalid);
            }
        }
        return;
    }
    // Later passes touch few tiles.  A wave looks at the flags of all its tiles at once, one tile per lane (the
    // same tiles it would walk over in pass 0, so that the tiles of a busy region are spread over many waves): a
    // tile is revisited only if it or one of its eight neighbours changed in the previous pass; the others just
    // clear their flag.
    const int lane = threadIdx.x & 63;
    const int tpf = ntiles * nwords;
    for (int j0 = 0; j0 * nwaves + wave < total_tiles; j0 += 64) {
        const int tile_id = (j0 + lane) * nwaves + wave;
        int f = 0;
        if (tile_id < total_tiles) {
            const int frame = tile_id / tpf;
            const int rem = tile_id - frame * tpf;
            const int tl = rem / nwords, wx = rem - tl * nwords;
            const int64_t tbase = (int64_t)frame * tpf;
            const uint8_t* fl_prev = tflag + (int64_t)((pass - 1) & 1) * tflag_half + tbase;
#pragma unroll
            for (int dl = -1; dl <= 1; ++dl)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int nl = tl + dl, nx = wx + dx;
                    if (nl >= 0 && nl < ntiles && nx >= 0 && nx < nwords) f |= fl_prev[nl * nwords + nx];
                }
            if (!f) tflag[(int64_t)(pass & 1) * tflag_half + tbase + rem] = 0;
        }
        unsigned long long todo = __ballot(f != 0);
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            hyst_tile(strong, cand, nlines, nwords, ntiles, (j0 + l) * nwaves + wave, changed, pass, tile_count, tflag, tflag_half, cvalid);
        }
    }
}

__device__ __forceinline__ bool hyst_tile(unsigned long long* __restrict__ strong, const unsigned long long* __restrict__ cand,
                                          int nlines, int nwords, int ntiles, int tile_id, unsigned int* changed, int pass,
                                          uint32_t* __restrict__ tile_count, uint8_t* tflag, int64_t tflag_half,
                                          const uint8_t* __restrict__ cvalid) {
    const int tpf = ntiles * nwords;
    const int frame = tile_id / tpf;
    const int rem = tile_id - frame * tpf;
    const int tl = rem / nwords, wx = rem - tl * nwords;
    const int lane = threadIdx.x & 63;
    const int64_t tbase = (int64_t)frame * ntiles * nwords;
    uint8_t* fl_cur = tflag + (int64_t)(pass & 1) * tflag_half + tbase;
    const int ln = tl * 64 + lane;
    // maps are stored [frame][word][line]: the 64 lines of a tile are contiguous
    unsigned long long* sb = strong + ((int64_t)frame * nwords + wx) * nlines;
    const unsigned long long* cb = cand + ((int64_t)frame * nwords + wx) * nlines;
    const bool lin = ln < nlines;
    const bool hasl = wx > 0, hasr = wx + 1 < nwords;  // previous / next word along the lines
    unsigned long long S = 0, C = 0, Lw = 0, Rw = 0;
    if (lin && (!cvalid || cvalid[tile_id])) C = cb[ln];
    if (!__any(C != 0)) {
        // no candidate in the tile (strong pixels are candidates too): nothing here can ever change
        if (lane == 0) {
            fl_cur[tl * nwords + wx] = 0;
            tile_count[tbase + tl * nwords + wx] = 0;
        }
        return false;
    }
    if (lin) {
        S = sb[ln];
        if (hasl) Lw = sb[ln - nlines];
        if (hasr) Rw = sb[ln + nlines];
    }
    // halo lines before and after the tile (own word and its two neighbours)
    unsigned long long To = 0, Tl = 0, Tr = 0, Bo = 0, Bl = 0, Br = 0;
    const int lt = tl * 64 - 1, lb = tl * 64 + 64;
    if (lt >= 0) {
        To = sb[lt];
        if (hasl) Tl = sb[lt - nlines];
        if (hasr) Tr = sb[lt + nlines];
    }
    if (lb < nlines) {
        Bo = sb[lb];
        if (hasl) Bl = sb[lb - nlines];
        if (hasr) Br = sb[lb + nlines];
    }
    if (!__any((S | Lw | Rw) != 0) && (To | Tl | Tr | Bo | Bl | Br) == 0) {
        // candidates but not one strong pixel in the tile or around it: nothing to grow from (yet)
        if (lane == 0) {
            fl_cur[tl * nwords + wx] = 0;
            tile_count[tbase + tl * nwords + wx] = 0;
        }
        return false;
    }
    const unsigned long long upL = from_lane_below(Lw, Tl), dnL = from_lane_above(Lw, Bl);
    const unsigned long long upR = from_lane_below(Rw, Tr), dnR = from_lane_above(Rw, Br);
    // pixels of the neighbouring words that touch this word's first / last pixel
    const unsigned long long carry = ((upL | Lw | dnL) >> 63) | (((upR | Rw | dnR) & 1ull) << 63);

    const unsigned long long orig = S;
    bool any;
    do {
        const unsigned long long up = from_lane_below(S, To), dn = from_lane_above(S, Bo);
        const unsigned long long A = S | up | dn;
        const unsigned long long D = A | (A << 1) | (A >> 1) | carry;
        const unsigned long long seeds = (S | (C & D)) & C;
        const unsigned long long nS = S | flood_runs(C, seeds);
        any = __any(nS != S);
        S = nS;
    } while (any);

    const bool ch = S != orig;
    if (ch) sb[ln] = S;
    const bool tile_changed = __any(ch);
    // the tile's count: shifts by 1, 2, 4, 8 inside each row of 16 lanes, lane 15 of a row into the next row, lane 31
    // into the upper half; lane 63 ends up with the total and does the bookkeeping
    uint32_t cnt = __popcll(S);
    cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x111, 0xf, 0xf, false);
    cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x112, 0xf, 0xf, false);
    cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x114, 0xf, 0xf, false);
    cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x118, 0xf, 0xf, false);
    cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x142, 0xa, 0xf, false);
    cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x143, 0xc, 0xf, false);
    if (lane == 63) {
        // a flag, not a count: every writer stores the same value, so a plain store is enough
        if (tile_changed && changed) changed[pass] = 1u;
        fl_cur[tl * nwords + wx] = tile_changed ? 1 : 0;
        tile_count[tbase + tl * nwords + wx] = cnt;
    }
    return tile_changed;
}

// The tail of the hysteresis: after a few grid-wide passes only a handful of tiles per frame are still
// growing, and a pass costs its launch.  Here one workgroup per frame repeats the pass over its own frame
// (frames do not interact) until nothing changes, with a barrier instead of a launch between rounds.  Counts
// as pass number `pass`; rounds alternate the flag parity like passes do.  A frame that is still changing
// after HYST_FINISH_ROUNDS rounds (an odd number, so that its flags end in this pass's parity) reports it in
// changed[pass] and the host queues the kernel again; a frame that is done clears both parities.
constexpr int HYST_FINISH_ROUNDS = 63;
__global__ __launch_bounds__(1024) void hyst_finish_kernel(unsigned long long* __restrict__ strong,
                                                           const unsigned long long* __restrict__ cand, int nlines,
                                                           int nwords, int ntiles, unsigned int* changed, int pass,
                                                           uint32_t* __restrict__ tile_count, uint8_t* tflag,
                                                           int64_t tflag_half, const uint8_t* __restrict__ cvalid) {
    if (changed[pass - 1] == 0) return;
    __shared__ int round_changed[2];
    const int frame = blockIdx.x;
    const int tpf = ntiles * nwords;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int64_t tbase = (int64_t)frame * tpf;
    if (threadIdx.x < 2) round_changed[threadIdx.x] = 0;
    __syncthreads();
    bool done = false;
    for (int round = 0; round < HYST_FINISH_ROUNDS; ++round) {
        const int p = pass + round;
        const uint8_t* fl_prev = tflag + (int64_t)((p - 1) & 1) * tflag_half + tbase;
        uint8_t* fl_cur = tflag + (int64_t)(p & 1) * tflag_half + tbase;
        bool any = false;
        // lane l of wave w looks at tile (j0 + l) * nwaves + w, so that the tiles of a busy region, which are
        // neighbours in this numbering, are spread over the waves instead of queueing up in one
        for (int j0 = 0; j0 * nwaves + wave < tpf; j0 += 64) {
            const int rem = (j0 + lane) * nwaves + wave;
            int f = 0;
            if (rem < tpf) {
                const int tl = rem / nwords, wx = rem - tl * nwords;
#pragma unroll
                for (int dl = -1; dl <= 1; ++dl)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int nl = tl + dl, nx = wx + dx;
                        if (nl >= 0 && nl < ntiles && nx >= 0 && nx < nwords) f |= fl_prev[nl * nwords + nx];
                    }
                if (!f) fl_cur[rem] = 0;
            }
            unsigned long long todo = __ballot(f != 0);
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                any |= hyst_tile(strong, cand, nlines, nwords, ntiles, (int)(tbase + (j0 + l) * nwaves + wave), nullptr, p, tile_count,
                                 tflag, tflag_half, cvalid);
            }
        }
        if (any && lane == 0) round_changed[round & 1] = 1;
        __syncthreads();  // also orders this round's bitmap and flag stores before the next round's loads
        const bool again = round_changed[round & 1] != 0;
        if (threadIdx.x == 0) round_changed[(round + 1) & 1] = 0;
        __syncthreads();
        if (!again) { done = true; break; }
    }
    if (done) {
        for (int i = threadIdx.x; i < tpf; i += blockDim.x) {
            tflag[tbase + i] = 0;
            tflag[tflag_half + tbase + i] = 0;
        }
    } else if (threadIdx.x == 0) {
        changed[pass] = 1u;
    }
}

// Non-maximum suppression after the fact.  When the stencil pass leaves the RAW double threshold in the maps
// (magnitude above low / above high, no suppression), a flood of those maps reaches a superset of the pixels
// cv2.Canny (passive_checker.py:116) can mark: an edge pixel is a suppressed candidate connected to a suppressed
// strong pixel through suppressed candidates, all of which are raw candidates connected to a raw strong pixel.  On
// camera-like frames that superset is a fraction of a percent of the image (a seventh of the pixels exceed the low
// threshold, a thousandth the high one, and 8-connected clusters at that density stay small), so the suppression --
// three Sobel magnitudes, the direction tests and three comparisons per pixel -- runs here on those pixels only,
// recomputed from the gray plane, instead of on every pixel in the stencil pass.  Afterwards `cand` holds the
// suppressed candidates inside the flooded set, `strong` the suppressed strong pixels, and the ordinary hysteresis
// passes run on them from pass 0.
//
// Two kernels.  nms_collect_kernel looks at the per-tile counts the flood left (64 tiles at a time, one per lane) and,
// for every tile that holds flooded pixels, a wave (lane = pixel column, word = 64 rows) appends them to the frame's
// list -- one atomic per tile reserves the room -- and clears the tile's words of `strong` and the tiles' flags.  nms_apply_kernel takes one list entry per
// thread, so the work is spread evenly however the pixels cluster; a pixel that passes sets its bit in `cand2` (and in
// `strong`, and sets its tile's flag) with an atomic or.  Nobody clears `cand2` as a whole: the collecting kernel
// zeroes the words of the tiles that hold flooded pixels and records in `cvalid`, a byte per tile, which tiles those
// are; the flood that follows takes every other tile's words as empty without reading them.  A frame whose list would overflow (the
// flooded set is larger than 4 % of the frame) raises *unsettled and the caller runs the chunk the ordinary way.
constexpr int NMS_WAVES = 4;
// L1 Sobel magnitude of the replicate-bordered gray image at (y, x); 0 outside the image
__device__ __forceinline__ int nms_mag(const uint8_t* gp, int gpitch, int H, int W, int y, int x, int* gxo, int* gyo) {
    if (y < 0 || y >= H || x < 0 || x >= W) return 0;
    const int ym = y > 0 ? y - 1 : 0, yp = y + 1 < H ? y + 1 : H - 1;
    const int xm = x > 0 ? x - 1 : 0, xp = x + 1 < W ? x + 1 : W - 1;
    const uint8_t* r0 = gp + (int64_t)(ym + GRAY_YOFF) * gpitch + GRAY_XOFF;
    const uint8_t* r1 = gp + (int64_t)(y + GRAY_YOFF) * gpitch + GRAY_XOFF;
    const uint8_t* r2 = gp + (int64_t)(yp + GRAY_YOFF) * gpitch + GRAY_XOFF;
    const int a = r0[xm], b = r0[x], c = r0[xp], d = r1[xm], f = r1[xp], g = r2[xm], h = r2[x], i = r2[xp];
    const int gx = (c - a) + 2 * (f - d) + (i - g);
    const int gy = (g - a) + 2 * (h - b) + (i - c);
    if (gxo) { *gxo = gx; *gyo = gy; }
    return abs(gx) + abs(gy);
}
// the same from three rows of a patch held in registers: bytes 0..2 of r0, r1, r2 are the three columns
__device__ __forceinline__ int patch_mag(uint32_t r0, uint32_t r1, uint32_t r2, int* gxo, int* gyo) {
    const int a = r0 & 255u, b = (r0 >> 8) & 255u, c = (r0 >> 16) & 255u, d = r1 & 255u, f = (r1 >> 16) & 255u;
    const int g = r2 & 255u, h = (r2 >> 8) & 255u, i = (r2 >> 16) & 255u;
    const int gx = (c - a) + 2 * (f - d) + (i - g);
    const int gy = (g - a) + 2 * (h - b) + (i - c);
    if (gxo) { *gxo = gx; *gyo = gy; }
    return abs(gx) + abs(gy);
}
__global__ __launch_bounds__(64 * NMS_WAVES) void nms_collect_kernel(unsigned long long* __restrict__ strong, int nlines,
                                                                     int nwords, int ntiles, int total_tiles,
                                                                     uint32_t* __restrict__ tile_count,
                                                                     uint8_t* __restrict__ sflag,
                                                                     unsigned long long* __restrict__ cand2,
                                                                     uint8_t* __restrict__ cvalid,
                                                                     uint32_t* __restrict__ list, int list_cap,
                                                                     unsigned int* __restrict__ count,
                                                                     unsigned int* __restrict__ unsettled) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * NMS_WAVES + (threadIdx.x >> 6), nwaves = gridDim.x * NMS_WAVES;
    const int tpf = ntiles * nwords;
    // lane l of wave w looks at tile (j0 + l) * nwaves + w: the tiles of a busy region are spread over many waves
    for (int j0 = 0; j0 * nwaves + wave < total_tiles; j0 += 64) {
        const int mine = (j0 + lane) * nwaves + wave;
        uint32_t had = 0;
        if (mine < total_tiles) {
            had = tile_count[mine];
            // the flood that follows counts from zero and starts from the flags the suppression sets
            tile_count[mine] = 0;
            sflag[mine] = 0;
            cvalid[mine] = had != 0 ? 1 : 0;
        }
        unsigned long long todo = __ballot(had != 0);
        while (todo) {
            const int tile_id = (j0 + __ffsll((long long)todo) - 1) * nwaves + wave;
            todo &= todo - 1;
            const int frame = tile_id / tpf;
            const int rem = tile_id - frame * tpf;
            const int tl = rem / nwords, wx = rem - tl * nwords;
            const int ln = tl * 64 + lane;
            const bool lin = ln < nlines;
            const int64_t idx = ((int64_t)frame * nwords + wx) * nlines + ln;
            const unsigned long long R = lin ? strong[idx] : 0ull;
            if (lin) cand2[idx] = 0ull;
            const uint32_t cnt = (uint32_t)__popcll(R);
            if (!__any(cnt != 0)) continue;
            if (R != 0) strong[idx] = 0ull;
            uint32_t inc = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o, 64);
                if (lane >= o) inc += t;
            }
            const uint32_t total = __shfl(inc, 63, 64);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&count[frame], total);
            base = __shfl(base, 0, 64);
            if (base + total > (uint32_t)list_cap) {  // wave-uniform
                if (lane == 0) *unsettled = 1u;
                continue;
            }
            uint32_t* dst = list + (int64_t)frame * list_cap + base + (inc - cnt);
            for (unsigned long long w = R; w != 0; w &= w - 1)
                *dst++ = ((uint32_t)(wx * 64 + (__ffsll((long long)w) - 1)) << 13) | (uint32_t)ln;
        }
    }
}
__global__ __launch_bounds__(256) void nms_apply_kernel(unsigned long long* __restrict__ strong,
                                                        unsigned long long* __restrict__ cand2,
                                                        const uint8_t* __restrict__ gray, int gpitch, int grows, int H,
                                                        int nlines, int nwords, int ntiles, int low, int high,
                                                        uint8_t* __restrict__ sflag, const uint32_t* __restrict__ list,
                                                        int list_cap, const unsigned int* __restrict__ count,
                                                        unsigned int* __restrict__ reached_total) {
    // grid: (blocks per frame, frames); a workgroup reads its frame's count once and walks over the list with the
    // grid's stride, so a frame with a short list costs the others nothing
    const int tpf = ntiles * nwords;
    const int frame = blockIdx.y;
    const unsigned int n = count[frame];
    // the size of the flooded set, for the caller's choice of path: one add per frame (adds to one address queue up:
    // one per tile cost the collecting kernel half of its time)
    if (blockIdx.x == 0 && threadIdx.x == 0 && n) atomicAdd(reached_total, n);
    if (n > (unsigned int)list_cap) return;  // (an overflowed frame is redone by the caller)
    const uint8_t* gp = gray + (int64_t)frame * grows * gpitch;
    const int W = nlines;
    const int lane = threadIdx.x & 63;
    for (unsigned int i0 = blockIdx.x * 256u + (threadIdx.x & ~63u); i0 < n; i0 += gridDim.x * 256u) {  // wave-uniform
        const unsigned int i = i0 + lane;
        const bool live = i < n;
        const uint32_t item = live ? list[(int64_t)frame * list_cap + i] : list[(int64_t)frame * list_cap + n - 1];
        const int x = (int)(item & 8191u), y = (int)(item >> 13);
        int m, na, nb, slack;
        if (y >= 2 && y + 2 < H && x >= 2 && x + 2 < W) {
            // the 5x5 patch around the pixel, five rows of two aligned words each: every load is issued before the
            // first is used, and a row costs one or two cache lines instead of a line per byte
            const uintptr_t a0 = reinterpret_cast<uintptr_t>(gp + (int64_t)(y - 2 + GRAY_YOFF) * gpitch + (x - 2 + GRAY_XOFF));
            const uint32_t* q = reinterpret_cast<const uint32_t*>(a0 & ~(uintptr_t)3);
            const uint32_t sh = (uint32_t)(a0 & 3u) * 8u;
            const int wpitch = gpitch >> 2;
            unsigned long long row[5];
#pragma unroll
            for (int r = 0; r < 5; ++r)
                row[r] = ((unsigned long long)q[r * wpitch] | ((unsigned long long)q[r * wpitch + 1] << 32)) >> sh;
            int gx, gy;
            m = patch_mag((uint32_t)(row[1] >> 8), (uint32_t)(row[2] >> 8), (uint32_t)(row[3] >> 8), &gx, &gy);
            const int ax = abs(gx), ay = abs(gy);
            const int q22 = (ax * 13573) >> 15;    // floor(|gx| tan 22.5 deg)
            const bool nothoriz = ay > q22;         // |gy| > tan 22.5 |gx|
            const bool vert = ay > 2 * ax + q22;    // |gy| > tan 67.5 |gx|
            const bool opp = (gx ^ gy) < 0;
            // first neighbour at (dy, dx), second at (-dy, -dx)
            int dy, dx;
            if (!nothoriz) { dy = 0; dx = -1; slack = 1; }
            else if (vert) { dy = -1; dx = 0; slack = 1; }
            else if (opp) { dy = -1; dx = 1; slack = 0; }
            else { dy = -1; dx = -1; slack = 0; }
            // rows of the first neighbour's 3x3 window are row[1+dy .. 3+dy], of the second row[1-dy .. 3-dy]
            const unsigned long long a_0 = dy ? row[0] : row[1], a_1 = dy ? row[1] : row[2], a_2 = dy ? row[2] : row[3];
            const unsigned long long b_0 = dy ? row[2] : row[1], b_1 = dy ? row[3] : row[2], b_2 = dy ? row[4] : row[3];
            const uint32_t sa = (uint32_t)(1 + dx) * 8u, sb = (uint32_t)(1 - dx) * 8u;
            na = patch_mag((uint32_t)(a_0 >> sa), (uint32_t)(a_1 >> sa), (uint32_t)(a_2 >> sa), nullptr, nullptr);
            nb = patch_mag((uint32_t)(b_0 >> sb), (uint32_t)(b_1 >> sb), (uint32_t)(b_2 >> sb), nullptr, nullptr);
        } else {
            int gx, gy;
            m = nms_mag(gp, gpitch, H, W, y, x, &gx, &gy);
            const int ax = abs(gx), ay = abs(gy);
            const int q22 = (ax * 13573) >> 15;
            const bool nothoriz = ay > q22;
            const bool vert = ay > 2 * ax + q22;
            const bool opp = (gx ^ gy) < 0;
            int ya, xa, yb, xb;
            if (!nothoriz) { ya = y; xa = x - 1; yb = y; xb = x + 1; slack = 1; }
            else if (vert) { ya = y - 1; xa = x; yb = y + 1; xb = x; slack = 1; }
            else if (opp) { ya = y - 1; xa = x + 1; yb = y + 1; xb = x - 1; slack = 0; }
            else { ya = y - 1; xa = x - 1; yb = y + 1; xb = x + 1; slack = 0; }
            na = nms_mag(gp, gpitch, H, W, ya, xa, nullptr, nullptr);
            nb = nms_mag(gp, gpitch, H, W, yb, xb, nullptr, nullptr);
        }
        // the second neighbour is compared with >= for horizontal / vertical gradients
        // Entries of one map word (a column's 64 rows) are neighbours in the list: their bits are combined across the
        // lanes first (a segmented or, six steps), and the last lane of a run issues the one atomic for it.
        const bool keep = live && m > low && m > na && m > nb - slack;  // the second neighbour is compared with >= for horizontal / vertical gradients
        const bool keep_s = keep && m > high;
        const int64_t idx = ((int64_t)frame * nwords + (y >> 6)) * nlines + x;
        const unsigned long long bit = 1ull << (y & 63);
        unsigned long long vc = keep ? bit : 0ull, vs = keep_s ? bit : 0ull;
        const uint32_t key = live ? item >> 19 << 13 | (item & 8191u) : 0xFFFFFFFFu;  // (word row, column)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t k2 = __shfl_up(key, o, 64);
            const unsigned long long c2 = __shfl_up(vc, o, 64), s2 = __shfl_up(vs, o, 64);
            if (lane >= o && k2 == key) { vc |= c2; vs |= s2; }
        }
        const uint32_t knext = __shfl_down(key, 1, 64);
        if (live && (lane == 63 || knext != key)) {
            if (vc) atomicOr(&cand2[idx], vc);
            if (vs) {
                atomicOr(&strong[idx], vs);
                sflag[(int64_t)frame * tpf + (x >> 6) * nwords + (y >> 6)] = 1;
            }
        }
    }
}

// Per-frame reduction of everything the other kernels left as partials, in a fixed order
// (bit-reproducible): stencil tile sums -> LBP histogram / high-pass / residual / gray sums,
// hysteresis tile counts -> edge count, FFT column partials -> log-magnitude sums.
__global__ __launch_bounds__(256) void frame_reduce_kernel(const uint32_t* __restrict__ spart, int stiles,
                                                           int lbp_from_tiles, const uint32_t* __restrict__ tile_count,
                                                           int htiles, const FftPart* __restrict__ fpart, int nfparts,
                                                           FrameAcc* __restrict__ acc) {
    __shared__ unsigned long long red[16][17];
    __shared__ double dred[5][4];
    __shared__ uint32_t ered[4];
    const int frame = blockIdx.x, tid = threadIdx.x;
    // 16 quantities x 16 tile lanes
    {
        const int q = tid & 15, lane16 = tid >> 4;
        unsigned long long s = 0;
        for (int t = lane16; t < stiles; t += 16) s += spart[((int64_t)frame * stiles + t) * 16 + q];
        red[q][lane16] = s;
    }
    uint32_t e = 0;
    for (int i = tid; i < htiles; i += 256) e += tile_count[(int64_t)frame * htiles + i];
    double fs = 0.0, fs2 = 0.0, ft1 = 0.0, ft2 = 0.0, fe2 = 0.0;
    for (int i = tid; i < nfparts; i += 256) {
        const FftPart p = fpart[(int64_t)frame * nfparts + i];
        fs += p.s;
        fs2 += p.s2;
        ft1 += p.t1;
        ft2 += p.t2;
        fe2 += p.e2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        e += __shfl_xor(e, o, 64);
        fs += __shfl_xor(fs, o, 64);
        fs2 += __shfl_xor(fs2, o, 64);
        ft1 += __shfl_xor(ft1, o, 64);
        ft2 += __shfl_xor(ft2, o, 64);
        fe2 += __shfl_xor(fe2, o, 64);
    }
    if ((tid & 63) == 0) {
        ered[tid >> 6] = e;
        dred[0][tid >> 6] = fs; dred[1][tid >> 6] = fs2; dred[2][tid >> 6] = ft1; dred[3][tid >> 6] = ft2; dred[4][tid >> 6] = fe2;
    }
    __syncthreads();
    FrameAcc* fa = acc + frame;
    if (tid < 14) {
        unsigned long long s = 0;
        for (int i = 0; i < 16; ++i) s += red[tid][i];
        if (tid < 10) {
            if (lbp_from_tiles) fa->lbp_hist[tid] = (uint32_t)s;
        } else if (tid == 10) fa->hp_sum = s;
        else if (tid == 11) fa->hp_sumsq = s;
        else if (tid == 12) fa->absdiff_sum = s;
        else fa->gray_sum = s;
    }
    if (tid == 64) {
        fa->edge_count = ered[0] + ered[1] + ered[2] + ered[3];
        fa->fft_sum = ((dred[0][0] + dred[0][1]) + dred[0][2]) + dred[0][3];
        fa->fft_sumsq = ((dred[1][0] + dred[1][1]) + dred[1][2]) + dred[1][3];
        fa->fft_t1 = ((dred[2][0] + dred[2][1]) + dred[2][2]) + dred[2][3];
        fa->fft_t2 = ((dred[3][0] + dred[3][1]) + dred[3][2]) + dred[3][3];
        fa->fft_e2 = ((dred[4][0] + dred[4][1]) + dred[4][2]) + dred[4][3];
    }
}

__global__ __launch_bounds__(256) void edges_expand_kernel(const unsigned long long* __restrict__ strong, int H, int W,
                                                           uint8_t* __restrict__ edges) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i % W);
    edges[i] = ((strong[(int64_t)(y >> 6) * W + x] >> (y & 63)) & 1ull) ? 255 : 0;
}

void launch_hyst_pass(hipStream_t s, unsigned long long* strong, const unsigned long long* cand, int n_frames,
                      const Geom& g, unsigned int* changed, int pass, uint32_t* tile_count, uint8_t* tflag, bool cand_sparse) {
    const int64_t half = (int64_t)n_frames * g.bm_tiles * g.bm_words;
    const int total = (int)half;
    int grid = (total + HYST_WAVES - 1) / HYST_WAVES;
    const int cap = pass <= 1 ? HYST_MAX_GRID : HYST_LATER_GRID;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(hyst_pass_kernel, dim3(grid), dim3(64 * HYST_WAVES), 0, s, strong,
                       cand, g.W, g.bm_words, g.bm_tiles, total, changed, pass, tile_count, tflag, half,
                       cand_sparse ? (const uint8_t*)(tflag + 2 * half) : nullptr);
}

void launch_hyst_finish(hipStream_t s, unsigned long long* strong, const unsigned long long* cand, int n_frames,
                        const Geom& g, unsigned int* changed, int pass, uint32_t* tile_count, uint8_t* tflag, bool cand_sparse) {
    const int64_t half = (int64_t)n_frames * g.bm_tiles * g.bm_words;
    hipLaunchKernelGGL(hyst_finish_kernel, dim3(n_frames), dim3(1024), 0, s, strong, cand, g.W, g.bm_words, g.bm_tiles,
                       changed, pass, tile_count, tflag, half, cand_sparse ? (const uint8_t*)(tflag + 2 * half) : nullptr);
}

int nms_list_cap(const Geom& g) {
    const int64_t px = (int64_t)g.H * g.W;
    const int64_t cap = px / 25;  // LIST_FRACTION: 4 % of the frame
    return (int)(cap < 1024 ? 1024 : cap);
}
void launch_nms_sparse(hipStream_t s, unsigned long long* strong, unsigned long long* cand2, const uint8_t* gray, int n_frames,
                       const Geom& g, int low, int high, uint32_t* tile_count, uint8_t* tflag, uint32_t* list,
                       unsigned int* count, unsigned int* unsettled, unsigned int* reached_total) {
    const int64_t half = (int64_t)n_frames * g.bm_tiles * g.bm_words;
    const int total = (int)half;
    const int cap = nms_list_cap(g);
    // eight tiles per wave: a tile with flooded pixels costs its wave two trips to memory and an atomic, one after the
    // other, and the waves that draw several such tiles are what the kernel waits for
    int grid = (total + 8 * NMS_WAVES - 1) / (8 * NMS_WAVES);
    if (grid > HYST_MAX_GRID) grid = HYST_MAX_GRID;
    hipLaunchKernelGGL(nms_collect_kernel, dim3(grid), dim3(64 * NMS_WAVES), 0, s, strong, g.W, g.bm_words, g.bm_tiles, total,
                       tile_count, tflag + half, cand2, tflag + 2 * half, list, cap, count, unsettled);
    // magnitudes are at most 2040: thresholds outside [-1, 4095] behave like the nearest bound (as in the stencil pass)
    const int lo = low < -1 ? -1 : (low > 4095 ? 4095 : low), hi = high < -1 ? -1 : (high > 4095 ? 4095 : high);
    const int bpf = (cap + 255) / 256;
    hipLaunchKernelGGL(nms_apply_kernel, dim3((unsigned)(bpf < 32 ? bpf : 32), (unsigned)n_frames), dim3(256), 0, s, strong, cand2,
                       gray, g.gpitch, g.grows, g.H, g.W, g.bm_words, g.bm_tiles, lo, hi, tflag + half, list, cap, count, reached_total);
}

// Everything a chunk needs cleared before its first kernel, in one launch: the per-frame accumulators, the pass
// counters of the hysteresis (and, for a chunk whose stencil pass leaves raw threshold maps, those of the flood of the
// raw maps, the per-frame list lengths, and the preset "the pass before changed something" of the flood after the
// suppression), and the flag half in which the stencil pass marks the tiles that hold a strong pixel.
__global__ __launch_bounds__(256) void chunk_init_kernel(uint32_t* __restrict__ acc_words, int n_acc_words,
                                                         unsigned int* __restrict__ changed, int n_changed,
                                                         unsigned int* __restrict__ raw_changed, int n_raw_changed,
                                                         unsigned int* __restrict__ preset_one,
                                                         unsigned int* __restrict__ nmscount, int n_frames,
                                                         uint32_t* __restrict__ sflag_words, int n_sflag_words) {
    const int stride = gridDim.x * 256;
    const int t = blockIdx.x * 256 + threadIdx.x;
    for (int i = t; i < n_acc_words; i += stride) acc_words[i] = 0u;
    for (int i = t; i < n_sflag_words; i += stride) sflag_words[i] = 0u;
    if (t < n_changed) changed[t] = 0u;
    if (raw_changed) {
        if (t < n_raw_changed) raw_changed[t] = 0u;
        for (int i = t; i < n_frames; i += stride) nmscount[i] = 0u;
        if (t == 0) *preset_one = 1u;
    }
}

void launch_chunk_init(hipStream_t s, FrameAcc* acc, int n_frames, unsigned int* changed, int n_changed,
                       unsigned int* raw_changed, int n_raw_changed, unsigned int* preset_one, unsigned int* nmscount,
                       uint8_t* sflag, size_t sflag_bytes) {
    static_assert(sizeof(FrameAcc) % 4 == 0, "cleared as 32-bit words");
    const int n_acc_words = (int)((size_t)n_frames * sizeof(FrameAcc) / 4);
    const int n_sflag_words = (int)((sflag_bytes + 3) / 4);  // (the buffer's size is rounded up to whole words)
    int grid = (std::max(n_acc_words, n_sflag_words) + 255) / 256;
    grid = grid < 1 ? 1 : (grid > 256 ? 256 : grid);
    hipLaunchKernelGGL(chunk_init_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<uint32_t*>(acc), n_acc_words, changed,
                       n_changed, raw_changed, n_raw_changed, preset_one, nmscount, n_frames,
                       reinterpret_cast<uint32_t*>(sflag), n_sflag_words);
}

void launch_frame_reduce(hipStream_t s, const uint32_t* spart, bool lbp_from_tiles, const uint32_t* tile_count,
                         const FftPart* fpart, int nfparts, int n_frames, const Geom& g, FrameAcc* acc) {
    hipLaunchKernelGGL(frame_reduce_kernel, dim3(n_frames), dim3(256), 0, s, spart, g.tiles_x * g.tiles_y,
                       lbp_from_tiles ? 1 : 0, tile_count, g.bm_tiles * g.bm_words, fpart, nfparts, acc);
}

void launch_edges_expand(hipStream_t s, const unsigned long long* strong, const Geom& g, uint8_t* edges) {
    const int64_t n = (int64_t)g.H * g.W;
    hipLaunchKernelGGL(edges_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, strong, g.H, g.W, edges);
}

}  // namespace stx
