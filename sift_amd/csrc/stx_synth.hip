// Deterministic synthetic BGR frames generated in device memory: the benchmark's input.
// Integer-only arithmetic, identical to sift_amd/synth.py, so a frame copied back to the
// host equals the one the Python generator makes.
#include "stx_internal.h"

namespace stx {

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ long long isin16(long long p) {
    p &= 0xFFFF;
    const long long q = p & 0x7FFF;
    const long long v = (q * (32768 - q)) >> 14;
    return (p & 0x8000) ? -v : v;
}

__device__ __forceinline__ long long floor_shr(long long v, int s) {
    return v >= 0 ? (v >> s) : -(((-v) + ((1ll << s) - 1)) >> s);
}

__device__ __forceinline__ uint8_t clamp_u8(long long v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__global__ __launch_bounds__(256) void synth_kernel(uint8_t* __restrict__ bgr, int kind, unsigned long long seed,
                                                    uint32_t video, uint32_t frame0, int h, int w) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const uint32_t frame = frame0 + blockIdx.z;
    if (x >= w) return;
    const long long f = frame;
    const unsigned long long idx = ((unsigned long long)video << 36) | ((unsigned long long)frame << 26) |
                                   ((unsigned long long)y << 13) | (unsigned long long)x;
    const unsigned long long hs = splitmix64(seed * 0xD1342543DE82EF95ull + idx);
    uint8_t* o = bgr + (((int64_t)blockIdx.z * h + y) * w + x) * 3;
    switch (kind) {
        case 0:
            o[0] = (uint8_t)hs; o[1] = (uint8_t)(hs >> 8); o[2] = (uint8_t)(hs >> 16);
            break;
        case 1: {
            const long long sx = isin16(((long long)x * 65536) / 97 + f * 1043);
            const long long sy = isin16(((long long)y * 65536) / 71 + 16384);
            long long base = 128 + floor_shr(60 * sx * sy, 28);
            const long long cx = w / 2 + 3 * f, cy = h / 2 + 2 * f, rad = h / 5;
            if ((x - cx) * (x - cx) + (y - cy) * (y - cy) < rad * rad) base += 40;
            for (int ch = 0; ch < 3; ++ch) {
                long long s = 0;
                for (int k = 0; k < 4; ++k) s += (long long)((hs >> (20 * ch + 5 * k)) & 31);
                o[ch] = clamp_u8(base + (((s * 55 + 128) >> 8) - 13) + 10 * (ch - 1));
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
        default:
            o[0] = o[1] = o[2] = clamp_u8(60 + 90 * ((((x + (int)frame) >> 4) + (y >> 4)) & 1) + ((long long)(hs & 3) - 1));
    }
}

// The `natural` kind, which the benchmark's clips are made of, at four pixels (twelve bytes, one store) per lane
// and SYNTH_ROWS image rows per wave.  Same function of (seed, video, frame, y, x) as synth_kernel, arranged so that
// what depends on the column only (the horizontal phase, the squared distance to the disc's centre) is computed once
// per lane, what depends on the row only sits in scalar registers, the three channels' noise sums come out of the
// hash by field-parallel adds on the 64-bit word, and everything but the hash itself is 32-bit arithmetic.
constexpr int SYNTH_ROWS = 8;

__device__ __forceinline__ int isin16_i32(int p) {
    p &= 0xFFFF;
    const int q = p & 0x7FFF;
    const int v = (q * (32768 - q)) >> 14;
    return (p & 0x8000) ? -v : v;
}

__global__ __launch_bounds__(64) void synth_natural4_kernel(uint32_t* __restrict__ out, unsigned long long seedmul,
                                                            uint32_t video, uint32_t frame0, int h, int w) {
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (x0 >= w) return;
    const uint32_t frame = frame0 + blockIdx.z;
    const int f = (int)frame;
    const int y0 = blockIdx.y * SYNTH_ROWS;
    const int cx = w / 2 + 3 * f, cy = h / 2 + 2 * f, rad = h / 5, rad2 = rad * rad;
    int sx[4], dx2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + k;
        sx[k] = isin16_i32((int)(((uint32_t)x * 65536u) / 97u) + f * 1043);
        dx2[k] = (x - cx) * (x - cx);
    }
    // 64-bit counter: everything above the column in one uniform word, the golden-ratio increment of the hash included
    const unsigned long long zrow0 = seedmul + 0x9E3779B97F4A7C15ull + (((unsigned long long)video << 36) |
                                                                        ((unsigned long long)frame << 26));
    uint32_t* o = out + (((int64_t)blockIdx.z * h + y0) * w + x0) / 4 * 3;
    const int64_t row_dwords = (int64_t)w / 4 * 3;
    for (int r = 0; r < SYNTH_ROWS && y0 + r < h; ++r, o += row_dwords) {
        const int y = y0 + r;
        const int sy = isin16_i32((int)(((uint32_t)y * 65536u) / 71u) + 16384);
        const int dy2 = (y - cy) * (y - cy);
        const unsigned long long zrow = zrow0 + ((unsigned long long)y << 13) + (unsigned long long)x0;
        uint32_t px[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned long long z = zrow + (unsigned long long)k;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            // twelve 5-bit fields, four per channel: adjacent fields added in 10-bit slots, then adjacent slots
            constexpr unsigned long long M = 0x007C1F07C1F07C1Full;  // fields 0, 2, 4 ... 10
            const unsigned long long t = (z & M) + ((z >> 5) & M);
            const unsigned long long u = t + (t >> 10);
            const int s0 = (int)((uint32_t)u & 0x7Fu);
            const int s1 = (int)(((uint32_t)u >> 20) & 0x7Fu);
            const int s2 = (int)(((uint32_t)(u >> 32) >> 8) & 0x7Fu);
            // 128 + floor(60 sx sy / 2^28): |sx sy| <= 2^28, so the product by 960 = 60 * 2^4 keeps its high word
            int base = 128 - 13 + __mulhi(sx[k] * sy, 960);
            if (dx2[k] + dy2 < rad2) base += 40;
            const int v0 = base - 10 + ((s0 * 55 + 128) >> 8);
            const int v1 = base + ((s1 * 55 + 128) >> 8);
            const int v2 = base + 10 + ((s2 * 55 + 128) >> 8);
            px[k][0] = (uint32_t)min(max(v0, 0), 255);
            px[k][1] = (uint32_t)min(max(v1, 0), 255);
            px[k][2] = (uint32_t)min(max(v2, 0), 255);
        }
        const uint32_t d0 = px[0][0] | (px[0][1] << 8) | (px[0][2] << 16) | (px[1][0] << 24);
        const uint32_t d1 = px[1][1] | (px[1][2] << 8) | (px[2][0] << 16) | (px[2][1] << 24);
        const uint32_t d2 = px[2][2] | (px[3][0] << 8) | (px[3][1] << 16) | (px[3][2] << 24);
        // one 12-byte store, marked non-temporal: nothing reads a frame before the whole clip is written
        __builtin_nontemporal_store(d0, o); __builtin_nontemporal_store(d1, o + 1); __builtin_nontemporal_store(d2, o + 2);
    }
}

void launch_synth(hipStream_t s, uint8_t* bgr, int kind, uint64_t seed, uint32_t video, uint32_t frame0, int n_frames,
                  int h, int w) {
    if (kind == 1 && w % 4 == 0 && ((uintptr_t)bgr & 3) == 0) {
        dim3 grid((w / 4 + 63) / 64, (h + SYNTH_ROWS - 1) / SYNTH_ROWS, n_frames);
        hipLaunchKernelGGL(synth_natural4_kernel, grid, dim3(64), 0, s, (uint32_t*)bgr,
                           (unsigned long long)seed * 0xD1342543DE82EF95ull, video, frame0, h, w);
        return;
    }
    dim3 grid((w + 255) / 256, h, n_frames);
    hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, s, bgr, kind, (unsigned long long)seed, video, frame0, h, w);
}

}  // namespace stx
