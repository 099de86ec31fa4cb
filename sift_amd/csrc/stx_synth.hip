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

void launch_synth(hipStream_t s, uint8_t* bgr, int kind, uint64_t seed, uint32_t video, uint32_t frame0, int n_frames,
                  int h, int w) {
    dim3 grid((w + 255) / 256, h, n_frames);
    hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, s, bgr, kind, (unsigned long long)seed, video, frame0, h, w);
}

}  // namespace stx
