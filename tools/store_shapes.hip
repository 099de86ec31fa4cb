// What a 1.2 GB store stream reaches by the shape of its instructions: bytes per lane (8 or 16), where the 128-byte
// lines of one instruction lie (next to each other, or far apart with consecutive instructions filling in between),
// and the order in which workgroups walk the buffer.   usage: store_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// B bytes per lane; an instruction writes LPI = 64 * B / 128 whole lines.  MODE 0: the lines of an instruction are
// adjacent and instructions follow each other; 1: line m of every instruction lies in the m-th LPI-th of the buffer;
// 2: as 0 with the workgroups' order scrambled; 3: as 0, but a workgroup's instructions are total/ITERS apart.
template <int B, int MODE, bool NT>
__global__ __launch_bounds__(64) void shape(unsigned char* out, long total_lines, int iters, unsigned nwg) {
    constexpr int LPI = 64 * B / 128, G = 128 / B;
    const int lane = threadIdx.x, m = lane / G, within = lane % G;
    unsigned w = blockIdx.x;
    if (MODE == 2) w = (unsigned)(((unsigned long long)w * 2654435761ull) % nwg);
    const long instrs = total_lines / LPI;
    for (int j = 0; j < iters; ++j) {
        long q = (MODE == 3) ? (long)j * (instrs / iters) + w : (long)w * iters + j;  // instruction number
        long line = (MODE == 1) ? (long)m * instrs + q : q * LPI + m;
        unsigned char* p = out + line * 128 + within * B;
        if (B == 16) {
            const f4 v = {(float)w, (float)lane, 1.f, 2.f};
            if (NT) __builtin_nontemporal_store(v, (f4*)p); else *(f4*)p = v;
        } else {
            const f2 v = {(float)w, (float)lane};
            if (NT) __builtin_nontemporal_store(v, (f2*)p); else *(f2*)p = v;
        }
    }
}

int main() {
    const long total_lines = 9600000;  // 1.2288 GB
    unsigned char* buf; CK(hipMalloc(&buf, total_lines * 128));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, int lpi) {
        const int iters = 8;
        const unsigned nwg = (unsigned)(total_lines / lpi / iters);
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(64), 0, 0, buf, total_lines, iters, nwg);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-64s %.3f ms  %.2f TB/s\n", name, best, total_lines * 128.0 / (best * 1e-3) / 1e12);
        return 0;
    };
    run("16 B/lane, adjacent lines, instructions in order", shape<16, 0, false>, 8);
    run("16 B/lane, adjacent lines, instructions in order, nt", shape<16, 0, true>, 8);
    run("16 B/lane, lines an eighth of the buffer apart", shape<16, 1, false>, 8);
    run("16 B/lane, lines an eighth of the buffer apart, nt", shape<16, 1, true>, 8);
    run("16 B/lane, adjacent lines, workgroups scrambled", shape<16, 2, false>, 8);
    run("16 B/lane, adjacent lines, a workgroup's instructions far apart", shape<16, 3, false>, 8);
    run("8 B/lane, adjacent lines, instructions in order", shape<8, 0, false>, 4);
    run("8 B/lane, adjacent lines, instructions in order, nt", shape<8, 0, true>, 4);
    run("8 B/lane, lines a quarter of the buffer apart", shape<8, 1, false>, 4);
    run("8 B/lane, lines a quarter of the buffer apart, nt", shape<8, 1, true>, 4);
    run("8 B/lane, adjacent lines, workgroups scrambled", shape<8, 2, false>, 4);
    run("8 B/lane, adjacent lines, a workgroup's instructions far apart", shape<8, 3, false>, 4);
    {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            (void)hipMemsetAsync(buf, 1, total_lines * 128, 0);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-64s %.3f ms  %.2f TB/s\n", "hipMemsetAsync", best, total_lines * 128.0 / (best * 1e-3) / 1e12);
    }
    return 0;
}
