// Traversal-order probe for the N x N uint8 matrix (pure stores), MI355X.
// sweep<RW,WPB>: a wave owns RW consecutive rows and sweeps all 1-KiB column segments left to right;
//                a block = WPB waves = WPB*RW consecutive rows; grid = one-shot over row groups or capped (grid-stride).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int RW, int WPB> __global__ __launch_bounds__(64 * WPB) void sweep(unsigned char *out, long n, long ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nseg = (n + 1023) / 1024;
    for (long rg = blockIdx.x; rg * (RW * WPB) < n; rg += gridDim.x) {
        const long r0 = (rg * WPB + wave) * RW;
        if (r0 >= n) continue;
        u32x4 v = {1, 2, 3, (unsigned)lane};
        for (long s = 0; s < nseg; ++s) {
            const long col = s * 1024 + lane * 16;
            if (col + 16 > n) break;
#pragma unroll
            for (int r = 0; r < RW; ++r)
                if (r0 + r < n) { v.x += r; *(u32x4 *)(out + (r0 + r) * ld + col) = v; }
        }
    }
}
// colblock<R>: a block of 4 waves owns R rows x 4 KiB (waves side by side), one-shot, x = column fastest
template <int R> __global__ __launch_bounds__(256) void colblock(unsigned char *out, long n, long ld) {
    const long col = (long)blockIdx.x * 4096 + threadIdx.x * 16;
    if (col + 16 > n) return;
    u32x4 v = {1, 2, 3, threadIdx.x};
    for (int r = 0; r < R; ++r) { long row = (long)blockIdx.y * R + r; if (row < n) { v.x += r; *(u32x4 *)(out + row * ld + col) = v; } }
}
int main() {
    const long n = 50000, ld = 50176;
    unsigned char *out; hipMalloc(&out, (size_t)n * ld + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, int blocks, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-26s blocks=%6d %7.3f ms  %7.1f GB/s\n", name, blocks, ms, (double)n * n / ms / 1e6);
    };
#define SW(RW, WPB, CAP) { int full = (int)((n + RW * WPB - 1) / (RW * WPB)); int b = (CAP) ? (CAP) : full; \
        run("sweep RW=" #RW " WPB=" #WPB, b, [&] { sweep<RW, WPB><<<b, 64 * WPB>>>(out, n, ld); }); }
    SW(1, 4, 0) SW(2, 4, 0) SW(4, 4, 0) SW(8, 4, 0) SW(4, 1, 0) SW(4, 2, 0) SW(4, 8, 0) SW(16, 4, 0)
    SW(4, 4, 256) SW(4, 4, 512) SW(4, 4, 1024) SW(1, 4, 256) SW(1, 4, 512) SW(8, 4, 256) SW(2, 8, 256) SW(1, 16, 256)
#define CB(R) run("colblock R=" #R, 0, [&] { dim3 g(13, (unsigned)((n + R - 1) / R)); colblock<R><<<g, 256>>>(out, n, ld); });
    CB(1) CB(2) CB(4) CB(8) CB(16) CB(64)
    return 0;
}
