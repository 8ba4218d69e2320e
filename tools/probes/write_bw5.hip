// Row pitch / block shape sweep for the N x N uint8 matrix stores (pure stores), MI355X.
//   kern<RPW,WPB>: the Hamming kernel's shape -- wave = RPW rows x 1 KiB, block = WPB waves stacked in rows, grid x = column segment
//   colb<R,W>:     block = W waves side by side (W KiB contiguous) x R rows, grid x = column block
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int RPW, int WPB> __global__ __launch_bounds__(64 * WPB) void kern(unsigned char *out, long n, long ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long col = (long)blockIdx.x * 1024 + lane * 16;
    const long r0 = ((long)blockIdx.y * WPB + wave) * RPW;
    if (col + 16 > n) return;
    u32x4 v = {1, 2, 3, (unsigned)lane};
#pragma unroll
    for (int r = 0; r < RPW; ++r)
        if (r0 + r < n) { v.x += r; __builtin_nontemporal_store(v, (u32x4 *)(out + (r0 + r) * ld + col)); }
}
template <int R, int W> __global__ __launch_bounds__(64 * W) void colb(unsigned char *out, long n, long ld) {
    const long col = (long)blockIdx.x * (1024 * W) + threadIdx.x * 16;
    if (col + 16 > n) return;
    u32x4 v = {1, 2, 3, threadIdx.x};
#pragma unroll
    for (int r = 0; r < R; ++r) {
        long row = (long)blockIdx.y * R + r;
        if (row < n) { v.x += r; __builtin_nontemporal_store(v, (u32x4 *)(out + row * ld + col)); }
    }
}
int main() {
    const long n = 50000;
    unsigned char *out; hipMalloc(&out, (size_t)n * 57344 + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, long ld, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-18s ld=%6ld %7.3f ms  %7.1f GB/s\n", name, ld, ms, (double)n * n / ms / 1e6);
    };
    for (long ld : {50176L, 51200L, 53248L, 57344L}) {
#define KN(RPW, WPB) run("kern RPW=" #RPW " WPB=" #WPB, ld, [&] { dim3 g((unsigned)((n + 1023) / 1024), (unsigned)((n + RPW * WPB - 1) / (RPW * WPB))); kern<RPW, WPB><<<g, 64 * WPB>>>(out, n, ld); });
        KN(8, 2) KN(4, 2) KN(4, 4) KN(8, 1) KN(2, 4) KN(16, 1)
#define CB(R, W) run("colb R=" #R " W=" #W, ld, [&] { dim3 g((unsigned)((n + 1024 * W - 1) / (1024 * W)), (unsigned)((n + R - 1) / R)); colb<R, W><<<g, 64 * W>>>(out, n, ld); });
        CB(1, 4) CB(4, 4) CB(8, 4) CB(8, 2) CB(4, 8) CB(2, 16)
    }
    return 0;
}
