// Persistent column-fixed blocks: block = W waves side by side (W KiB of one row), grid (colblocks, G); each block walks
// rows y, y+G, y+2G, ... so that at any time the chip writes ~G consecutive rows (a linear window at 4 KiB granularity)
// while a lane's columns (its hashes in the real kernel) stay fixed.  Pure stores, MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int W, bool NT> __global__ __launch_bounds__(64 * W) void pers(unsigned char *out, long n, long ld, int G) {
    const long col = (long)blockIdx.x * (1024 * W) + threadIdx.x * 16;
    if (col + 16 > n) return;
    u32x4 v = {1, 2, 3, threadIdx.x};
    for (long row = blockIdx.y; row < n; row += G) {
        v.x += 1;
        if (NT) __builtin_nontemporal_store(v, (u32x4 *)(out + row * ld + col));
        else *(u32x4 *)(out + row * ld + col) = v;
    }
}
int main() {
    const long n = 50000;
    unsigned char *out; hipMalloc(&out, (size_t)n * 57344 + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, long ld, int G, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-14s ld=%6ld G=%5d %7.3f ms  %7.1f GB/s\n", name, ld, G, ms, (double)n * n / ms / 1e6);
    };
    for (long ld : {50176L, 53248L}) {
        for (int G : {40, 80, 160, 320, 640, 1280}) {
#define PS(W, NT) run("pers W=" #W " NT=" #NT, ld, G, [&] { dim3 g((unsigned)((n + 1024 * W - 1) / (1024 * W)), G); pers<W, NT><<<g, 64 * W>>>(out, n, ld, G); });
            PS(4, true) PS(4, false) PS(1, true) PS(2, true) PS(8, true)
        }
    }
    return 0;
}
