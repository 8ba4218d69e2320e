// Persistent blocks over XCD-affine tiles: block p takes tiles p, p+P, p+2P, ... (P multiple of 8, so the XCD residue of
// its chunks never changes); tile = R rows (8 apart) x one 4-KiB column block.  Pure stores, MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int R, bool WAIT> __global__ __launch_bounds__(256) void pers(unsigned char *out, long n, long ld, int cb, long tiles) {
    const int cpr = (int)(ld >> 12);
    int inv = 1;
    for (int t = 1; t < 8; t += 2) if ((cpr * t & 7) == 1) inv = t;
    u32x4 v = {1, 2, 3, threadIdx.x};
    for (long b = blockIdx.x; b < tiles; b += gridDim.x) {
        const int x = (int)(b & 7);
        const long q = b >> 3;
        const int c = (int)(q % cb);
        const long g = q / cb;
        const int rho = (int)((((x - c) % 8 + 8) % 8) * inv & 7);
        const long col = (long)c * 4096 + threadIdx.x * 16;
        if (col + 16 > n) continue;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const long row = g * (8 * R) + rho + 8 * j;
            if (row < n) { v.x += j; __builtin_nontemporal_store(v, (u32x4 *)(out + row * ld + col)); }
        }
        if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
int main() {
    const long n = 50000, ld = 53248;
    unsigned char *out; hipMalloc(&out, (size_t)n * ld + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, int P, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-16s P=%5d %7.3f ms  %7.1f GB/s\n", name, P, ms, (double)n * n / ms / 1e6);
    };
    const int cb = (int)((n + 4095) / 4096);
    for (int P : {256, 512, 768, 1024, 2048, 4096}) {
#define PS(R) { long tiles = (n + 8 * R - 1) / (8 * R) * cb * 8; run("pers R=" #R, P, [&] { pers<R, false><<<P, 256>>>(out, n, ld, cb, tiles); }); \
        run("pers+wait R=" #R, P, [&] { pers<R, true><<<P, 256>>>(out, n, ld, cb, tiles); }); }
        PS(2) PS(4) PS(8) PS(16)
    }
    return 0;
}
