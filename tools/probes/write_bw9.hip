// Persistent blocks that take XCD-affine tiles from per-XCD atomic counters (in-order, like the hardware dispatcher) and
// bound their outstanding stores with s_waitcnt vmcnt(R): can software-pipelined blocks keep the one-shot store rate?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int R, int WAITN> __global__ __launch_bounds__(256) void pers(unsigned char *out, long n, long ld, int cb, unsigned tiles_per_x,
                                                                        unsigned *ctr) {
    __shared__ unsigned tsh;
    const int cpr = (int)(ld >> 12);
    int inv = 1;
    for (int t = 1; t < 8; t += 2) if ((cpr * t & 7) == 1) inv = t;
    const int x = blockIdx.x & 7;
    u32x4 v = {1, 2, 3, threadIdx.x};
    for (;;) {
        if (threadIdx.x == 0) tsh = atomicAdd(&ctr[x * 32], 1u);
        __syncthreads();
        const unsigned q = tsh;
        __syncthreads();
        if (q >= tiles_per_x) break;
        const int c = (int)(q % cb);
        const long g = q / cb;
        const int rho = (int)((((x - c) % 8 + 8) % 8) * inv & 7);
        const long col = (long)c * 4096 + threadIdx.x * 16;
        if (col + 16 <= n) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const long row = g * (8 * R) + rho + 8 * j;
                if (row < n) { v.x += j; __builtin_nontemporal_store(v, (u32x4 *)(out + row * ld + col)); }
            }
        }
        if (WAITN == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (WAITN == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (WAITN == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
}
int main() {
    const long n = 50000, ld = 53248;
    unsigned char *out; hipMalloc(&out, (size_t)n * ld + 4096);
    unsigned *ctr; hipMalloc(&ctr, 8 * 32 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, int P, auto launch) {
        float tot = 0;
        for (int i = 0; i < 23; ++i) {
            hipMemsetAsync(ctr, 0, 8 * 32 * 4, nullptr);
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (i >= 3) tot += ms;
        }
        printf("%-18s P=%5d %7.3f ms  %7.1f GB/s\n", name, P, tot / 20, (double)n * n / (tot / 20) / 1e6);
    };
    const int cb = (int)((n + 4095) / 4096);
    for (int P : {512, 1024, 2048}) {
#define PS(R, W) { unsigned tpx = (unsigned)((n + 8 * R - 1) / (8 * R) * cb); run("pers R=" #R " wait=" #W, P, [&] { pers<R, W><<<P, 256>>>(out, n, ld, cb, tpx, ctr); }); }
        PS(4, 0) PS(4, 4) PS(4, 8) PS(4, 99) PS(8, 0) PS(8, 8) PS(1, 0) PS(2, 0)
    }
    return 0;
}
