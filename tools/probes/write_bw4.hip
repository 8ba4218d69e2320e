// Why does a row-pitched 2-D store pattern (5.3 TB/s) trail a one-shot linear fill (6.9 TB/s)?  Isolation probe, MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// one-shot linear, block = 4 KiB contiguous; DATA: 0 constant, 1 lane-dependent, 2 address-dependent
template <int DATA> __global__ void lin(u32x4 *out, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    u32x4 v = {1, 2, 3, 4};
    if (DATA == 1) v.w = threadIdx.x;
    if (DATA == 2) { v.x = (unsigned)i * 2654435761u; v.y = v.x ^ (unsigned)(i >> 7); v.z = v.x + 77; v.w = ~v.x; }
    out[i] = v;
}
// 2-D: block = 4 KiB of one row; grid (ceil(n/4096), n) or flattened 1-D; skips columns >= n
template <bool FLAT> __global__ void rows(unsigned char *out, long n, long ld, int cb) {
    long bx = FLAT ? blockIdx.x % cb : blockIdx.x, by = FLAT ? blockIdx.x / cb : blockIdx.y;
    const long col = bx * 4096 + threadIdx.x * 16;
    if (col + 16 > n) return;
    u32x4 v = {1, 2, 3, 4};
    *(u32x4 *)(out + by * ld + col) = v;
}
int main() {
    const long n = 50000, ld = 50176;
    unsigned char *out; hipMalloc(&out, (size_t)ld * ld + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, double bytes, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-44s %7.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    size_t n16 = (size_t)n * n / 16;
    unsigned g = (unsigned)((n16 + 255) / 256);
    run("linear one-shot const", 16.0 * n16, [&] { lin<0><<<g, 256>>>((u32x4 *)out, n16); });
    run("linear one-shot lane data", 16.0 * n16, [&] { lin<1><<<g, 256>>>((u32x4 *)out, n16); });
    run("linear one-shot hashed data", 16.0 * n16, [&] { lin<2><<<g, 256>>>((u32x4 *)out, n16); });
    for (long nn : {50000L, 49152L, 50176L}) {
        int cb = (int)((nn + 4095) / 4096);
        char nm[96];
        snprintf(nm, 96, "rows 2-D grid n=%ld ld=%ld", nn, ld);
        run(nm, (double)nn * nn, [&] { rows<false><<<dim3(cb, (unsigned)nn), 256>>>(out, nn, ld, cb); });
        snprintf(nm, 96, "rows flat grid n=%ld ld=%ld", nn, ld);
        run(nm, (double)nn * nn, [&] { rows<true><<<(unsigned)(cb * nn), 256>>>(out, nn, ld, cb); });
        snprintf(nm, 96, "rows 2-D grid n=%ld ld=n (dense)", nn);
        if (nn % 16 == 0) run(nm, (double)nn * nn, [&] { rows<false><<<dim3(cb, (unsigned)nn), 256>>>(out, nn, nn, cb); });
    }
    return 0;
}
