// XCD-affine chunk ownership: does it matter WHICH XCD writes a given 4-KiB chunk?  Pure stores, MI355X.
// Hypothesis from write_bw4/5: a one-shot linear fill (block i writes chunk i, blocks round-robin over the 8 XCDs) is
// fast because XCD x only ever writes chunks with index % 8 == x.  Here a block (linear id b, XCD b % 8) writes R rows of
// one 4-KiB column block, the rows chosen 8 apart so that every chunk it writes has (chunk index + SHIFT) % 8 == b % 8.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int R> __global__ __launch_bounds__(256) void xcd(unsigned char *out, long n, long ld, int cb, int shift, int mode) {
    const long b = blockIdx.x;
    const int x = (int)(b & 7);
    const long q = b >> 3;
    const long ngroups = (n + 8 * R - 1) / (8 * R);
    const int c = (mode == 2) ? (int)(q / ngroups) : (int)(q % cb);      // mode 2: column block slowest (hashes stay in L1)
    const long g = (mode == 2) ? q % ngroups : q / cb;
    const int cpr = (int)(ld >> 12);                       // chunks per row
    int rho;
    if (mode != 1) {                                       // affine: (cpr * r + c + shift) % 8 == x
        int inv = 1;                                       // inverse of cpr mod 8 (cpr odd)
        for (int t = 1; t < 8; t += 2) if ((cpr * t & 7) == 1) inv = t;
        rho = (int)((((x - c - shift) % 8 + 8) % 8) * inv & 7);
    } else rho = x;                                        // naive: residue = XCD id regardless of the column block
    const long col = (long)c * 4096 + threadIdx.x * 16;
    if (col + 16 > n) return;
    u32x4 v = {1, 2, 3, threadIdx.x};
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const long row = g * (8 * R) + rho + 8 * j;
        if (row < n) { v.x += j; __builtin_nontemporal_store(v, (u32x4 *)(out + row * ld + col)); }
    }
}
int main() {
    const long n = 50000;
    unsigned char *out; hipMalloc(&out, (size_t)n * 61440 + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, long ld, int shift, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-22s ld=%6ld shift=%d %7.3f ms  %7.1f GB/s\n", name, ld, shift, ms, (double)n * n / ms / 1e6);
    };
    for (long ld : {53248L}) {
        const int cb = (int)((n + 4095) / 4096);
#define XC(R, MODE, SH) { long groups = (n + 8 * R - 1) / (8 * R); unsigned nb = (unsigned)(groups * cb * 8); \
        run("xcd R=" #R " mode=" #MODE, ld, SH, [&] { xcd<R><<<nb, 256>>>(out, n, ld, cb, SH, MODE); }); }
        XC(1, 0, 0) XC(2, 0, 0) XC(4, 0, 0) XC(8, 0, 0) XC(16, 0, 0)
        XC(4, 0, 1) XC(4, 0, 2) XC(4, 0, 3) XC(4, 0, 4) XC(4, 0, 5) XC(4, 0, 6) XC(4, 0, 7)
        XC(1, 1, 0) XC(4, 1, 0) XC(8, 1, 0)
        XC(1, 2, 0) XC(2, 2, 0) XC(4, 2, 0) XC(8, 2, 0)
#define XL(R, LDS) { long groups = (n + 8 * R - 1) / (8 * R); unsigned nb = (unsigned)(groups * cb * 8); \
        hipFuncSetAttribute((const void *)xcd<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        run("xcd R=" #R " lds=" #LDS "K", ld, 0, [&] { xcd<R><<<nb, 256, LDS * 1024>>>(out, n, ld, cb, 0, 0); }); }
        XL(1, 20) XL(1, 40) XL(1, 80) XL(4, 20) XL(4, 40) XL(4, 80) XL(4, 159) XL(8, 40) XL(8, 80) XL(8, 159) XL(16, 80) XL(16, 159)
    }
    return 0;
}
