// Write-bandwidth probe for the Hamming-matrix store pattern (MI355X): `./write_bw [linear|tile|xcd] [ld]` (default: all).
//   linear : grid-stride / one-shot dwordx4 fill and hipMemsetAsync (the ceiling)
//   tile   : wave = 1 KiB of a row x R rows (row stride = ld), WPB waves side by side on columns or stacked on rows
//   xcd    : XCD-affine chunk ownership -- a block writes R rows of one 4-KiB column block, the rows chosen 8 apart so that every
//            chunk it writes has (chunk index + shift) % 8 == its XCD (block b runs on XCD b % 8); + the dynamic-LDS throttle sweep.
//            This is the finding hamdist_tile_kernel is built on (one residue class of chunks per XCD, few chunks in flight).
// The generations in between (one-thread-many-chunks fills, traversal orders, pitch sweeps, persistent column blocks, persistent
// XCD-affine blocks with counters: write_bw2 .. 6, 8, 9) found nothing the kernel uses and live in the git history (rounds 1 - 3).
//   hipcc --offload-arch=gfx950 -O3 -o probe_write_bw write_bw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void st(u32x4 *p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __global__ void linear(u32x4 *out, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    u32x4 v = {1, 2, 3, (unsigned)i};
    for (; i < n16; i += s) st<NT>(out + i, v);
}
// COLS_W: waves of a block side by side along a row (contiguous span = COLS_W KiB); ROWS: rows per wave
template <bool NT, int COLS_W, int ROWS_W> __global__ __launch_bounds__(256) void tile(unsigned char *out, long n, long ld, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wc = wave % COLS_W, wr = wave / COLS_W;
    const long col0 = ((long)blockIdx.x * COLS_W + wc) * 1024 + lane * 16;
    const long r0 = ((long)blockIdx.y * ROWS_W + wr) * rows;
    if (col0 + 16 > n) return;
    unsigned char *p = out + r0 * ld + col0;
    u32x4 v = {1, 2, 3, (unsigned)lane};
    for (int r = 0; r < rows && r0 + r < n; ++r, p += ld) { v.x += r; st<NT>((u32x4 *)p, v); }
}
// interleaved rows: wave (colseg, phase) writes rows phase, phase+P, phase+2P, ... so that at any
// time the resident waves cover a compact window of ~P full rows (like the linear fill)
template <bool NT, int COLS_W> __global__ __launch_bounds__(256) void tileI(unsigned char *out, long n, long ld, int P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wc = wave % COLS_W, wr = wave / COLS_W;
    const long col0 = ((long)blockIdx.x * COLS_W + wc) * 1024 + lane * 16;
    const long phase = (long)blockIdx.y * (4 / COLS_W) + wr;
    if (col0 + 16 > n || phase >= P) return;
    u32x4 v = {1, 2, 3, (unsigned)lane};
    for (long r = phase; r < n; r += P) { v.x += (unsigned)r; st<NT>((u32x4 *)(out + r * ld + col0), v); }
}
static int main_tile(long ld_arg, bool do_linear, bool do_tile) {
    const long n = 50000, ld = ld_arg;   // ld % 16 == 0
    unsigned char *out; hipMalloc(&out, (size_t)n * ld + 4096); printf("ld=%ld\n", ld);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-34s %7.3f ms  %7.1f GB/s\n", name, ms, (double)n * n / ms / 1e6);
    };
    size_t n16 = (size_t)n * n / 16;
    if (do_linear) {
    run("linear default 2048 blk", [&] { linear<false><<<2048, 256>>>((u32x4 *)out, n16); });
    run("linear nt      2048 blk", [&] { linear<true><<<2048, 256>>>((u32x4 *)out, n16); });
    run("linear default 8192 blk", [&] { linear<false><<<8192, 256>>>((u32x4 *)out, n16); });
    run("linear nt      8192 blk", [&] { linear<true><<<8192, 256>>>((u32x4 *)out, n16); });
    run("hipMemsetAsync", [&] { hipMemsetAsync(out, 1, (size_t)n * n, nullptr); });
    }
    if (!do_tile) return 0;
#define T(NT, CW, RW, R) run("tile nt=" #NT " colsW=" #CW " rowsW=" #RW " R=" #R, [&] { \
        dim3 g((unsigned)((n + 1024 * CW - 1) / (1024 * CW)), (unsigned)((n + RW * R - 1) / (RW * R))); tile<NT, CW, RW><<<g, 256>>>(out, n, ld, R); });
    T(false, 1, 4, 64) T(true, 1, 4, 64) T(false, 4, 1, 64) T(true, 4, 1, 64) T(false, 4, 1, 16) T(true, 4, 1, 16)
    T(false, 4, 1, 256) T(true, 4, 1, 256) T(false, 2, 2, 64) T(true, 2, 2, 64) T(true, 1, 4, 16) T(true, 1, 4, 256)
#define TI(NT, CW, P) run("tileI nt=" #NT " colsW=" #CW " P=" #P, [&] { \
        dim3 g((unsigned)((n + 1024 * CW - 1) / (1024 * CW)), (unsigned)((P + (4 / CW) - 1) / (4 / CW))); tileI<NT, CW><<<g, 256>>>(out, n, ld, P); });
    TI(false, 1, 168) TI(true, 1, 168) TI(false, 4, 168) TI(true, 4, 168) TI(false, 1, 84) TI(true, 1, 84) TI(false, 1, 336) TI(true, 1, 336)
    TI(false, 1, 42) TI(false, 4, 84) TI(false, 1, 672)
    return 0;
}

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
static int main_xcd() {
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

int main(int argc, char **argv) {
    const char *mode = argc > 1 ? argv[1] : "all";
    const long ld = argc > 2 ? atol(argv[2]) : 50000;      // ld % 16 == 0
    const bool all = !strcmp(mode, "all");
    if (all || !strcmp(mode, "linear") || !strcmp(mode, "tile")) main_tile(ld, all || !strcmp(mode, "linear"), all || !strcmp(mode, "tile"));
    if (all || !strcmp(mode, "xcd")) main_xcd();
    return 0;
}
