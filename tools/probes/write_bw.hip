// Write-bandwidth probe for the Hamming-matrix store pattern (MI355X).
//   linear : grid-stride dwordx4 fill (the ceiling)
//   tile   : wave = 1 KiB of a row x R rows (row stride = ld), WPB waves side by side on columns or stacked on rows
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
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
int main(int argc, char **argv) {
    const long n = 50000, ld = argc > 1 ? atol(argv[1]) : 50000;   // ld % 16 == 0
    unsigned char *out; hipMalloc(&out, (size_t)n * ld + 4096); printf("ld=%ld\n", ld);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-34s %7.3f ms  %7.1f GB/s\n", name, ms, (double)n * n / ms / 1e6);
    };
    size_t n16 = (size_t)n * n / 16;
    run("linear default 2048 blk", [&] { linear<false><<<2048, 256>>>((u32x4 *)out, n16); });
    run("linear nt      2048 blk", [&] { linear<true><<<2048, 256>>>((u32x4 *)out, n16); });
    run("linear default 8192 blk", [&] { linear<false><<<8192, 256>>>((u32x4 *)out, n16); });
    run("linear nt      8192 blk", [&] { linear<true><<<8192, 256>>>((u32x4 *)out, n16); });
    run("hipMemsetAsync", [&] { hipMemsetAsync(out, 1, (size_t)n * n, nullptr); });
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
