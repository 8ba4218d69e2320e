// tools/probes/clock_under_load.hip -- the shader clock a VALU-heavy kernel really runs at: s_memtime (shader cycles) against
// s_memrealtime (100 MHz) around a loop of packed-f32 / DPP work, every SIMD busy with W waves, for kernels of ~0.1 .. ~20 ms.
//   hipcc --offload-arch=gfx950 -O3 -o clock_under_load clock_under_load.hip && ./clock_under_load
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k(unsigned long long *out, float seed, int iters) {
    f32x2 p[8];
    float a[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 0.5f}; }
    const f32x2 c2 = {seed * 0.999f, seed * 1.001f};
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
            asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c2.x));
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[7] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

int main() {
    unsigned long long *out, h[2];
    CK(hipMalloc(&out, 64));
    int cus = 0, clk = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    printf("%d CUs, nominal %d kHz\n", cus, clk);
    for (int w = 1; w <= 3; ++w)
        for (int iters : {2000, 20000, 200000}) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
                k<<<cus * w, 256>>>(out, 1.0f, iters);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
                printf("W=%d iters %6d rep %d: %8.3f ms, %llu shader cycles / %llu ticks of 100 MHz -> %.0f MHz; %.2f cycles per instruction pair per wave\n", w, iters, rep, ms,
                       h[0], h[1], h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0, (double)h[0] / ((double)iters * 8));
            }
        }
    return 0;
}
