// tools/probes/valu_rate.hip -- issue cost of the VALU instruction classes the force kernels are made of, on gfx950, in SIMD cycles
// per wave64 instruction: independent streams of one instruction class (8 accumulators), W waves per SIMD (1 .. 4).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096, UNR = 8;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float a[UNR];
    f32x2 p[UNR];
    for (int i = 0; i < UNR; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 0.5f}; }
    const float c = seed * 0.999f;
    const f32x2 c2 = {c, c * 1.001f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
            if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            if (OP == 7) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c));
            if (OP == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 9) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a[i]) : "v"(c));
            if (OP == 10) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(c));
            if (OP == 11) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        }
    }
    float s = 0;
    for (int i = 0; i < UNR; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}
// dependent chain of one class (one accumulator): latency
template <int OP>
__global__ __launch_bounds__(256) void kdep(float *out, float seed) {
    float a = seed + threadIdx.x;
    f32x2 p = {a, a + 0.5f};
    const float c = seed * 0.999f;
    const f32x2 c2 = {c, c * 1.001f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(c2));
            if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
            if (OP == 7) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(c));
            if (OP == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(c));
            if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(c2));
            if (OP == 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
            if (OP == 13) asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(a), "+v"(p.x) : "v"(c));   // two chains, alternating
        }
    }
    if (a + p.x + p.y == 12345.678f) out[0] = a;
}

template <typename F>
double run(F launch, int waves_per_simd) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(waves_per_simd);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(waves_per_simd);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    float *out;
    CK(hipMalloc(&out, 64));
    int clk_khz = 0, cus = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("clock %d kHz, %d CUs\n", clk_khz, cus);
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_rcp_f32", "v_log_f32", "v_med3_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_add_f32_dpp", "v_mul_f32",
                           "v_bitop3_b32", "v_alignbit_b32", "v_xor_b32", "v_add_f32", "2x v_add_f32 (two chains)"};
    // warm the clocks
    for (int r = 0; r < 20; ++r) k<0><<<cus * 4, 256>>>(out, 1.0f);
    CK(hipDeviceSynchronize());
    printf("independent streams: SIMD cycles per wave64 instruction at W waves per SIMD (a block = 4 waves = one per SIMD; W blocks per CU)\n");
#define ROW(OP)                                                                                                          \
    {                                                                                                                    \
        printf("%-16s", names[OP]);                                                                                      \
        for (int w = 1; w <= 4; ++w) {                                                                                   \
            const double ms = run([&](int W) { k<OP><<<cus * W, 256>>>(out, 1.0f); }, w);                               \
            printf("  W=%d: %5.2f", w, ms * 1e-3 * clk_khz * 1e3 / ((double)ITER * UNR * w));                            \
        }                                                                                                                \
        printf("\n");                                                                                                    \
    }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11)
    printf("dependent chain (latency), one wave per SIMD: cycles per instruction\n");
#define DROW(OP) { const double ms = run([&](int W) { kdep<OP><<<cus * W, 256>>>(out, 1.0f); }, 1); printf("%-16s %5.2f\n", names[OP], ms * 1e-3 * clk_khz * 1e3 / ((double)ITER * UNR)); }
    DROW(0) DROW(1) DROW(2) DROW(7) DROW(8) DROW(6) DROW(12) DROW(13)
    return 0;
}
