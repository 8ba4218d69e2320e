// LDS atomic throughput on gfx950: lane-atomics per clock per CU for returning / non-returning OR, ADD, CAS and plain
// read / write, with addresses random over a table, bank-distinct, or all equal.  Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int OP, int PAT>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, int words) {
    extern __shared__ unsigned tab[];
    for (int i = threadIdx.x; i < words; i += 256) tab[i] = 0;
    __syncthreads();
    unsigned x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        unsigned a = PAT == 0 ? (x >> 8) & (words - 1) : PAT == 1 ? ((x >> 8) & (words - 1) & ~63u) | lane : 5u;
        if (OP == 0) acc += atomicOr(&tab[a], 1u << (x & 31));
        if (OP == 1) atomicOr(&tab[a], 1u << (x & 31));
        if (OP == 2) acc += atomicAdd(&tab[a], 1u);
        if (OP == 3) atomicAdd(&tab[a], 1u);
        if (OP == 4) acc += atomicCAS(&tab[a], 0u, x | 1u);
        if (OP == 5) acc += ((volatile unsigned *)tab)[a];
        if (OP == 6) ((volatile unsigned *)tab)[a] = x;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP, int PAT>
void run(const char *name, int cus, int blocks_per_cu) {
    const int iters = 2000, words = 2048;
    unsigned *out;
    hipMalloc(&out, (size_t)cus * blocks_per_cu * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<OP, PAT><<<cus * blocks_per_cu, 256, words * 4>>>(out, 10, words);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP, PAT><<<cus * blocks_per_cu, 256, words * 4>>>(out, iters, words);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double lane_ops = (double)cus * blocks_per_cu * 256 * iters;
    const double clk = 2.4e9 * ms * 1e-3;
    printf("%-44s %2d blocks/CU: %7.3f ms  %6.2f lane-ops / clk / CU (at 2.4 GHz)\n", name, blocks_per_cu, ms, lane_ops / clk / cus);
    hipFree(out);
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    for (int bpc : {2, 8}) {
        run<0, 0>("or  rtn   random 2048 words", cus, bpc);
        run<0, 1>("or  rtn   bank-distinct", cus, bpc);
        run<0, 2>("or  rtn   one address", cus, bpc);
        run<1, 0>("or  nortn random", cus, bpc);
        run<1, 1>("or  nortn bank-distinct", cus, bpc);
        run<2, 0>("add rtn   random", cus, bpc);
        run<3, 0>("add nortn random", cus, bpc);
        run<3, 1>("add nortn bank-distinct", cus, bpc);
        run<4, 0>("cas rtn   random", cus, bpc);
        run<5, 0>("read      random", cus, bpc);
        run<5, 1>("read      bank-distinct", cus, bpc);
        run<6, 0>("write     random", cus, bpc);
    }
    return 0;
}
