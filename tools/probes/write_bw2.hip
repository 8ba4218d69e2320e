// Which linear fill shape reaches hipMemset's rate?  (MI355X write ceiling probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// each thread writes U consecutive 16-byte chunks (U*16 B contiguous per lane)
template <int U> __global__ void lin_u(u32x4 *out, size_t n16) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * U, s = (size_t)gridDim.x * blockDim.x * U;
    u32x4 v = {1, 2, 3, 4};
    for (; i + U <= n16; i += s)
#pragma unroll
        for (int u = 0; u < U; ++u) out[i + u] = v;
}
// wave-contiguous: a wave writes U consecutive KiB (lane-interleaved, each store instr = 1 KiB contiguous)
template <int U> __global__ void lin_w(u32x4 *out, size_t n16) {
    size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
    size_t nw = (size_t)gridDim.x * blockDim.x / 64;
    u32x4 v = {1, 2, 3, 4};
    for (size_t base = w * 64 * U; base + 64 * U <= n16; base += nw * 64 * U)
#pragma unroll
        for (int u = 0; u < U; ++u) out[base + u * 64 + lane] = v;
}
// dword stores (4 B per lane)
__global__ void lin_d(unsigned *out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += s) out[i] = 7;
}
int main() {
    const size_t bytes = 2500000000ull / 4096 * 4096;
    unsigned char *out; hipMalloc(&out, bytes + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, int blocks, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-22s blocks=%6d %7.3f ms  %7.1f GB/s\n", name, blocks, ms, (double)bytes / ms / 1e6);
    };
    size_t n16 = bytes / 16;
    for (int b : {256, 512, 1024, 2048, 4096, 16384, 65536}) {
        run("lin_u<1>", b, [&] { lin_u<1><<<b, 256>>>((u32x4 *)out, n16); });
        run("lin_u<4>", b, [&] { lin_u<4><<<b, 256>>>((u32x4 *)out, n16); });
        run("lin_w<4>", b, [&] { lin_w<4><<<b, 256>>>((u32x4 *)out, n16); });
        run("lin_w<16>", b, [&] { lin_w<16><<<b, 256>>>((u32x4 *)out, n16); });
    }
    run("lin_d", 2048, [&] { lin_d<<<2048, 256>>>((unsigned *)out, bytes / 4); });
    run("lin_u<1> 1024thr", 1024, [&] { lin_u<1><<<1024, 1024>>>((u32x4 *)out, n16); });
    run("lin_u<1> 64thr", 8192, [&] { lin_u<1><<<8192, 64>>>((u32x4 *)out, n16); });
    run("one-shot lin_u<1>", (int)(n16 / 256), [&] { lin_u<1><<<(unsigned)(n16 / 256), 256>>>((u32x4 *)out, n16); });
    run("hipMemsetAsync", 0, [&] { hipMemsetAsync(out, 1, bytes, nullptr); });
    run("hipMemsetD32Async", 0, [&] { hipMemsetD32Async((hipDeviceptr_t)out, 1, bytes / 4, nullptr); });
    return 0;
}
