#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void wr(unsigned *out) { if (threadIdx.x == 0) out[blockIdx.x] = blockIdx.x + 1; }
__global__ void wr_after_read(const unsigned *in, unsigned *out) {
    __shared__ unsigned s;
    unsigned v = in[blockIdx.x * 2048 + threadIdx.x * 8];
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    atomicAdd(&s, v);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s + blockIdx.x + 1;
}
int check(const char *tag, unsigned *d, int n) {
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += (h[i] != (unsigned)i + 1);
    printf("%s: n=%d bad=%d first: %u %u %u %u %u %u %u %u %u\n", tag, n, bad, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
    return bad;
}
int main() {
    for (int n : {8, 32, 512}) {
        unsigned *a, *b, *in;
        hipMalloc(&a, n * 4);
        hipMemset(a, 0, n * 4);
        wr<<<n, 256>>>(a);
        hipDeviceSynchronize();
        check("hipMalloc      ", a, n);
        hipStream_t st = nullptr;
        hipMallocAsync((void **)&b, n * 4, st);
        wr<<<n, 256, 0, st>>>(b);
        hipStreamSynchronize(st);
        check("hipMallocAsync ", b, n);
        hipMalloc(&in, n * 2048 * 4);
        hipMemsetAsync(in, 0, n * 2048 * 4, st);
        unsigned *c;
        hipMallocAsync((void **)&c, n * 4, st);
        wr_after_read<<<n, 256, 0, st>>>(in, c);
        hipStreamSynchronize(st);
        check("async+memset   ", c, n);
        unsigned long long *tot;
        hipMallocAsync((void **)&tot, 8, st);
        (void)tot;
    }
    return 0;
}
