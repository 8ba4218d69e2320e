#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void zero(unsigned *b, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) b[i] = 0; }
__global__ void hist(const unsigned *h, int n, unsigned *bins) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) atomicAdd(&bins[h[i]], 1u); }
__global__ void cnt(const unsigned *bins, unsigned *out) {
    __shared__ unsigned s; if (threadIdx.x == 0) s = 0; __syncthreads();
    unsigned m = 0; for (int j = 0; j < 8; ++j) m += bins[(blockIdx.x * 256 + threadIdx.x) * 8 + j] != 0;
    atomicAdd(&s, m); __syncthreads(); if (threadIdx.x == 0) out[blockIdx.x] = s; }
int main() {
    const int nb = 32, nbins = nb * 2048, n = 500;
    std::vector<unsigned> h(n); for (int i = 0; i < n; ++i) h[i] = (unsigned)((i * 2654435761u) % nbins);
    unsigned *dh, *out; hipMalloc(&dh, n * 4); hipMemcpy(dh, h.data(), n * 4, hipMemcpyHostToDevice); hipMalloc(&out, nb * 4);
    for (int mode = 0; mode < 4; ++mode) {
        unsigned *bins; hipMalloc(&bins, nbins * 4);
        hipMemset(bins, 0xFF, nbins * 4); hipDeviceSynchronize();   // make stale garbage obvious
        if (mode == 0) hipMemsetAsync(bins, 0, nbins * 4, nullptr);
        if (mode == 1) { hipMemsetAsync(bins, 0, nbins * 4, nullptr); hipStreamSynchronize(nullptr); }
        if (mode == 2) zero<<<nbins / 256, 256>>>(bins, nbins);
        if (mode == 3) { hipMemsetAsync(bins, 0, nbins * 4, nullptr); hipDeviceSynchronize(); }
        hist<<<2, 256>>>(dh, n, bins);
        cnt<<<nb, 256>>>(bins, out);
        hipDeviceSynchronize();
        std::vector<unsigned> o(nb); hipMemcpy(o.data(), out, nb * 4, hipMemcpyDeviceToHost);
        unsigned tot = 0; printf("mode %d:", mode); for (int i = 0; i < nb; ++i) { printf(" %u", o[i]); tot += o[i]; } printf(" | total %u\n", tot);
        hipFree(bins);
    }
}
