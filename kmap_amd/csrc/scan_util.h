// scan_util.h -- small shared device utilities (included by counts.hip and scan.hip)
#pragma once
#include "common.h"

namespace {
// exclusive scan of n uint32 values into uint64 offsets by ONE block (n up to a few million);
// total written to *total
__global__ __launch_bounds__(1024) void scan_single_block_kernel(const uint32_t *__restrict__ in, int64_t n,
                                                                 uint64_t *__restrict__ out,
                                                                 uint64_t *__restrict__ total) {
    __shared__ uint64_t part[1024];
    const int t = threadIdx.x;
    const int64_t chunk = (n + 1023) / 1024;
    const int64_t lo = (int64_t)t * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    uint64_t s = 0;
    for (int64_t i = lo; i < hi; ++i) s += in[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan
        uint64_t v = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = (t == 0) ? 0 : part[t - 1];
    for (int64_t i = lo; i < hi; ++i) {
        out[i] = run;
        run += in[i];
    }
    if (t == 1023) *total = part[1023];
}

}  // namespace
