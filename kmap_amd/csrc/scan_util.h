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


// ---- multi-block exclusive scan: uint32 in -> uint64 out (+ total) -------------------------------------------
// phase 1: per-tile sums; phase 2: single-block scan of the tile sums; phase 3: per-tile scan + tile offset.
constexpr int SCAN_TPB = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_TPB * SCAN_ITEMS;

__global__ __launch_bounds__(SCAN_TPB) void scan_tile_sums_kernel(const uint32_t *__restrict__ in, int64_t n,
                                                                  uint32_t *__restrict__ tile_sums) {
    __shared__ uint32_t ws[SCAN_TPB / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) s += in[base + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];   // < 2^32 per tile by construction
}

__global__ __launch_bounds__(SCAN_TPB) void scan_tiles_kernel(const uint32_t *__restrict__ in, int64_t n,
                                                              const uint64_t *__restrict__ tile_off,
                                                              uint64_t *__restrict__ out) {
    __shared__ uint32_t ws[SCAN_TPB / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0u;
        s += v[i];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = s;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) ws[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += ws[w];
    uint64_t run = tile_off[blockIdx.x] + woff + (inc - s);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
}

// out[0..n) = exclusive prefix sums of in, out[n] = total.  tile scratch from the arena (slots C and D).
inline int exclusive_scan_u32(const uint32_t *in, int64_t n, uint64_t *out, hipStream_t st) {
    if (n <= 4096) {
        scan_single_block_kernel<<<1, 1024, 0, st>>>(in, n, out, out + n);
        return KMAP_OK;
    }
    const int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t *tsum = nullptr;
    uint64_t *toff = nullptr;
    KMAP_TRY(kmap_scratch((void **)&tsum, (size_t)tiles * 4, st, KMAP_SLOT_C));
    KMAP_TRY(kmap_scratch((void **)&toff, ((size_t)tiles + 1) * 8, st, KMAP_SLOT_D));
    scan_tile_sums_kernel<<<(unsigned)tiles, SCAN_TPB, 0, st>>>(in, n, tsum);
    scan_single_block_kernel<<<1, 1024, 0, st>>>(tsum, tiles, toff, out + n);   // total lands in out[n]
    scan_tiles_kernel<<<(unsigned)tiles, SCAN_TPB, 0, st>>>(in, n, toff, out);
    return KMAP_OK;
}
}  // namespace
