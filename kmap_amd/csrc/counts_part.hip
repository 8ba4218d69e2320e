// counts_part.hip -- partitioned histogram for k = 15, 16 (hashes as uint32) and the dispatch of the partitioned paths.
//
// A 4^k-bin table (4 / 16 GiB) does not fit LDS, and device-scope atomics into it run at ~20 G updates/s (k = 16: 112 ms for
// 1.5e9 k-mers, random bins: every update is a read-modify-write of a DRAM sector).  Instead:
//   (1) count the valid hashes per bucket (bucket = top 10 bits of the 2k-bit hash; 1024-bin LDS histogram per block);
//   (2) exclusive scan -> bucket offsets; (3) counting-sort tiles of 32768 hashes into bucket order (LDS counts, one global
//   fetch-add per non-empty bucket per tile, LDS cursors) -> a bucket-ordered key array (4 B per valid k-mer);
//   (4) a second level splits every bucket into S = 32 / 128 sub-buckets of 32768 bins with the same tile-staged counting sort,
//   run on 32768-key tiles that never cross a bucket border;
//   (5) one block per sub-bucket: stream its keys, LDS histogram, plain coalesced stores of the LDS bins into the table.  Every
//   bin of the table is written exactly once (no memset, no global atomics).
// 10 <= k <= 14 take counts_fine.hip (4096 buckets of <= 65 536 bins, 16-bit keys, one level); this file's one-level forms for
// those k (1024 buckets, 4-byte keys, 2 - 8 range passes over every bucket) were removed in round 4 (CHANGELOG.md).
// The table then goes through the usual compaction / reverse-complement merge (counts.hip).
#include <stdlib.h>

#include "common.h"
#include "counts_internal.h"
#include "packed_keys.h"
#include "scan_util.h"

namespace {
constexpr int PB = 10, NBK = 1 << PB;
constexpr int PT_TPB = 256, PT_PER = 128, PT_TILE = PT_TPB * PT_PER;   // 32768 hashes per tile
constexpr int PH_TPB = 1024, PH_BINS = 32768;
constexpr uint32_t INV32 = 0xFFFFFFFFu;

__global__ __launch_bounds__(PT_TPB) void part_count_kernel(const uint32_t *__restrict__ h, int64_t n, int shift,
                                                            uint32_t *__restrict__ gcount) {
    __shared__ uint32_t cnt[NBK];
    for (int b = threadIdx.x; b < NBK; b += PT_TPB) cnt[b] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * PT_TPB;
    for (int64_t i = (int64_t)blockIdx.x * PT_TPB + threadIdx.x; i < n; i += stride) {
        const uint32_t v = h[i];
        if (v != INV32) atomicAdd(&cnt[v >> shift], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < NBK; b += PT_TPB)
        if (cnt[b]) atomicAdd(&gcount[b], cnt[b]);
}

// the same count with the keys taken from the packed reads (thread = one 16-position group); k = 16: the all-T 16-mer's valid
// windows are counted here (all_ones), once
__global__ __launch_bounds__(PT_TPB) void part_count_packed_kernel(const uint32_t *__restrict__ codes, const uint16_t *__restrict__ inval,
                                                                   const uint32_t *__restrict__ skip, int64_t n, int k, int shift,
                                                                   uint32_t *__restrict__ gcount, unsigned long long *__restrict__ all_ones) {
    __shared__ uint32_t cnt[NBK + 64];
    for (int b = threadIdx.x; b < NBK + 64; b += PT_TPB) cnt[b] = 0;
    __syncthreads();
    const int64_t n_groups = (n + 15) >> 4, stride = (int64_t)gridDim.x * PT_TPB;
    const uint32_t dummy = (uint32_t)NBK + (threadIdx.x & 63u);           // invalid keys: the lane's private counter
    uint32_t ones = 0;
    for (int64_t g = (int64_t)blockIdx.x * PT_TPB + threadIdx.x; g < n_groups; g += stride) {
        uint32_t keys[16], n1;
        packed_group_keys(codes, inval, skip, n, k, g, keys, n1);
        ones += n1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t b = keys[i] == INV32 ? dummy : keys[i] >> shift;
            atomicAdd(&cnt[b], 1u);
        }
    }
    if (all_ones && ones) atomicAdd(all_ones, (unsigned long long)ones);
    __syncthreads();
    for (int b = threadIdx.x; b < NBK; b += PT_TPB)
        if (cnt[b]) atomicAdd(&gcount[b], cnt[b]);
}

__global__ void part_init_cursor_kernel(const uint64_t *__restrict__ goff, unsigned long long *__restrict__ cursor) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < NBK) cursor[b] = goff[b];
}

// Counting sort of one 32768-hash tile per block iteration, staged in LDS so that every bucket's run leaves the CU as
// consecutive addresses (scattering single dwords straight to the buckets left partially written lines to be evicted from
// L2: 27 ms for 1.5e9 hashes against 6 GB of output).  LDS: sorted tile 128 KiB + offsets 4 KiB + global bases 8 KiB.
constexpr int PS_TPB = 1024, PS_PER = PT_TILE / PS_TPB;   // 32 hashes per thread
static_assert(PS_TPB == NBK, "one thread per bucket in the scan / reservation steps");
template <bool PACKED>   // PACKED: keys hashed on the fly from the packed reads (h = codes), thread = two 16-position groups of the tile
__global__ __launch_bounds__(PS_TPB) void part_scatter_kernel(const uint32_t *__restrict__ h, const uint16_t *__restrict__ inval,
                                                              const uint32_t *__restrict__ skip, int k, int64_t n, int shift,
                                                              unsigned long long *__restrict__ cursor, uint32_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sorted[];      // PT_TILE entries
    __shared__ uint32_t cnt[NBK];                                           // counts -> exclusive offsets -> running cursors
    __shared__ uint32_t loff[NBK];                                          // exclusive offsets of the buckets inside the tile
    __shared__ unsigned long long base[NBK];
    __shared__ uint32_t wsum[PS_TPB / 64];
    const int64_t n_tiles = (n + PT_TILE - 1) / PT_TILE;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t t0 = tile * PT_TILE;
        cnt[threadIdx.x] = 0;                                               // NBK == PS_TPB
        __syncthreads();
        uint32_t v[PS_PER];
        if (PACKED) {
            static_assert(PS_PER == 32, "two groups of 16 keys per thread");
            const int64_t g0 = (t0 >> 4) + 2 * (int64_t)threadIdx.x;       // PT_TILE is a multiple of 16
            uint32_t n1;
            packed_group_keys(h, inval, skip, n, k, g0, v, n1);            // groups behind the array: all keys invalid (n test inside;
            packed_group_keys(h, inval, skip, n, k, g0 + 1, v + 16, n1);   //  the two padding groups keep the loads in bounds)
#pragma unroll
            for (int j = 0; j < PS_PER; ++j)
                if (v[j] != INV32) atomicAdd(&cnt[v[j] >> shift], 1u);
        } else {
#pragma unroll
            for (int j = 0; j < PS_PER; ++j) {
                const int64_t i = t0 + (int64_t)j * PS_TPB + threadIdx.x;
                v[j] = (i < n) ? h[i] : INV32;
                if (v[j] != INV32) atomicAdd(&cnt[v[j] >> shift], 1u);
            }
        }
        __syncthreads();
        // exclusive scan of the 1024 counts (one per thread): wave scan + wave sums
        const uint32_t c = cnt[threadIdx.x];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t inc = c;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const uint32_t excl = woff + inc - c;
        __syncthreads();
        loff[threadIdx.x] = excl;
        cnt[threadIdx.x] = excl;                                            // running cursor of the bucket inside the tile
        base[threadIdx.x] = c ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)c) : 0ull;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PS_PER; ++j)
            if (v[j] != INV32) sorted[atomicAdd(&cnt[v[j] >> shift], 1u)] = v[j];
        __syncthreads();
        const uint32_t n_valid = cnt[NBK - 1];                              // the last bucket's cursor ended at the tile's valid count
        for (uint32_t p = threadIdx.x; p < n_valid; p += PS_TPB) {
            const uint32_t key = sorted[p];
            const uint32_t b = key >> shift;
            out[base[b] + (p - loff[b])] = key;                             // consecutive p of a bucket -> consecutive addresses
        }
        __syncthreads();
    }
}

// block = sub-bucket (`bins_per_bucket` = 32768 consecutive bins): LDS histogram of its keys, then plain stores into the table
__global__ __launch_bounds__(PH_TPB) void part_hist_kernel(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ goff,
                                                           uint32_t bins_per_bucket, uint32_t *__restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lb[];
    const uint32_t bucket = blockIdx.x;
    for (uint32_t j = threadIdx.x; j < bins_per_bucket; j += PH_TPB) lb[j] = 0;
    __syncthreads();
    const uint64_t lo = goff[bucket], hi = goff[bucket + 1];
    const uint32_t low = bins_per_bucket - 1u;
    // 16-byte loads over the 4-key-aligned interior of [lo, hi), scalar head and tail
    const uint64_t lo4 = (lo + 3) & ~(uint64_t)3, hi4 = hi & ~(uint64_t)3;
    if (lo4 < hi4) {
        for (uint64_t i = lo + threadIdx.x; i < lo4; i += PH_TPB) atomicAdd(&lb[keys[i] & low], 1u);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *k4 = reinterpret_cast<const u32x4 *>(keys);
        // two loads ahead of the keys being counted (one block per CU: 16 waves; a plain load -> count loop exposed one memory
        // round trip per 16 keys and thread).  Clamped, unconditional loads so that the compiler keeps them in flight.
        const uint64_t qend = hi4 >> 2, qlast = qend - 1;
        uint64_t q = (lo4 >> 2) + threadIdx.x;
        u32x4 k0 = k4[q < qend ? q : qlast], k1 = k4[q + PH_TPB < qend ? q + PH_TPB : qlast];
        for (; q < qend; q += PH_TPB) {
            const u32x4 kv = k0;
            k0 = k1;
            k1 = k4[q + 2 * PH_TPB < qend ? q + 2 * PH_TPB : qlast];
            atomicAdd(&lb[kv.x & low], 1u);
            atomicAdd(&lb[kv.y & low], 1u);
            atomicAdd(&lb[kv.z & low], 1u);
            atomicAdd(&lb[kv.w & low], 1u);
        }
        for (uint64_t i = hi4 + threadIdx.x; i < hi; i += PH_TPB) atomicAdd(&lb[keys[i] & low], 1u);
    } else {
        for (uint64_t i = lo + threadIdx.x; i < hi; i += PH_TPB) atomicAdd(&lb[keys[i] & low], 1u);
    }
    __syncthreads();
    uint32_t *dst = table + (size_t)bucket * bins_per_bucket;
    for (uint32_t j = threadIdx.x; j < bins_per_bucket; j += PH_TPB) dst[j] = lb[j];
}
// ---- second level: tiles of <= 32768 keys inside one first-level bucket ---------------------------------------------------
constexpr int P2_MAX = 128;        // sub-buckets per bucket (k = 16); 32 at k = 15
__global__ void part2_ntiles_kernel(const uint64_t *__restrict__ goff, uint32_t *__restrict__ ntile) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < NBK) ntile[b] = (uint32_t)((goff[b + 1] - goff[b] + PT_TILE - 1) / PT_TILE);
}
// tile -> (bucket, key range); tile_off[NBK] = number of tiles.  One thread searches, the block reads the result from LDS.
struct TileRange {
    uint32_t bucket;
    uint64_t lo, hi;
};
__device__ __forceinline__ TileRange find_tile(const uint64_t *__restrict__ tile_off, const uint64_t *__restrict__ goff, uint64_t tile,
                                               TileRange *sh) {
    if (threadIdx.x == 0) {
        int lo = 0, hi = NBK;                          // largest b with tile_off[b] <= tile
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (tile_off[mid] <= tile) lo = mid;
            else hi = mid;
        }
        TileRange r;
        r.bucket = (uint32_t)lo;
        r.lo = goff[lo] + (tile - tile_off[lo]) * PT_TILE;
        r.hi = r.lo + PT_TILE < goff[lo + 1] ? r.lo + PT_TILE : goff[lo + 1];
        *sh = r;
    }
    __syncthreads();
    const TileRange r = *sh;
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(PS_TPB) void part2_count_kernel(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ goff,
                                                             const uint64_t *__restrict__ tile_off, int shift2, int S,
                                                             uint32_t *__restrict__ gcount2) {
    __shared__ uint32_t cnt[P2_MAX];
    __shared__ TileRange tr;
    const uint64_t n_tiles = tile_off[NBK];
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const TileRange r = find_tile(tile_off, goff, tile, &tr);
        if ((int)threadIdx.x < S) cnt[threadIdx.x] = 0;
        __syncthreads();
        for (uint64_t i = r.lo + threadIdx.x; i < r.hi; i += PS_TPB) atomicAdd(&cnt[(keys[i] >> shift2) & (uint32_t)(S - 1)], 1u);
        __syncthreads();
        if ((int)threadIdx.x < S && cnt[threadIdx.x]) atomicAdd(&gcount2[(size_t)r.bucket * S + threadIdx.x], cnt[threadIdx.x]);
        __syncthreads();
    }
}
__global__ void part2_init_cursor_kernel(const uint64_t *__restrict__ goff2, int64_t m, unsigned long long *__restrict__ cursor2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) cursor2[i] = goff2[i];
}
// the tile-staged counting sort of part_scatter_kernel with S sub-buckets inside one bucket
__global__ __launch_bounds__(PS_TPB) void part2_scatter_kernel(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ goff,
                                                               const uint64_t *__restrict__ tile_off, int shift2, int S,
                                                               unsigned long long *__restrict__ cursor2, uint32_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sorted[];      // PT_TILE entries
    __shared__ uint32_t cnt[P2_MAX], loff[P2_MAX];
    __shared__ unsigned long long base[P2_MAX];
    __shared__ TileRange tr;
    const uint64_t n_tiles = tile_off[NBK];
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const TileRange r = find_tile(tile_off, goff, tile, &tr);
        if ((int)threadIdx.x < S) cnt[threadIdx.x] = 0;
        __syncthreads();
        uint32_t v[PS_PER];
        bool live[PS_PER];
#pragma unroll
        for (int j = 0; j < PS_PER; ++j) {
            const uint64_t i = r.lo + (uint64_t)j * PS_TPB + threadIdx.x;
            live[j] = i < r.hi;
            v[j] = live[j] ? keys[i] : 0u;
            if (live[j]) atomicAdd(&cnt[(v[j] >> shift2) & (uint32_t)(S - 1)], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {                       // exclusive scan of the S <= 128 counts by the first wave (two per lane)
            const int lane = threadIdx.x;
            const uint32_t c0 = (2 * lane < S) ? cnt[2 * lane] : 0u, c1 = (2 * lane + 1 < S) ? cnt[2 * lane + 1] : 0u;
            uint32_t inc = c0 + c1;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            const uint32_t e0 = inc - c0 - c1, e1 = e0 + c0;
            if (2 * lane < S) {
                loff[2 * lane] = e0;
                base[2 * lane] = c0 ? atomicAdd(&cursor2[(size_t)r.bucket * S + 2 * lane], (unsigned long long)c0) : 0ull;
            }
            if (2 * lane + 1 < S) {
                loff[2 * lane + 1] = e1;
                base[2 * lane + 1] = c1 ? atomicAdd(&cursor2[(size_t)r.bucket * S + 2 * lane + 1], (unsigned long long)c1) : 0ull;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < S) cnt[threadIdx.x] = loff[threadIdx.x];      // running cursor of the sub-bucket inside the tile
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PS_PER; ++j)
            if (live[j]) sorted[atomicAdd(&cnt[(v[j] >> shift2) & (uint32_t)(S - 1)], 1u)] = v[j];
        __syncthreads();
        const uint32_t n_keys = (uint32_t)(r.hi - r.lo);
        for (uint32_t p = threadIdx.x; p < n_keys; p += PS_TPB) {
            const uint32_t key = sorted[p];
            const uint32_t sb = (key >> shift2) & (uint32_t)(S - 1);
            out[base[sb] + (p - loff[sb])] = key;                              // consecutive p of a sub-bucket -> consecutive addresses
        }
        __syncthreads();
    }
}
// k = 16: the all-T 16-mer's hash is 0xFFFFFFFF = the uint32 invalid marker; the hash kernel counts its valid windows aside and
// this adds them to their bin
__global__ void part_add_bin_kernel(uint32_t *__restrict__ table, size_t bin, const unsigned long long *__restrict__ extra) {
    table[bin] += (uint32_t)*extra;
}
}  // namespace

// k = 10: 32 LDS passes over the reads (13.7 ms at C3) against ~5 ms partitioned; k = 9: 8 passes (3.7 ms) win.  Few keys: the
// fixed cost of the partition (scans, 4096-block histogram pass) exceeds device atomics.
bool kmap_counts_part_applies(int k, int64_t n) { return k >= 10 && k <= 16 && n >= ((int64_t)1 << 20); }

// bins of c <- histogram of the valid (!= 0xFFFFFFFF) keys; the whole table is written (no prior memset needed).  Keys: a hash
// array (hash_dev), or -- hash_dev null -- hashed on the fly from the packed reads in the count and the scatter pass (no 4 B /
// position array written and read twice: 18 GB of the ~36 GB a k = 14 count pass moved at C3)
static int part_hist_any(kmap_counts *c, const uint32_t *hash_dev, const uint32_t *codes_dev, const uint16_t *inval_dev,
                         const uint32_t *skip_dev, int64_t n, int k, hipStream_t st) {
    if (kmap_counts_fine_applies(k)) return kmap_counts_fine_hist(c, hash_dev, codes_dev, inval_dev, skip_dev, n, k, st);
    const size_t n_bins = (size_t)1 << (2 * k);
    KMAP_TRY(kmap_counts_reserve_bins(c, k));
    const int shift = 2 * k - PB;
    uint32_t *gcount = nullptr, *keys = nullptr;
    uint64_t *goff = nullptr;
    unsigned long long *cursor = nullptr;
    void *small = nullptr;
    KMAP_TRY(kmap_scratch(&small, (size_t)NBK * 4 + ((size_t)NBK + 1) * 8 + (size_t)NBK * 8 + 16, st, KMAP_SLOT_A));
    goff = reinterpret_cast<uint64_t *>(small);                       // 8-byte aligned parts first
    cursor = reinterpret_cast<unsigned long long *>(goff + NBK + 1);
    unsigned long long *all_ones = cursor + NBK;                      // k = 16, packed source: valid windows of the all-T 16-mer
    gcount = reinterpret_cast<uint32_t *>(all_ones + 1);
    KMAP_TRY(kmap_scratch((void **)&keys, (size_t)n * 4, st, KMAP_SLOT_PART));
    KMAP_CHECK_HIP(hipMemsetAsync(all_ones, 0, 8 + (size_t)NBK * 4, st));
    const bool packed = hash_dev == nullptr;
    if (packed) {
        int64_t g = (((n + 15) >> 4) + PT_TPB - 1) / PT_TPB;
        if (g > 2048) g = 2048;
        part_count_packed_kernel<<<(unsigned)g, PT_TPB, 0, st>>>(codes_dev, inval_dev, skip_dev, n, k, shift, gcount, k == 16 ? all_ones : nullptr);
    } else {
        int64_t g = (n + PT_TPB - 1) / PT_TPB;
        if (g > 2048) g = 2048;
        part_count_kernel<<<(unsigned)g, PT_TPB, 0, st>>>(hash_dev, n, shift, gcount);
    }
    KMAP_TRY(exclusive_scan_u32(gcount, NBK, goff, st));              // goff[NBK] = number of valid hashes
    part_init_cursor_kernel<<<NBK / 256, 256, 0, st>>>(goff, cursor);
    int64_t tiles = (n + PT_TILE - 1) / PT_TILE;
    if (tiles > 1024) tiles = 1024;
    if (packed) {
        KMAP_TRY(kmap_allow_lds((const void *)part_scatter_kernel<true>, PT_TILE * 4));
        part_scatter_kernel<true><<<(unsigned)tiles, PS_TPB, (size_t)PT_TILE * 4, st>>>(codes_dev, inval_dev, skip_dev, k, n, shift, cursor, keys);
    } else {
        KMAP_TRY(kmap_allow_lds((const void *)part_scatter_kernel<false>, PT_TILE * 4));
        part_scatter_kernel<false><<<(unsigned)tiles, PS_TPB, (size_t)PT_TILE * 4, st>>>(hash_dev, nullptr, nullptr, k, n, shift, cursor, keys);
    }
    const uint32_t bins_per_bucket = (uint32_t)(n_bins >> PB);
    const int passes = (int)(bins_per_bucket / (uint32_t)PH_BINS);
    KMAP_REQUIRE(k >= 15 && passes <= P2_MAX, "counts: the two-level partition serves k = 15, 16 (k=%d)", k);
    KMAP_TRY(kmap_allow_lds((const void *)part_hist_kernel, (PH_BINS + 64) * 4));
    // second level: S = passes sub-buckets of 32768 bins per bucket; keys re-sorted tile by tile inside their bucket
    const int S = passes, shift2 = 15;                               // sub-bucket = bits [15, 15 + log2 S) of the hash
    uint32_t *gcount2 = nullptr, *keys2 = nullptr, *ntile = nullptr;
    uint64_t *goff2 = nullptr, *tile_off = nullptr;
    unsigned long long *cursor2 = nullptr;
    const size_t m = (size_t)NBK * S;
    void *aux = nullptr;
    KMAP_TRY(kmap_scratch(&aux, (m + 1) * 8 + m * 8 + ((size_t)NBK + 1) * 8 + m * 4 + (size_t)NBK * 4, st, KMAP_SLOT_B));
    goff2 = reinterpret_cast<uint64_t *>(aux);
    cursor2 = reinterpret_cast<unsigned long long *>(goff2 + m + 1);
    tile_off = reinterpret_cast<uint64_t *>(cursor2 + m);
    gcount2 = reinterpret_cast<uint32_t *>(tile_off + NBK + 1);
    ntile = gcount2 + m;
    KMAP_TRY(kmap_scratch((void **)&keys2, (size_t)n * 4, st, KMAP_SLOT_HASH));   // the hash array is dead: its slot takes the re-sorted keys
    KMAP_CHECK_HIP(hipMemsetAsync(gcount2, 0, m * 4, st));
    part2_ntiles_kernel<<<NBK / 256, 256, 0, st>>>(goff, ntile);
    KMAP_TRY(exclusive_scan_u32(ntile, NBK, tile_off, st));
    part2_count_kernel<<<2048, PS_TPB, 0, st>>>(keys, goff, tile_off, shift2, S, gcount2);
    KMAP_TRY(exclusive_scan_u32(gcount2, (int64_t)m, goff2, st));
    part2_init_cursor_kernel<<<(unsigned)((m + 255) / 256), 256, 0, st>>>(goff2, (int64_t)m, cursor2);
    KMAP_TRY(kmap_allow_lds((const void *)part2_scatter_kernel, PT_TILE * 4));
    part2_scatter_kernel<<<1024, PS_TPB, (size_t)PT_TILE * 4, st>>>(keys, goff, tile_off, shift2, S, cursor2, keys2);
    // one pass per sub-bucket: "bucket" = sub-bucket index, 32768 bins each
    part_hist_kernel<<<(unsigned)m, PH_TPB, (size_t)(PH_BINS + 64) * 4, st>>>(keys2, goff2, (uint32_t)PH_BINS, c->bins);
    if (packed && k == 16) part_add_bin_kernel<<<1, 1, 0, st>>>(c->bins, (size_t)0xFFFFFFFFu, all_ones);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

int kmap_counts_part_hist_u32(kmap_counts *c, const uint32_t *hash_dev, int64_t n, int k, hipStream_t st) {
    return part_hist_any(c, hash_dev, nullptr, nullptr, nullptr, n, k, st);
}
int kmap_counts_part_hist_packed(kmap_counts *c, const uint32_t *codes_dev, const uint16_t *inval_dev, const uint32_t *skip_dev,
                                 int64_t n, int k, hipStream_t st) {
    return part_hist_any(c, nullptr, codes_dev, inval_dev, skip_dev, n, k, st);
}

int kmap_counts_part_add_bin(kmap_counts *c, size_t bin, const unsigned long long *extra_dev, hipStream_t st) {
    part_add_bin_kernel<<<1, 1, 0, st>>>(c->bins, bin, extra_dev);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}
