// counts_sort.hip -- k-mer counting for 17 <= k < 32, where a direct 4^k histogram is impossible: the counterpart of
// np.unique(return_counts=True) in count_uniq_hash (kmer_count.py:476-491) as a hand-written LSD radix sort of the valid uint64
// hashes + run-length encoding, then the reverse-complement merge (kmer_count.py:643-685) by binary search in the sorted unique
// keys and an order-preserving compaction.  Off the headline path (the reference's default k range is 6..16).
//
// Sort: invalid hashes (all ones) are dropped first (flag, scan, scatter), then ceil(2k / 8) stable passes over 8-bit digits.
// A pass: (1) every wave histograms its tile of 1024 keys (64 lanes x 16, striped) into 256 LDS counters -> counts[digit][tile];
// (2) exclusive scan of the digit-major counts; (3) every wave ranks its tile's keys again -- lanes holding the same digit find each
// other with eight ballots (one per digit bit), a lane's rank is the digit's running count + the lanes of its group in front of it,
// the group's first lane bumps the count: stable, no atomics -- and stores key i at offset[digit][tile] + rank.
#include <cstring>
#include <string.h>

#include "counts_internal.h"
#include "scan_util.h"

namespace {
constexpr int BLK = 256;

__device__ __forceinline__ int64_t lower_bound(const uint64_t *__restrict__ a, int64_t n, uint64_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t m = (lo + hi) >> 1;
        if (a[m] < v) lo = m + 1;
        else hi = m;
    }
    return lo;
}

// flag[i] = 1 if entry i is emitted; okey/ocnt hold the emitted values (see bin_entry in counts.hip)
__global__ __launch_bounds__(BLK) void merge_decide_kernel(const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ cnt,
                                                           int64_t n, int k, int merge, uint32_t *__restrict__ flag,
                                                           uint64_t *__restrict__ okey, uint32_t *__restrict__ ocnt) {
    const int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    const uint64_t x = uniq[i];
    uint32_t c = cnt[i];
    uint64_t key = x;
    uint32_t keep = 1;
    if (merge) {
        const uint64_t r = revcom_hash(x, k);
        if (r == x) {
            c = c + c;
        } else {
            const int64_t j = lower_bound(uniq, n, r);
            const bool present = (j < n && uniq[j] == r);
            if (present && x > r) keep = 0;
            else {
                key = x > r ? r : x;
                if (present) c += cnt[j];
            }
        }
    }
    flag[i] = keep;
    okey[i] = key;
    ocnt[i] = c;
}

// ---- LSD radix sort of uint64 keys, 8-bit digits ---------------------------------------------------------------------------
constexpr int RS_ITEMS = 16, RS_TILE = KMAP_WAVE * RS_ITEMS, RS_WAVES = 4;   // 1024 keys per wave; four independent waves per block
__global__ __launch_bounds__(BLK) void rs_flag_valid_kernel(const uint64_t *__restrict__ h, int64_t n, uint32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (i < n) flag[i] = h[i] != ~0ull;
}
__global__ __launch_bounds__(BLK) void rs_compact_kernel(const uint64_t *__restrict__ h, const uint32_t *__restrict__ flag,
                                                         const uint64_t *__restrict__ off, int64_t n, uint64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (i < n && flag[i]) out[off[i]] = h[i];
}
__global__ __launch_bounds__(KMAP_WAVE *RS_WAVES) void rs_hist_kernel(const uint64_t *__restrict__ keys, int64_t n, int shift, int64_t n_tiles,
                                                                      uint32_t *__restrict__ counts) {
    __shared__ uint32_t cnt[RS_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * RS_WAVES + wave;
    for (int d = lane; d < 256; d += 64) cnt[wave][d] = 0;
    __builtin_amdgcn_wave_barrier();
    if (tile < n_tiles) {
#pragma unroll
        for (int i = 0; i < RS_ITEMS; ++i) {
            const int64_t idx = tile * RS_TILE + (int64_t)i * KMAP_WAVE + lane;
            if (idx < n) atomicAdd(&cnt[wave][(keys[idx] >> shift) & 255u], 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (tile < n_tiles)
        for (int d = lane; d < 256; d += 64) counts[(int64_t)d * n_tiles + tile] = cnt[wave][d];
}
__global__ __launch_bounds__(KMAP_WAVE *RS_WAVES) void rs_scatter_kernel(const uint64_t *__restrict__ keys, int64_t n, int shift, int64_t n_tiles,
                                                                         const uint64_t *__restrict__ offs, uint64_t *__restrict__ out) {
    __shared__ uint32_t cnt[RS_WAVES][256];
    __shared__ uint64_t base[RS_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * RS_WAVES + wave;
    if (tile >= n_tiles) return;                                          // wave-uniform; no block-wide barrier below
    for (int d = lane; d < 256; d += 64) {
        cnt[wave][d] = 0;
        base[wave][d] = offs[(int64_t)d * n_tiles + tile];
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll 4
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int64_t idx = tile * RS_TILE + (int64_t)i * KMAP_WAVE + lane;
        const bool live = idx < n;
        const uint64_t key = live ? keys[idx] : 0ull;
        const uint32_t dg = (uint32_t)(key >> shift) & 255u;
        unsigned long long same = __ballot(live);                         // lanes with this lane's digit (and a key)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((dg >> b) & 1u);
            same &= ((dg >> b) & 1u) ? bal : ~bal;
        }
        if (live) {
            const uint32_t old = cnt[wave][dg];                           // the group's lanes all read the count before its leader bumps it
            const uint32_t rank = old + (uint32_t)__popcll(same & lt);
            if ((same & lt) == 0ull) cnt[wave][dg] = old + (uint32_t)__popcll(same);
            out[base[wave][dg] + rank] = key;
        }
        __builtin_amdgcn_wave_barrier();                                  // LDS operations of a wave execute in order: item i + 1 sees the bump
    }
}
// run-length encoding of sorted keys: flag = first of its run; start[run] = index of the run's first key
__global__ __launch_bounds__(BLK) void rle_flag_kernel(const uint64_t *__restrict__ keys, int64_t n, uint32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]);
}
__global__ __launch_bounds__(BLK) void rle_emit_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ flag,
                                                       const uint64_t *__restrict__ off, int64_t n, uint64_t *__restrict__ uniq,
                                                       uint64_t *__restrict__ start) {
    const int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (i < n && flag[i]) {
        uniq[off[i]] = keys[i];
        start[off[i]] = (uint64_t)i;
    }
}
__global__ __launch_bounds__(BLK) void rle_count_kernel(const uint64_t *__restrict__ start, int64_t runs, int64_t n, uint32_t *__restrict__ cnt) {
    const int64_t r = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (r < runs) cnt[r] = (uint32_t)((r + 1 < runs ? start[r + 1] : (uint64_t)n) - start[r]);
}

__global__ __launch_bounds__(BLK) void scatter_kernel(const uint32_t *__restrict__ flag, const uint64_t *__restrict__ off,
                                                      const uint64_t *__restrict__ key, const uint32_t *__restrict__ cnt,
                                                      int64_t n, uint64_t *__restrict__ okey, uint32_t *__restrict__ ocnt) {
    const int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (i >= n || !flag[i]) return;
    okey[off[i]] = key[i];
    ocnt[off[i]] = cnt[i];
}
}  // namespace

int kmap_counts_sort_path(kmap_counts *c, const uint64_t *hash_dev, int64_t n, int k, int merge, int64_t *n_uniq,
                          hipStream_t st) {
    c->k = k;
    c->narrow = 0;
    c->n_uniq = 0;
    if (n_uniq) *n_uniq = 0;
    if (n == 0) return KMAP_OK;
    KMAP_REQUIRE(n < (int64_t)1 << 32, "counts (k >= 17): more than 2^32 positions per call are not supported");
    uint64_t *sorted = nullptr, *ru = nullptr, *mkey = nullptr, *off = nullptr;
    uint32_t *rc = nullptr, *nruns = nullptr, *flag = nullptr, *mcnt = nullptr;
    void *tmp = nullptr;
    auto cleanup = [&]() {
        void *ptrs[] = {sorted, ru, mkey, off, rc, nruns, flag, mcnt, tmp};
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
    };
    auto fail = [&](hipError_t e, const char *what) {
        kmap_set_error("counts (k >= 17): %s: %s", what, hipGetErrorString(e));
        cleanup();
        return e == hipErrorOutOfMemory ? KMAP_E_NOMEM : KMAP_E_HIP;
    };
#define TRYH(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail(_e, #expr); } while (0)
    // ---- valid hashes only (np.unique result minus invalid, kmer_count.py:485-487)
    const unsigned gridn = (unsigned)((n + BLK - 1) / BLK);
    TRYH(hipMalloc((void **)&flag, (size_t)n * 4));
    TRYH(hipMalloc((void **)&off, ((size_t)n + 1) * 8));
    rs_flag_valid_kernel<<<gridn, BLK, 0, st>>>(hash_dev, n, flag);
    { int r = exclusive_scan_u32(flag, n, off, st); if (r != KMAP_OK) { cleanup(); return r; } }
    uint64_t n_valid = 0;
    TRYH(hipMemcpyAsync(&n_valid, off + n, 8, hipMemcpyDeviceToHost, st));
    TRYH(hipStreamSynchronize(st));
    if (n_valid == 0) { cleanup(); return KMAP_OK; }
    const int64_t nv = (int64_t)n_valid;
    TRYH(hipMalloc((void **)&sorted, (size_t)nv * 8));
    TRYH(hipMalloc((void **)&ru, (size_t)nv * 8));                        // ping-pong partner of `sorted`, later the unique keys
    rs_compact_kernel<<<gridn, BLK, 0, st>>>(hash_dev, flag, off, n, sorted);
    // ---- LSD radix sort over the 2 k significant bits
    {
        const int64_t n_tiles = (nv + RS_TILE - 1) / RS_TILE;
        const unsigned gridt = (unsigned)((n_tiles + RS_WAVES - 1) / RS_WAVES);
        uint32_t *counts = nullptr;
        uint64_t *offs = nullptr;
        TRYH(hipMalloc((void **)&counts, (size_t)256 * n_tiles * 4));
        tmp = counts;                                                     // freed by cleanup()
        hipError_t e2 = hipMalloc((void **)&offs, ((size_t)256 * n_tiles + 1) * 8);
        if (e2 != hipSuccess) return fail(e2, "hipMalloc(radix offsets)");
        uint64_t *src = sorted, *dst = ru;
        int rcode = KMAP_OK;
        for (int shift = 0; shift < 2 * k && rcode == KMAP_OK; shift += 8) {
            rs_hist_kernel<<<gridt, KMAP_WAVE * RS_WAVES, 0, st>>>(src, nv, shift, n_tiles, counts);
            rcode = exclusive_scan_u32(counts, 256 * n_tiles, offs, st);
            if (rcode != KMAP_OK) break;
            rs_scatter_kernel<<<gridt, KMAP_WAVE * RS_WAVES, 0, st>>>(src, nv, shift, n_tiles, offs, dst);
            uint64_t *t = src; src = dst; dst = t;
        }
        hipError_t e3 = hipGetLastError();
        if (e3 == hipSuccess) e3 = hipStreamSynchronize(st);
        (void)hipFree(offs);
        if (rcode != KMAP_OK) { cleanup(); return rcode; }
        if (e3 != hipSuccess) return fail(e3, "radix sort");
        if (src != sorted) { uint64_t *t = sorted; sorted = ru; ru = t; }   // `sorted` names the buffer that holds the result
    }
    // ---- run-length encoding: unique keys (ru) + counts (rc)
    const unsigned gridv = (unsigned)((nv + BLK - 1) / BLK);
    rle_flag_kernel<<<gridv, BLK, 0, st>>>(sorted, nv, flag);
    { int r = exclusive_scan_u32(flag, nv, off, st); if (r != KMAP_OK) { cleanup(); return r; } }
    uint64_t runs64 = 0;
    TRYH(hipMemcpyAsync(&runs64, off + nv, 8, hipMemcpyDeviceToHost, st));
    TRYH(hipStreamSynchronize(st));
    int64_t m = (int64_t)runs64;
    TRYH(hipMalloc((void **)&rc, (size_t)m * 4));
    TRYH(hipMalloc((void **)&nruns, ((size_t)m + 1) * 8));                // run starts
    rle_emit_kernel<<<gridv, BLK, 0, st>>>(sorted, flag, off, nv, ru, reinterpret_cast<uint64_t *>(nruns));
    rle_count_kernel<<<(unsigned)((m + BLK - 1) / BLK), BLK, 0, st>>>(reinterpret_cast<const uint64_t *>(nruns), m, nv, rc);
    TRYH(hipGetLastError());
    TRYH(hipStreamSynchronize(st));
    (void)hipFree(flag); flag = nullptr;
    (void)hipFree(off); off = nullptr;
    if (m == 0) { cleanup(); return KMAP_OK; }
    TRYH(hipMalloc((void **)&flag, (size_t)m * 4));
    TRYH(hipMalloc((void **)&mkey, (size_t)m * 8));
    TRYH(hipMalloc((void **)&mcnt, (size_t)m * 4));
    TRYH(hipMalloc((void **)&off, ((size_t)m + 1) * 8));
    const unsigned grid = (unsigned)((m + BLK - 1) / BLK);
    merge_decide_kernel<<<grid, BLK, 0, st>>>(ru, rc, m, k, merge, flag, mkey, mcnt);
    { int r = exclusive_scan_u32(flag, m, off, st); if (r != KMAP_OK) { cleanup(); return r; } }
    uint64_t total = 0;
    TRYH(hipMemcpyAsync(&total, off + m, 8, hipMemcpyDeviceToHost, st));
    TRYH(hipStreamSynchronize(st));
    if (c->cap < total || !c->uniq) {
        if (c->uniq) (void)hipFree(c->uniq);
        if (c->cnt) (void)hipFree(c->cnt);
        c->uniq = nullptr; c->cnt = nullptr; c->cap = 0;
        TRYH(hipMalloc(&c->uniq, (size_t)(total ? total : 1) * 8));
        TRYH(hipMalloc((void **)&c->cnt, (size_t)(total ? total : 1) * 4));
        c->cap = total ? total : 1;
    }
    scatter_kernel<<<grid, BLK, 0, st>>>(flag, off, mkey, mcnt, m, (uint64_t *)c->uniq, c->cnt);
    TRYH(hipGetLastError());
    TRYH(hipStreamSynchronize(st));
#undef TRYH
    cleanup();
    c->n_uniq = (int64_t)total;
    if (n_uniq) *n_uniq = (int64_t)total;
    return KMAP_OK;
}
