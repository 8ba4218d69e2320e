// counts_sort.hip -- k-mer counting for 17 <= k < 32, where a direct 4^k histogram is impossible:
// radix sort of the uint64 hashes + run-length encode (rocPRIM device primitives, AMD's own library -- the
// counterpart of np.unique(return_counts=True) in count_uniq_hash, kmer_count.py:476-491), then the
// reverse-complement merge (kmer_count.py:643-685) by binary search in the sorted unique keys and an
// order-preserving compaction.  Off the headline path (the reference's default k range is 6..16).
#include <cstring>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>

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
    TRYH(hipMalloc((void **)&sorted, (size_t)n * 8));
    TRYH(hipMalloc((void **)&ru, (size_t)n * 8));
    TRYH(hipMalloc((void **)&rc, (size_t)n * 4));
    TRYH(hipMalloc((void **)&nruns, 8));
    size_t ta = 0, tb = 0;
    // all 64 bits: the invalid hash (all ones) must sort last
    TRYH(rocprim::radix_sort_keys(nullptr, ta, const_cast<uint64_t *>(hash_dev), sorted, (size_t)n, 0, 64, st));
    TRYH(rocprim::run_length_encode(nullptr, tb, sorted, (size_t)n, ru, rc, nruns, st));
    TRYH(hipMalloc(&tmp, (ta > tb ? ta : tb) + 16));
    TRYH(rocprim::radix_sort_keys(tmp, ta, const_cast<uint64_t *>(hash_dev), sorted, (size_t)n, 0, 64, st));
    TRYH(rocprim::run_length_encode(tmp, tb, sorted, (size_t)n, ru, rc, nruns, st));
    uint32_t runs = 0;
    TRYH(hipMemcpyAsync(&runs, nruns, 4, hipMemcpyDeviceToHost, st));
    TRYH(hipStreamSynchronize(st));
    int64_t m = runs;
    if (m > 0) {   // drop the run of invalid hashes (np.unique result minus invalid, kmer_count.py:485-487)
        uint64_t last = 0;
        TRYH(hipMemcpy(&last, ru + (m - 1), 8, hipMemcpyDeviceToHost));
        if (last == ~0ull) --m;
    }
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
