// scan.hip -- per-read motif occurrence scan ("Hamming-ball scan over reads", BASELINE config 5).
// Replaces get_motif_occurence (reference motif_discovery.py:1422-1477), which per read and per
// consensus launches one hash kernel + two Hamming kernels from Python.  Here: one wave per read,
// hash windows rolled from the uint8 array (1 B per position), min(fwd, revcom) distance, wave-min of
// the hits, ballot-ordered compaction of the positions at the read's minimum distance.
//
// Roofline: HBM-bound, 1 B per array position + sparse hit output.
#include <algorithm>

#include "common.h"
#include "scan_util.h"

namespace {

constexpr int SC_WAVES = 4;

// number of candidate positions = len(hash_arr[0 : L-k+1]) with Python slice semantics (negative stop wraps)
__device__ __forceinline__ int64_t slice_stop(int64_t L, int k) {
    int64_t stop = L - k + 1;
    if (stop < 0) {
        stop += L;
        if (stop < 0) stop = 0;
    }
    return stop > L ? L : stop;
}

// min(fwd, rc) Hamming distance of the k-mer starting at read[p] (invalid window = all ones, compared as is)
__device__ __forceinline__ int window_dist(const uint8_t *__restrict__ read, int64_t L, int64_t p, int k, uint64_t m,
                                           uint64_t cons, uint64_t rcc, int revcom) {
    uint64_t h = 0;
    bool bad = (p + k > L);
    for (int i = 0; i < k; ++i) {
        const uint32_t b = (p + i < L) ? read[p + i] : 255u;
        bad |= (b == 255u);
        h = (h << 2) + b;
    }
    h = bad ? m : (h & m);
    int d = popc2((h ^ cons) & m);
    if (revcom) {
        const int d2 = popc2((h ^ rcc) & m);
        d = d2 < d ? d2 : d;
    }
    return d;
}

template <bool WRITE>
__global__ __launch_bounds__(KMAP_WAVE *SC_WAVES) void scan_kernel(const uint8_t *__restrict__ seq, int64_t n,
                                                                    const int64_t *__restrict__ borders, int64_t n_seq,
                                                                    int k, uint64_t cons, uint64_t rcc, int radius,
                                                                    int revcom, int32_t *__restrict__ hits,
                                                                    int8_t *__restrict__ min_dist,
                                                                    const uint64_t *__restrict__ offs,
                                                                    int32_t *__restrict__ pos_out) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * SC_WAVES + (threadIdx.x >> 6);
    if (s >= n_seq) return;
    int64_t st = borders[2 * s], en = borders[2 * s + 1];
    if (st < 0) st = 0;
    if (en > n) en = n;
    const int64_t L = en > st ? en - st : 0;
    const uint8_t *read = seq + st;
    const int64_t stop = slice_stop(L, k);
    const uint64_t m = low_mask<uint64_t>(k);
    int best;
    if (!WRITE) {
        best = 1 << 30;
        for (int64_t p = lane; p < stop; p += 64) {
            const int d = window_dist(read, L, p, k, m, cons, rcc, revcom);
            if (d <= radius && d < best) best = d;
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int v = __shfl_xor(best, o);
            best = v < best ? v : best;
        }
    } else {
        best = min_dist[s];
        if (best < 0) return;
    }
    int count = 0;
    uint64_t base = WRITE ? offs[s] : 0;
    if (best <= radius) {
        for (int64_t p0 = 0; p0 < stop; p0 += 64) {
            const int64_t p = p0 + lane;
            const bool hit = (p < stop) && (window_dist(read, L, p, k, m, cons, rcc, revcom) == best);
            const unsigned long long mask = __ballot(hit);
            if (WRITE && hit) pos_out[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)p;
            const int c = __popcll(mask);
            count += c;
            base += c;
        }
    }
    if (!WRITE && lane == 0) {
        hits[s] = count;
        min_dist[s] = (int8_t)((best <= radius) ? best : -1);
    }
}

__global__ __launch_bounds__(256) void scan_summary_kernel(const int32_t *__restrict__ hits, int64_t n_seq,
                                                           unsigned long long *__restrict__ stat) {
    unsigned long long cnt = 0, mx = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_seq; i += (int64_t)gridDim.x * 256) {
        const int32_t h = hits[i];
        cnt += h > 0;
        mx = (unsigned long long)h > mx && h > 0 ? (unsigned long long)h : mx;
    }
    for (int o = 32; o; o >>= 1) {
        cnt += __shfl_down(cnt, o);
        const unsigned long long m2 = __shfl_down(mx, o);
        mx = m2 > mx ? m2 : mx;
    }
    __shared__ unsigned long long wc[4], wm[4];                          // one pair of global atomics per block: thousands of
    if ((threadIdx.x & 63) == 0) {                                        // same-address 64-bit atomics cost 200 us, not 20
        wc[threadIdx.x >> 6] = cnt;
        wm[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long c = (wc[0] + wc[1]) + (wc[2] + wc[3]);
        unsigned long long m = wm[0] > wm[1] ? wm[0] : wm[1];
        m = wm[2] > m ? wm[2] : m;
        m = wm[3] > m ? wm[3] : m;
        if (c) atomicAdd(&stat[0], c);
        if (m) atomicMax(&stat[1], m);
    }
}

// thread = 4 reads: int32 hit counts -> bytes (saturating)
__global__ __launch_bounds__(256) void narrow_hits_kernel(const int32_t *__restrict__ hits, int64_t n_seq, uint8_t *__restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    for (int j = 0; j < 4; ++j)
        if (i + j < n_seq) {
            const int32_t h = hits[i + j];
            out[i + j] = (uint8_t)(h < 0 ? 0 : h > 255 ? 255 : h);
        }
}

}  // namespace

#include "scan_internal.h"

int kmap_scan_reserve(kmap_scan *s, int64_t n_seq) {
    if (s->cap_seq < n_seq) {
        void *ptrs[] = {s->hits, s->mind, s->offs};
        for (void *p : ptrs)
            if (p) KMAP_CHECK_HIP(hipFree(p));
        s->hits = nullptr; s->mind = nullptr; s->offs = nullptr; s->cap_seq = 0;
        KMAP_CHECK_HIP(hipMalloc((void **)&s->hits, (size_t)n_seq * 4));
        KMAP_CHECK_HIP(hipMalloc((void **)&s->mind, (size_t)n_seq));
        KMAP_CHECK_HIP(hipMalloc((void **)&s->offs, ((size_t)n_seq + 1) * 8 + 128));   // + slack: doubles as scratch of kmap_scan_summary / _fetch_stream_u8 (64 B header + n_seq bytes)
        s->cap_seq = n_seq;
    }
    return KMAP_OK;
}
int kmap_scan_reserve_pos(kmap_scan *s, uint64_t total) {
    if (s->cap_pos < (int64_t)total || !s->pos) {
        if (s->pos) KMAP_CHECK_HIP(hipFree(s->pos));
        s->pos = nullptr;
        s->cap_pos = 0;
        const size_t cap = total ? (size_t)total + (size_t)total / 8 : 1;
        KMAP_CHECK_HIP(hipMalloc((void **)&s->pos, cap * 4));
        s->cap_pos = (int64_t)cap;
    }
    return KMAP_OK;
}

extern "C" {

int kmap_scan_create(kmap_scan **s) {
    KMAP_REQUIRE(s, "scan_create: null");
    *s = new kmap_scan();
    return KMAP_OK;
}
int kmap_scan_destroy(kmap_scan *s) {
    if (!s) return KMAP_OK;
    void *ptrs[] = {s->hits, s->mind, s->offs, s->pos};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete s;
    return KMAP_OK;
}

int kmap_scan_run_dev(kmap_scan *s, const uint8_t *seq_dev, int64_t n, const int64_t *borders_dev, int64_t n_seq, int k,
                      uint64_t cons, int radius, int revcom, int64_t *total_hits, void *stream) {
    KMAP_REQUIRE(s, "scan_run: null handle");
    KMAP_REQUIRE(k > 0 && k < 32, "scan_run: k=%d out of range", k);
    KMAP_REQUIRE(n >= 0 && n_seq >= 0 && radius >= 0, "scan_run: negative size");
    s->n_seq = n_seq;
    s->total = 0;
    if (total_hits) *total_hits = 0;
    if (n_seq == 0) return KMAP_OK;
    KMAP_REQUIRE(seq_dev && borders_dev, "scan_run: null pointer");
    hipStream_t st = as_stream(stream);
    KMAP_TRY(kmap_scan_reserve(s, n_seq));
    const uint64_t m = low_mask<uint64_t>(k);
    const uint64_t c = cons & m;
    // reverse complement on the host (same arithmetic as revcom_hash; u32 wrap for k < 16 is moot: c < 4^k)
    uint64_t com = m - c, rcc = com & 3u;
    for (int i = 0; i < k - 1; ++i) { rcc <<= 2; com >>= 2; rcc += com & 3u; }
    const unsigned grid = (unsigned)((n_seq + SC_WAVES - 1) / SC_WAVES);
    scan_kernel<false><<<grid, KMAP_WAVE * SC_WAVES, 0, st>>>(seq_dev, n, borders_dev, n_seq, k, c, rcc, radius, revcom,
                                                              s->hits, s->mind, nullptr, nullptr);
    KMAP_TRY(exclusive_scan_u32(reinterpret_cast<const uint32_t *>(s->hits), n_seq, s->offs, st));
    uint64_t total = 0;
    KMAP_CHECK_HIP(hipMemcpyAsync(&total, s->offs + n_seq, 8, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));
    KMAP_TRY(kmap_scan_reserve_pos(s, total));
    if (total)
        scan_kernel<true><<<grid, KMAP_WAVE * SC_WAVES, 0, st>>>(seq_dev, n, borders_dev, n_seq, k, c, rcc, radius, revcom,
                                                                 s->hits, s->mind, s->offs, s->pos);
    KMAP_CHECK_HIP(hipGetLastError());
    s->total = (int64_t)total;
    if (total_hits) *total_hits = (int64_t)total;
    return KMAP_OK;
}

// fetch of the last run's lists on a stream of the caller's (a worker thread's own non-blocking stream: the copies then neither
// wait for nor hold up the launching thread's null-stream work).  The lists must be complete: call kmap_scan_summary first
// (it synchronises behind the kernels that write them); the handle must not run again before this returns.
int kmap_scan_fetch_stream(kmap_scan *s, int32_t *hits_per_read, int32_t *positions, void *stream) {
    KMAP_REQUIRE(s, "scan_fetch_stream: null handle");
    hipStream_t st = as_stream(stream);
    if (s->n_seq && hits_per_read)
        KMAP_CHECK_HIP(hipMemcpyAsync(hits_per_read, s->hits, (size_t)s->n_seq * 4, hipMemcpyDeviceToHost, st));
    if (s->total && positions)
        KMAP_CHECK_HIP(hipMemcpyAsync(positions, s->pos, (size_t)s->total * 4, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));
    return KMAP_OK;
}

// device addresses of the last run's lists (valid until the handle's next run): a multi-GPU caller gathers them with a
// device collective instead of fetching, exchanging and re-uploading them
int kmap_scan_result_dev(kmap_scan *s, void **hits_dev, void **pos_dev, int64_t *n_seq, int64_t *total) {
    KMAP_REQUIRE(s && hits_dev && pos_dev, "scan_result_dev: null");
    *hits_dev = s->n_seq ? (void *)s->hits : nullptr;
    *pos_dev = s->total ? (void *)s->pos : nullptr;
    if (n_seq) *n_seq = s->n_seq;
    if (total) *total = s->total;
    return KMAP_OK;
}

// what gen_motif_occurence_file's caller needs of a hit list without fetching it: reads with >= 1 hit, largest hit count
int kmap_scan_summary(kmap_scan *s, int64_t *reads_with_hits, int32_t *max_hits, void *stream) {
    KMAP_REQUIRE(s && reads_with_hits && max_hits, "scan_summary: null");
    *reads_with_hits = 0;
    *max_hits = 0;
    if (s->n_seq == 0) return KMAP_OK;
    hipStream_t st = as_stream(stream);
    unsigned long long *stat = reinterpret_cast<unsigned long long *>(s->offs);   // offs is dead once the positions are written
    KMAP_CHECK_HIP(hipMemsetAsync(stat, 0, 16, st));
    scan_summary_kernel<<<(unsigned)std::min<int64_t>((s->n_seq + 2047) / 2048, 512), 256, 0, st>>>(s->hits, s->n_seq, stat);
    KMAP_CHECK_HIP(hipGetLastError());
    unsigned long long host[2] = {0, 0};
    KMAP_CHECK_HIP(hipMemcpyAsync(host, stat, 16, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));
    *reads_with_hits = (int64_t)host[0];
    *max_hits = (int32_t)host[1];
    return KMAP_OK;
}

// the same with the hit counts narrowed to bytes on the device first (staged in the handle's dead offset array): for lists
// whose kmap_scan_summary max_hits is <= 255 -- larger counts would saturate at 255
int kmap_scan_fetch_stream_u8(kmap_scan *s, uint8_t *hits_u8, int32_t *positions, void *stream) {
    KMAP_REQUIRE(s, "scan_fetch_stream_u8: null handle");
    hipStream_t st = as_stream(stream);
    if (s->n_seq && hits_u8) {
        uint8_t *stage = reinterpret_cast<uint8_t *>(s->offs) + 64;
        narrow_hits_kernel<<<(unsigned)((s->n_seq + 1023) / 1024), 256, 0, st>>>(s->hits, s->n_seq, stage);
        KMAP_CHECK_HIP(hipGetLastError());
        KMAP_CHECK_HIP(hipMemcpyAsync(hits_u8, stage, (size_t)s->n_seq, hipMemcpyDeviceToHost, st));
    }
    if (s->total && positions)
        KMAP_CHECK_HIP(hipMemcpyAsync(positions, s->pos, (size_t)s->total * 4, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));
    return KMAP_OK;
}

int kmap_scan_fetch(kmap_scan *s, int32_t *hits_per_read, int8_t *min_dist, int32_t *positions) {
    KMAP_REQUIRE(s, "scan_fetch: null handle");
    KMAP_CHECK_HIP(hipDeviceSynchronize());
    if (s->n_seq) {
        if (hits_per_read) KMAP_CHECK_HIP(hipMemcpy(hits_per_read, s->hits, (size_t)s->n_seq * 4, hipMemcpyDeviceToHost));
        if (min_dist) KMAP_CHECK_HIP(hipMemcpy(min_dist, s->mind, (size_t)s->n_seq, hipMemcpyDeviceToHost));
    }
    if (s->total && positions) KMAP_CHECK_HIP(hipMemcpy(positions, s->pos, (size_t)s->total * 4, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

}  // extern "C"
