// kmer_ops.hip -- array operators of the k-mer path: hashing, per-read de-duplication,
// reverse complement, Hamming 1-vs-N (full / head / tail) and Hamming-ball masking.
// Each replaces one numpy-in/numpy-out operator of the reference (cited at the entry points).
// All of these are HBM-streaming integer kernels: 1 B (sequence) or 4/8 B (hash) per element.
#include <type_traits>

#include "common.h"

namespace {

constexpr int BLK = 256;

static inline unsigned grid_for(int64_t n, int per_block) {
    int64_t g = (n + per_block - 1) / per_block;
    return (unsigned)(g < 1 ? 1 : g);
}

// ---- hash at every position (taichi_core.py:3-61) ------------------------------------------------
// A thread owns POS consecutive positions and rolls the 2-bit window across them: it reads
// POS+k-1 bytes once instead of POS*k.  `bad` counts down the bases until the window is clear of
// the most recent 255 byte / end of array.
template <typename H, int POS>
__global__ __launch_bounds__(BLK) void hash_kernel(const uint8_t *__restrict__ seq, int64_t n, int k, H mask,
                                                   H *__restrict__ out) {
    const int64_t p0 = ((int64_t)blockIdx.x * BLK + threadIdx.x) * POS;
    if (p0 >= n) return;
    H h = 0;
    int bad = 0;   // > 0: window still contains an invalid byte
    // prime the window with the first k-1 bases
    for (int i = 0; i < k - 1; ++i) {
        const int64_t q = p0 + i;
        const uint32_t b = (q < n) ? seq[q] : 255u;
        h = (H)((h << 2) + b);
        bad = (b == 255u) ? k : (bad > 0 ? bad - 1 : 0);
    }
#pragma unroll 4
    for (int i = 0; i < POS; ++i) {
        const int64_t p = p0 + i;
        if (p >= n) break;
        const int64_t q = p + k - 1;
        const uint32_t b = (q < n) ? seq[q] : 255u;
        h = (H)(((h << 2) + b) & mask);
        bad = (b == 255u) ? k : (bad > 0 ? bad - 1 : 0);
        out[p] = bad ? (H)~(H)0 : h;
    }
}

// ---- reverse complement (taichi_core.py:181-224) -------------------------------------------------
template <typename H>
__global__ __launch_bounds__(BLK) void revcom_kernel(const H *__restrict__ in, int64_t n, int k, H *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (i < n) out[i] = revcom_hash(in[i], k);
}

// ---- Hamming 1-vs-N (taichi_core.py:63-177) ------------------------------------------------------
template <typename H>
__global__ __launch_bounds__(BLK) void ham1vN_kernel(const H *__restrict__ h, int64_t n, H cons, int shift, H cmask,
                                                     uint8_t *__restrict__ out) {
    const int64_t i0 = ((int64_t)blockIdx.x * BLK + threadIdx.x) * 4;
    if (i0 + 4 <= n) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) w |= (uint32_t)popc2((H)(((h[i0 + j] >> shift) ^ cons) & cmask)) << (8 * j);
        if ((reinterpret_cast<uintptr_t>(out) & 3) == 0) *reinterpret_cast<uint32_t *>(out + i0) = w;
        else
            for (int j = 0; j < 4; ++j) out[i0 + j] = (uint8_t)(w >> (8 * j));
    } else {
        for (int64_t i = i0; i < n; ++i) out[i] = (uint8_t)popc2((H)(((h[i] >> shift) ^ cons) & cmask));
    }
}

// ---- per-read de-duplication (kmer_count.py:743-760) ---------------------------------------------
// Short reads: one wave per read with a private open-addressing hash set in LDS (key -> smallest position).
// Positions are inserted 64 at a time in read order (atomicCAS on the key, atomicMin on the position), and after each
// chunk a position survives only if it is the stored minimum: the first occurrence wins, O(L) LDS operations.
constexpr int DD_CAP = 512;             // longest read handled here; table = next_pow2(2L) <= 1024 slots per wave
constexpr int DD_SLOTS = 2 * DD_CAP;
constexpr int DD_WAVES = 4;
template <typename H>
__device__ __forceinline__ uint64_t mix(H v) {
    uint64_t x = (uint64_t)v * 0x9E3779B97F4A7C15ull;
    return x ^ (x >> 29);
}
template <typename H>
__global__ __launch_bounds__(KMAP_WAVE *DD_WAVES) void dedupe_short_kernel(H *__restrict__ hash,
                                                                            const int64_t *__restrict__ borders,
                                                                            int64_t n_seq, int64_t n) {
    typedef typename std::conditional<sizeof(H) == 4, unsigned int, unsigned long long>::type K;
    __shared__ K keys[DD_WAVES][DD_SLOTS];
    __shared__ unsigned int minpos[DD_WAVES][DD_SLOTS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t s = (int64_t)blockIdx.x * DD_WAVES + wave;
    if (s >= n_seq) return;
    int64_t st = borders[2 * s], en = borders[2 * s + 1];
    if (st < 0) st = 0;
    if (en > n) en = n;
    const int L = (int)((en - st > DD_CAP) ? 0 : (en - st));   // long reads: dedupe_long_kernel
    if (L <= 1) return;
    int T = 64;
    while (T < 2 * L) T <<= 1;
    K *kt = keys[wave];
    unsigned int *pt = minpos[wave];
    const K EMPTY = (K)~(K)0;                                    // the invalid hash is never inserted
    for (int t = lane; t < T; t += 64) {
        kt[t] = EMPTY;
        pt[t] = 0xFFFFFFFFu;
    }
    __builtin_amdgcn_wave_barrier();
    for (int p0 = 0; p0 < L; p0 += 64) {
        const int p = p0 + lane;
        const bool act = p < L;
        const K v = act ? (K)hash[st + p] : EMPTY;
        int slot = -1;
        if (v != EMPTY) {
            slot = (int)(mix((H)v) & (uint64_t)(T - 1));
            for (;;) {
                const K prev = atomicCAS(&kt[slot], EMPTY, v);
                if (prev == EMPTY || prev == v) break;
                slot = (slot + 1) & (T - 1);
            }
            atomicMin(&pt[slot], (unsigned int)p);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0): this wave's LDS atomics have landed
        if (slot >= 0 && pt[slot] != (unsigned int)p) hash[st + p] = (H)~(H)0;
    }
}

// Long reads: one block per read, open-addressing table (key -> smallest position) in global
// scratch sized >= 2*L; phase A inserts with atomicMin on the position, phase B keeps a
// position only if it is the stored minimum.
template <typename H>
__global__ __launch_bounds__(BLK) void dedupe_long_kernel(H *__restrict__ hash, const int64_t *__restrict__ borders,
                                                          const int64_t *__restrict__ long_ids,
                                                          const int64_t *__restrict__ tab_off,
                                                          unsigned long long *__restrict__ keys,
                                                          unsigned long long *__restrict__ minpos, int64_t n) {
    const int64_t s = long_ids[blockIdx.x];
    int64_t st = borders[2 * s], en = borders[2 * s + 1];
    if (st < 0) st = 0;
    if (en > n) en = n;
    const int64_t L = en - st;
    const int64_t off = tab_off[blockIdx.x];
    const uint64_t cap = (uint64_t)(tab_off[blockIdx.x + 1] - off);   // power of two
    unsigned long long *K = keys + off, *P = minpos + off;
    const unsigned long long EMPTY = ~0ull;
    const H inval = (H)~(H)0;
    for (int64_t i = threadIdx.x; i < L; i += BLK) {
        const H v = hash[st + i];
        if (v == inval) continue;
        uint64_t slot = mix(v) & (cap - 1);
        for (;;) {
            unsigned long long prev = atomicCAS(&K[slot], EMPTY, (unsigned long long)v);
            if (prev == EMPTY || prev == (unsigned long long)v) {
                atomicMin(&P[slot], (unsigned long long)i);
                break;
            }
            slot = (slot + 1) & (cap - 1);
        }
    }
    __threadfence();
    __syncthreads();
    for (int64_t i = threadIdx.x; i < L; i += BLK) {
        const H v = hash[st + i];
        if (v == inval) continue;
        uint64_t slot = mix(v) & (cap - 1);
        while (__hip_atomic_load(&K[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)v)
            slot = (slot + 1) & (cap - 1);
        if (__hip_atomic_load(&P[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i)
            hash[st + i] = inval;
    }
}

// ---- masking (kmer_count.py:580-610) -------------------------------------------------------------
struct ConsTab {
    uint64_t cons[32];
    int32_t radius[32];
    int n;
};
// flag[p] = 1 iff the k-mer hash at p (invalid = all ones, compared like any value) is within
// radius of any consensus
__global__ __launch_bounds__(BLK) void mask_flag_kernel(const uint8_t *__restrict__ seq, int64_t n, int k, ConsTab t,
                                                        uint8_t *__restrict__ flag) {
    const int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (p >= n) return;
    uint64_t h = 0;
    bool bad = (p + k > n);
    for (int i = 0; i < k; ++i) {
        const uint32_t b = (p + i < n) ? seq[p + i] : 255u;
        bad |= (b == 255u);
        h = (h << 2) + b;
    }
    const uint64_t m = low_mask<uint64_t>(k);
    h = bad ? m : (h & m);   // invalid hash: every compared bit set
    uint8_t f = 0;
    for (int c = 0; c < t.n; ++c) f |= (popc2((h ^ t.cons[c]) & m) <= t.radius[c]);
    flag[p] = f;
}
__global__ __launch_bounds__(BLK) void mask_apply_kernel(uint8_t *__restrict__ seq, int64_t n, int k,
                                                         const uint8_t *__restrict__ flag) {
    const int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (p >= n) return;
    uint8_t f = 0;
    for (int i = 0; i < k; ++i)
        if (p - i >= 0) f |= flag[p - i];
    if (f) seq[p] = 255;
}

template <typename H>
int hash_launch(const uint8_t *seq_dev, int64_t n, int k, H *out_dev, void *stream) {
    KMAP_REQUIRE(k > 0 && k < 32 && 2 * k <= (int)(8 * sizeof(H)), "hash_kmers: k=%d out of range", k);
    KMAP_REQUIRE(n >= 0, "hash_kmers: n<0");
    if (n == 0) return KMAP_OK;
    KMAP_REQUIRE(seq_dev && out_dev, "hash_kmers: null pointer");
    constexpr int POS = 16;
    hash_kernel<H, POS><<<grid_for(n, BLK * POS), BLK, 0, as_stream(stream)>>>(seq_dev, n, k, low_mask<H>(k), out_dev);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

template <typename H>
int revcom_launch(const H *in_dev, int64_t n, int k, H *out_dev, void *stream) {
    KMAP_REQUIRE(k > 0 && k < 32 && 2 * k <= (int)(8 * sizeof(H)), "revcom: k=%d out of range", k);
    if (n <= 0) return KMAP_OK;
    KMAP_REQUIRE(in_dev && out_dev, "revcom: null pointer");
    revcom_kernel<H><<<grid_for(n, BLK), BLK, 0, as_stream(stream)>>>(in_dev, n, k, out_dev);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

template <typename H>
int ham_launch(const H *h_dev, int64_t n, H cons, int shift_bits, int clen, uint8_t *out_dev, void *stream) {
    KMAP_REQUIRE(clen > 0 && 2 * clen <= (int)(8 * sizeof(H)) && clen < 32, "hamdist_1vN: clen=%d out of range", clen);
    KMAP_REQUIRE(shift_bits >= 0 && shift_bits < (int)(8 * sizeof(H)) && (shift_bits & 1) == 0,
                 "hamdist_1vN: bad shift %d", shift_bits);
    if (n <= 0) return KMAP_OK;
    KMAP_REQUIRE(h_dev && out_dev, "hamdist_1vN: null pointer");
    ham1vN_kernel<H><<<grid_for(n, BLK * 4), BLK, 0, as_stream(stream)>>>(h_dev, n, cons, shift_bits,
                                                                          low_mask<H>(clen), out_dev);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

__global__ void classify_reads_kernel(const int64_t *__restrict__ borders, int64_t n_seq, int64_t n, int cap,
                                      int64_t *__restrict__ long_ids, unsigned long long *__restrict__ n_long) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seq) return;
    int64_t st = borders[2 * s], en = borders[2 * s + 1];
    if (st < 0) st = 0;
    if (en > n) en = n;
    if (en - st > cap) long_ids[atomicAdd(n_long, 1ull)] = s;
}

template <typename H>
int dedupe_launch(H *hash_dev, int64_t n, const int64_t *borders_dev, int64_t n_seq, void *stream) {
    KMAP_REQUIRE(n >= 0 && n_seq >= 0, "dedupe: negative size");
    if (n == 0 || n_seq == 0) return KMAP_OK;
    KMAP_REQUIRE(hash_dev && borders_dev, "dedupe: null pointer");
    hipStream_t st = as_stream(stream);
    dedupe_short_kernel<H><<<grid_for(n_seq, DD_WAVES), KMAP_WAVE * DD_WAVES, 0, st>>>(hash_dev, borders_dev, n_seq, n);
    KMAP_CHECK_HIP(hipGetLastError());
    // reads longer than DD_CAP (rare for kmap's short-read inputs): hash-set path
    int64_t *long_ids = nullptr;
    unsigned long long *n_long_dev = nullptr;
    KMAP_TRY(kmap_scratch((void **)&long_ids, (size_t)n_seq * 8 + 8, st, KMAP_SLOT_C));
    n_long_dev = (unsigned long long *)(long_ids + n_seq);
    KMAP_CHECK_HIP(hipMemsetAsync(n_long_dev, 0, 8, st));
    classify_reads_kernel<<<grid_for(n_seq, BLK), BLK, 0, st>>>(borders_dev, n_seq, n, DD_CAP, long_ids, n_long_dev);
    unsigned long long n_long = 0;
    KMAP_CHECK_HIP(hipMemcpyAsync(&n_long, n_long_dev, 8, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));
    int rc = KMAP_OK;
    if (n_long > 0) {
        // host-side table layout: needs the long reads' lengths
        int64_t *ids_h = (int64_t *)malloc((size_t)n_long * 8);
        int64_t *off_h = (int64_t *)malloc(((size_t)n_long + 1) * 8);
        int64_t *bh = (int64_t *)malloc((size_t)n_long * 16);
        KMAP_CHECK_HIP(hipMemcpy(ids_h, long_ids, (size_t)n_long * 8, hipMemcpyDeviceToHost));
        // atomics give an arbitrary order; any order is fine (reads are independent)
        for (unsigned long long i = 0; i < n_long; ++i)
            KMAP_CHECK_HIP(hipMemcpy(bh + 2 * i, borders_dev + 2 * ids_h[i], 16, hipMemcpyDeviceToHost));
        off_h[0] = 0;
        for (unsigned long long i = 0; i < n_long; ++i) {
            int64_t L = bh[2 * i + 1] - bh[2 * i];
            int64_t cap = 64;
            while (cap < 2 * L) cap <<= 1;
            off_h[i + 1] = off_h[i] + cap;
        }
        const size_t tot = (size_t)off_h[n_long];
        unsigned long long *tab = nullptr;
        int64_t *off_d = nullptr;
        hipError_t e = hipMalloc((void **)&tab, tot * 16);
        if (e == hipSuccess) e = hipMalloc((void **)&off_d, ((size_t)n_long + 1) * 8);
        if (e == hipSuccess) e = hipMemsetAsync(tab, 0xFF, tot * 16, st);
        if (e == hipSuccess) e = hipMemcpyAsync(off_d, off_h, ((size_t)n_long + 1) * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            dedupe_long_kernel<H><<<(unsigned)n_long, BLK, 0, st>>>(hash_dev, borders_dev, long_ids, off_d, tab,
                                                                    tab + tot, n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            kmap_set_error("dedupe (long reads): %s", hipGetErrorString(e));
            rc = (e == hipErrorOutOfMemory) ? KMAP_E_NOMEM : KMAP_E_HIP;
        }
        if (tab) (void)hipFree(tab);
        if (off_d) (void)hipFree(off_d);
        free(ids_h);
        free(off_h);
        free(bh);
    }
    return rc;
}

}  // namespace

// mask on a device array; shared with counts.hip
int kmap_mask_launch(uint8_t *seq_dev, int64_t n, int k, const uint64_t *cons, const int32_t *radius, int n_cons,
                     hipStream_t st) {
    KMAP_REQUIRE(k > 0 && k < 32, "mask_hamball: k=%d out of range", k);
    KMAP_REQUIRE(n_cons >= 0 && (n_cons == 0 || (cons && radius)), "mask_hamball: null consensus list");
    if (n <= 0 || n_cons == 0) return KMAP_OK;
    KMAP_REQUIRE(seq_dev, "mask_hamball: null pointer");
    uint8_t *flag = nullptr;
    KMAP_TRY(kmap_scratch((void **)&flag, (size_t)n, st, KMAP_SLOT_A));
    // the reference hashes ONCE and then applies the consensuses in turn (kmer_count.py:605-607),
    // so the final mask is the union over consensuses; batches of 32 OR into the same flag array
    // would need a read-modify-write, so run flag+apply per batch on the ORIGINAL sequence copy.
    uint8_t *orig = nullptr;
    if (n_cons > 32) {
        KMAP_TRY(kmap_scratch((void **)&orig, (size_t)n, st, KMAP_SLOT_B));
        KMAP_CHECK_HIP(hipMemcpyAsync(orig, seq_dev, (size_t)n, hipMemcpyDeviceToDevice, st));
    }
    for (int c0 = 0; c0 < n_cons; c0 += 32) {
        ConsTab t;
        t.n = (n_cons - c0 < 32) ? (n_cons - c0) : 32;
        for (int c = 0; c < t.n; ++c) {
            t.cons[c] = cons[c0 + c] & low_mask<uint64_t>(k);
            t.radius[c] = radius[c0 + c];
        }
        mask_flag_kernel<<<grid_for(n, BLK), BLK, 0, st>>>(orig ? orig : seq_dev, n, k, t, flag);
        mask_apply_kernel<<<grid_for(n, BLK), BLK, 0, st>>>(seq_dev, n, k, flag);
    }
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

// internal launchers used by counts.hip
int kmap_hash_launch_u32(const uint8_t *seq, int64_t n, int k, uint32_t *out, void *stream) {
    return hash_launch<uint32_t>(seq, n, k, out, stream);
}
int kmap_hash_launch_u64(const uint8_t *seq, int64_t n, int k, uint64_t *out, void *stream) {
    return hash_launch<uint64_t>(seq, n, k, out, stream);
}

// ---- blocking host-pointer wrappers -------------------------------------------------------------
namespace {
template <typename Tin, typename Tout, typename F>
int host_map(const Tin *in, int64_t n_in, Tout *out, int64_t n_out, F &&f) {
    DevBuf di, dout;
    KMAP_TRY(di.alloc((size_t)n_in * sizeof(Tin)));
    KMAP_TRY(dout.alloc((size_t)n_out * sizeof(Tout)));
    if (n_in) KMAP_CHECK_HIP(hipMemcpy(di.p, in, (size_t)n_in * sizeof(Tin), hipMemcpyHostToDevice));
    KMAP_TRY(f(di.as<Tin>(), dout.as<Tout>()));
    KMAP_CHECK_HIP(hipStreamSynchronize(nullptr));
    if (n_out) KMAP_CHECK_HIP(hipMemcpy(out, dout.p, (size_t)n_out * sizeof(Tout), hipMemcpyDeviceToHost));
    return KMAP_OK;
}
}  // namespace

namespace {
template <typename H>
int dedupe_host(H *hash, int64_t n, const int64_t *borders, int64_t n_seq) {
    KMAP_REQUIRE(n >= 0 && n_seq >= 0, "dedupe: negative size");
    if (n == 0 || n_seq == 0) return KMAP_OK;
    KMAP_REQUIRE(hash && borders, "dedupe: null pointer");
    DevBuf dh, db;
    KMAP_TRY(dh.alloc((size_t)n * sizeof(H)));
    KMAP_TRY(db.alloc((size_t)n_seq * 16));
    KMAP_CHECK_HIP(hipMemcpy(dh.p, hash, (size_t)n * sizeof(H), hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(db.p, borders, (size_t)n_seq * 16, hipMemcpyHostToDevice));
    KMAP_TRY(dedupe_launch<H>(dh.as<H>(), n, db.as<int64_t>(), n_seq, nullptr));
    KMAP_CHECK_HIP(hipMemcpy(hash, dh.p, (size_t)n * sizeof(H), hipMemcpyDeviceToHost));
    return KMAP_OK;
}
}  // namespace

extern "C" {

int kmap_hash_kmers_u32_dev(const uint8_t *seq_dev, int64_t n, int k, uint32_t *out_dev, void *stream) {
    KMAP_REQUIRE(k < 16, "hash_kmers_u32: k=%d needs the u64 entry point (kmer_count.py:359-365)", k);
    return hash_launch<uint32_t>(seq_dev, n, k, out_dev, stream);
}
int kmap_hash_kmers_u64_dev(const uint8_t *seq_dev, int64_t n, int k, uint64_t *out_dev, void *stream) {
    return hash_launch<uint64_t>(seq_dev, n, k, out_dev, stream);
}
int kmap_hash_kmers_u32(const uint8_t *seq, int64_t n, int k, uint32_t *out) {
    KMAP_REQUIRE(n == 0 || (seq && out), "hash_kmers_u32: null pointer");
    return host_map(seq, n, out, n, [&](const uint8_t *d, uint32_t *o) { return kmap_hash_kmers_u32_dev(d, n, k, o, nullptr); });
}
int kmap_hash_kmers_u64(const uint8_t *seq, int64_t n, int k, uint64_t *out) {
    KMAP_REQUIRE(n == 0 || (seq && out), "hash_kmers_u64: null pointer");
    return host_map(seq, n, out, n, [&](const uint8_t *d, uint64_t *o) { return kmap_hash_kmers_u64_dev(d, n, k, o, nullptr); });
}

int kmap_dedupe_per_read_u32_dev(uint32_t *hash_dev, int64_t n, const int64_t *borders_dev, int64_t n_seq, void *stream) {
    return dedupe_launch<uint32_t>(hash_dev, n, borders_dev, n_seq, stream);
}
int kmap_dedupe_per_read_u64_dev(uint64_t *hash_dev, int64_t n, const int64_t *borders_dev, int64_t n_seq, void *stream) {
    return dedupe_launch<uint64_t>(hash_dev, n, borders_dev, n_seq, stream);
}
int kmap_dedupe_per_read_u32(uint32_t *hash, int64_t n, const int64_t *borders, int64_t n_seq) {
    return dedupe_host<uint32_t>(hash, n, borders, n_seq);
}
int kmap_dedupe_per_read_u64(uint64_t *hash, int64_t n, const int64_t *borders, int64_t n_seq) {
    return dedupe_host<uint64_t>(hash, n, borders, n_seq);
}

int kmap_revcom_u32_dev(const uint32_t *in_dev, int64_t n, int k, uint32_t *out_dev, void *stream) {
    KMAP_REQUIRE(k < 16, "revcom_u32: k=%d needs the u64 entry point", k);
    return revcom_launch<uint32_t>(in_dev, n, k, out_dev, stream);
}
int kmap_revcom_u64_dev(const uint64_t *in_dev, int64_t n, int k, uint64_t *out_dev, void *stream) {
    return revcom_launch<uint64_t>(in_dev, n, k, out_dev, stream);
}
int kmap_revcom_u32(const uint32_t *in, int64_t n, int k, uint32_t *out) {
    KMAP_REQUIRE(n <= 0 || (in && out), "revcom_u32: null pointer");
    if (n <= 0) return KMAP_OK;
    return host_map(in, n, out, n, [&](const uint32_t *d, uint32_t *o) { return kmap_revcom_u32_dev(d, n, k, o, nullptr); });
}
int kmap_revcom_u64(const uint64_t *in, int64_t n, int k, uint64_t *out) {
    KMAP_REQUIRE(n <= 0 || (in && out), "revcom_u64: null pointer");
    if (n <= 0) return KMAP_OK;
    return host_map(in, n, out, n, [&](const uint64_t *d, uint64_t *o) { return kmap_revcom_u64_dev(d, n, k, o, nullptr); });
}

int kmap_hamdist_1vN_u32_dev(const uint32_t *h_dev, int64_t n, uint32_t cons, int shift_bits, int clen, uint8_t *out_dev,
                             void *stream) {
    return ham_launch<uint32_t>(h_dev, n, cons, shift_bits, clen, out_dev, stream);
}
int kmap_hamdist_1vN_u64_dev(const uint64_t *h_dev, int64_t n, uint64_t cons, int shift_bits, int clen, uint8_t *out_dev,
                             void *stream) {
    return ham_launch<uint64_t>(h_dev, n, cons, shift_bits, clen, out_dev, stream);
}
int kmap_hamdist_1vN_u32(const uint32_t *h, int64_t n, uint32_t cons, int shift_bits, int clen, uint8_t *out) {
    KMAP_REQUIRE(n <= 0 || (h && out), "hamdist_1vN_u32: null pointer");
    if (n <= 0) return KMAP_OK;
    return host_map(h, n, out, n, [&](const uint32_t *d, uint8_t *o) {
        return kmap_hamdist_1vN_u32_dev(d, n, cons, shift_bits, clen, o, nullptr);
    });
}
int kmap_hamdist_1vN_u64(const uint64_t *h, int64_t n, uint64_t cons, int shift_bits, int clen, uint8_t *out) {
    KMAP_REQUIRE(n <= 0 || (h && out), "hamdist_1vN_u64: null pointer");
    if (n <= 0) return KMAP_OK;
    return host_map(h, n, out, n, [&](const uint64_t *d, uint8_t *o) {
        return kmap_hamdist_1vN_u64_dev(d, n, cons, shift_bits, clen, o, nullptr);
    });
}

int kmap_mask_hamball_dev(uint8_t *seq_dev, int64_t n, int k, const uint64_t *cons, const int32_t *radius, int n_cons,
                          void *stream) {
    return kmap_mask_launch(seq_dev, n, k, cons, radius, n_cons, as_stream(stream));
}
int kmap_mask_hamball(uint8_t *seq, int64_t n, int k, const uint64_t *cons, const int32_t *radius, int n_cons) {
    KMAP_REQUIRE(n <= 0 || seq, "mask_hamball: null pointer");
    if (n <= 0) return KMAP_OK;
    DevBuf d;
    KMAP_TRY(d.alloc((size_t)n));
    KMAP_CHECK_HIP(hipMemcpy(d.p, seq, (size_t)n, hipMemcpyHostToDevice));
    KMAP_TRY(kmap_mask_launch(d.as<uint8_t>(), n, k, cons, radius, n_cons, nullptr));
    KMAP_CHECK_HIP(hipStreamSynchronize(nullptr));
    KMAP_CHECK_HIP(hipMemcpy(seq, d.p, (size_t)n, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

}  // extern "C"
