// counts.hip -- k-mer counting on device: count_uniq_hash (kmer_count.py:476-491) fused with
// remove_duplicate_hash_per_seq (:743-760) and merge_revcom (:643-685), the Hamming-ball mass
// of find_motif (motif_discovery.py:666-673) and the per-read motif occurrence scan
// (motif_discovery.py:1422-1477).
//
// Counting uses a direct-address histogram of 4^k uint32 bins in HBM for k <= 16 (MI355X has
// 288 GB: even the 16 GiB table of k = 16 is resident), followed by an order-preserving
// compaction that applies the reverse-complement merge on the fly.  A uint32 bin wraps exactly
// like the reference's int64 -> int32 cast of np.unique counts.
#include <errno.h>
#include <string.h>
#include <unistd.h>

#include <type_traits>
#include <algorithm>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "common.h"
#include "counts_internal.h"
#include "scan_util.h"

int kmap_hash_launch_u32(const uint8_t *seq, int64_t n, int k, uint32_t *out, void *stream);
int kmap_hash_launch_u64(const uint8_t *seq, int64_t n, int k, uint64_t *out, void *stream);

namespace {

constexpr int BLK = 256;
static inline unsigned grid_for(int64_t n, int64_t per_block) {
    int64_t g = (n + per_block - 1) / per_block;
    return (unsigned)(g < 1 ? 1 : g);
}

// ---- histogram ----------------------------------------------------------------------------------
template <typename H>
__global__ __launch_bounds__(BLK) void hist_kernel(const H *__restrict__ h, int64_t n, uint32_t *__restrict__ bins) {
    const int64_t stride = (int64_t)gridDim.x * BLK;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
        const H v = h[i];
        if (v != (H)~(H)0) atomicAdd(&bins[(uint64_t)v], 1u);
    }
}

// LDS-privatised histogram for small tables: a block owns HL_BINS consecutive bins of the pass's range in LDS,
// streams the whole hash array (grid-stride, 16-byte loads) and counts only hashes of that range with LDS atomics;
// one flush of the non-zero LDS bins per block at the end.  4^k / HL_BINS passes re-read the hash array (4 B per
// position per pass), which beats ~20 G scattered device-scope atomics per second up to ~8 passes (k <= 9).
constexpr int HL_BINS = 32768;      // 128 KiB of uint32 per block (one block per CU)
constexpr int HL_TPB = 1024;
__global__ __launch_bounds__(HL_TPB) void hist_lds_kernel(const uint32_t *__restrict__ h, int64_t n, uint32_t bin0,
                                                          uint32_t *__restrict__ bins) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lb[];
    for (int b = threadIdx.x; b < HL_BINS; b += HL_TPB) lb[b] = 0;
    __syncthreads();
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    const int64_t n4 = n / 4;
    const int64_t stride = (int64_t)gridDim.x * HL_TPB;
    const u32x4v *h4 = reinterpret_cast<const u32x4v *>(h);
    for (int64_t i = (int64_t)blockIdx.x * HL_TPB + threadIdx.x; i < n4; i += stride) {
        const u32x4v v = h4[i];
        const uint32_t a = v.x - bin0, b = v.y - bin0, c = v.z - bin0, d = v.w - bin0;   // invalid (all ones) never lands in range
        if (a < (uint32_t)HL_BINS) atomicAdd(&lb[a], 1u);
        if (b < (uint32_t)HL_BINS) atomicAdd(&lb[b], 1u);
        if (c < (uint32_t)HL_BINS) atomicAdd(&lb[c], 1u);
        if (d < (uint32_t)HL_BINS) atomicAdd(&lb[d], 1u);
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += HL_TPB) {
            const uint32_t a = h[i] - bin0;
            if (a < (uint32_t)HL_BINS) atomicAdd(&lb[a], 1u);
        }
    __syncthreads();
    for (int b = threadIdx.x; b < HL_BINS; b += HL_TPB) {
        const uint32_t c = lb[b];
        if (c) atomicAdd(&bins[bin0 + b], c);
    }
}

// ---- order-preserving compaction with optional revcom merge ------------------------------------
constexpr int CT_PER_THREAD = 8;
constexpr int CT_TILE = BLK * CT_PER_THREAD;   // bins per block

__device__ __forceinline__ uint64_t rc_bits(uint64_t x, int k) { return revcom_hash(x, k); }

// keep/emit decision for bin x (see merge_revcom, kmer_count.py:643-685): returns true if an
// entry is emitted; key/cnt are the emitted values.
// merge 3 (key-space-sharded counting, counts_internal.h): `bins` is a rank's range-mode table -- T1 = bins[0, len) holds the counts of
// the positions lo .. lo + len, T2 = bins[half, half + len) the counts of their reverse complements that lie OUTSIDE the range (a
// partner inside the range was counted into T1 at its own position).
__device__ __forceinline__ bool bin_entry(const uint32_t *__restrict__ bins, uint64_t x, uint32_t c, int k, int merge, uint64_t &key,
                                          uint32_t &cnt, const kmap_key_range &kr = kmap_key_range{}) {
    if (c == 0) return false;
    key = x;
    cnt = c;
    if (!merge) return true;
    const uint64_t r = rc_bits(x, k);
    if (merge == 3) {
        if (r == x) {
            cnt = c + c;
            return true;
        }
        const uint64_t ro = r - kr.lo;
        const uint32_t cr = ro < (uint64_t)kr.len ? bins[ro] : bins[(uint64_t)kr.half + (x - kr.lo)];
        if (cr > 0 && x > r) return false;
        key = (x > r) ? r : x;
        cnt = c + cr;
        return true;
    }
    if (merge == 2) {   // table already merged in place by rc_merge_tiles_kernel: a surviving x > rc(x) had no partner
        key = (x > r) ? r : x;
        return true;
    }
    if (r == x) {   // palindrome: its own partner, the reference adds the count to itself
        cnt = c + c;
        return true;
    }
    const uint32_t cr = bins[r];
    if (cr > 0 && x > r) return false;   // higher member of a present pair is deleted
    key = (x > r) ? r : x;               // partner absent and x > rc(x): replaced in place, not re-sorted
    cnt = c + cr;
    return true;
}

// ---- reverse-complement merge of the whole table as a tiled transpose (k >= 11) ------------------------------------------
// bin_entry's partner lookup is one random 4-byte read per non-empty bin (22 G/s: 6.7 + 7.8 ms per compaction at k = 14).
// Split x into (a, m, b) with a / b the top / bottom three bases: rc(x) = (rc3(b), rc(m), rc3(a)), so the 64 x 64 entries
// that share m pair up with the 64 x 64 entries that share rc(m), transposed.  One block loads both tiles (64 rows of 256
// contiguous bytes each), merges them through LDS and writes both back in place: bins[x] becomes the merged count of a
// kept entry, 0 for a deleted or empty one.  The compaction then runs without any gather (merge mode 2).
__device__ __forceinline__ uint32_t rc3(uint32_t v) {   // reverse complement of a 3-base group (6 bits)
    v = 63u - v;
    return ((v & 3u) << 4) | (v & 12u) | (v >> 4);
}
// PRES (key-range-sharded multi-GPU counting): `bins` holds ONE rank's counts and `nib` says, one nibble per bin (bin x: byte x / 2,
// low nibble for even x), on how many ranks the bin is non-empty.  Which member of a pair survives is decided by that global
// presence; the value written is the local part of the merged count (own + partner, both local), so that the SUM over the ranks of
// the merged tables is the merged table of the summed counts: bins[x] = survives(x) ? c(x) + c(rc x) : 0.
// tile_counts (optional): the compaction's per-tile survivor counts (one per CT_TILE bins), accumulated here -- a 64-bin row segment
// lies inside one tile -- so that the compaction's counting pass over the table is not needed (0.2 ms at k = 14, 3 ms at k = 16).
template <bool PRES>
__global__ __launch_bounds__(BLK) void rc_merge_tiles_kernel(uint32_t *__restrict__ bins, int k, const uint8_t *__restrict__ nib,
                                                             uint32_t *__restrict__ tile_counts) {
    __shared__ uint32_t A[64][65], B[64][65];
    __shared__ uint8_t PA[PRES ? 64 : 1][68], PB[PRES ? 64 : 1][68];
    const int mg = k - 6;                                     // middle groups (k >= 7)
    const uint64_t m = blockIdx.x;
    const uint64_t m2 = revcom_hash(m, mg);
    if (m > m2) return;                                      // the pair is handled by the block of the smaller middle
    const bool self = (m == m2);
    const uint64_t row_stride = (uint64_t)1 << (2 * (k - 3));
    // 256 threads: 16 lanes x 16 B cover one 256-byte row; 16 rows per step
    const int lr = threadIdx.x >> 4, lc = (threadIdx.x & 15) * 4;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int a = s * 16 + lr;
        const u32x4 va = *reinterpret_cast<const u32x4 *>(bins + (uint64_t)a * row_stride + (m << 6) + lc);
        A[a][lc] = va.x; A[a][lc + 1] = va.y; A[a][lc + 2] = va.z; A[a][lc + 3] = va.w;
        if (PRES) {
            const uint32_t pa = *reinterpret_cast<const uint16_t *>(nib + (((uint64_t)a * row_stride + (m << 6) + lc) >> 1));
            PA[a][lc] = pa & 15u; PA[a][lc + 1] = (pa >> 4) & 15u; PA[a][lc + 2] = (pa >> 8) & 15u; PA[a][lc + 3] = pa >> 12;
        }
        if (!self) {
            const u32x4 vb = *reinterpret_cast<const u32x4 *>(bins + (uint64_t)a * row_stride + (m2 << 6) + lc);
            B[a][lc] = vb.x; B[a][lc + 1] = vb.y; B[a][lc + 2] = vb.z; B[a][lc + 3] = vb.w;
            if (PRES) {
                const uint32_t pb = *reinterpret_cast<const uint16_t *>(nib + (((uint64_t)a * row_stride + (m2 << 6) + lc) >> 1));
                PB[a][lc] = pb & 15u; PB[a][lc + 1] = (pb >> 4) & 15u; PB[a][lc + 2] = (pb >> 8) & 15u; PB[a][lc + 3] = pb >> 12;
            }
        }
    }
    __syncthreads();
    // entry (a, mm, b) against its partner (rc3(b), mo, rc3(a)): compare the tuples lexicographically
    auto merged = [&](uint32_t (&own)[64][65], uint32_t (&oth)[64][65], uint8_t (&pown)[PRES ? 64 : 1][68], uint8_t (&poth)[PRES ? 64 : 1][68],
                      uint64_t mm, uint64_t mo, int a, int b) -> uint32_t {
        const uint32_t c = own[a][b];
        if (PRES ? pown[a][b] == 0 : c == 0) return 0u;
        const uint32_t pa = rc3((uint32_t)b), pb = rc3((uint32_t)a);
        const bool eq = ((uint32_t)a == pa) && (mm == mo) && ((uint32_t)b == pb);
        if (eq) return c + c;                                // palindrome: its own partner
        const bool greater = ((uint32_t)a != pa) ? ((uint32_t)a > pa) : (mm != mo) ? (mm > mo) : ((uint32_t)b > pb);
        const uint32_t cr = oth[pa][pb];
        const bool partner = PRES ? poth[pa][pb] != 0 : cr > 0;
        return (partner && greater) ? 0u : c + cr;           // higher member of a present pair is deleted
    };
    uint32_t ra[4][4], rb[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a = s * 16 + lr, b = lc + j;
            ra[s][j] = self ? merged(A, A, PA, PA, m, m, a, b) : merged(A, B, PA, PB, m, m2, a, b);
            rb[s][j] = self ? 0u : merged(B, A, PB, PA, m2, m, a, b);
        }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int a = s * 16 + lr;
        *reinterpret_cast<u32x4 *>(bins + (uint64_t)a * row_stride + (m << 6) + lc) = u32x4{ra[s][0], ra[s][1], ra[s][2], ra[s][3]};
        if (!self)
            *reinterpret_cast<u32x4 *>(bins + (uint64_t)a * row_stride + (m2 << 6) + lc) = u32x4{rb[s][0], rb[s][1], rb[s][2], rb[s][3]};
        if (tile_counts) {                                   // survivors of the row's 64 bins: 16 lanes x 4 bins
            uint32_t na = (ra[s][0] != 0) + (ra[s][1] != 0) + (ra[s][2] != 0) + (ra[s][3] != 0);
            uint32_t nb = (rb[s][0] != 0) + (rb[s][1] != 0) + (rb[s][2] != 0) + (rb[s][3] != 0);
            uint32_t both = na | (nb << 16);
            for (int o = 8; o > 0; o >>= 1) both += __shfl_xor(both, o);
            if ((threadIdx.x & 15) == 0) {
                if (both & 0xFFFFu) atomicAdd(&tile_counts[((uint64_t)a * row_stride + (m << 6)) / CT_TILE], both & 0xFFFFu);
                if (both >> 16) atomicAdd(&tile_counts[((uint64_t)a * row_stride + (m2 << 6)) / CT_TILE], both >> 16);
            }
        }
    }
}
// presence nibbles of the table: thread = eight bins -> four bytes
__global__ __launch_bounds__(BLK) void presence_nibbles_kernel(const uint32_t *__restrict__ bins, uint64_t n_bins, uint32_t *__restrict__ nib) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t t = (uint64_t)blockIdx.x * BLK + threadIdx.x;
    if (t * 8 >= n_bins) return;                              // n_bins = 4^k, k >= 2: a multiple of 8
    const u32x4 a = *reinterpret_cast<const u32x4 *>(bins + t * 8), b = *reinterpret_cast<const u32x4 *>(bins + t * 8 + 4);
    nib[t] = (uint32_t)(a.x != 0) | ((uint32_t)(a.y != 0) << 4) | ((uint32_t)(a.z != 0) << 8) | ((uint32_t)(a.w != 0) << 12) |
             ((uint32_t)(b.x != 0) << 16) | ((uint32_t)(b.y != 0) << 20) | ((uint32_t)(b.z != 0) << 24) | ((uint32_t)(b.w != 0) << 28);
}

// the CT_PER_THREAD consecutive bins of a thread: 16-byte loads (element loads through the `x < n_bins` guards were eight
// 4-byte loads per thread at a 32-byte lane stride: every line fetched by eight instructions -- 2 TB/s on a pure streaming read)
__device__ __forceinline__ void load_bins(const uint32_t *__restrict__ bins, uint64_t x0, uint64_t n_bins, uint32_t (&c)[CT_PER_THREAD]) {
    static_assert(CT_PER_THREAD == 8, "two 16-byte loads");
    if (x0 + CT_PER_THREAD <= n_bins) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 a = *reinterpret_cast<const u32x4 *>(bins + x0), b = *reinterpret_cast<const u32x4 *>(bins + x0 + 4);
        c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w;
        c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < CT_PER_THREAD; ++j) c[j] = (x0 + j < n_bins) ? bins[x0 + j] : 0u;
    }
}
// A block walks CT_TPB consecutive tiles (the per-tile counts / offsets keep their meaning): with one 8-KiB tile per block the
// 16-GiB table of k = 16 is 2 M blocks and the pass ran at the dispatch rate (2 TB/s), not at the memory's.
constexpr int CT_TPB = 8;
// x_base: `bins` is the slice [x_base, x_base + n_bins) of the table (key-range-sharded counting; merge 0 or 2 only: no partner gathers)
__global__ __launch_bounds__(BLK) void compact_count_kernel(const uint32_t *__restrict__ bins, uint64_t n_bins, int k,
                                                            int merge, uint32_t *__restrict__ block_counts, unsigned n_tiles, uint64_t x_base,
                                                            kmap_key_range kr) {
    __shared__ uint32_t wsum[CT_TPB][BLK / 64];
    uint32_t m[CT_TPB];
#pragma unroll
    for (int t = 0; t < CT_TPB; ++t) {
        const uint64_t tile = (uint64_t)blockIdx.x * CT_TPB + t;
        const uint64_t x0 = (tile * BLK + threadIdx.x) * CT_PER_THREAD;
        m[t] = 0;
        uint32_t c8[CT_PER_THREAD];
        load_bins(bins, x0, n_bins, c8);
#pragma unroll
        for (int j = 0; j < CT_PER_THREAD; ++j) {
            uint64_t key;
            uint32_t cnt;
            m[t] += bin_entry(bins, x_base + x0 + j, c8[j], k, merge, key, cnt, kr);  // bins past the end were loaded as 0
        }
    }
#pragma unroll
    for (int t = 0; t < CT_TPB; ++t) {
        for (int o = 32; o > 0; o >>= 1) m[t] += __shfl_down(m[t], o);
        if ((threadIdx.x & 63) == 0) wsum[t][threadIdx.x >> 6] = m[t];
    }
    __syncthreads();
    if (threadIdx.x < CT_TPB) {
        const unsigned tile = blockIdx.x * CT_TPB + threadIdx.x;
        if (tile < n_tiles) block_counts[tile] = wsum[threadIdx.x][0] + wsum[threadIdx.x][1] + wsum[threadIdx.x][2] + wsum[threadIdx.x][3];
    }
}

template <typename H>
__global__ __launch_bounds__(BLK) void compact_write_kernel(const uint32_t *__restrict__ bins, uint64_t n_bins, int k,
                                                            int merge, const uint64_t *__restrict__ block_off,
                                                            H *__restrict__ uniq, uint32_t *__restrict__ cnt_out, uint64_t x_base,
                                                            kmap_key_range kr) {
    __shared__ uint32_t wsum[BLK / 64];
    __shared__ H skey[CT_TILE];
    __shared__ uint32_t scnt[CT_TILE];
    static_assert(BLK / 64 == 4, "n_out below adds four wave sums");
  for (int t = 0; t < CT_TPB; ++t) {                                      // the block's tiles, one after the other
    const uint64_t tile = (uint64_t)blockIdx.x * CT_TPB + t;
    if (tile * CT_TILE >= n_bins) break;                                  // block-uniform
    const uint64_t x0 = (tile * BLK + threadIdx.x) * CT_PER_THREAD;
    uint64_t keys[CT_PER_THREAD];
    uint32_t cnts[CT_PER_THREAD];
    uint32_t flags = 0, m = 0;
    uint32_t c8[CT_PER_THREAD];
    load_bins(bins, x0, n_bins, c8);
#pragma unroll
    for (int j = 0; j < CT_PER_THREAD; ++j) {
        if (bin_entry(bins, x_base + x0 + j, c8[j], k, merge, keys[j], cnts[j], kr)) {
            flags |= 1u << j;
            ++m;
        }
    }
    // exclusive scan of m across the block: wave scan + wave sums
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = m;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    // The tile's entries meet in LDS at their rank inside the tile and leave as contiguous rows.  (Each thread storing its own up to
    // eight entries wrote 4 bytes per lane at a stride of ~4 entries: 16 partly used store instructions per tile; k = 14: 0.63 ms.)
    const uint32_t local = woff + (inc - m);                              // rank of the thread's first entry inside the tile
    uint32_t at = local;
#pragma unroll
    for (int j = 0; j < CT_PER_THREAD; ++j) {
        if (flags & (1u << j)) {
            skey[at] = (H)keys[j];
            scnt[at] = cnts[j];
            ++at;
        }
    }
    __syncthreads();
    const uint32_t n_out = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const uint64_t base = block_off[tile];
    for (uint32_t i = threadIdx.x; i < n_out; i += BLK) {
        uniq[base + i] = skey[i];
        cnt_out[base + i] = scnt[i];
    }
    __syncthreads();                                                      // wsum / the staging rows are re-used by the next tile
  }
}

__global__ __launch_bounds__(BLK) void sum_counts_kernel(const uint32_t *__restrict__ cnt, int64_t n, int as_signed,
                                                         unsigned long long *__restrict__ total) {
    // the reference sums int32 counts as Python ints (find_motif :648): sign-extend for k < 16
    long long s = 0;
    const int64_t stride = (int64_t)gridDim.x * BLK;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += stride)
        s += as_signed ? (long long)(int32_t)cnt[i] : (long long)cnt[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(total, (unsigned long long)s);
}

// ---- top-k by count ---------------------------------------------------------------------------
// key = (count << 32) | ~index : the maximum key is the largest count, lowest index.  Every thread keeps its own top
// TK of a grid-strided slice, the block merges them by TK rounds of a block-wide max, the host merges the blocks.
constexpr int TK = 16;
__global__ __launch_bounds__(BLK) void topk_kernel(const uint32_t *__restrict__ cnt, int64_t n, int as_signed, int top_k,
                                                   unsigned long long *__restrict__ out) {
    __shared__ unsigned long long red[BLK];
    unsigned long long best[TK];
#pragma unroll
    for (int t = 0; t < TK; ++t) best[t] = 0;
    const int64_t stride = (int64_t)gridDim.x * BLK;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
        const long long cv = as_signed ? (long long)(int32_t)cnt[i] : (long long)cnt[i];
        if (cv <= 0) continue;
        unsigned long long key = ((unsigned long long)cv << 32) | (0xFFFFFFFFull - (unsigned long long)i);
        if (key > best[top_k - 1]) {   // insertion into the descending list
#pragma unroll
            for (int t = 0; t < TK; ++t) {
                if (t < top_k && key > best[t]) {
                    const unsigned long long tmp = best[t];
                    best[t] = key;
                    key = tmp;
                }
            }
        }
    }
    int head = 0;
    for (int round = 0; round < top_k; ++round) {
        unsigned long long mine = 0;
#pragma unroll
        for (int t = 0; t < TK; ++t)
            if (t == head) mine = best[t];
        red[threadIdx.x] = (head < top_k) ? mine : 0;
        __syncthreads();
        for (int o = BLK / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o && red[threadIdx.x + o] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + o];
            __syncthreads();
        }
        const unsigned long long win = red[0];
        __syncthreads();
        if (threadIdx.x == 0) out[(size_t)blockIdx.x * top_k + round] = win;
        if (win != 0 && mine == win) ++head;   // keys are unique (they embed the index)
    }
}

// ---- Hamming-ball mass -----------------------------------------------------------------------
struct CandTab {
    uint64_t fwd[16];
    uint64_t rc[16];
    int n;
};
template <typename H>
__global__ __launch_bounds__(BLK) void mass_kernel(const H *__restrict__ uniq, const uint32_t *__restrict__ cnt, int64_t n,
                                                   int k, CandTab t, int radius, int revcom, int as_signed,
                                                   unsigned long long *__restrict__ mass) {
    long long acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0;
    const uint64_t m = low_mask<uint64_t>(k);
    const int64_t stride = (int64_t)gridDim.x * BLK;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
        const uint64_t u = (uint64_t)uniq[i];
        const long long w = as_signed ? (long long)(int32_t)cnt[i] : (long long)cnt[i];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < t.n) {
                int d = popc2((u ^ t.fwd[c]) & m);
                if (revcom) {
                    int d2 = popc2((u ^ t.rc[c]) & m);
                    d = d2 < d ? d2 : d;
                }
                if (d <= radius) acc[c] += w;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c < t.n) {   // wave-uniform
            long long s = acc[c];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
            if ((threadIdx.x & 63) == 0 && s != 0) atomicAdd(&mass[c], (unsigned long long)s);
        }
    }
}

static uint64_t host_revcom(uint64_t h, int k, int narrow) {
    // same arithmetic as the device helper (u32 wrap for k < 16)
    if (narrow) {
        uint32_t mask = (uint32_t)((1ull << (2 * k)) - 1), com = mask - (uint32_t)h, r = com & 3u;
        for (int i = 0; i < k - 1; ++i) { r <<= 2; com >>= 2; r += com & 3u; }
        return r;
    }
    uint64_t mask = (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1), com = mask - h, r = com & 3u;
    for (int i = 0; i < k - 1; ++i) { r <<= 2; com >>= 2; r += com & 3u; }
    return r;
}

}  // namespace

#include <map>
#include <mutex>

#include "counts_internal.h"
#include "host_pool.h"

namespace {

// The 4^k-bin histogram is transient (one count call, or hist -> all-reduce -> finish in the sharded flow), so all handles of a
// device share ONE table from the scratch arena instead of owning one each: a fresh handle per k and per find_motif round paid
// ~1 s of hipMalloc / hipFree for the 16-GiB table at k = 16.  (Counting is single-threaded per device, like the arena.)
// Ownership: every (re)fill takes a new generation number of the device's table and stamps the handle with it.  The split API
// hist -> kmap_counts_bins -> (caller's all-reduce) -> kmap_counts_finish checks the stamp, so a count through another handle in
// between (which overwrites the table, or regrows the arena and frees it) is an error instead of a silently wrong compaction.
std::mutex g_bins_mu;
std::map<int, uint64_t> g_bins_gen;     // device -> generation of the table's current contents

int ensure_bins(kmap_counts *c, size_t n_bins) {
    void *p = nullptr;
    const int rc = kmap_scratch(&p, n_bins * 4, (hipStream_t) nullptr, KMAP_SLOT_BINS);
    if (rc != KMAP_OK) {
        kmap_set_error("counts: cannot allocate %zu-bin histogram (%.1f GiB)", n_bins, n_bins * 4.0 / (1 << 30));
        return rc;
    }
    int dev = 0;
    KMAP_CHECK_HIP(hipGetDevice(&dev));
    c->bins = (uint32_t *)p;
    c->bins_cap = n_bins;
    c->bins_dev = dev;
    std::lock_guard<std::mutex> lock(g_bins_mu);
    c->bins_gen = ++g_bins_gen[dev];
    return KMAP_OK;
}

}  // namespace

void kmap_counts_bins_invalidate(int dev) {
    std::lock_guard<std::mutex> lock(g_bins_mu);
    ++g_bins_gen[dev];
}

int kmap_counts_bins_check(const kmap_counts *c, const char *who) {
    std::lock_guard<std::mutex> lock(g_bins_mu);
    auto it = g_bins_gen.find(c->bins_dev);
    if (!c->bins || it == g_bins_gen.end() || it->second != c->bins_gen) {
        kmap_set_error("%s: the device's shared histogram table no longer holds this handle's counts (another handle counted, or "
                       "the scratch arena was released, since kmap_counts_hist_packed_dev)", who);
        return KMAP_E_STATE;
    }
    return KMAP_OK;
}

int kmap_counts_reserve_bins(kmap_counts *c, int k) {
    KMAP_REQUIRE(k > 0 && k <= 16, "counts: direct histogram needs k <= 16 (k=%d)", k);
    return ensure_bins(c, (size_t)1 << (2 * k));
}

int kmap_counts_prepare_bins(kmap_counts *c, int k, hipStream_t st) {
    KMAP_REQUIRE(k > 0 && k <= 16, "counts: direct histogram needs k <= 16 (k=%d)", k);
    const size_t n_bins = (size_t)1 << (2 * k);
    KMAP_TRY(ensure_bins(c, n_bins));
    KMAP_CHECK_HIP(hipMemsetAsync(c->bins, 0, n_bins * 4, st));
    return KMAP_OK;
}

// order-preserving compaction of the bins [first, first + n_bins) of the table into the handle's uniq/cnt arrays; merge: 0, 1 (partner
// gathers over the WHOLE table: first must be 0), 2 (table merged in place beforehand)
static int compact_range(kmap_counts *c, int k, int merge, uint64_t first, uint64_t n_bins, int64_t *n_uniq, hipStream_t st,
                         bool merge_tiles_first = false, kmap_key_range kr = kmap_key_range{}) {
    const unsigned nb = grid_for((int64_t)n_bins, CT_TILE);
    uint32_t *bc = nullptr;
    uint64_t *boff = nullptr;
    KMAP_TRY(kmap_scratch((void **)&bc, (size_t)nb * 4, st, KMAP_SLOT_A));
    KMAP_TRY(kmap_scratch((void **)&boff, ((size_t)nb + 1) * 8, st, KMAP_SLOT_B));
    const uint32_t *bins = kr.len ? c->bins : c->bins + first;          // a range-mode table (kr) starts at the range's first position
    const unsigned nblk = (nb + CT_TPB - 1) / CT_TPB;
    if (merge_tiles_first) {   // whole table, k >= 11: merge it in place; the merge counts the survivors per compaction tile as it goes
        KMAP_CHECK_HIP(hipMemsetAsync(bc, 0, (size_t)nb * 4, st));
        rc_merge_tiles_kernel<false><<<(unsigned)((size_t)1 << (2 * (k - 6))), BLK, 0, st>>>(c->bins, k, nullptr, bc);
    } else {
        compact_count_kernel<<<nblk, BLK, 0, st>>>(bins, n_bins, k, merge, bc, nb, first, kr);
    }
    KMAP_TRY(exclusive_scan_u32(bc, nb, boff, st));
    uint64_t total = 0;
    KMAP_CHECK_HIP(hipMemcpyAsync(&total, boff + nb, 8, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));
    if (c->cap < total || !c->uniq) {
        if (c->uniq) KMAP_CHECK_HIP(hipFree(c->uniq));
        if (c->cnt) KMAP_CHECK_HIP(hipFree(c->cnt));
        c->uniq = nullptr;
        c->cnt = nullptr;
        c->cap = 0;
        const size_t cap = total ? total : 1;
        KMAP_CHECK_HIP(hipMalloc(&c->uniq, cap * 8));
        KMAP_CHECK_HIP(hipMalloc((void **)&c->cnt, cap * 4));
        c->cap = cap;
    }
    if (total) {
        if (k < 16) compact_write_kernel<uint32_t><<<nblk, BLK, 0, st>>>(bins, n_bins, k, merge, boff, (uint32_t *)c->uniq, c->cnt, first, kr);
        else compact_write_kernel<uint64_t><<<nblk, BLK, 0, st>>>(bins, n_bins, k, merge, boff, (uint64_t *)c->uniq, c->cnt, first, kr);
    }
    KMAP_CHECK_HIP(hipGetLastError());
    c->k = k;
    c->narrow = (k < 16);
    c->n_uniq = (int64_t)total;
    if (n_uniq) *n_uniq = (int64_t)total;
    return KMAP_OK;
}

// order-preserving compaction of the filled histogram (+ revcom merge) into the handle's uniq/cnt arrays
int kmap_counts_finish_hist(kmap_counts *c, int k, int merge, int64_t *n_uniq, hipStream_t st) {
    if (merge == 1 && k >= 11)                 // merge the table in place first; the compaction then needs no partner gathers
        return compact_range(c, k, 2, 0, (uint64_t)1 << (2 * k), n_uniq, st, true);
    return compact_range(c, k, merge, 0, (uint64_t)1 << (2 * k), n_uniq, st);
}

// compaction of a rank's range-mode table (counts_internal.h: kmap_key_range): positions lo .. lo + len in key order
int kmap_counts_finish_key_range(kmap_counts *c, int k, kmap_key_range r, int64_t *n_uniq, hipStream_t st) {
    return compact_range(c, k, r.half ? 3 : 0, r.lo, r.len, n_uniq, st, false, r);
}
// the whole table is in c->bins: merge it in place (k >= 11), then compact positions [first, first + n_bins)
int kmap_counts_finish_hist_slice(kmap_counts *c, int k, int merge, uint64_t first, uint64_t n_bins, int64_t *n_uniq, hipStream_t st) {
    if (merge) {
        KMAP_REQUIRE(k >= 11, "counts: a merged slice needs k >= 11 (k=%d)", k);
        rc_merge_tiles_kernel<false><<<(unsigned)((size_t)1 << (2 * (k - 6))), BLK, 0, st>>>(c->bins, k, nullptr, nullptr);
    }
    return compact_range(c, k, merge ? 2 : 0, first, n_bins, n_uniq, st);
}

namespace {

template <typename H>
int hist_hashes(kmap_counts *c, const H *hash_dev, int64_t n, int k, hipStream_t st) {
    const size_t n_bins = (size_t)1 << (2 * k);
    if (sizeof(H) == 4 && kmap_counts_part_applies(k, n))
        return kmap_counts_part_hist_u32(c, (const uint32_t *)hash_dev, n, k, st);
    KMAP_TRY(kmap_counts_prepare_bins(c, k, st));
    if (n > 0) {
        const size_t passes = (n_bins + HL_BINS - 1) / HL_BINS;
        if (sizeof(H) == 4 && passes <= 32 && n >= (1 << 20) && ((uintptr_t)hash_dev % 16) == 0) {
            KMAP_TRY(kmap_allow_lds((const void *)hist_lds_kernel, HL_BINS * 4));
            for (size_t p = 0; p < passes; ++p)
                hist_lds_kernel<<<256, HL_TPB, HL_BINS * 4, st>>>((const uint32_t *)hash_dev, n, (uint32_t)(p * HL_BINS),
                                                                 c->bins);
        } else {
            int64_t g = (n + BLK - 1) / BLK;
            if (g > 256 * 32) g = 256 * 32;
            hist_kernel<H><<<(unsigned)g, BLK, 0, st>>>(hash_dev, n, c->bins);
        }
        KMAP_CHECK_HIP(hipGetLastError());
    }
    return KMAP_OK;
}

template <typename H>
int run_hashes(kmap_counts *c, const H *hash_dev, int64_t n, int k, int merge, int64_t *n_uniq, hipStream_t st) {
    if (k > 16) {
        if constexpr (sizeof(H) == 8) return kmap_counts_sort_path(c, (const uint64_t *)hash_dev, n, k, merge, n_uniq, st);
        KMAP_REQUIRE(false, "counts: k=%d needs uint64 hashes", k);
    }
    KMAP_TRY(hist_hashes<H>(c, hash_dev, n, k, st));
    return kmap_counts_finish_hist(c, k, merge, n_uniq, st);
}

}  // namespace

int kmap_counts_hist_hashes(kmap_counts *c, const void *hash_dev, int64_t n, int k, hipStream_t st) {
    KMAP_REQUIRE(k > 0 && k <= 16, "counts_hist_hashes: k=%d", k);
    if (k < 16) return hist_hashes<uint32_t>(c, (const uint32_t *)hash_dev, n, k, st);
    return hist_hashes<uint64_t>(c, (const uint64_t *)hash_dev, n, k, st);
}

extern "C" {

int kmap_counts_create(kmap_counts **c) {
    KMAP_REQUIRE(c, "counts_create: null");
    *c = new kmap_counts();
    return KMAP_OK;
}
int kmap_counts_destroy(kmap_counts *c) {
    if (!c) return KMAP_OK;
    if (c->uniq) (void)hipFree(c->uniq);
    if (c->cnt) (void)hipFree(c->cnt);
    delete c;   // the histogram table belongs to the scratch arena
    return KMAP_OK;
}

int kmap_counts_run_hashes_dev(kmap_counts *c, const void *hash_dev, int64_t n, int k, int merge_revcom, int64_t *n_uniq,
                               void *stream) {
    KMAP_REQUIRE(c, "counts_run_hashes: null handle");
    KMAP_REQUIRE(k > 0 && k < 32, "counts_run_hashes: k=%d out of range", k);
    KMAP_REQUIRE(n >= 0 && (n == 0 || hash_dev), "counts_run_hashes: bad input");
    if (k < 16) return run_hashes<uint32_t>(c, (const uint32_t *)hash_dev, n, k, merge_revcom, n_uniq, as_stream(stream));
    return run_hashes<uint64_t>(c, (const uint64_t *)hash_dev, n, k, merge_revcom, n_uniq, as_stream(stream));
}

int kmap_counts_run_seq_dev(kmap_counts *c, const uint8_t *seq_dev, int64_t n, const int64_t *borders_dev, int64_t n_seq,
                            int k, int dedupe_per_read, int merge_revcom, int64_t *n_uniq, void *stream) {
    KMAP_REQUIRE(c, "counts_run_seq: null handle");
    KMAP_REQUIRE(k > 0 && k < 32, "counts_run_seq: k=%d out of range", k);
    KMAP_REQUIRE(n >= 0 && (n == 0 || seq_dev), "counts_run_seq: bad input");
    KMAP_REQUIRE(!dedupe_per_read || n_seq == 0 || borders_dev, "counts_run_seq: dedupe needs borders");
    hipStream_t st = as_stream(stream);
    const size_t hb = (k < 16) ? 4 : 8;
    void *hash = nullptr;   // cached arena buffer: 4-8 B per position, reused across k and rounds
    KMAP_TRY(kmap_scratch(&hash, (size_t)(n ? n : 1) * hb, st, KMAP_SLOT_HASH));
    int rc;
    if (k < 16) {
        rc = kmap_hash_launch_u32(seq_dev, n, k, (uint32_t *)hash, stream);
        if (rc == KMAP_OK && dedupe_per_read) rc = kmap_dedupe_per_read_u32_dev((uint32_t *)hash, n, borders_dev, n_seq, stream);
        if (rc == KMAP_OK) rc = run_hashes<uint32_t>(c, (const uint32_t *)hash, n, k, merge_revcom, n_uniq, st);
    } else {
        rc = kmap_hash_launch_u64(seq_dev, n, k, (uint64_t *)hash, stream);
        if (rc == KMAP_OK && dedupe_per_read) rc = kmap_dedupe_per_read_u64_dev((uint64_t *)hash, n, borders_dev, n_seq, stream);
        if (rc == KMAP_OK) rc = run_hashes<uint64_t>(c, (const uint64_t *)hash, n, k, merge_revcom, n_uniq, st);
    }
    return rc;
}

int kmap_counts_load(kmap_counts *c, const void *uniq, const void *cnt, int64_t n_uniq, int k) {
    KMAP_REQUIRE(c && k > 0 && k < 32 && n_uniq >= 0, "counts_load: bad arguments");
    KMAP_REQUIRE(n_uniq == 0 || (uniq && cnt), "counts_load: null pointer");
    const int narrow = (k < 16);
    if (c->cap < (size_t)n_uniq || !c->uniq) {
        if (c->uniq) KMAP_CHECK_HIP(hipFree(c->uniq));
        if (c->cnt) KMAP_CHECK_HIP(hipFree(c->cnt));
        c->uniq = nullptr; c->cnt = nullptr; c->cap = 0;
        const size_t cap = n_uniq ? (size_t)n_uniq : 1;
        KMAP_CHECK_HIP(hipMalloc(&c->uniq, cap * 8));
        KMAP_CHECK_HIP(hipMalloc((void **)&c->cnt, cap * 4));
        c->cap = cap;
    }
    if (n_uniq) {
        KMAP_CHECK_HIP(hipMemcpy(c->uniq, uniq, (size_t)n_uniq * (narrow ? 4 : 8), hipMemcpyHostToDevice));
        if (narrow) {
            KMAP_CHECK_HIP(hipMemcpy(c->cnt, cnt, (size_t)n_uniq * 4, hipMemcpyHostToDevice));
        } else {   // int64 counts are kept as uint32 on the device (see header: < 2^32 per k-mer)
            uint32_t *tmp = (uint32_t *)malloc((size_t)n_uniq * 4);
            KMAP_REQUIRE(tmp, "counts_load: host malloc");
            for (int64_t i = 0; i < n_uniq; ++i) tmp[i] = (uint32_t)((const int64_t *)cnt)[i];
            hipError_t e = hipMemcpy(c->cnt, tmp, (size_t)n_uniq * 4, hipMemcpyHostToDevice);
            free(tmp);
            KMAP_CHECK_HIP(e);
        }
    }
    c->k = k;
    c->narrow = narrow;
    c->n_uniq = n_uniq;
    return KMAP_OK;
}

/* ---- key-range-sharded multi-GPU counting (11 <= k <= 16; the caller owns the collectives) -------------------------------------
 * kmap_counts_hist_packed_dev (local table) -> [merge_revcom: kmap_counts_presence_dev -> SUM all-reduce of the nibbles ->
 * kmap_counts_merge_presence_dev] -> reduce the table slice by slice to its owner (SUM) -> kmap_counts_finish_range on the own
 * slice -> all-gather of the (uniq, cnt) shards -> kmap_counts_adopt_dev.  Against the all-reduce of the whole table every rank
 * receives one slice instead of the table, plus half a byte per bin of presence. */
int kmap_counts_presence_dev(kmap_counts *c, int k, void *nib_dev, void *stream) {
    KMAP_REQUIRE(c && c->bins && nib_dev && k >= 2 && k <= 16 && c->bins_cap >= ((size_t)1 << (2 * k)), "counts_presence: no histogram for k=%d", k);
    KMAP_TRY(kmap_counts_bins_check(c, "counts_presence"));
    const uint64_t n_bins = (uint64_t)1 << (2 * k);
    presence_nibbles_kernel<<<(unsigned)((n_bins / 8 + BLK - 1) / BLK), BLK, 0, as_stream(stream)>>>(c->bins, n_bins, (uint32_t *)nib_dev);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}
int kmap_counts_merge_presence_dev(kmap_counts *c, int k, const void *nib_dev, void *stream) {
    KMAP_REQUIRE(c && c->bins && nib_dev && k >= 11 && k <= 16 && c->bins_cap >= ((size_t)1 << (2 * k)),
                 "counts_merge_presence: needs a histogram and 11 <= k <= 16 (k=%d)", k);
    KMAP_TRY(kmap_counts_bins_check(c, "counts_merge_presence"));
    rc_merge_tiles_kernel<true><<<(unsigned)((size_t)1 << (2 * (k - 6))), BLK, 0, as_stream(stream)>>>(c->bins, k, (const uint8_t *)nib_dev, nullptr);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}
int kmap_counts_finish_range(kmap_counts *c, int k, int merged, uint64_t first_bin, uint64_t n_bins, int64_t *n_uniq, void *stream) {
    KMAP_REQUIRE(c && c->bins && k > 0 && k <= 16 && c->bins_cap >= ((size_t)1 << (2 * k)), "counts_finish_range: no histogram for k=%d", k);
    KMAP_REQUIRE(first_bin + n_bins <= ((uint64_t)1 << (2 * k)) && first_bin % 8 == 0, "counts_finish_range: slice [%llu, +%llu) outside the table or not 8-aligned",
                 (unsigned long long)first_bin, (unsigned long long)n_bins);
    KMAP_TRY(kmap_counts_bins_check(c, "counts_finish_range"));
    return compact_range(c, k, merged ? 2 : 0, first_bin, n_bins, n_uniq, as_stream(stream));
}
/* the handle's table <- device arrays (uint32 / uint64 keys as k < 16 / k >= 16, uint32 counts), copied */
int kmap_counts_adopt_dev(kmap_counts *c, const void *uniq_dev, const void *cnt_dev, int64_t n_uniq, int k) {
    KMAP_REQUIRE(c && k > 0 && k < 32 && n_uniq >= 0, "counts_adopt: bad arguments");
    KMAP_REQUIRE(n_uniq == 0 || (uniq_dev && cnt_dev), "counts_adopt: null pointer");
    const int narrow = (k < 16);
    if (n_uniq && uniq_dev != c->uniq) {
        void *u = nullptr;
        uint32_t *q = nullptr;
        KMAP_CHECK_HIP(hipMalloc(&u, (size_t)n_uniq * 8));
        if (hipMalloc((void **)&q, (size_t)n_uniq * 4) != hipSuccess) {
            (void)hipFree(u);
            kmap_set_error("counts_adopt: hipMalloc(%zu) failed", (size_t)n_uniq * 4);
            return KMAP_E_NOMEM;
        }
        hipError_t e = hipMemcpy(u, uniq_dev, (size_t)n_uniq * (narrow ? 4 : 8), hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(q, cnt_dev, (size_t)n_uniq * 4, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            (void)hipFree(u);
            (void)hipFree(q);
            KMAP_CHECK_HIP(e);
        }
        if (c->uniq) KMAP_CHECK_HIP(hipFree(c->uniq));
        if (c->cnt) KMAP_CHECK_HIP(hipFree(c->cnt));
        c->uniq = u;
        c->cnt = q;
        c->cap = (size_t)n_uniq;
    }
    c->k = k;
    c->narrow = narrow;
    c->n_uniq = n_uniq;
    return KMAP_OK;
}

int kmap_counts_fetch(kmap_counts *c, void *uniq_out, void *cnt_out) {
    KMAP_REQUIRE(c && c->k > 0, "counts_fetch: nothing counted yet");
    if (c->n_uniq == 0) return KMAP_OK;
    KMAP_REQUIRE(uniq_out && cnt_out, "counts_fetch: null output");
    const size_t n = (size_t)c->n_uniq;
    KMAP_CHECK_HIP(hipMemcpy(uniq_out, c->uniq, n * (c->narrow ? 4 : 8), hipMemcpyDeviceToHost));
    if (c->narrow) {
        KMAP_CHECK_HIP(hipMemcpy(cnt_out, c->cnt, n * 4, hipMemcpyDeviceToHost));   // uint32 bits == int32 wrap
    } else {
        uint32_t *tmp = (uint32_t *)malloc(n * 4);
        KMAP_REQUIRE(tmp, "counts_fetch: host malloc");
        hipError_t e = hipMemcpy(tmp, c->cnt, n * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess)   // widen uint32 -> int64 on several host threads (10^9 entries at k = 16)
            kmap_convert_pool<uint32_t, int64_t>((int64_t *)cnt_out, tmp, n, std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
        free(tmp);
        KMAP_CHECK_HIP(e);
    }
    return KMAP_OK;
}

}  // extern "C"

namespace {
// device -> host copy of n elements through two pinned staging buffers on `st`, converted on host threads while the next chunk is
// in flight: SRC (device element) -> DST (host element), e.g. uint32 -> int64.  A plain hipMemcpy into pageable numpy memory runs
// at ~16 GB/s and, on the null stream, would also serialise with the kernels of the trials that follow; this path keeps the
// transfer on its own stream (SDMA next to the kernels) and reaches the host-side memory bandwidth.
// pinned staging buffers are kept in a process-wide free list (hipHostMalloc costs ~75 ms per 256 MiB: more than the whole copy
// of a 1-GB table); concurrent fetches (two TableSaver threads) each take their own pair
constexpr size_t STAGE_BYTES = (size_t)256 << 20;
std::mutex g_stage_mu;
std::vector<void *> g_stage_free;
struct StagePair {
    void *buf[2] = {nullptr, nullptr};
    StagePair() {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        for (int b = 0; b < 2; ++b) {
            if (!g_stage_free.empty()) {
                buf[b] = g_stage_free.back();
                g_stage_free.pop_back();
            } else if (hipHostMalloc(&buf[b], STAGE_BYTES, hipHostMallocDefault) != hipSuccess) {
                buf[b] = nullptr;
            }
        }
    }
    bool ok() const { return buf[0] && buf[1]; }
    ~StagePair() {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        for (int b = 0; b < 2; ++b)
            if (buf[b]) g_stage_free.push_back(buf[b]);
    }
};

template <typename SRC, typename DST>
int staged_fetch(DST *dst, const SRC *src_dev, size_t n, hipStream_t st) {
    if (n == 0) return KMAP_OK;
    StagePair sp;                                                 // two pinned 256-MiB buffers from the process-wide pool
    if (!sp.ok()) {
        kmap_set_error("counts_fetch: pinned staging allocation failed");
        return KMAP_E_NOMEM;
    }
    const size_t chunk = STAGE_BYTES / sizeof(SRC);               // elements per chunk
    SRC *stage[2] = {(SRC *)sp.buf[0], (SRC *)sp.buf[1]};
    const unsigned nt = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    hipError_t err = hipSuccess;
    size_t off = 0;
    int b = 0;
    size_t len = std::min(chunk, n);
    err = hipMemcpyAsync(stage[0], src_dev, len * sizeof(SRC), hipMemcpyDeviceToHost, st);
    while (err == hipSuccess && off < n) {
        err = hipStreamSynchronize(st);                           // chunk `b` has landed
        if (err != hipSuccess) break;
        const size_t next_off = off + len, next_len = next_off < n ? std::min(chunk, n - next_off) : 0;
        if (next_len) err = hipMemcpyAsync(stage[b ^ 1], src_dev + next_off, next_len * sizeof(SRC), hipMemcpyDeviceToHost, st);
        kmap_convert_pool<SRC, DST>(dst + off, stage[b], len, nt);  // the destination may be an unaligned view into a memory-mapped pickle file (TableSaver)
        off = next_off;
        len = next_len;
        b ^= 1;
    }
    if (err != hipSuccess) (void)hipStreamSynchronize(st);
    KMAP_CHECK_HIP(err);
    return KMAP_OK;
}

// counts widened to the reference's int64 on the device (k >= 16): the bytes that cross PCIe are the bytes of the file
__global__ __launch_bounds__(256) void widen_counts_kernel(const uint32_t *__restrict__ in, int64_t n, int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int64_t)in[i];
}

// write() in full (short writes, EINTR); false + errno on failure
static bool pwrite_all(int fd, const char *buf, size_t len, int64_t off) {
    while (len) {
        const ssize_t w = pwrite(fd, buf, len, (off_t)off);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        buf += w;
        off += w;
        len -= (size_t)w;
    }
    return true;
}

// `n` elements of a device array -> file bytes at `file_off`, in the file's dtype DST: chunk i + 1 crosses PCIe into one pinned
// staging buffer while chunk i is written from the other -- no pageable copy, no conversion pass on the host (WIDEN: the uint32
// counts become int64 on the device, in a scratch buffer the size of a chunk)
template <typename SRC, typename DST>
int staged_write(int fd, int64_t file_off, const SRC *src_dev, size_t n, hipStream_t st) {
    if (n == 0) return KMAP_OK;
    constexpr bool WIDEN = !std::is_same<SRC, DST>::value;
    StagePair sp;
    if (!sp.ok()) {
        kmap_set_error("counts_write: pinned staging allocation failed");
        return KMAP_E_NOMEM;
    }
    const size_t chunk = STAGE_BYTES / sizeof(DST);
    DevBuf wide[2];
    if (WIDEN) {
        KMAP_TRY(wide[0].alloc(std::min(chunk, n) * sizeof(DST)));
        KMAP_TRY(wide[1].alloc(std::min(chunk, n) * sizeof(DST)));
    }
    auto issue = [&](int b, size_t off, size_t len) -> hipError_t {
        const void *from = src_dev + off;
        if (WIDEN) {
            widen_counts_kernel<<<(unsigned)((len + 255) / 256), 256, 0, st>>>((const uint32_t *)(src_dev + off), (int64_t)len, wide[b].as<int64_t>());
            from = wide[b].p;
        }
        return hipMemcpyAsync(sp.buf[b], from, len * sizeof(DST), hipMemcpyDeviceToHost, st);
    };
    size_t off = 0, len = std::min(chunk, n);
    int b = 0;
    hipError_t err = issue(0, 0, len);
    while (err == hipSuccess && off < n) {
        err = hipStreamSynchronize(st);                           // chunk `b` has landed
        if (err != hipSuccess) break;
        const size_t next_off = off + len, next_len = next_off < n ? std::min(chunk, n - next_off) : 0;
        if (next_len) err = issue(b ^ 1, next_off, next_len);
        if (!pwrite_all(fd, (const char *)sp.buf[b], len * sizeof(DST), file_off + (int64_t)(off * sizeof(DST)))) {
            const int en = errno;
            (void)hipStreamSynchronize(st);
            kmap_set_error("counts_write: pwrite failed: %s", strerror(en));
            return KMAP_E_IO;
        }
        off = next_off;
        len = next_len;
        b ^= 1;
    }
    if (err != hipSuccess) (void)hipStreamSynchronize(st);
    KMAP_CHECK_HIP(err);
    return KMAP_OK;
}

// uint64 keys that fit 32 bits (k = 16): narrowed on the device so that half the bytes cross PCIe
__global__ __launch_bounds__(256) void narrow_keys_kernel(const uint64_t *__restrict__ in, int64_t n, uint32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (uint32_t)in[i];
}
}  // namespace

extern "C" {

/* device addresses of the resident table (uniq: uint32 for k < 16 else uint64; cnt: uint32 whatever k), valid until the next
 * count / load / destroy on this handle */
int kmap_counts_table_dev(kmap_counts *c, void **uniq_dev, void **cnt_dev, int64_t *n_uniq) {
    KMAP_REQUIRE(c && c->k > 0 && uniq_dev && cnt_dev && n_uniq, "counts_table_dev: nothing counted yet / null output");
    *uniq_dev = c->uniq;
    *cnt_dev = (void *)c->cnt;
    *n_uniq = c->n_uniq;
    return KMAP_OK;
}

/* kmap_counts_fetch on a caller-chosen stream (so that a background host thread can drain a finished table while the null
 * stream keeps counting into ANOTHER handle): pinned staging, conversion on host threads.  Blocks until the arrays are complete. */
int kmap_counts_fetch_stream(kmap_counts *c, void *uniq_out, void *cnt_out, void *stream) {
    KMAP_REQUIRE(c && c->k > 0, "counts_fetch: nothing counted yet");
    if (c->n_uniq == 0) return KMAP_OK;
    KMAP_REQUIRE(uniq_out && cnt_out, "counts_fetch: null output");
    hipStream_t st = as_stream(stream);
    const size_t n = (size_t)c->n_uniq;
    if (c->narrow) {
        KMAP_TRY((staged_fetch<uint32_t, uint32_t>((uint32_t *)uniq_out, (const uint32_t *)c->uniq, n, st)));
        KMAP_TRY((staged_fetch<uint32_t, uint32_t>((uint32_t *)cnt_out, c->cnt, n, st)));   // uint32 bits == int32 wrap
    } else {
        if (c->k <= 16) {
            DevBuf k32;
            KMAP_TRY(k32.alloc(n * 4));
            narrow_keys_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((const uint64_t *)c->uniq, (int64_t)n, k32.as<uint32_t>());
            KMAP_CHECK_HIP(hipGetLastError());
            KMAP_TRY((staged_fetch<uint32_t, uint64_t>((uint64_t *)uniq_out, k32.as<uint32_t>(), n, st)));
        } else {
            KMAP_TRY((staged_fetch<uint64_t, uint64_t>((uint64_t *)uniq_out, (const uint64_t *)c->uniq, n, st)));
        }
        KMAP_TRY((staged_fetch<uint32_t, int64_t>((int64_t *)cnt_out, c->cnt, n, st)));
    }
    return KMAP_OK;
}

/* a range of one array of the table, converted to the reference's dtype: which = 0 the unique hashes (uint32 for k < 16, uint64
 * otherwise), which = 1 the counts (int32 / int64).  Same stream / staging path as kmap_counts_fetch_stream; lets a writer
 * stream a multi-GB table to a file chunk by chunk without holding it in host memory. */
int kmap_counts_fetch_range(kmap_counts *c, int which, int64_t first, int64_t count, void *out, void *stream) {
    KMAP_REQUIRE(c && c->k > 0 && (which == 0 || which == 1), "counts_fetch_range: nothing counted yet / bad selector");
    KMAP_REQUIRE(first >= 0 && count >= 0 && first + count <= c->n_uniq, "counts_fetch_range: range outside the table");
    if (count == 0) return KMAP_OK;
    KMAP_REQUIRE(out, "counts_fetch_range: null output");
    hipStream_t st = as_stream(stream);
    const size_t n = (size_t)count;
    if (which == 0) {
        if (c->narrow) return staged_fetch<uint32_t, uint32_t>((uint32_t *)out, (const uint32_t *)c->uniq + first, n, st);
        return staged_fetch<uint64_t, uint64_t>((uint64_t *)out, (const uint64_t *)c->uniq + first, n, st);
    }
    if (c->narrow) return staged_fetch<uint32_t, uint32_t>((uint32_t *)out, c->cnt + first, n, st);   // uint32 bits == int32 wrap
    return staged_fetch<uint32_t, int64_t>((int64_t *)out, c->cnt + first, n, st);
}

/* kmap_counts_fetch_range straight into a file: the range's bytes, in the reference's dtype, are written at `file_offset` of the open
 * descriptor `fd` (pwrite: the descriptor's position is not used) from pinned staging buffers, the next chunk in flight while the
 * current one is written.  The k{k}.pkl writers place the array payloads of a pickle whose layout they know this way. */
int kmap_counts_write_range(kmap_counts *c, int which, int64_t first, int64_t count, int fd, int64_t file_offset, void *stream) {
    KMAP_REQUIRE(c && c->k > 0 && (which == 0 || which == 1), "counts_write_range: nothing counted yet / bad selector");
    KMAP_REQUIRE(first >= 0 && count >= 0 && first + count <= c->n_uniq, "counts_write_range: range outside the table");
    KMAP_REQUIRE(fd >= 0 && file_offset >= 0, "counts_write_range: bad file descriptor / offset");
    if (count == 0) return KMAP_OK;
    hipStream_t st = as_stream(stream);
    const size_t n = (size_t)count;
    if (which == 0) {
        if (c->narrow) return staged_write<uint32_t, uint32_t>(fd, file_offset, (const uint32_t *)c->uniq + first, n, st);
        return staged_write<uint64_t, uint64_t>(fd, file_offset, (const uint64_t *)c->uniq + first, n, st);
    }
    if (c->narrow) return staged_write<uint32_t, uint32_t>(fd, file_offset, c->cnt + first, n, st);   // uint32 bits == int32 wrap
    return staged_write<uint32_t, int64_t>(fd, file_offset, c->cnt + first, n, st);
}

int kmap_counts_total(kmap_counts *c, int64_t *total) {
    KMAP_REQUIRE(c && c->k > 0 && total, "counts_total: nothing counted yet");
    *total = 0;
    if (c->n_uniq == 0) return KMAP_OK;
    DevBuf t;
    KMAP_TRY(t.alloc(8));
    KMAP_CHECK_HIP(hipMemset(t.p, 0, 8));
    int64_t g = (c->n_uniq + BLK - 1) / BLK;
    if (g > 4096) g = 4096;
    sum_counts_kernel<<<(unsigned)g, BLK>>>(c->cnt, c->n_uniq, c->narrow, t.as<unsigned long long>());
    KMAP_CHECK_HIP(hipMemcpy(total, t.p, 8, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

int kmap_counts_topk(kmap_counts *c, int top_k, int64_t *idx_out, uint64_t *kh_out, int64_t *cnt_out, int *n_found) {
    KMAP_REQUIRE(c && c->k > 0, "counts_topk: nothing counted yet");
    KMAP_REQUIRE(top_k > 0 && top_k <= TK && idx_out && kh_out && cnt_out && n_found, "counts_topk: bad arguments (top_k <= %d)", TK);
    KMAP_REQUIRE(c->n_uniq < ((int64_t)1 << 32), "counts_topk: more than 2^32 unique k-mers");
    *n_found = 0;
    if (c->n_uniq == 0) return KMAP_OK;
    int64_t g = (c->n_uniq + BLK - 1) / BLK;
    if (g > 1024) g = 1024;
    DevBuf out;
    KMAP_TRY(out.alloc((size_t)g * top_k * 8));
    topk_kernel<<<(unsigned)g, BLK>>>(c->cnt, c->n_uniq, c->narrow, top_k, out.as<unsigned long long>());
    KMAP_CHECK_HIP(hipGetLastError());
    std::vector<unsigned long long> keys((size_t)g * top_k);
    KMAP_CHECK_HIP(hipMemcpy(keys.data(), out.p, keys.size() * 8, hipMemcpyDeviceToHost));
    std::sort(keys.begin(), keys.end(), [](unsigned long long a, unsigned long long b) { return a > b; });
    int m = 0;
    for (; m < top_k && m < (int)keys.size() && keys[(size_t)m] != 0; ++m) {
        const int64_t idx = (int64_t)(0xFFFFFFFFull - (keys[(size_t)m] & 0xFFFFFFFFull));
        idx_out[m] = idx;
        cnt_out[m] = (int64_t)(keys[(size_t)m] >> 32);
        if (c->narrow) {
            uint32_t h = 0;
            KMAP_CHECK_HIP(hipMemcpy(&h, (const uint32_t *)c->uniq + idx, 4, hipMemcpyDeviceToHost));
            kh_out[m] = h;
        } else {
            KMAP_CHECK_HIP(hipMemcpy(&kh_out[m], (const uint64_t *)c->uniq + idx, 8, hipMemcpyDeviceToHost));
        }
    }
    *n_found = m;
    return KMAP_OK;
}

int kmap_counts_hamball_mass(kmap_counts *c, const uint64_t *cands, int n_cand, int radius, int revcom, double *mass_out) {
    KMAP_REQUIRE(c && c->k > 0, "hamball_mass: nothing counted yet");
    KMAP_REQUIRE(n_cand >= 0 && (n_cand == 0 || (cands && mass_out)), "hamball_mass: null pointer");
    DevBuf m;
    KMAP_TRY(m.alloc(16 * 8));
    for (int c0 = 0; c0 < n_cand; c0 += 16) {
        CandTab t;
        t.n = (n_cand - c0 < 16) ? n_cand - c0 : 16;
        for (int i = 0; i < t.n; ++i) {
            t.fwd[i] = cands[c0 + i];
            t.rc[i] = host_revcom(cands[c0 + i], c->k, c->narrow);
        }
        KMAP_CHECK_HIP(hipMemset(m.p, 0, 16 * 8));
        if (c->n_uniq > 0) {
            int64_t g = (c->n_uniq + BLK - 1) / BLK;
            if (g > 2048) g = 2048;
            if (c->narrow)
                mass_kernel<uint32_t><<<(unsigned)g, BLK>>>((const uint32_t *)c->uniq, c->cnt, c->n_uniq, c->k, t, radius,
                                                            revcom, 1, m.as<unsigned long long>());
            else
                mass_kernel<uint64_t><<<(unsigned)g, BLK>>>((const uint64_t *)c->uniq, c->cnt, c->n_uniq, c->k, t, radius,
                                                            revcom, 0, m.as<unsigned long long>());
            KMAP_CHECK_HIP(hipGetLastError());
        }
        long long host[16];
        KMAP_CHECK_HIP(hipMemcpy(host, m.p, 16 * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < t.n; ++i) mass_out[c0 + i] = (double)host[i];
    }
    return KMAP_OK;
}

}  // extern "C"
