// counts_fine.hip -- partitioned histogram for 10 <= k <= 14 with 16-bit keys and XCD-private bucket streams.
//
// counts_part.hip's first version of this path sorted 4-byte keys into 1024 buckets; at k = 14 a bucket then spans 2^18 bins
// and the LDS histogram needed four range passes over every bucket's keys (3.3 of the count pass's 8.6 ms at C3, all of it L2
// re-reads), and the key array was 4 B per k-mer written once and read four times.  Here
//   * a bucket spans at most 65 536 bins (4096 buckets), so the histogram step is ONE pass;
//   * a key inside its bucket is 16 bits: the key array is 2 B per k-mer (bucket = position in the array);
//   * more buckets mean shorter runs per tile (a 32 768-window tile holds ~8 keys = 16 B of every one of 4096 buckets).  Short
//     appends to a stream that all eight XCDs write leave partially written lines in eight L2s (tools/probes/append_streams.hip:
//     16-B appends to 4096 shared streams 1.0 TB/s); with one stream per (bucket, XCD class), class = tile mod 8 = the XCD a
//     tile's block runs on when the grid is a multiple of 8, a line is completed inside ONE L2 before it leaves it (1.6 TB/s at
//     16 B, 2.2 at 32 B).  The class is a function of the tile, not of the hardware: a different dispatch order costs speed only;
//   * the counting sort inside a tile keeps 16-bit keys in LDS; the bucket of a sorted position is recovered from a bitmap of run
//     starts (rank = popcount prefix) that indexes a dense list of the non-empty buckets' global bases.
// Steps: (1) valid keys per (class, bucket); (2) exclusive scan in (bucket, class) order (one block); (3) tile-staged counting sort into the
// (bucket, class) regions; (4) one block per bucket: LDS histogram of its keys, plain stores into the 4^k table (every bin
// written once: no memset, no global atomics on the table).  The table then goes through counts.hip's merge / compaction.
#include <stdlib.h>

#include "common.h"
#include "counts_internal.h"
#include "packed_keys.h"

namespace {
constexpr int FC = 8;                                   // XCD classes
constexpr int FS_TPB = 1024;
constexpr uint32_t INV32 = 0xFFFFFFFFu;
constexpr uint32_t FH_LIMIT = 0x4000u;                  // 16-bit counters: spill threshold (see counts_part.hip's half kernel)

// the KPT = 16 GPT keys of a thread in tile `tile` (FT = 1024 * KPT windows); invalid / absent windows come back as INV32
template <bool PACKED, int GPT>
__device__ __forceinline__ void tile_keys(const uint32_t *__restrict__ h, const uint16_t *__restrict__ inval, const uint32_t *__restrict__ skip,
                                          int k, int64_t n, int64_t tile, uint32_t (&v)[16 * GPT]) {
    constexpr int KPT = 16 * GPT;
    const int64_t t0 = tile * (int64_t)(FS_TPB * KPT);
    if (PACKED) {
        const int64_t g0 = (t0 >> 4) + (int64_t)GPT * threadIdx.x;
        uint32_t n1;
#pragma unroll
        for (int j = 0; j < GPT; ++j) packed_group_keys(h, inval, skip, n, k, g0 + j, *reinterpret_cast<uint32_t(*)[16]>(&v[16 * j]), n1);
    } else {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int64_t i = t0 + (int64_t)j * FS_TPB + threadIdx.x;
            v[j] = (i < n) ? h[i] : INV32;
        }
    }
}

// ---- the packed source, two groups per thread, as RAW loads + a separate key computation --------------------------------------
// A tile's keys come from four loads per thread (8 B of codes + 4 B, 8 B of invalid flags, 4 B of skip bits: the thread's groups g0,
// g0 + 1 are an aligned pair).  Split from the arithmetic so that the loads of the block's NEXT tile can be issued at the top of the
// current one -- unconditionally, on clamped addresses: a block's 16 waves all reach a tile's first use together (the barriers
// keep them in step), and without the prefetch every tile began with one exposed memory round trip (r03: ~0.4 ms of the 3.0).
struct RawPair {
    uint32_t c0, c1, c2;            // codes of groups g0, g0 + 1, g0 + 2
    uint32_t f01, f23;              // invalid flags of groups g0 .. g0 + 3 (little-endian pairs)
    uint32_t sk;                    // skip bits of groups g0 (high half), g0 + 1 (low half)
};
__device__ __forceinline__ void raw_pair_load(RawPair &r, const uint32_t *__restrict__ codes, const uint16_t *__restrict__ inval,
                                              const uint32_t *__restrict__ skip, int64_t n, int64_t g0) {
    const int64_t glast = (((n + 15) >> 4) - 1) & ~(int64_t)1;            // last aligned pair that holds data (kmap_packed_groups: g + 3 stays inside)
    const int64_t g = g0 < glast ? g0 : (glast > 0 ? glast : 0);
    const uint2 c = *reinterpret_cast<const uint2 *>(codes + g);
    r.c0 = c.x;
    r.c1 = c.y;
    r.c2 = codes[g + 2];
    const uint2 f = *reinterpret_cast<const uint2 *>(inval + g);
    r.f01 = f.x;
    r.f23 = f.y;
    r.sk = skip ? skip[g >> 1] : 0u;
}
__device__ __forceinline__ void group_keys_from(uint32_t hi, uint32_t lo, uint64_t bad, uint32_t skip16, int64_t left, int k, uint32_t (&keys)[16]) {
    for (int have = 1; have < k;) {
        const int step = (have <= k - have) ? have : k - have;
        bad |= bad << step;
        have += step;
    }
    uint32_t drop16 = ((uint32_t)(bad >> 32) & 0xFFFFu) | skip16;          // windows 0..15 in bits 15..0
    if (left < 16) drop16 |= left <= 0 ? 0xFFFFu : ((1u << (16 - (int)left)) - 1u);   // windows that start inside the array
    const int sh = 32 - 2 * k;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t top = (i == 0) ? hi : __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * i);
        keys[i] = (top >> sh) | (uint32_t)__builtin_amdgcn_sbfe((int)drop16, 15 - i, 1);   // all ones when dropped (k <= 14: never a hash)
    }
}
__device__ __forceinline__ void raw_pair_keys(const RawPair &r, int64_t n, int k, int64_t g0, uint32_t (&v)[32]) {
    const uint32_t f0 = r.f01 & 0xFFFFu, f1 = r.f01 >> 16, f2 = r.f23 & 0xFFFFu, f3 = r.f23 >> 16;
    group_keys_from(r.c0, r.c1, ((uint64_t)f0 << 32) | ((uint64_t)f1 << 16) | f2, r.sk >> 16, n - 16 * g0, k,
                    *reinterpret_cast<uint32_t(*)[16]>(&v[0]));
    group_keys_from(r.c1, r.c2, ((uint64_t)f1 << 32) | ((uint64_t)f2 << 16) | f3, r.sk & 0xFFFFu, n - 16 * (g0 + 1), k,
                    *reinterpret_cast<uint32_t(*)[16]>(&v[16]));
}

// (1) valid keys per (class, bucket): gcount[class * NB + bucket].  (Measured: a bucket-only extraction -- five instead of eleven
// vector instructions per window, byte offsets straight into the counters -- 0.48 against 0.43 ms: the pass sits at the LDS atomic rate.)  The grid is a multiple of 8, so all tiles of a block share
// their class (tile = blockIdx + i * gridDim).
template <bool PACKED, int GPT>
__global__ __launch_bounds__(FS_TPB) void fine_count_kernel(const uint32_t *__restrict__ h, const uint16_t *__restrict__ inval,
                                                            const uint32_t *__restrict__ skip, int k, int64_t n, int low_bits, int NB,
                                                            uint32_t *__restrict__ gcount) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];         // NB + 64 (the lanes' private counters for invalid keys)
    constexpr int KPT = 16 * GPT;
    for (int b = threadIdx.x; b < NB + 64; b += FS_TPB) cnt[b] = 0;
    __syncthreads();
    const int64_t n_tiles = (n + FS_TPB * KPT - 1) / (FS_TPB * KPT);
    const uint32_t dummy = (uint32_t)NB + (threadIdx.x & 63u);
    if constexpr (PACKED && GPT == 2) {
        auto g0_of = [&](int64_t tile) { return ((tile * (int64_t)(FS_TPB * KPT)) >> 4) + 2 * (int64_t)threadIdx.x; };
        RawPair cur, nxt;
        raw_pair_load(cur, h, inval, skip, n, g0_of(blockIdx.x));
        for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
            raw_pair_load(nxt, h, inval, skip, n, g0_of(tile + gridDim.x));          // next tile's loads in flight (clamped behind the array)
            uint32_t v[KPT];
            raw_pair_keys(cur, n, k, g0_of(tile), v);
#pragma unroll
            for (int j = 0; j < KPT; ++j) atomicAdd(&cnt[v[j] == INV32 ? dummy : v[j] >> low_bits], 1u);
            cur = nxt;
        }
    } else {
        for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
            uint32_t v[KPT];
            tile_keys<PACKED, GPT>(h, inval, skip, k, n, tile, v);
#pragma unroll
            for (int j = 0; j < KPT; ++j) atomicAdd(&cnt[v[j] == INV32 ? dummy : v[j] >> low_bits], 1u);
        }
    }
    __syncthreads();
    const int cls = (int)(blockIdx.x & (FC - 1));
    for (int b = threadIdx.x; b < NB; b += FS_TPB)
        if (cnt[b]) atomicAdd(&gcount[(size_t)cls * NB + b], cnt[b]);
}

// (2) (class, bucket) counts -> exclusive offsets in (bucket, class) order (goff[NB * 8 + 1]) and the per-(class, bucket) cursors
// that start there; and the work list of the histogram step.  One block; thread t scans the 8 NB / 1024 consecutive (bucket,
// class) entries it owns.  (scan_util.h's multi-tile scan takes arena slots C / D for its tile sums: slot C holds the
// per-read-dedupe skip bits this count pass reads.)
// Work list: a bucket with more than four times the average number of keys (a planted motif, poly-A: its block would run alone for
// a multiple of everybody else's time -- 12x at C3's k = 14 for the bucket of the motif's first six bases) is cut into slices of
// two average buckets; slices come first in the list, add their LDS bins to the table with device atomics, and their buckets' table
// segments are zeroed beforehand (fine_zero_heavy_kernel).  plan: [0] items, [1] heavy buckets, [2..3] slice length (uint64),
// [4 ..) items = bucket | slice << 12 | heavy << 31, then (at 4 + 2 NB) the heavy buckets' ids.
constexpr uint32_t FP_HEAVY = 0x80000000u;
__global__ __launch_bounds__(FS_TPB) void fine_offsets_kernel(const uint32_t *__restrict__ gcb, int NB, uint64_t *__restrict__ goff,
                                                              unsigned long long *__restrict__ cursor, uint32_t *__restrict__ plan) {
    __shared__ uint64_t wsum[FS_TPB / 64], total_s;
    const int per = NB * FC / FS_TPB, bpt = NB / FS_TPB;                     // 8 or 32 entries, 1 or 4 buckets per thread
    const int i0 = threadIdx.x * per;
    uint64_t s = 0;
    for (int j = 0; j < per; ++j) {
        const int i = i0 + j;
        s += gcb[(size_t)(i % FC) * NB + i / FC];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = s;
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint64_t run = inc - s;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    uint64_t size[4] = {0, 0, 0, 0};                                        // keys of the thread's buckets (bpt <= 4)
    for (int j = 0; j < per; ++j) {
        const int i = i0 + j;
        const int b = i / FC;
        goff[i] = run;
        cursor[(size_t)(i % FC) * NB + (size_t)(b % bpt) * FS_TPB + b / bpt] = run;   // slot order of fine_scatter_kernel's reservations
        const uint32_t c = gcb[(size_t)(i % FC) * NB + b];
        run += c;
        size[j / FC] += c;
    }
    if (threadIdx.x == FS_TPB - 1) {
        goff[NB * FC] = run;
        total_s = run;
    }
    __syncthreads();
    const uint64_t avg = total_s / (uint64_t)NB;
    const uint64_t heavy_min = 4 * avg > 131072 ? 4 * avg : 131072;
    const uint64_t slice_len = ((2 * avg > 65536 ? 2 * avg : 65536) + 7) & ~(uint64_t)7;
    // exclusive scan of (unsplit buckets, slices, heavy buckets) packed in 16 + 16 + 16 bits (each total <= 4096)
    uint64_t mine = 0;
    for (int j = 0; j < bpt; ++j) {
        if (size[j] > heavy_min) mine += (((size[j] + slice_len - 1) / slice_len) << 16) + ((uint64_t)1 << 32);
        else mine += 1;
    }
    uint64_t pinc = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t t = __shfl_up(pinc, o);
        if (lane >= o) pinc += t;
    }
    __syncthreads();                                                         // wsum is reused
    if (lane == 63) wsum[wave] = pinc;
    __syncthreads();
    uint64_t before = pinc - mine, all = 0;
    for (int w = 0; w < FS_TPB / 64; ++w) {
        if (w < wave) before += wsum[w];
        all += wsum[w];
    }
    const uint32_t n_slices = (uint32_t)(all >> 16) & 0xFFFFu, n_plain = (uint32_t)all & 0xFFFFu, n_heavy = (uint32_t)(all >> 32);
    uint32_t at_plain = n_slices + ((uint32_t)before & 0xFFFFu), at_slice = (uint32_t)(before >> 16) & 0xFFFFu, at_heavy = (uint32_t)(before >> 32);
    uint32_t *items = plan + 4, *heavy = plan + 4 + 2 * NB;
    for (int j = 0; j < bpt; ++j) {
        const uint32_t b = (uint32_t)(threadIdx.x * bpt + j);
        if (size[j] > heavy_min) {
            const uint32_t ns = (uint32_t)((size[j] + slice_len - 1) / slice_len);
            for (uint32_t sl = 0; sl < ns; ++sl) items[at_slice++] = b | (sl << 12) | FP_HEAVY;
            heavy[at_heavy++] = b;
        } else {
            items[at_plain++] = b;
        }
    }
    if (threadIdx.x == 0) {
        plan[0] = n_slices + n_plain;
        plan[1] = n_heavy;
        plan[2] = (uint32_t)slice_len;
        plan[3] = (uint32_t)(slice_len >> 32);
    }
}

// (3) counting sort of one tile per block iteration.  Phase times at C3, k = 14 (3.0 ms): key extraction + counting 0.72, scan +
// reservations 0.45, placement 0.63, write-out 1.38 -- the last is the rate the memory system takes 16-byte appends at (the
// probe's 1.6 TB/s): halving the write-out's instructions (rank from a per-word popcount prefix), halving its active lanes
// (4-byte stores) and overlapping it with the next tile's extraction (software-pipelined loop, two counter sets: 2.95 ms) each
// left the pass where it was; so did branch-free atomics (invalid windows to per-lane spare counters instead of an exec-mask
// save / restore around each of the 2 x 32 atomics).  LDS: sorted 16-bit keys (2 B x FT), bucket counters / cursors (NB), dense
// bases of the non-empty buckets (8 B x NB), run-start bitmap (FT bits) + its words paired with their popcount prefix (8 B / word).
template <int GPT, int BPT>
constexpr size_t fine_scatter_lds() {
    return (size_t)FS_TPB * 16 * GPT * 2 + (size_t)BPT * 1024 * 4 + (size_t)BPT * 1024 * 8 + 3 * (size_t)FS_TPB * 16 * GPT / 8 + 3 * 64;
}
template <bool PACKED, int GPT, int BPT>
__global__ __launch_bounds__(FS_TPB) void fine_scatter_kernel(const uint32_t *__restrict__ h, const uint16_t *__restrict__ inval,
                                                              const uint32_t *__restrict__ skip, int k, int64_t n, int low_bits,
                                                              unsigned long long *__restrict__ cursor, uint16_t *__restrict__ out) {
    constexpr int KPT = 16 * GPT, FT = FS_TPB * KPT, NB = BPT * 1024, SW = FT / 32, WPW = SW / 16;   // bitmap words, words per wave
    static_assert(FT < 65536, "tile positions and run ranks share a 16 + 16 bit scan word");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint16_t *sorted = reinterpret_cast<uint16_t *>(lds);
    uint32_t *cnt = lds + FT / 2;
    unsigned long long *based = reinterpret_cast<unsigned long long *>(cnt + NB);
    uint2 *pairs = reinterpret_cast<uint2 *>(based + NB);                   // {bitmap word, run starts in the wave's words before it}
    uint32_t *startbits = reinterpret_cast<uint32_t *>(pairs + SW);
    uint32_t *wsum = startbits + SW, *wrank = wsum + 16, *total = wrank + 16;
    static_assert(WPW == 64, "a wave scans one bitmap word per lane and writes out the 2048 positions they cover");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t low_mask = (1u << low_bits) - 1u;
    const int64_t n_tiles = (n + FT - 1) / FT;
    constexpr bool PREF = PACKED && GPT == 2;
    auto g0_of = [&](int64_t tile) { return ((tile * (int64_t)FT) >> 4) + 2 * (int64_t)threadIdx.x; };
    RawPair raw_cur, raw_nxt;
    if constexpr (PREF) raw_pair_load(raw_cur, h, inval, skip, n, g0_of(blockIdx.x));
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int cls = (int)(tile & (FC - 1));
        if constexpr (PREF) raw_pair_load(raw_nxt, h, inval, skip, n, g0_of(tile + gridDim.x));   // next tile's loads in flight
#pragma unroll
        for (int j = 0; j < BPT; ++j) cnt[threadIdx.x * BPT + j] = 0;
        for (int w = threadIdx.x; w < SW; w += FS_TPB) startbits[w] = 0;
        __syncthreads();
        uint32_t v[KPT];
        if constexpr (PREF) raw_pair_keys(raw_cur, n, k, g0_of(tile), v);
        else tile_keys<PACKED, GPT>(h, inval, skip, k, n, tile, v);
        // (taking each key's rank from the returning form of this atomic, so that the placement below is a plain read instead of a
        // second atomic, needs 16 more registers per thread at the 128 this block size allows: spills, 6.8 instead of 5.9 ms)
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if (v[j] != INV32) atomicAdd(&cnt[v[j] >> low_bits], 1u);
        __syncthreads();
        // exclusive scan over the buckets (BPT consecutive ones per thread) of (keys, non-empty buckets) packed as low / high half
        uint32_t c[BPT], s = 0;
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            c[j] = cnt[threadIdx.x * BPT + j];
            s += c[j] + ((c[j] != 0u) << 16);
        }
        uint32_t inc = s;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        uint32_t e = (woff + inc - s) & 0xFFFFu, r = (woff + inc - s) >> 16;
        if (threadIdx.x == FS_TPB - 1) *total = (woff + inc) & 0xFFFFu;
        // the buckets' global reservations are issued here and consumed after the placement phase: their round trips (device-scope
        // returning atomics, four per thread at k = 14) hide behind the LDS work instead of standing between two barriers
        unsigned long long g[BPT];
        uint32_t e0[BPT];
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            const int b = threadIdx.x * BPT + j;
            cnt[b] = e;                                                      // running cursor of the bucket inside the tile
            e0[j] = e;
            // cursor slots are laid out [class][j][thread]: the lanes of a wave reserve in consecutive words (with [class][bucket] a
            // wave's four reservations each touched sixteen lines, four lanes apiece, and the pass took 7.3 instead of ... ms)
            g[j] = atomicAdd(&cursor[(size_t)cls * NB + (size_t)j * FS_TPB + threadIdx.x], (unsigned long long)c[j]);   // c = 0: harmless, rare
            if (c[j]) atomicOr(&startbits[e >> 5], 1u << (e & 31u));
            e += c[j];
        }
        __syncthreads();
        {   // every wave: popcount prefix over its 64 bitmap words (= the 2048 sorted positions it writes out below)
            const uint32_t bits = startbits[wave * WPW + lane];
            const uint32_t pc = (uint32_t)__builtin_popcount(bits);
            uint32_t ps = pc;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(ps, o);
                if (lane >= o) ps += t;
            }
            pairs[wave * WPW + lane] = make_uint2(bits, ps - pc);
            if (lane == 63) wrank[wave] = ps;
        }
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if (v[j] != INV32) sorted[atomicAdd(&cnt[v[j] >> low_bits], 1u)] = (uint16_t)(v[j] & low_mask);
#pragma unroll
        for (int j = 0; j < BPT; ++j)
            if (c[j]) based[r++] = g[j] - e0[j];                             // global index of tile position 0 as seen from this run
        __syncthreads();
        const uint32_t n_valid = *total;
        uint32_t rank0 = 0xFFFFFFFFu;                                         // run starts in front of the wave's positions, minus one
        for (int w = 0; w < wave; ++w) rank0 += wrank[w];
        const uint32_t pm = (2u << (lane & 31)) - 1u;                         // bits of the positions up to the lane's inside its word
#pragma unroll 2
        for (int i = 0; i < WPW / 2; ++i) {                                   // 64 positions (two bitmap words) per step
            const uint32_t P = (uint32_t)wave * (FT / 16) + (uint32_t)i * 64u;
            if (P >= n_valid) break;                                         // wave-uniform
            const uint2 pr = pairs[wave * WPW + 2 * i + (lane >> 5)];
            const uint32_t rank = rank0 + pr.y + (uint32_t)__builtin_popcount(pr.x & pm);
            const uint32_t p = P + (uint32_t)lane;
            if (p < n_valid) out[based[rank] + p] = sorted[p];
        }
        __syncthreads();
        if constexpr (PREF) raw_cur = raw_nxt;
    }
}

// (4) block = work item (a bucket, or a slice of a heavy one): LDS histogram of its keys (a bucket's eight class regions are
// adjacent), then plain stores into the table -- a slice adds its non-empty bins with device atomics instead.
// HALF (low_bits = 16): two 16-bit counters per LDS word; the thread whose returning add sets bit 14 of a counter takes 0x4000 out
// again and notes the bin in the spill list (bounded; applied by fine_spill_kernel).
__global__ __launch_bounds__(256) void fine_zero_heavy_kernel(const uint32_t *__restrict__ plan, int NB, int low_bits, uint32_t *__restrict__ table) {
    const uint32_t n_heavy = plan[1];
    const uint32_t *heavy = plan + 4 + 2 * NB;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (uint32_t h = blockIdx.x; h < n_heavy; h += gridDim.x) {
        uint32_t *dst = table + ((size_t)heavy[h] << low_bits);
        for (uint32_t j = threadIdx.x * 4; j < (1u << low_bits); j += 256 * 4) *reinterpret_cast<u32x4 *>(dst + j) = u32x4{0u, 0u, 0u, 0u};
    }
}
template <bool HALF>
__global__ __launch_bounds__(FS_TPB) void fine_hist_kernel(const uint16_t *__restrict__ keys, const uint64_t *__restrict__ goff,
                                                           const uint32_t *__restrict__ plan, int low_bits,
                                                           uint32_t *__restrict__ table, unsigned long long *__restrict__ spill_n,
                                                           uint32_t *__restrict__ spill, unsigned long long spill_cap) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lb[];
    if (blockIdx.x >= plan[0]) return;
    const uint32_t item = plan[4 + blockIdx.x];
    const uint32_t bucket = item & 0xFFFu;
    const bool heavy = (item & FP_HEAVY) != 0u;
    const uint32_t words = HALF ? (1u << (low_bits - 1)) : (1u << low_bits);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    for (uint32_t j = threadIdx.x * 4; j < words; j += FS_TPB * 4) *reinterpret_cast<u32x4 *>(lb + j) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    uint64_t lo = goff[(size_t)bucket * FC], hi = goff[(size_t)bucket * FC + FC];
    if (heavy) {
        const uint64_t slice_len = ((uint64_t)plan[3] << 32) | plan[2];
        lo += (uint64_t)((item >> 12) & 0xFFFu) * slice_len;
        hi = lo + slice_len < hi ? lo + slice_len : hi;
    }
    const uint32_t bin_base = bucket << low_bits;
    auto count = [&](uint32_t key) {
        if (HALF) {
            const int hs = (int)(key & 1u) * 16;
            const uint32_t old = atomicAdd(&lb[key >> 1], 1u << hs);
            const uint32_t o16 = (old >> hs) & 0xFFFFu;
            if (~o16 & (o16 + 1u) & FH_LIMIT) {                              // this add set bit 14
                atomicSub(&lb[key >> 1], FH_LIMIT << hs);
                const unsigned long long at = atomicAdd(spill_n, 1ull);
                if (at < spill_cap) spill[at] = bin_base + key;
            }
        } else {
            atomicAdd(&lb[key], 1u);
        }
    };
    const uint64_t lo8 = (lo + 7) & ~(uint64_t)7, hi8 = hi & ~(uint64_t)7;
    if (lo8 < hi8) {
        for (uint64_t i = lo + threadIdx.x; i < lo8; i += FS_TPB) count(keys[i]);
        const u32x4 *k8 = reinterpret_cast<const u32x4 *>(keys);            // eight keys per 16-byte load, two loads ahead
        const uint64_t qend = hi8 >> 3, qlast = qend - 1;
        uint64_t q = (lo8 >> 3) + threadIdx.x;
        u32x4 k0 = k8[q < qend ? q : qlast], k1 = k8[q + FS_TPB < qend ? q + FS_TPB : qlast];
        for (; q < qend; q += FS_TPB) {
            const u32x4 kv = k0;
            k0 = k1;
            k1 = k8[q + 2 * FS_TPB < qend ? q + 2 * FS_TPB : qlast];
            count(kv.x & 0xFFFFu); count(kv.x >> 16);
            count(kv.y & 0xFFFFu); count(kv.y >> 16);
            count(kv.z & 0xFFFFu); count(kv.z >> 16);
            count(kv.w & 0xFFFFu); count(kv.w >> 16);
        }
        for (uint64_t i = hi8 + threadIdx.x; i < hi; i += FS_TPB) count(keys[i]);
    } else {
        for (uint64_t i = lo + threadIdx.x; i < hi; i += FS_TPB) count(keys[i]);
    }
    __syncthreads();
    uint32_t *dst = table + ((size_t)bucket << low_bits);
    if (heavy) {                                                             // a slice: the bins it touched, added to the zeroed segment
        for (uint32_t j = threadIdx.x; j < words; j += FS_TPB) {
            const uint32_t w = lb[j];
            if (HALF) {
                if (w & 0xFFFFu) atomicAdd(&dst[2 * j], w & 0xFFFFu);
                if (w >> 16) atomicAdd(&dst[2 * j + 1], w >> 16);
            } else if (w) {
                atomicAdd(&dst[j], w);
            }
        }
    } else if (HALF) {
        for (uint32_t j = threadIdx.x * 2; j < words; j += FS_TPB * 2) {     // two LDS words -> four bins (16 bytes)
            const uint32_t w0 = lb[j], w1 = lb[j + 1];
            *reinterpret_cast<u32x4 *>(dst + 2 * j) = u32x4{w0 & 0xFFFFu, w0 >> 16, w1 & 0xFFFFu, w1 >> 16};
        }
    } else {
        for (uint32_t j = threadIdx.x * 4; j < words; j += FS_TPB * 4) *reinterpret_cast<u32x4 *>(dst + j) = *reinterpret_cast<const u32x4 *>(lb + j);
    }
}
__global__ void fine_spill_kernel(uint32_t *__restrict__ table, const unsigned long long *__restrict__ spill_n,
                                  const uint32_t *__restrict__ spill, unsigned long long spill_cap) {
    unsigned long long m = *spill_n;
    if (m > spill_cap) m = spill_cap;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (unsigned long long)gridDim.x * blockDim.x)
        atomicAdd(&table[spill[i]], FH_LIMIT);
}

template <bool PACKED, int GPT, int BPT>
int fine_scatter_launch(const uint32_t *src, const uint16_t *inval, const uint32_t *skip, int k, int64_t n, int low_bits,
                        unsigned long long *cursor, uint16_t *keys, unsigned grid, hipStream_t st) {
    constexpr size_t lds = fine_scatter_lds<GPT, BPT>();
    KMAP_TRY(kmap_allow_lds((const void *)fine_scatter_kernel<PACKED, GPT, BPT>, (int)lds));
    fine_scatter_kernel<PACKED, GPT, BPT><<<grid, FS_TPB, lds, st>>>(src, inval, skip, k, n, low_bits, cursor, keys);
    return KMAP_OK;
}

template <bool PACKED, int GPT>
int fine_hist(kmap_counts *c, const uint32_t *src, const uint16_t *inval, const uint32_t *skip, int64_t n, int k, hipStream_t st) {
    const int nb_bits = 12, NB = 1 << nb_bits, low_bits = 2 * k - nb_bits;   // 4096 buckets of 4^k / 4096 <= 65 536 bins
    KMAP_REQUIRE(k >= 10 && k <= 14, "counts: fine partition needs 10 <= k <= 14 (k=%d)", k);
    KMAP_TRY(kmap_counts_reserve_bins(c, k));
    const size_t m = (size_t)NB * FC;
    void *small = nullptr;
    KMAP_TRY(kmap_scratch(&small, (m + 1) * 8 + m * 8 + m * 4 + 16 + (4 + (size_t)3 * NB) * 4, st, KMAP_SLOT_A));
    uint64_t *goff = reinterpret_cast<uint64_t *>(small);
    unsigned long long *cursor = reinterpret_cast<unsigned long long *>(goff + m + 1);
    unsigned long long *spill_n = cursor + m;
    uint32_t *gcb = reinterpret_cast<uint32_t *>(spill_n + 2), *plan = gcb + m;   // plan: 4 + 2 NB items + NB heavy ids
    uint16_t *keys = nullptr;
    KMAP_TRY(kmap_scratch((void **)&keys, (size_t)n * 2 + 64, st, KMAP_SLOT_PART));
    KMAP_CHECK_HIP(hipMemsetAsync(spill_n, 0, 16 + m * 4, st));             // spill counter + the (class, bucket) counts
    constexpr int64_t FT = (int64_t)FS_TPB * 16 * GPT;
    int64_t tiles = (n + FT - 1) / FT;
    tiles = (tiles + FC - 1) / FC * FC;                                      // a multiple of 8: the tiles of a block share their class
    const unsigned grid = (unsigned)(tiles > 1024 ? 1024 : tiles);
    KMAP_TRY(kmap_allow_lds((const void *)fine_count_kernel<PACKED, GPT>, (NB + 64) * 4));
    fine_count_kernel<PACKED, GPT><<<grid, FS_TPB, (size_t)(NB + 64) * 4, st>>>(src, inval, skip, k, n, low_bits, NB, gcb);
    fine_offsets_kernel<<<1, FS_TPB, 0, st>>>(gcb, NB, goff, cursor, plan);        // goff[m] = number of valid keys
    KMAP_TRY((fine_scatter_launch<PACKED, GPT, 4>(src, inval, skip, k, n, low_bits, cursor, keys, grid, st)));
    if (low_bits == 16) {
        void *sp = nullptr;
        const size_t cap = (size_t)(n / (int64_t)FH_LIMIT) + 16;
        KMAP_TRY(kmap_scratch(&sp, cap * 4, st, KMAP_SLOT_B));
        uint32_t *spill = reinterpret_cast<uint32_t *>(sp);
        KMAP_TRY(kmap_allow_lds((const void *)fine_hist_kernel<true>, 32768 * 4));
        fine_zero_heavy_kernel<<<64, 256, 0, st>>>(plan, NB, low_bits, c->bins);
        fine_hist_kernel<true><<<(unsigned)(2 * NB), FS_TPB, (size_t)32768 * 4, st>>>(keys, goff, plan, low_bits, c->bins, spill_n, spill, (unsigned long long)cap);
        fine_spill_kernel<<<64, 256, 0, st>>>(c->bins, spill_n, spill, (unsigned long long)cap);
    } else {
        const size_t lds = ((size_t)4 << low_bits) < 16 ? 16 : ((size_t)4 << low_bits);
        KMAP_TRY(kmap_allow_lds((const void *)fine_hist_kernel<false>, (int)lds));
        fine_zero_heavy_kernel<<<64, 256, 0, st>>>(plan, NB, low_bits, c->bins);
        fine_hist_kernel<false><<<(unsigned)(2 * NB), FS_TPB, lds, st>>>(keys, goff, plan, low_bits, c->bins, nullptr, nullptr, 0ull);
    }
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}
}  // namespace

bool kmap_counts_fine_applies(int k) { return k >= 10 && k <= 14; }
int kmap_counts_fine_hist(kmap_counts *c, const uint32_t *hash_dev, const uint32_t *codes_dev, const uint16_t *inval_dev,
                          const uint32_t *skip_dev, int64_t n, int k, hipStream_t st) {
    // two 16-position groups per thread: 32 768-window tiles (three: 24-byte runs at k = 14, but 29 spilled registers -- 6.9 against 6.5 ms)
    if (hash_dev) return fine_hist<false, 2>(c, hash_dev, nullptr, nullptr, n, k, st);
    return fine_hist<true, 2>(c, codes_dev, inval_dev, skip_dev, n, k, st);
}
