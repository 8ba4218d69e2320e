// counts_range.hip -- key-space-sharded counting (11 <= k <= 16, multi-GPU): the front end that turns ALL reads into the dense list of
// the keys ONE rank's key range needs (counts_internal.h: kmap_key_range).
//
// A rank owns positions [lo, lo + len) of the 4^k table.  The entry at position y of the merged table depends on c(y) and c(rc y) only,
// so of a window with k-mer x the rank keeps x - lo when x lies in the range (table T1) or, failing that, half + rc(x) - lo when rc(x)
// does (table T2: the partner counts of the range's positions, already in the partner's place: no reverse-complement transpose of a
// table afterwards) -- about 2 / G of all windows.  The partitioned histogram passes (counts_fine.hip / counts_part.hip) cost per
// (tile, bucket) as much as per key, so running them over all windows with most keys dropped gained nothing (first form of this
// file's job, measured at C3, k = 14, G = 8: 4.4 ms against 5.7 ms for the whole table on one GPU).  Instead ONE pass over the
// packed reads writes the kept virtual keys densely, and the histogram passes of a table of 4^vk <= 2 len bins run on that list
// alone -- everything behind this pass shrinks with the number of ranks.
//   * thread = one 16-position group per step; the 16 k-mers by funnel shifts of the two code words, their reverse complements by
//     funnel shifts of the reverse complement of the whole 32-base pair (two v_bfrev + pair swaps per GROUP, 2 instructions per window);
//   * kept keys meet in an LDS buffer (a fixed quarter per wave, a lane's slot from the ballot of "window i kept" -- no scan, no
//     reservation, no branch per key) and leave as full rows into the
//     block's current 32 768-key chunk of the output; chunks are handed out by one global counter (one device atomic per 32 768 keys),
//     a step's keys may straddle two chunks; a block pads its last chunk with the invalid marker, which the histogram passes skip.
#include "common.h"
#include "counts_internal.h"

namespace {
constexpr int RS_TPB = 256;                            // four waves: many independent blocks per CU hide each other's barriers
constexpr int RS_CHUNK = 32768;                        // keys per output chunk (= the histogram passes' tile)
constexpr uint32_t INV32 = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t rev_pairs(uint32_t w) {      // the sixteen 2-bit groups of w in reverse order
    uint32_t t = __builtin_bitreverse32(w);
    return ((t >> 1) & 0x55555555u) | ((t & 0x55555555u) << 1);
}

// ALIGNED: the range is a block of keys with a common prefix of `t` bits (len a power of two, lo a multiple of it: what an equal split
// over 2 / 4 / 8 ... ranks gives).  "x in range" is then "the first t bits of the window are the prefix" and "rc(x) in range" is "the
// complements of its last bits, read backwards, are" -- both as masks over the group's 16 windows from two 64-bit shifts and two
// v_bitop3 per prefix bit, instead of a subtraction and a compare per window and strand.  Only the kept windows (2 / G of them) are
// hashed: a lane walks the set bits of its mask (the wave as long as its busiest lane: ~8.5 of 16 at G = 8), the slot of a key is
// again "kept so far + kept lanes below".
template <bool ALIGNED>
__global__ __launch_bounds__(RS_TPB) void range_stage_kernel(const uint32_t *__restrict__ codes, const uint16_t *__restrict__ inval,
                                                             const uint32_t *__restrict__ skip, int64_t n, int k, kmap_key_range kr,
                                                             int64_t groups_per_block, uint32_t *__restrict__ out,
                                                             unsigned int *__restrict__ chunk_counter) {
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[17 * RS_TPB];   // the kept keys of one step (4096 windows) + a trash word per thread
    __shared__ unsigned int s_cnt[RS_TPB / 64], s_chunk[2];               // keys each wave staged in this step; current chunk, the next one
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_groups = (n + 15) >> 4;
    const int64_t g_begin = (int64_t)blockIdx.x * groups_per_block;
    int64_t g_end = g_begin + groups_per_block;
    if (g_end > n_groups) g_end = n_groups;
    if (threadIdx.x == 0) s_chunk[0] = atomicAdd(chunk_counter, 1u);
    __syncthreads();
    unsigned int fill = 0;                                                 // keys in the current chunk (block-uniform)
    const uint32_t kmask = k == 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    const int sh = 32 - 2 * k;
    for (int64_t g0 = g_begin; g0 < g_end; g0 += RS_TPB) {
        const int64_t g = g0 + threadIdx.x;
        const int64_t gl = g < g_end ? g : g_end - 1;                      // clamped: every lane loads, lanes behind the range keep nothing
        const uint32_t hi = codes[gl], lo = codes[gl + 1];
        uint64_t bad = ((uint64_t)inval[gl] << 32) | ((uint64_t)inval[gl + 1] << 16) | inval[gl + 2];
        for (int have = 1; have < k;) {
            const int step = (have <= k - have) ? have : k - have;
            bad |= bad << step;
            have += step;
        }
        uint32_t drop16 = (uint32_t)(bad >> 32) & 0xFFFFu;                 // windows 0..15 in bits 15..0
        if (skip) drop16 |= (skip[gl >> 1] >> ((gl & 1) ? 0 : 16)) & 0xFFFFu;
        const int64_t left = n - 16 * gl;
        if (left < 16) drop16 |= left <= 0 ? 0xFFFFu : ((1u << (16 - (int)left)) - 1u);
        if (g >= g_end) drop16 = 0xFFFFu;
        // reverse complement of the 32 bases (hi : lo) as (rhi : rlo): window i's reverse complement is its bits [2 i, 2 i + 2 k)
        const uint32_t rlo = rev_pairs(~hi), rhi = rev_pairs(~lo);
        // Every wave fills its own quarter of the step buffer (no reservation: a wave's slot base is fixed), a lane's slot comes from the
        // ballot of "this window is kept" (run so far + kept lanes below), and every lane stores every time -- a window that is not kept goes
        // to the lane's own trash word behind the buffer -- so the stores need no per-key branch (16 LDS stores per thread and step are ~2 %
        // of the LDS pipe).  The predicates are combined with & and | on purpose: && / || made the compiler materialise every boolean as
        // 0 / 1 in a vector register and compare it again (40 vector instructions per window instead of ~16).
        const bool merge = kr.half != 0u;
        char *const sb = reinterpret_cast<char *>(s_buf);
        const unsigned int trash4 = 4u * (16u * RS_TPB + threadIdx.x);
        const unsigned int run0 = (unsigned int)__builtin_amdgcn_readfirstlane(wave) * (16u * 64u * 4u);
        unsigned int run4 = run0;                                          // byte offset of the wave's next free slot (scalar)
        if constexpr (ALIGNED) {
            // masks with window i at bit 31 - 2 i: bit b of the window's key is stream bit 2 i + b, bit b of its reverse complement
            // the complement of stream bit 2 i + 2 (k - 1 - b / 2) + (b & 1)
            const int p = 31 - __builtin_clz(kr.len), t = 2 * k - p;
            const uint32_t pfx = kr.lo >> p;
            const uint64_t stream = ((uint64_t)hi << 32) | lo;
            uint32_t own_m = 0xAAAAAAAAu, par_m = merge ? 0xAAAAAAAAu : 0u;
            for (int b = 0; b < t; ++b) {                                  // t <= 8, uniform
                const uint32_t cm = 0u - ((pfx >> (t - 1 - b)) & 1u);      // scalar: all ones when the prefix bit is set
                const uint32_t tb = (uint32_t)((stream << b) >> 32);
                const uint32_t ub = (uint32_t)((stream << (2 * (k - 1 - (b >> 1)) + (b & 1))) >> 32);
                own_m &= ~(tb ^ cm);
                par_m &= ub ^ cm;
            }
            uint32_t dead = drop16;                                        // window i: bit 15 - i -> bit 31 - 2 i
            dead = (dead | (dead << 8)) & 0x00FF00FFu;
            dead = (dead | (dead << 4)) & 0x0F0F0F0Fu;
            dead = (dead | (dead << 2)) & 0x33333333u;
            dead = ((dead | (dead << 1)) & 0x55555555u) << 1;
            uint32_t cand = (own_m | par_m) & ~dead;
            // two kept windows per trip, every lane stores both (a lane that has none left: to its trash word) -- no branch inside
            while (__builtin_amdgcn_ballot_w64(cand != 0u)) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t live = cand;
                    const int z = __builtin_clz(cand | 1u);                // 2 i of the lane's next kept window (31: none, bit 0 is never a window)
                    cand &= ~(0x80000000u >> z);
                    const uint32_t x = (uint32_t)((stream << z) >> 32) >> sh;
                    const uint32_t rx = __builtin_amdgcn_alignbit(rhi, rlo, z) & kmask;
                    const uint32_t own = x - kr.lo, par = rx - kr.lo;
                    const bool is_own = own < kr.len;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(live != 0u);   // the masks are exact: a candidate is kept
                    const unsigned int below = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
                    const unsigned int at = (below << 2) + run4;
                    *reinterpret_cast<uint32_t *>(sb + (live != 0u ? at : trash4)) = is_own ? own : kr.half + par;
                    run4 += 4u * (unsigned int)__builtin_popcountll(m);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t x = ((i == 0) ? hi : __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * i)) >> sh;
                const uint32_t rx = ((i == 0) ? rlo : __builtin_amdgcn_alignbit(rhi, rlo, 2 * i)) & kmask;
                const uint32_t own = x - kr.lo, par = rx - kr.lo;
                const bool is_own = own < kr.len, is_par = par < kr.len;
                const bool live = (drop16 & (0x8000u >> i)) == 0u;
                const bool kp = live & (is_own | (merge & is_par));
                const unsigned long long m = __builtin_amdgcn_ballot_w64(kp);
                const unsigned int below = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
                *reinterpret_cast<uint32_t *>(sb + (kp ? (below << 2) + run4 : trash4)) = is_own ? own : kr.half + par;
                run4 += 4u * (unsigned int)__builtin_popcountll(m);
            }
        }
        if (lane == 0) s_cnt[wave] = (run4 - run0) >> 2;
        __syncthreads();
        unsigned int T = 0, toff[RS_TPB / 64];
#pragma unroll
        for (int w = 0; w < RS_TPB / 64; ++w) {
            toff[w] = T;
            T += s_cnt[w];
        }
        const bool straddle = fill + T > (unsigned int)RS_CHUNK;           // block-uniform
        if (threadIdx.x == 0 && straddle) s_chunk[1] = atomicAdd(chunk_counter, 1u);
        __syncthreads();
        const unsigned int cA = s_chunk[0], cB = s_chunk[1];
        uint32_t *const oa = out + (size_t)cA * RS_CHUNK + fill;
        if (!straddle) {                                                   // nearly every step: the keys of the four waves, row by row
#pragma unroll
            for (int w = 0; w < RS_TPB / 64; ++w) {
                const unsigned int tw = s_cnt[w];
                for (unsigned int e = threadIdx.x; e < tw; e += RS_TPB) oa[toff[w] + e] = s_buf[w * (16 * 64) + e];
            }
        } else {
            uint32_t *const ob = out + (size_t)cB * RS_CHUNK;
#pragma unroll
            for (int w = 0; w < RS_TPB / 64; ++w) {
                const unsigned int tw = s_cnt[w];
                for (unsigned int e = threadIdx.x; e < tw; e += RS_TPB) {
                    const unsigned int p = fill + toff[w] + e;
                    if (p < (unsigned int)RS_CHUNK) oa[toff[w] + e] = s_buf[w * (16 * 64) + e];
                    else ob[p - RS_CHUNK] = s_buf[w * (16 * 64) + e];
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && fill + T > (unsigned int)RS_CHUNK) s_chunk[0] = cB;
        fill = fill + T > (unsigned int)RS_CHUNK ? fill + T - RS_CHUNK : fill + T;
    }
    __syncthreads();
    const unsigned int cA = s_chunk[0];
    for (unsigned int p = fill + threadIdx.x; p < (unsigned int)RS_CHUNK; p += RS_TPB) out[(size_t)cA * RS_CHUNK + p] = INV32;
}
}  // namespace

// keys_out (KMAP_SLOT_HASH of the stream's scratch arena): *n_keys uint32 entries, the kept virtual keys of all windows in no
// particular order, padded with 0xFFFFFFFF to whole chunks
int kmap_counts_range_stage(const uint32_t *codes_dev, const uint16_t *inval_dev, const uint32_t *skip_dev, int64_t n, int k, kmap_key_range kr,
                            uint32_t **keys_out, int64_t *n_keys, hipStream_t st) {
    const int64_t n_groups = (n + 15) >> 4;
    int64_t blocks = (n_groups + RS_TPB - 1) / RS_TPB;
    if (blocks > 2048) blocks = 2048;                  // persistent: 8 blocks of 4 waves per CU; a partly filled chunk per block at the end
    if (blocks < 1) blocks = 1;
    const int64_t gpb = ((n_groups + blocks - 1) / blocks + RS_TPB - 1) / RS_TPB * RS_TPB;     // whole steps per block
    // worst case: every window kept (+ a partly filled last chunk per block, + a step's straddle)
    const size_t cap_chunks = (size_t)((n + RS_CHUNK - 1) / RS_CHUNK) + 2 * (size_t)blocks + 1;
    uint32_t *keys = nullptr;
    KMAP_TRY(kmap_scratch((void **)&keys, cap_chunks * RS_CHUNK * 4 + 64, st, KMAP_SLOT_HASH));
    unsigned int *counter = nullptr;
    KMAP_TRY(kmap_scratch((void **)&counter, 64, st, KMAP_SLOT_D));
    KMAP_CHECK_HIP(hipMemsetAsync(counter, 0, 4, st));
    // a range that is a block of keys with a common prefix of 1 .. 8 bits (equal splits over 2, 4, 8 ... ranks): the masked form
    const bool pow2 = kr.len != 0u && (kr.len & (kr.len - 1u)) == 0u && (kr.lo & (kr.len - 1u)) == 0u;
    const int tbits = pow2 ? 2 * k - (31 - __builtin_clz(kr.len)) : 0;
    const char *form = getenv("KMAP_RANGE_STAGE");                          // "plain": the per-window form for every range (tests compare the two)
    if (tbits >= 1 && tbits <= 8 && !(form && !strcmp(form, "plain")))
        range_stage_kernel<true><<<(unsigned)blocks, RS_TPB, 0, st>>>(codes_dev, inval_dev, skip_dev, n, k, kr, gpb, keys, counter);
    else
        range_stage_kernel<false><<<(unsigned)blocks, RS_TPB, 0, st>>>(codes_dev, inval_dev, skip_dev, n, k, kr, gpb, keys, counter);
    KMAP_CHECK_HIP(hipGetLastError());
    unsigned int used = 0;
    KMAP_CHECK_HIP(hipMemcpyAsync(&used, counter, 4, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));
    KMAP_REQUIRE((size_t)used <= cap_chunks, "counts: key-range stage wrote %u chunks into room for %zu", used, cap_chunks);
    *keys_out = keys;
    *n_keys = (int64_t)used * RS_CHUNK;
    return KMAP_OK;
}
