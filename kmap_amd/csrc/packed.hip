// packed.hip -- 2-bit-packed reads resident in HBM and the kernels that work on them.
//
// Packed layout (one "group" = 16 consecutive positions of the reference's uint8 array, kmer_count.py:244-347):
//   codes[g] : uint32, base i of the group in bits [31-2i, 30-2i]   (first base most significant, like the hash)
//   inval[g] : uint16, bit (15-i) set when byte i is not A/C/G/T (255: N or read separator) or lies past the end
// plus two all-invalid halo groups, so every kernel may read groups g, g+1, g+2 unguarded.  0.375 B per position
// instead of 1 B, and a k-mer window is a funnel shift instead of a k-step byte loop.  Masking (mask_input,
// kmer_count.py:580-610) only ever turns positions into 255, i.e. it ORs bits into `inval`: the codes are immutable
// and "restore the unmasked array" (motif_discovery.py:263) is a copy of n/8 bytes.
//
// Kernels: pack / unpack, hash materialisation, histogram (LDS-privatised passes or global atomics) straight from the
// packed stream, Hamming-ball mask (flag + coverage), and the per-read occurrence scan.
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"
#include "counts_internal.h"
#include "scan_internal.h"
#include "scan_util.h"

namespace {

constexpr int BLK = 256;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- window extraction -------------------------------------------------------------------------------------
struct Win {
    uint64_t t0;    // bases 0..31 of the 48-base stream (group g and g+1), base 0 in bits 63:62
    uint32_t c2;    // bases 32..47 (group g+2)
    uint64_t m;     // 48 invalid flags, position 0 in bit 47
};
__device__ __forceinline__ Win load_win(const uint32_t *__restrict__ codes, const uint16_t *__restrict__ inval, int64_t g) {
    Win w;
    const uint32_t c0 = codes[g], c1 = codes[g + 1];
    w.t0 = ((uint64_t)c0 << 32) | c1;
    w.c2 = codes[g + 2];
    w.m = ((uint64_t)inval[g] << 32) | ((uint64_t)inval[g + 1] << 16) | inval[g + 2];
    return w;
}
// hash of the k bases starting at offset i (0..15) of the stream; invalid windows return all ones in the low 2k bits
// (the value the reference's invalid hash has under its "compare like any value" rule); `bad` reports invalidity.
template <bool WIDE>   // WIDE: k may exceed 16 (needs the third group)
__device__ __forceinline__ uint64_t win_hash(const Win &w, int i, int k, uint64_t kmask, bool &bad) {
    uint64_t v = w.t0 << (2 * i);
    if (WIDE && i > 0) v |= (uint64_t)w.c2 >> (32 - 2 * i);
    const uint64_t h = v >> (64 - 2 * k);
    bad = ((w.m >> (48 - i - k)) & ((1ull << k) - 1ull)) != 0;
    return bad ? kmask : h;
}

// ---- pack / unpack -------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack4(uint32_t wd, uint32_t &code, uint32_t &flags) {
    const uint32_t t = wd & 0x03030303u;
    code = ((t << 6) | (t >> 4) | (t >> 14) | (t >> 24)) & 0xFFu;
    uint32_t f = wd & 0xFCFCFCFCu;                 // any of bits 7:2 set -> not a base
    f |= f >> 1; f |= f >> 2; f |= f >> 4;         // smear into bit 0 of every byte (cross-byte smear only goes downward
    f &= 0x01010101u;                              //  from a byte that is itself non-zero, so bit 0 of byte j stays exact)
    flags = ((f << 3) | (f >> 6) | (f >> 15) | (f >> 24)) & 0xFu;
}
__global__ __launch_bounds__(BLK) void pack_kernel(const uint8_t *__restrict__ seq, int64_t n, uint32_t *__restrict__ codes,
                                                   uint16_t *__restrict__ inval, int64_t n_groups, int aligned) {
    const int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (g >= n_groups) return;
    const int64_t p0 = g * 16;
    uint32_t wd[4];
    if (p0 + 16 <= n && aligned) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(seq + p0);
        wd[0] = v.x; wd[1] = v.y; wd[2] = v.z; wd[3] = v.w;
    } else {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int64_t p = p0 + 4 * d + b;
                x |= (uint32_t)((p < n) ? seq[p] : 255u) << (8 * b);
            }
            wd[d] = x;
        }
    }
    uint32_t c = 0, m = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t cd, fd;
        pack4(wd[d], cd, fd);
        c |= cd << (24 - 8 * d);
        m |= fd << (12 - 4 * d);
    }
    codes[g] = c;
    inval[g] = (uint16_t)m;
}
__global__ __launch_bounds__(BLK) void unpack_kernel(const uint32_t *__restrict__ codes, const uint16_t *__restrict__ inval,
                                                     int64_t n, uint8_t *__restrict__ seq) {
    const int64_t p = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (p >= n) return;
    const int64_t g = p >> 4;
    const int i = (int)(p & 15);
    const bool bad = (inval[g] >> (15 - i)) & 1;
    seq[p] = bad ? 255 : (uint8_t)((codes[g] >> (30 - 2 * i)) & 3u);
}

// ---- hash materialisation (one thread per group, 16 hashes, 64/128 contiguous bytes out) -------------------------
// skip bits of per-read de-duplication (dedupe_bitmap_packed_kernel): word w covers positions 32w .. 32w+31, position 32w+j in bit
// 31-j; a set bit = "the k-mer starting here already occurred in its read".  Group g's 16 bits, window i in bit 15-i:
__device__ __forceinline__ uint32_t skip16_of(const uint32_t *__restrict__ skip, int64_t g) {
    return skip ? ((skip[g >> 1] >> ((g & 1) ? 0 : 16)) & 0xFFFFu) : 0u;
}

template <typename H, bool WIDE>
__global__ __launch_bounds__(BLK) void hash_packed_kernel(const uint32_t *__restrict__ codes, const uint16_t *__restrict__ inval,
                                                          int64_t n, int k, H *__restrict__ out, const uint32_t *__restrict__ skip,
                                                          unsigned long long *__restrict__ all_ones) {
    const int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const int64_t p0 = g * 16;
    if (p0 >= n) return;
    const Win w = load_win(codes, inval, g);
    const uint64_t kmask = low_mask<uint64_t>(k);
    const uint32_t sk = skip16_of(skip, g);
    H hs[16];
    uint32_t n_ones = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        bool bad;
        const uint64_t h = win_hash<WIDE>(w, i, k, kmask, bad);
        const bool drop = bad || ((sk >> (15 - i)) & 1u);
        hs[i] = drop ? (H)~(H)0 : (H)h;
        // 16-mers as uint32 (partitioned counting): the all-T k-mer's hash IS the invalid marker -- its valid windows are counted
        // aside (all_ones) and leave the array as invalid
        if (all_ones && !drop && (H)h == (H)~(H)0) ++n_ones;
    }
    if (n_ones) atomicAdd(all_ones, (unsigned long long)n_ones);
    if (p0 + 16 <= n && ((uintptr_t)out % 16) == 0) {
        u32x4 *o = reinterpret_cast<u32x4 *>(out + p0);
        if constexpr (sizeof(H) == 4) {
#pragma unroll
            for (int v = 0; v < 4; ++v) o[v] = u32x4{(uint32_t)hs[4 * v], (uint32_t)hs[4 * v + 1], (uint32_t)hs[4 * v + 2], (uint32_t)hs[4 * v + 3]};
        } else {
#pragma unroll
            for (int v = 0; v < 8; ++v)
                o[v] = u32x4{(uint32_t)hs[2 * v], (uint32_t)((uint64_t)hs[2 * v] >> 32), (uint32_t)hs[2 * v + 1],
                             (uint32_t)((uint64_t)hs[2 * v + 1] >> 32)};
        }
    } else {
        for (int i = 0; i < 16 && p0 + i < n; ++i) out[p0 + i] = hs[i];
    }
}

// ---- histogram straight from the packed stream (no per-read dedupe) ---------------------------------------------
constexpr int HP_BINS = 32768;   // uint32 LDS bins per block (128 KiB)
constexpr int HP_TPB = 1024;
template <bool WIDE, bool LDSMODE>
__global__ __launch_bounds__(LDSMODE ? HP_TPB : BLK) void hist_packed_kernel(const uint32_t *__restrict__ codes,
                                                                             const uint16_t *__restrict__ inval, int64_t n,
                                                                             int k, uint64_t bin0, uint32_t *__restrict__ bins,
                                                                             const uint32_t *__restrict__ skip) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lb[];
    if (LDSMODE) {
        for (int b = threadIdx.x; b < HP_BINS; b += blockDim.x) lb[b] = 0;
        __syncthreads();
    }
    const uint64_t kmask = low_mask<uint64_t>(k);
    const int64_t n_groups = (n + 15) >> 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // software pipeline: the loads of the thread's NEXT group are issued before the 16 windows of the current one are counted
    // (raw registers, nothing derived from them before their turn, the same number of loads on every path -- otherwise the
    // compiler waits for them where they are issued).  Without it every iteration exposed one memory round trip: 360
    // iterations x ~1.6 us = the 0.59 ms a pass took, at 50 % VALU utilisation and 75 % of the wave-cycles waiting (PMC).
    const int64_t g_first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g_last = n_groups - 1;
    uint32_t nc0 = 0, nc1 = 0, nc2 = 0, nsk = 0;
    uint16_t nf0 = 0, nf1 = 0, nf2 = 0;
    const uint32_t *skp = skip ? skip : codes;    // the skip word travels with the windows; without skip bits: any valid word, ignored
    if (n_groups > 0) {
        const int64_t gl = g_first < n_groups ? g_first : g_last;
        nc0 = codes[gl]; nc1 = codes[gl + 1]; nc2 = codes[gl + 2];
        nf0 = inval[gl]; nf1 = inval[gl + 1]; nf2 = inval[gl + 2];
        nsk = skp[gl >> 1];
    }
    const uint32_t dummy = (uint32_t)HP_BINS + (threadIdx.x & 63u);   // LDSMODE: this lane's private bin behind the table
    for (int64_t g = g_first; g < n_groups; g += stride) {
        Win w;
        w.t0 = ((uint64_t)nc0 << 32) | nc1;
        w.c2 = nc2;
        w.m = ((uint64_t)nf0 << 32) | ((uint64_t)nf1 << 16) | nf2;
        const uint32_t sk16 = skip ? ((nsk >> ((g & 1) ? 0 : 16)) & 0xFFFFu) : 0u;   // skip16_of(skip, g)
        {
            const int64_t gn = g + stride < n_groups ? g + stride : g_last;      // clamped: the last round re-reads a valid group
            nc0 = codes[gn]; nc1 = codes[gn + 1]; nc2 = codes[gn + 2];
            nf0 = inval[gn]; nf1 = inval[gn + 1]; nf2 = inval[gn + 2];
            nsk = skp[gn >> 1];
        }
        if ((w.m >> 32) == 0xFFFFull) continue;   // group entirely invalid (cheap skip of masked regions)
        if constexpr (LDSMODE && !WIDE) {
            // 32-bit fast path (k <= 16): window i = bits [63-2i, 64-2i-2k) of t0 -> one v_alignbit + one shift; the 16
            // "window touches an invalid position" flags come from one doubling pass over the 48-bit invalid stream
            // (bit 47-p of `bad` = OR of m[p .. p+k-1]) instead of a 64-bit shift-and-mask per window
            uint64_t bad = w.m;
            for (int have = 1; have < k;) {
                const int step = (have <= k - have) ? have : k - have;
                bad |= bad << step;
                have += step;
            }
            const uint32_t bad16 = (uint32_t)(bad >> 32) | sk16;   // windows 0..15 in bits 15..0 (+ per-read duplicates)
            const uint32_t hi = (uint32_t)(w.t0 >> 32), lo = (uint32_t)w.t0;
            const uint32_t b0 = (uint32_t)bin0;
            const int sh = 32 - 2 * k;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t top = (i == 0) ? hi : __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * i);
                // no branch, no exec mask per window (a pass was bound by instruction issue: ~10 instructions per window, two of
                // them scalar): a window that is invalid (sign-extended flag bit ORed in) or belongs to another pass's bin range
                // lands, by one unsigned min, in the lane's private bin behind the table
                // (key ^ b0) | flag in one v_bitop3: b0 is a multiple of the 32 768 bins of a pass, so inside the pass's range the
                // XOR is the subtraction, and outside it leaves a high bit set
                uint32_t a = __builtin_amdgcn_bitop3_b32(top >> sh, b0, (uint32_t)__builtin_amdgcn_sbfe((int)bad16, 15 - i, 1), 0xBE);
                a = a < dummy ? a : dummy;
                atomicAdd(&lb[a], 1u);
            }
            continue;
        }
        const uint32_t sk = sk16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            bool bad;
            const uint64_t h = win_hash<WIDE>(w, i, k, kmask, bad);
            if (bad || ((sk >> (15 - i)) & 1u)) continue;
            if (LDSMODE) {
                const uint64_t a = h - bin0;
                if (a < (uint64_t)HP_BINS) atomicAdd(&lb[a], 1u);
            } else {
                atomicAdd(&bins[h], 1u);
            }
        }
    }
    if (LDSMODE) {
        __syncthreads();
        for (int b = threadIdx.x; b < HP_BINS; b += blockDim.x) {
            const uint32_t c = lb[b];
            if (c) atomicAdd(&bins[bin0 + b], c);
        }
    }
}

// ---- the same with 16-bit LDS counters: 65 536 bins per pass (k = 8 in ONE pass instead of two, k = 9 in four instead of eight) ----
// Two counters per LDS word, plain (non-returning) adds of 1 or 1 << 16.  A block adds at most 1024 x 16 = 16 384 windows per round of its
// loop; every HP16_ROUNDS = 3 rounds the block meets at a barrier and sweeps the table (32 words per thread): a word with a half at or
// above 0x4000 is emptied (atomic exchange) into the global bins.  A second barrier BEHIND the sweep keeps the waves that finish it early
// from adding the next interval's windows before a slower wave has looked at its words: between two looks at a word exactly one
// interval's adds (at most 49 152) can land on it, so a half stays below 0x4000 + 49 152 = 65 536 and no carry ever reaches the
// neighbour -- also when ONE k-mer takes every window of a block (poly-A: tests/test_gpu_packed.py::test_hist16_single_kmer_no_carry).
// (The returning form of the add with a check of the
// returned word was measured first: 0.97 against 0.94 ms for the two 32-bit passes -- the returned data costs what the second pass did.)
constexpr int HP16_BINS = 65536;
constexpr int HP16_ROUNDS = 3;
__global__ __launch_bounds__(HP_TPB) void hist_packed16_kernel(const uint32_t *__restrict__ codes, const uint16_t *__restrict__ inval,
                                                               int64_t n, int k, uint32_t bin0, uint32_t *__restrict__ bins,
                                                               const uint32_t *__restrict__ skip) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lb[];      // HP16_BINS / 2 words + 64 lane-private words for dropped windows
    for (int b = threadIdx.x; b < HP16_BINS / 2 + 64; b += blockDim.x) lb[b] = 0;
    __syncthreads();
    const int64_t n_groups = (n + 15) >> 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t g_first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g_last = n_groups - 1;
    uint32_t nc0 = 0, nc1 = 0, nsk = 0;
    uint16_t nf0 = 0, nf1 = 0, nf2 = 0;
    const uint32_t *skp = skip ? skip : codes;
    if (n_groups > 0) {
        const int64_t gl = g_first < n_groups ? g_first : g_last;
        nc0 = codes[gl]; nc1 = codes[gl + 1];
        nf0 = inval[gl]; nf1 = inval[gl + 1]; nf2 = inval[gl + 2];
        nsk = skp[gl >> 1];
    }
    const uint32_t dummy = (uint32_t)HP16_BINS + 2u * (threadIdx.x & 63u);   // bin index of the lane's private word (low half)
    const int sh = 32 - 2 * k;
    // every thread of the block runs the same number of rounds (the barriers): a thread behind the last group counts nothing
    const int64_t rounds = (n_groups - (int64_t)blockIdx.x * blockDim.x + stride - 1) / stride;      // of thread 0 = the block's maximum
    auto sweep = [&]() {
        typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
        for (int q = threadIdx.x; q < HP16_BINS / 8; q += HP_TPB) {        // 4 words = 8 bins per step
            const u32x4v v = *reinterpret_cast<const u32x4v *>(lb + 4 * q);
            if (((v.x | v.y | v.z | v.w) & 0xC000C000u) == 0u) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w0 = j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
                if ((w0 & 0xC000C000u) == 0u) continue;
                const uint32_t w = atomicExch(&lb[4 * q + j], 0u);
                const uint32_t b = (uint32_t)(8 * q + 2 * j);
                if (w & 0xFFFFu) atomicAdd(&bins[bin0 + b], w & 0xFFFFu);
                if (w >> 16) atomicAdd(&bins[bin0 + b + 1], w >> 16);
            }
        }
        if (threadIdx.x < 64) lb[HP16_BINS / 2 + threadIdx.x] = 0;          // the private words only absorb: nobody reads them
    };
    int64_t g = g_first;
    for (int64_t r = 0; r < rounds; ++r, g += stride) {
        if (r && r % HP16_ROUNDS == 0) {
            __syncthreads();
            sweep();
            __syncthreads();    // no wave adds for the next interval before every word has been looked at (see above)
        }
        const uint32_t hi = nc0, lo = nc1;
        uint64_t bad = ((uint64_t)nf0 << 32) | ((uint64_t)nf1 << 16) | nf2;
        const uint32_t sk16 = skip ? ((nsk >> ((g & 1) ? 0 : 16)) & 0xFFFFu) : 0u;
        {
            const int64_t gn = g + stride < n_groups ? g + stride : g_last;      // clamped: the last round re-reads a valid group
            nc0 = codes[gn]; nc1 = codes[gn + 1];
            nf0 = inval[gn]; nf1 = inval[gn + 1]; nf2 = inval[gn + 2];
            nsk = skp[gn >> 1];
        }
        if (g >= n_groups || (bad >> 32) == 0xFFFFull) continue;   // behind the array / group entirely invalid
        for (int have = 1; have < k;) {
            const int step = (have <= k - have) ? have : k - have;
            bad |= bad << step;
            have += step;
        }
        const uint32_t bad16 = (uint32_t)(bad >> 32) | sk16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t top = (i == 0) ? hi : __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * i);
            // (key ^ bin0) | flag: inside the pass's range the XOR is the subtraction, outside it (or dropped) a high bit is set
            uint32_t a = __builtin_amdgcn_bitop3_b32(top >> sh, bin0, (uint32_t)__builtin_amdgcn_sbfe((int)bad16, 15 - i, 1), 0xBE);
            a = a < dummy ? a : dummy;
            atomicAdd(lb + (a >> 1), 1u << ((a & 1u) << 4));
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < HP16_BINS / 2; b += blockDim.x) {
        const uint32_t w = lb[b];
        if (w & 0xFFFFu) atomicAdd(&bins[bin0 + 2 * b], w & 0xFFFFu);
        if (w >> 16) atomicAdd(&bins[bin0 + 2 * b + 1], w >> 16);
    }
}

// ---- per-read de-duplication as skip bits (remove_duplicate_hash_per_seq, kmer_count.py:743-760, fused with counting) -----------
// The reference invalidates every repeated hash of a read before counting; only the COUNTS are used afterwards, so which of the
// equal windows survives does not matter.  One wave per read: the windows are hashed straight from the packed codes, 64 at a
// time, and looked up in a per-wave set in LDS.  The duplicates leave as one bit per position (ballot -> two 32-bit words per
// step; words that straddle a read border are shared with the neighbouring read's wave and are ORed atomically, the others are
// plain stores into the zeroed array), which the histogram / hash kernels OR into their invalid-window masks.  No 4-8 B/position
// hash array is written, de-duplicated in place and read back.
// (The round-2 kernel kept the set as an open-addressing table filled by LDS compare-and-swap: 349 scalar + 206 vector
// instructions per read against 18 LDS instructions -- the CU's single scalar unit was the bound, not the LDS atomics:
// wave-uniform values were computed per lane in 64 bits under divergent loops, and every probe round is a dozen mask operations.
// A set without atomics -- store, read back, lanes decide who owns the slot -- was measured too: 3.3 rounds per 64 windows,
// 7.5 ms.  Both are gone; CHANGELOG.md has the numbers.)
constexpr int DS_CAP = 512;             // longest read handled here (longer ones: the hash-array path of kmer_ops.hip)
constexpr int DS_WAVES = 4;
// Here: * the set is a BITMAP in LDS indexed by the k-mer itself (4^k bits, 3 <= k <= 8: exact) or by 15 hashed bits (other k),
//         one returning ds_or per window: the lane that finds its bit clear keeps the k-mer, no probing, no loop;
//       * hashed mode: a lane that finds its bit set is only a candidate (143 windows in 32 768 bits: ~0.3 false positives
//         per read).  Candidates are confirmed exactly, one at a time, against all windows of the read up to this step:
//         duplicate iff another window with the same k-mer claimed a bit, or starts earlier;
//       * waves are persistent (a grid-stride loop over the reads), everything wave-uniform is scalar, positions are 32-bit
//         offsets from the read's first skip word.
// Round 4: the kernel was bound by instruction issue (104 vector + ~80 scalar instructions per read, PMC issue utilisation 0.77),
// so the common read (at most DB_NB = 3 steps of 64 windows, not at the very end of the arrays) now runs a form with ~9 vector
// instructions per step:
//       * the step's loads have NO per-lane address arithmetic: lane l's windows start at offsets 64c + l from a 32-aligned
//         position, so its group index is 4c + (l >> 4) -- a loop-invariant lane offset + an immediate on a scalar base;
//       * the window's bits leave the two code words by ONE 64-bit shift with a per-lane constant; in exact mode the bitmap's word
//         comes from the k-mer's LOW bits and the bit from its high five, so address and bit are one and-or and two shifts of
//         the shifted pair (the k-mer itself is never formed);
//       * "window touches an invalid position" is one AND of the raw flag word with a per-lane constant mask (the window's k
//         flags, rotated into the word's little-endian half order once, outside the loop);
//       * "window starts inside the read" is a scalar 64-bit mask per step (from the read's [lo, hi)), ANDed with the ballot;
//         invalid lanes OR a zero bit into the set (no exec masking, no result register to pre-clear);
//       * the words touched are zeroed by all lanes, valid or not (every other word of the bitmap is zero already);
//       * a read's geometry reaches the scalar registers as TWO v_readlane (its first skip word's index and one packed word), the
//         three base addresses are scalar adds.
//       Reads with more steps, or whose loads could run past the arrays' padding, take the general form below (one step at a
//       time, clamped loads).
constexpr int DB_HASH_WORDS = 1024;     // hashed bitmap: 32 768 bits per wave
constexpr int DB_MAXSTEPS = (DS_CAP + 31 + 63) / 64;
constexpr int DB_NB = 3;                // steps of the fast form
constexpr int DB_TAIL_GROUPS = 4 * DB_NB;   // the fast form loads groups 0 .. 4 DB_NB of the read's frame
struct DbRead {                         // a read as the dedupe kernel sees it (all wave-uniform)
    const uint32_t *crd;                // codes of the group holding the read's first skip word
    const uint16_t *ird;
    uint32_t *srd;                      // the read's first skip word
    int lo, hi;                         // the read's positions as offsets from that word's first position: [lo, hi)
    int gmax;                           // last group (offset) a window of the read starts in
    int nsteps;                         // 64-position steps; 0 = nothing to do
    bool fast;                          // nsteps <= DB_NB and the unclamped loads stay inside the arrays
};
struct DbRaw {
    uint32_t c0, c1;                    // codes of groups g, g + 1
    uint32_t fw;                        // their invalid flags as loaded: f0 | f1 << 16
};
// The read's pointers reach the wave through v_readlane, so the compiler no longer knows them to be global and emits FLAT loads --
// which count in lgkmcnt as well as vmcnt: every wait for an LDS atomic's result then also waited for the window loads prefetched
// for the NEXT read.  The address-space casts make them global_load again.
typedef const __attribute__((address_space(1))) uint32_t *db_g32;
typedef const __attribute__((address_space(1))) uint16_t *db_g16;
typedef uint32_t __attribute__((aligned(2))) db_u32a2;                   // two neighbouring 16-bit flag words as one (2-byte aligned) load
typedef const __attribute__((address_space(1))) db_u32a2 *db_g32a2;
typedef __attribute__((address_space(3))) uint32_t *db_l32;
// general form: any offset, clamped into the read's groups (lanes behind the read come out invalid)
__device__ __forceinline__ void db_load_clamped(const DbRead &g, int o, DbRaw &w) {
    const uint32_t gi = (uint32_t)min(o >> 4, g.gmax);
    const db_g32 crd = (db_g32)g.crd;
    const db_g16 ird = (db_g16)g.ird;
    w.c0 = crd[gi];
    w.c1 = crd[gi + 1];
    w.fw = *(db_g32a2)(ird + gi);
}
__device__ __forceinline__ bool db_window(const DbRead &g, int o, const DbRaw &w, int k, uint32_t kbits, uint64_t kones, uint32_t &h) {
    const int i = o & 15;
    const uint64_t t0 = ((uint64_t)w.c0 << 32) | w.c1;
    const uint32_t fl = (w.fw << 16) | (w.fw >> 16);                      // 32 invalid flags, position 0 in bit 31
    h = (uint32_t)(t0 >> (64 - 2 * i - 2 * k)) & kbits;                   // k <= 16: the window lies in groups g, g + 1
    const bool bad = ((fl >> (32 - i - k)) & (uint32_t)kones) != 0;       // 1 <= 32 - i - k <= 31
    return o >= g.lo && o < g.hi && !bad;
}
template <bool EXACT>
__global__ __launch_bounds__(KMAP_WAVE *DS_WAVES) void dedupe_bitmap_packed_kernel(const uint32_t *__restrict__ codes,
                                                                                   const uint16_t *__restrict__ inval, int64_t n,
                                                                                   const int64_t *__restrict__ borders, int64_t n_seq,
                                                                                   int k, uint32_t *__restrict__ skip, int bw) {
    extern __shared__ uint4 db_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // LDS: the waves' bitmaps first (bw words each, bw a power of two: a wave's bitmap is aligned to its size, so that a word's
    // address is `offset | base`), then the hashed mode's claim masks
    uint32_t *bm = reinterpret_cast<uint32_t *>(db_raw) + (size_t)wave * bw;
    unsigned long long *claims = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint32_t *>(db_raw) + (size_t)DS_WAVES * bw) + (size_t)wave * (DB_MAXSTEPS + 1);
    const uint32_t bm_base = (uint32_t)(uintptr_t)(db_l32)bm;            // LDS byte address of the wave's bitmap
    if (bm_base & ((uint32_t)bw * 4u - 1u)) __builtin_trap();             // (dynamic LDS starts at 0 in this kernel: no static LDS)
    const uint32_t kbits = k < 16 ? (1u << (2 * k)) - 1u : ~0u;
    const uint64_t kones = (1ull << k) - 1ull;
    const int64_t n_waves = (int64_t)gridDim.x * DS_WAVES;
    const int64_t last_group = ((n + 15) >> 4) + 1;                       // the arrays hold at least (n + 15) / 16 + 2 groups (kmap_packed_groups)
    // per-lane constants of the fast form
    const int li = lane & 15;
    const uint32_t lane_g = (uint32_t)lane >> 4;
    const uint32_t sh_h = (uint32_t)(64 - 2 * li - 2 * k);                // k-mer = low 2k bits of (c0:c1) >> sh_h
    uint32_t lane_bad;                                                    // the window's k flags in the raw flag word
    {
        const uint32_t m = (uint32_t)kones << (32 - li - k);
        lane_bad = (m << 16) | (m >> 16);
    }
    const int wbits = EXACT ? 2 * k - 5 : 10;                             // exact: word = low 2k - 5 bits of the k-mer, bit = its high 5
    uint32_t amask = ((1u << wbits) - 1u) << 2;
    asm volatile("v_mov_b32 %0, %0" : "+v"(amask));                       // a vector register: (x & amask) | base is then ONE v_and_or (one scalar operand per instruction)
    uint32_t off[DB_NB];                                                  // the lane's window offset in each step of the frame
#pragma unroll
    for (int c = 0; c < DB_NB; ++c) off[c] = (uint32_t)(64 * c + lane);
    // a wave takes 64 CONSECUTIVE reads per batch -- reads (b n_waves + w) 64 .. + 63 in batch b -- so that the batch's border load
    // is one coalesced KiB and the window loads walk through one contiguous stretch of the packed array
    const int64_t wave_global = (int64_t)blockIdx.x * DS_WAVES + wave;
    int64_t base = wave_global * 64;                                      // first read of the current batch
    if (base >= n_seq) return;
    const int64_t batch_step = n_waves * 64;                              // first read of the wave's next batch - of this one
    {   // the bitmap is zeroed once; after a read the words it touched are zeroed again
        uint4 *b4 = reinterpret_cast<uint4 *>(bm);
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        for (int t = lane; t < bw / 4; t += 64) b4[t] = zero;
    }
    __builtin_amdgcn_wave_barrier();
    // Borders come through VECTOR loads, 64 reads at a time (lane l holds the read this wave handles l iterations into the batch)
    // and reach the scalar registers by v_readlane: as scalar loads they shared the lgkm counter with the LDS atomics.  The
    // geometry of a read is computed by the lane that loaded the borders -- 64 reads per vector instruction.
    struct Batch {
        uint32_t w5;                    // index of the read's first skip word (position >> 5): 2^37 positions
        uint32_t pk;                    // lo | hi << 5 | gmax << 16 | nsteps << 24 | fast << 30 | work << 31
    };
    auto batch = [&](int64_t first, Batch &B) {
        const int64_t rr = first + lane;
        int64_t st = 0, en = 0;
        if (rr < n_seq) {
            st = borders[2 * rr];
            en = borders[2 * rr + 1];
        }
        if (st < 0) st = 0;
        if (en > n) en = n;
        const int64_t a0 = st & ~(int64_t)31;
        const int lo = (int)(st - a0), hi = (int)(en - a0);
        const int gmax = hi > 0 ? (hi - 1) >> 4 : 0;
        const int nsteps = en - st <= 1 ? 0 : (hi + 63) >> 6;             // a read of one window has no duplicate
        const bool fast = nsteps <= DB_NB && (a0 >> 4) + DB_TAIL_GROUPS <= last_group;
        B.w5 = (uint32_t)(a0 >> 5);
        B.pk = (uint32_t)lo | ((uint32_t)hi << 5) | ((uint32_t)gmax << 16) | ((uint32_t)nsteps << 24) | (fast ? 0x40000000u : 0u) |
               (nsteps ? 0x80000000u : 0u);
    };
    // what the wave keeps of a read: two scalars and the prefetched windows of its (up to) DB_NB steps
    struct Rd {
        uint32_t w5, pk;
        DbRaw W[DB_NB];
    };
    auto pick = [&](const Batch &B, int l, Rd &R) {
        R.w5 = (uint32_t)__builtin_amdgcn_readlane((int)B.w5, l);
        R.pk = (uint32_t)__builtin_amdgcn_readlane((int)B.pk, l);
    };
    // the steps' loads, unclamped, on every path the same number (with a load count that depends on a branch the compiler waits
    // with vmcnt(0) before the current read's windows are used, i.e. for the loads just issued); a read that takes the general form
    // loads for itself: its prefetch reads the head of the arrays instead (never used)
    auto prefetch = [&](Rd &R) {
        const uint32_t w5p = (R.pk & 0x40000000u) ? R.w5 : 0u;
        const db_g32 crd = (db_g32)codes + 2 * (size_t)w5p;
        db_g32 crd1 = (db_g32)codes + 1 + 2 * (size_t)w5p;
        asm volatile("" : "+s"(crd1));
        const db_g16 ird = (db_g16)inval + 2 * (size_t)w5p;
#pragma unroll
        for (int c = 0; c < DB_NB; ++c) {
            // two 4-byte code loads on purpose: merged into one 8-byte load the pair arrives as c1:c0 and has to be swapped (a
            // v_pk_mov per step) before the 64-bit shift; `crd1` is `crd + 1` behind an empty asm, so that the compiler cannot
            // see they are neighbours.  Group 4c + (lane >> 4): a loop-invariant lane offset + an immediate on a scalar base.
            const uint32_t gi = lane_g + 4u * c;
            R.W[c].c0 = crd[gi];
            R.W[c].c1 = crd1[gi];
            R.W[c].fw = *(db_g32a2)(ird + gi);
        }
    };
    auto geometry = [&](const Rd &R) -> DbRead {                          // the full geometry: general form and skip-word writes only
        DbRead g;
        g.crd = codes + 2 * (size_t)R.w5;
        g.ird = inval + 2 * (size_t)R.w5;
        g.srd = skip + (size_t)R.w5;
        g.lo = (int)(R.pk & 31u);
        g.hi = (int)((R.pk >> 5) & 2047u);
        g.gmax = (int)((R.pk >> 16) & 255u);
        g.nsteps = (int)((R.pk >> 24) & 63u);
        g.fast = (R.pk & 0x40000000u) != 0;
        return g;
    };
    // duplicates of step cs (lane l -> window 64 cs + l) leave as two skip words; words that straddle a read border are shared with
    // the neighbouring read's wave and are ORed atomically, the others are plain stores into the zeroed array
    auto write_skip = [&](const DbRead &G, int cs, unsigned long long m) {
        if (lane < 2) {
            const uint32_t bits = __builtin_bitreverse32(lane ? (uint32_t)(m >> 32) : (uint32_t)m);   // lane l of the half -> bit 31-l
            const int w0 = cs * 64 + 32 * lane;                           // first position (offset) of this word
            if (bits) {
                typedef __attribute__((address_space(1))) uint32_t *db_gw32;      // global, not flat (see above)
                const db_gw32 sw = (db_gw32)G.srd + (w0 >> 5);
                if (w0 < G.lo || w0 + 32 > G.hi) __hip_atomic_fetch_or(sw, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *sw = bits;
            }
        }
    };
    auto process = [&](const Rd &R) {
        if (!(R.pk & 0x80000000u)) return;                                // no step: nothing to do
        if (R.pk & 0x40000000u) {
            const uint32_t lo = R.pk & 31u, hi = (R.pk >> 5) & 2047u;
            uint32_t addr[DB_NB], bit[DB_NB], old[DB_NB], h[DB_NB];
            unsigned long long vm[DB_NB];                                 // valid windows of the step, as a lane mask
#pragma unroll
            for (int c = 0; c < DB_NB; ++c) {
                const DbRaw &w = R.W[c];
                const uint64_t t0 = ((uint64_t)w.c0 << 32) | w.c1;
                uint32_t x, s;
                if (EXACT) {
                    x = (uint32_t)(t0 >> (sh_h - 2));                     // k-mer << 2 (+ the bases before it above)
                    s = x >> (2 + wbits);                                 // its high five bits (the shift below ignores the rest)
                    h[c] = 0;
                } else {
                    h[c] = (uint32_t)(t0 >> sh_h) & kbits;
                    const uint32_t y = h[c] * 0x9E3779B1u;
                    x = y >> 15;                                          // 10 hashed bits << 2
                    s = y >> 27;
                }
                addr[c] = (x & amask) | bm_base;
                // "starts inside the read" per lane (one compare per step, two in the first): as scalar masks built from lo / hi
                // these were ~9 scalar instructions per step, and the CU's ONE scalar unit was the kernel's bound (94 scalar
                // against 44 vector instructions per read, PMC)
                unsigned long long ok = __ballot((w.fw & lane_bad) == 0) & __ballot(off[c] < hi);
                if (c == 0) ok &= __ballot(off[0] >= lo);
                vm[c] = ok;
                bit[c] = __builtin_amdgcn_inverse_ballot_w64(ok) ? 1u << (s & 31u) : 0u;     // invalid lanes OR a zero into the set
                old[c] = __hip_atomic_fetch_or((db_l32)(uintptr_t)addr[c], bit[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            unsigned long long m[DB_NB], any = 0;
#pragma unroll
            for (int c = 0; c < DB_NB; ++c) {
                m[c] = __ballot((old[c] & bit[c]) != 0);
                any |= m[c];
            }
            if (any) {                                                    // wave-uniform; most reads have no repeated k-mer: one branch per read
                const DbRead G = geometry(R);
                unsigned long long claim[DB_NB];
#pragma unroll
                for (int c = 0; c < DB_NB; ++c) {
                    unsigned long long mc = m[c];
                    if (!EXACT) {
                        claim[c] = vm[c] & ~mc;
                        unsigned long long cand = mc;
                        mc = 0;
                        while (cand) {                                    // scalar loop, rarely entered
                            const int y = __builtin_ctzll(cand);
                            cand &= cand - 1;
                            const uint32_t hc = (uint32_t)__builtin_amdgcn_readlane((int)h[c], y);
                            unsigned long long found = 0;
#pragma unroll
                            for (int c2 = 0; c2 <= c; ++c2) {             // an equal window that claimed its bit, or an earlier one
                                const unsigned long long before = c2 < c ? ~0ull : ((1ull << y) - 1ull);
                                const unsigned long long self = c2 == c ? (1ull << y) : 0ull;
                                found |= __ballot(h[c2] == hc) & vm[c2] & (claim[c2] | before) & ~self;
                            }
                            if (found) mc |= 1ull << y;
                        }
                    }
                    if (mc) write_skip(G, c, mc);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < DB_NB; ++c) *(db_l32)(uintptr_t)addr[c] = 0u;   // all lanes: the other words are zero anyway
            __builtin_amdgcn_wave_barrier();
            return;
        }
        // general form: one step at a time, clamped loads
        const DbRead G = geometry(R);
        for (int cs = 0; cs < G.nsteps; ++cs) {
            DbRaw w;
            db_load_clamped(G, cs * 64 + lane, w);
            uint32_t hh;
            const bool valid = db_window(G, cs * 64 + lane, w, k, kbits, kones, hh);
            const uint32_t idx = EXACT ? hh : (hh * 0x9E3779B1u) >> 17;
            const uint32_t b = 1u << (idx & 31);
            uint32_t o = 0;
            if (valid) o = atomicOr(&bm[idx >> 5], b);
            const bool saw_set = valid && (o & b);
            unsigned long long m = __ballot(saw_set);
            if (!EXACT) {
                const unsigned long long cl = __ballot(valid && !saw_set);
                if (lane == 0) claims[cs] = cl;
                __builtin_amdgcn_wave_barrier();
                unsigned long long cand = m;
                m = 0;
                while (cand) {
                    const int y = __builtin_ctzll(cand);
                    cand &= cand - 1;
                    const uint32_t hc = (uint32_t)__builtin_amdgcn_readlane((int)hh, y);
                    const int py = cs * 64 + y;
                    bool found = false;
                    for (int c2 = 0; c2 <= cs && !found; ++c2) {
                        const int o2 = c2 * 64 + lane;
                        DbRaw w2;
                        db_load_clamped(G, o2, w2);
                        uint32_t h2;
                        const bool v2 = db_window(G, o2, w2, k, kbits, kones, h2);
                        const unsigned long long cl2 = claims[c2];
                        found = __any(v2 && h2 == hc && o2 != py && (((cl2 >> lane) & 1ull) || o2 < py));
                    }
                    if (found) m |= 1ull << y;
                }
            }
            if (m) write_skip(G, cs, m);
        }
        __builtin_amdgcn_wave_barrier();
        uint4 *b4 = reinterpret_cast<uint4 *>(bm);
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        for (int t = lane; t < bw / 4; t += 64) b4[t] = zero;
        __builtin_amdgcn_wave_barrier();
    };
    // (Measured and dropped in r04: three register sets with the loads two reads ahead -- 1.30 against 1.27 ms, the waits are not
    // on the window loads; and, on top of that, the next read's bitmap words / bits / masks computed between the issue of this
    // read's LDS atomics and the use of their results -- 1.41 ms: the prepared state travels through 18 more registers and the
    // compiler's copies of it cost more than the LDS round trip they hide.)
    // Two-deep software pipeline over the wave's reads, unrolled by two (reads alternate between A and B: no register copies from
    // "next" to "current"): while a read is inserted, the window loads of the next one and the border load of the next BATCH are
    // in flight.  A batch holds 64 reads (an even number) unless it is the wave's last one.
    Batch cur, nxt;
    batch(base, cur);
    batch(base + batch_step, nxt);
    int cnt = (int)(n_seq - base < 64 ? n_seq - base : 64);               // reads of the current batch
    Rd A, B;
    pick(cur, 0, A);
    prefetch(A);
    int l = 0;                                                            // A's index in the batch (even)
    for (;;) {
        const bool in_batch = l + 1 < cnt;                                // otherwise: an odd count, i.e. the wave's last batch ends with A
        if (in_batch) pick(cur, l + 1, B);
        else B = A;
        prefetch(B);
        process(A);
        if (!in_batch) break;
        l += 2;
        bool more = true;
        if (l >= cnt) {                                                   // the batch ends with B
            base += batch_step;
            if (base >= n_seq) more = false;
            else {
                cur = nxt;
                batch(base + batch_step, nxt);
                cnt = (int)(n_seq - base < 64 ? n_seq - base : 64);
                l = 0;
            }
        }
        if (more) pick(cur, l, A);
        else A = B;
        prefetch(A);
        process(B);
        if (!more) break;
    }
}
// (r03 built and measured a quarter-wave form -- 16 lanes per read, four reads per wave in lock-step, 143 / 160 lane utilisation,
// the read geometry paid once per four reads, each read's set an open-addressing table of its k-mers filled by LDS
// compare-and-swap, window words fetched one iteration ahead: correct on every counting test, 5.2 ms at C3 against this
// kernel's 3.0 ms.  The returning CAS and its divergent probe loop cost more than the lane utilisation gains -- the same
// finding as for r02's first CAS-set kernel; an exact bitmap per read (8 KiB at k = 8) does not fit four reads per wave at a
// useful occupancy.  Dropped.)
__global__ __launch_bounds__(BLK) void max_read_len_kernel(const int64_t *__restrict__ borders, int64_t n_seq, int64_t n,
                                                           unsigned long long *__restrict__ out) {
    unsigned long long m = 0;
    for (int64_t s = (int64_t)blockIdx.x * BLK + threadIdx.x; s < n_seq; s += (int64_t)gridDim.x * BLK) {
        int64_t st = borders[2 * s], en = borders[2 * s + 1];
        if (st < 0) st = 0;
        if (en > n) en = n;
        if (en - st > (int64_t)m) m = (unsigned long long)(en - st);
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_down(m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// ---- Hamming-ball mask on the packed stream (mask_input, kmer_count.py:580-610) ---------------------------------------
struct ConsTabP {
    uint64_t cons[32];
    int32_t radius[32];
    int n;
};
// hit16[g]: bit (15-i) set when the window at position 16g+i (invalid = all ones, compared as is) is within radius of
// any consensus.  Reads the CURRENT invalid mask; the coverage pass below writes it.
// k > 16 only: k <= 16 is tested bit-sliced on the reads' bit planes (bitslice.hip)
__global__ __launch_bounds__(BLK) void mask_flag_packed_kernel(const uint32_t *__restrict__ codes,
                                                               const uint16_t *__restrict__ inval, int64_t n, int k,
                                                               ConsTabP t, uint16_t *__restrict__ hit16) {
    const int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const int64_t n_groups = (n + 15) >> 4;
    if (g >= n_groups) return;
    const Win w = load_win(codes, inval, g);
    const uint64_t kmask = low_mask<uint64_t>(k);
    uint32_t hits = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        bool bad;
        const uint64_t h = win_hash<true>(w, i, k, kmask, bad);
        bool f = false;
        for (int c = 0; c < t.n; ++c) f |= (popc2((h ^ t.cons[c]) & kmask) <= t.radius[c]);
        if (16 * g + i >= n) f = false;           // positions past the end do not exist
        hits |= (uint32_t)f << (15 - i);
    }
    hit16[g] = (uint16_t)hits;
}
// position q becomes invalid when a hit starts in [q-k+1, q]; k <= 31 reaches at most two groups back
__device__ __forceinline__ uint32_t cover16(uint64_t h2, uint64_t h1, uint64_t h0, int k) {
    // 48-bit stream of hits: groups g-2, g-1, g (position 0 of g-2 in bit 47); cover = OR_{j=0}^{k-1} (s >> j) by doubling
    uint64_t cover = (h2 << 32) | (h1 << 16) | h0;
    int have = 1;
    while (have < k) {
        const int step = (have <= k - have) ? have : k - have;
        cover |= cover >> step;
        have += step;
    }
    return (uint32_t)(cover & 0xFFFFull);
}
// thread = four groups (one 8-byte load of hits, one of the mask, one store); the hit array of a consensus batch starts 8-byte
// aligned.  (One group per thread moved two bytes per lane and access: 0.42 ms for 0.57 GB at C3.)
__global__ __launch_bounds__(BLK) void mask_cover_packed_kernel(const uint16_t *__restrict__ hit16, int64_t n, int k,
                                                                uint16_t *__restrict__ inval) {
    const int64_t g0 = ((int64_t)blockIdx.x * BLK + threadIdx.x) * 4;
    const int64_t n_groups = (n + 15) >> 4;
    if (g0 >= n_groups) return;
    if (g0 + 4 <= n_groups) {
        const uint64_t hq = *reinterpret_cast<const uint64_t *>(hit16 + g0);
        const uint32_t hp = g0 ? *reinterpret_cast<const uint32_t *>(hit16 + g0 - 2) : 0u;   // groups g0 - 2 (low half), g0 - 1
        const uint64_t h[6] = {hp & 0xFFFFu, hp >> 16, hq & 0xFFFFull, (hq >> 16) & 0xFFFFull, (hq >> 32) & 0xFFFFull, hq >> 48};
        uint64_t add = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) add |= (uint64_t)cover16(h[j], h[j + 1], h[j + 2], k) << (16 * j);
        if (add) {
            uint64_t *p = reinterpret_cast<uint64_t *>(inval + g0);
            *p |= add;
        }
        return;
    }
    for (int64_t g = g0; g < n_groups; ++g) {
        const uint64_t h2 = (g >= 2) ? hit16[g - 2] : 0, h1 = (g >= 1) ? hit16[g - 1] : 0, h0 = hit16[g];
        const uint16_t add = (uint16_t)cover16(h2, h1, h0, k);
        if (add) inval[g] = (uint16_t)(inval[g] | add);
    }
}

// ---- occurrence scan on the packed stream (get_motif_occurence, motif_discovery.py:1422-1477) ------------------------
constexpr int SC_WAVES = 4;
__device__ __forceinline__ int64_t slice_stop(int64_t L, int k) {
    int64_t stop = L - k + 1;
    if (stop < 0) {
        stop += L;
        if (stop < 0) stop = 0;
    }
    return stop > L ? L : stop;
}
__device__ __forceinline__ int pos_dist(const uint32_t *__restrict__ codes, const uint16_t *__restrict__ inval, int64_t p,
                                        int k, uint64_t kmask, uint64_t cons, uint64_t rcc, int revcom) {
    const Win w = load_win(codes, inval, p >> 4);
    bool bad;
    const uint64_t h = win_hash<true>(w, (int)(p & 15), k, kmask, bad);
    int d = popc2((h ^ cons) & kmask);
    if (revcom) {
        const int d2 = popc2((h ^ rcc) & kmask);
        d = d2 < d ? d2 : d;
    }
    return d;
}
template <bool WRITE>
__global__ __launch_bounds__(KMAP_WAVE *SC_WAVES) void scan_packed_kernel(const uint32_t *__restrict__ codes,
                                                                           const uint16_t *__restrict__ inval, int64_t n,
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
    const int64_t stop = slice_stop(L, k);
    const uint64_t kmask = low_mask<uint64_t>(k);
    // the read's own end acts like a separator even if the caller's border does not sit on one
    constexpr int REG = 4;                       // distances kept in registers for reads up to 256 positions
    int dreg[REG];
    int best = 1 << 30;
#pragma unroll
    for (int r = 0; r < REG; ++r) {
        const int64_t p = (int64_t)r * 64 + lane;
        int d = 1 << 29;
        if (p < stop) {
            d = (p + k > L) ? popc2((kmask ^ cons) & kmask) : pos_dist(codes, inval, st + p, k, kmask, cons, rcc, revcom);
            if (p + k > L && revcom) { const int d2 = popc2((kmask ^ rcc) & kmask); d = d2 < d ? d2 : d; }
        }
        dreg[r] = d;
        if (d <= radius && d < best) best = d;
    }
    for (int64_t p = (int64_t)REG * 64 + lane; p < stop; p += 64) {
        int d = (p + k > L) ? popc2((kmask ^ cons) & kmask) : pos_dist(codes, inval, st + p, k, kmask, cons, rcc, revcom);
        if (p + k > L && revcom) { const int d2 = popc2((kmask ^ rcc) & kmask); d = d2 < d ? d2 : d; }
        if (d <= radius && d < best) best = d;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int v = __shfl_xor(best, o);
        best = v < best ? v : best;
    }
    int count = 0;
    uint64_t base = WRITE ? offs[s] : 0;
    if (best <= radius) {
        for (int64_t p0 = 0; p0 < stop; p0 += 64) {
            const int64_t p = p0 + lane;
            int d;
            if (p0 < (int64_t)REG * 64) d = dreg[p0 >> 6];
            else {
                d = 1 << 29;
                if (p < stop) {
                    d = (p + k > L) ? popc2((kmask ^ cons) & kmask) : pos_dist(codes, inval, st + p, k, kmask, cons, rcc, revcom);
                    if (p + k > L && revcom) { const int d2 = popc2((kmask ^ rcc) & kmask); d = d2 < d ? d2 : d; }
                }
            }
            const bool hit = (p < stop) && (d == best);
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


// ---- occurrence scan, flat formulation ---------------------------------------------------------------------------
// The wave-per-read kernel above spends its time on per-read latency chains (borders -> codes -> reduce -> ballot): 10^7
// waves of ~3 positions per lane.  Split instead into
// (k > 16 only: k <= 16 scans the bit-sliced hit words of bitslice.hip.)
//   (1) a flat pass, thread per 16-position group, that stores the capped distance of EVERY window as a nibble
//       (d <= radius ? d : 15; 8 B per group = 0.5 B per position) -- independent of read borders because a window that
//       the scan may use (p < L-k+1) lies entirely inside its read;
//       plus the smallest nibble of every group as one byte;
//   (2) a thread-per-read pass: minimum over the read = its two boundary words (masked) and the group minima of the words
//       in between (consecutive threads read consecutive bytes); then the number of positions at that minimum, decoding
//       only the words whose group minimum equals it;
//   (3) after the scan of the counts, a thread-per-read pass that writes those positions in ascending order.
// Reads longer than FL_LONG positions are handled by their whole wave (64 words per step) inside (2) and (3).
// Needs radius <= 14; larger radii take the wave-per-read kernel.
constexpr int FL_TPB = 256;
constexpr int FL_LONG = 1024;
constexpr uint64_t NIB_ONES = 0x1111111111111111ull;

__device__ __forceinline__ int nib_min(uint64_t x) {
    // pairwise minimum of the 16 nibbles (SWAR: compare 8 nibble pairs held in separate bytes, then fold)
    int m = 15;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int v = (int)((x >> (4 * i)) & 15);
        m = v < m ? v : m;
    }
    return m;
}
__global__ __launch_bounds__(BLK) void scan_nibble_kernel(const uint32_t *__restrict__ codes, const uint16_t *__restrict__ inval,
                                                          int64_t n, int k, uint64_t cons, uint64_t rcc, int radius, int revcom,
                                                          uint64_t *__restrict__ nib, uint8_t *__restrict__ wmin) {
    const int64_t g = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (g >= ((n + 15) >> 4)) return;
    const Win w = load_win(codes, inval, g);
    const uint64_t kmask = low_mask<uint64_t>(k);
    uint64_t out = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        bool bad;
        const uint64_t h = win_hash<true>(w, i, k, kmask, bad);
        int d = popc2((h ^ cons) & kmask);
        if (revcom) {
            const int d2 = popc2((h ^ rcc) & kmask);
            d = d2 < d ? d2 : d;
        }
        out |= (uint64_t)(d <= radius ? d : 15) << (4 * i);
    }
    nib[g] = out;
    wmin[g] = (uint8_t)nib_min(out);   // smallest nibble of the word: the per-read passes skip words that cannot matter
}

// nibbles of word wi restricted to absolute positions [a, b): everything else reads as 15.  `src` is the nibble array
// shifted so that src[wi - wsh] is word wi (global array: wsh = 0; block-staged LDS copy: wsh = first staged word).
__device__ __forceinline__ uint64_t nib_load(const uint64_t *src, int64_t wsh, int64_t wi, int64_t a, int64_t b) {
    uint64_t x = src[wi - wsh];
    const int64_t w0 = wi << 4;
    if (a > w0) x |= (1ull << (4 * (int)(a - w0))) - 1ull;
    if (b < w0 + 16) x |= ~0ull << (4 * (int)(b - w0));
    return x;
}
// 16-bit mask (bit i = position i of the word) of the nibbles equal to v
__device__ __forceinline__ uint32_t nib_eq_mask(uint64_t x, int v) {
    uint64_t y = x ^ (NIB_ONES * (uint64_t)v);         // zero nibble <=> equal
    y |= y >> 1;
    y |= y >> 2;
    y = ~y & NIB_ONES;                                 // bit 4i set <=> nibble i equal
    y = (y | (y >> 3)) & 0x0303030303030303ull;        // gather: 2 bits per byte
    y = (y | (y >> 6)) & 0x000F000F000F000Full;        // 4 bits per 16
    y = (y | (y >> 12)) & 0x000000FF000000FFull;       // 8 bits per 32
    return (uint32_t)((y | (y >> 24)) & 0xFFFFull);
}

template <bool WRITE>
__global__ __launch_bounds__(FL_TPB) void scan_reads_kernel(const uint64_t *__restrict__ nib, int64_t n,
                                                            const int64_t *__restrict__ borders, int64_t n_seq, int k, int d_inv,
                                                            int radius, int32_t *__restrict__ hits, int8_t *__restrict__ min_dist,
                                                            const uint64_t *__restrict__ offs, int32_t *__restrict__ pos_out,
                                                            const uint8_t *__restrict__ wmin) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t st = 0, stop = 0;
    bool quirk = false;
    if (s < n_seq) {
        st = borders[2 * s];
        int64_t en = borders[2 * s + 1];
        if (st < 0) st = 0;
        if (en > n) en = n;
        const int64_t L = en > st ? en - st : 0;
        quirk = (L - k + 1 < 0);                 // negative slice stop (motif_discovery.py:1443): every window runs off the read
        stop = slice_stop(L, k);
    }
    int best = 15, count = 0;
    uint64_t base = 0;
    if (WRITE && s < n_seq) {
        count = hits[s];
        best = min_dist[s];
        base = offs[s];
        if (count == 0) stop = 0;                // nothing to write for this read
    }
    if (quirk) {
        if (!WRITE) {
            best = d_inv <= radius ? d_inv : 15;
            count = d_inv <= radius ? (int)stop : 0;
        } else {
            for (int64_t p = 0; p < stop; ++p) pos_out[base + p] = (int32_t)p;
        }
        stop = 0;
    }
    const bool is_long = stop > FL_LONG;
    if (stop > 0 && !is_long) {
        const int64_t a = st, b = st + stop;
        const int64_t w0 = a >> 4, w1 = (b - 1) >> 4;
        auto edge = [&](int64_t wi) -> uint64_t {            // boundary word: nibbles outside [a, b) read as 15
            uint64_t x = nib[wi];
            const int64_t p0 = wi << 4;
            if (a > p0) x |= (1ull << (4 * (int)(a - p0))) - 1ull;
            if (b < p0 + 16) x |= ~0ull << (4 * (int)(b - p0));
            return x;
        };
        // interior words are judged by their precomputed minimum (1 byte, consecutive threads read consecutive bytes);
        // only the two boundary words and the words that hold the read's minimum are decoded
        const uint64_t xa = edge(w0), xb = (w1 > w0) ? edge(w1) : ~0ull;
        if (!WRITE) {
            best = min(nib_min(xa), nib_min(xb));
            for (int64_t wi = w0 + 1; wi < w1; ++wi) best = min(best, (int)wmin[wi]);
            if (best < 15) {
                count = __builtin_popcount(nib_eq_mask(xa, best)) + __builtin_popcount(nib_eq_mask(xb, best));
                for (int64_t wi = w0 + 1; wi < w1; ++wi)
                    if (wmin[wi] == best) count += __builtin_popcount(nib_eq_mask(nib[wi], best));
            }
        } else {
            auto emit = [&](uint64_t x, int64_t wi) {
                uint32_t m = nib_eq_mask(x, best);
                while (m) {
                    const int i = __builtin_ctz(m);
                    m &= m - 1;
                    pos_out[base++] = (int32_t)((wi << 4) + i - st);
                }
            };
            emit(xa, w0);
            for (int64_t wi = w0 + 1; wi < w1; ++wi)
                if (wmin[wi] == best) emit(nib[wi], wi);
            if (w1 > w0) emit(xb, w1);
        }
    }
    // long reads: the whole wave works on one read at a time, 64 words per step
    unsigned long long todo = __ballot(is_long);
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int64_t a = __shfl(st, src), b = a + __shfl(stop, src);
        const int64_t w0 = a >> 4, w1 = (b - 1) >> 4;
        if (!WRITE) {
            int m = 15;
            for (int64_t wi = w0 + lane; wi <= w1; wi += 64) {
                const int v = nib_min(nib_load(nib, 0, wi, a, b));
                m = v < m ? v : m;
            }
            for (int o = 32; o > 0; o >>= 1) {
                const int v = __shfl_xor(m, o);
                m = v < m ? v : m;
            }
            int c = 0;
            if (m < 15)
                for (int64_t wi = w0 + lane; wi <= w1; wi += 64) c += __builtin_popcount(nib_eq_mask(nib_load(nib, 0, wi, a, b), m));
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
            if (lane == src) {
                best = m;
                count = c;
            }
        } else {
            const int bst = __shfl(best, src);
            uint64_t wbase = __shfl(base, src);
            for (int64_t c0 = w0; c0 <= w1; c0 += 64) {
                const int64_t wi = c0 + lane;
                uint32_t m = (wi <= w1) ? nib_eq_mask(nib_load(nib, 0, wi, a, b), bst) : 0u;
                const int c = __builtin_popcount(m);
                int inc = c;
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(inc, o);
                    if (lane >= o) inc += v;
                }
                uint64_t at = wbase + (uint64_t)(inc - c);
                while (m) {
                    const int i = __builtin_ctz(m);
                    m &= m - 1;
                    pos_out[at++] = (int32_t)((wi << 4) + i - a);
                }
                wbase += (uint64_t)__shfl(inc, 63);
            }
        }
    }
    if (!WRITE && s < n_seq) {
        hits[s] = count;
        min_dist[s] = (int8_t)(best < 15 ? best : -1);
    }
}

static inline unsigned grid_for(int64_t n, int64_t per) {
    int64_t g = (n + per - 1) / per;
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace


extern "C" {

// data groups + two all-invalid halo groups, rounded up to an EVEN number of groups: kernels that take group pairs (bitslice.hip:
// thread = groups 2t .. 2t + 3) read whole aligned pairs without a bounds test
int64_t kmap_packed_groups(int64_t n) { return ((((n + 15) >> 4) + 2) + 1) & ~(int64_t)1; }

int kmap_pack_reads_dev(const uint8_t *seq_dev, int64_t n, uint32_t *codes_dev, uint16_t *inval_dev, void *stream) {
    KMAP_REQUIRE(n >= 0 && (n == 0 || seq_dev) && codes_dev && inval_dev, "pack_reads: bad arguments");
    const int64_t ng = kmap_packed_groups(n);
    pack_kernel<<<grid_for(ng, BLK), BLK, 0, as_stream(stream)>>>(seq_dev, n, codes_dev, inval_dev, ng,
                                                                   ((uintptr_t)seq_dev % 16) == 0);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

int kmap_unpack_reads_dev(const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n, uint8_t *seq_out_dev, void *stream) {
    KMAP_REQUIRE(n >= 0 && codes_dev && inval_dev && (n == 0 || seq_out_dev), "unpack_reads: bad arguments");
    if (n == 0) return KMAP_OK;
    unpack_kernel<<<grid_for(n, BLK), BLK, 0, as_stream(stream)>>>(codes_dev, inval_dev, n, seq_out_dev);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

int kmap_hash_kmers_packed_dev(const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n, int k, void *out_dev,
                               void *stream) {
    KMAP_REQUIRE(k > 0 && k < 32, "hash_kmers_packed: k=%d out of range", k);
    KMAP_REQUIRE(n >= 0 && codes_dev && inval_dev && (n == 0 || out_dev), "hash_kmers_packed: bad arguments");
    if (n == 0) return KMAP_OK;
    const unsigned g = grid_for((n + 15) >> 4, BLK);
    hipStream_t st = as_stream(stream);
    if (k < 16) hash_packed_kernel<uint32_t, false><<<g, BLK, 0, st>>>(codes_dev, inval_dev, n, k, (uint32_t *)out_dev, nullptr, nullptr);
    else if (k == 16) hash_packed_kernel<uint64_t, false><<<g, BLK, 0, st>>>(codes_dev, inval_dev, n, k, (uint64_t *)out_dev, nullptr, nullptr);
    else hash_packed_kernel<uint64_t, true><<<g, BLK, 0, st>>>(codes_dev, inval_dev, n, k, (uint64_t *)out_dev, nullptr, nullptr);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

namespace {
// skip bits of the per-read de-duplication for k <= 16; *skip_out stays null when some read is longer than DS_CAP (the caller
// then takes the hash-array path, which handles any length)
int dedupe_skip_bits(const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n, const int64_t *borders_dev, int64_t n_seq, int k,
                     hipStream_t st, uint32_t **skip_out) {
    *skip_out = nullptr;
    if (n_seq == 0 || n == 0) return KMAP_OK;
    // arrays shorter than one read frame of the kernel's unclamped prefetch (13 groups = 208 positions): the hash-array path
    if (((n + 15) >> 4) + 2 <= DB_TAIL_GROUPS) return KMAP_OK;
    const size_t words = (size_t)((n + 31) >> 5) + 4;
    uint32_t *skip = nullptr;
    KMAP_TRY(kmap_scratch((void **)&skip, words * 4 + 16, st, KMAP_SLOT_C));
    unsigned long long *mx = (unsigned long long *)(skip + ((words + 1) & ~(size_t)1));
    KMAP_CHECK_HIP(hipMemsetAsync(skip, 0, words * 4 + 16, st));
    max_read_len_kernel<<<1024, BLK, 0, st>>>(borders_dev, n_seq, n, mx);
    unsigned long long max_len = 0;
    KMAP_CHECK_HIP(hipMemcpyAsync(&max_len, mx, 8, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));
    if (max_len > (unsigned long long)DS_CAP) return KMAP_OK;
    const bool exact = k >= 3 && k <= 8;                                  // 4^k bits fit the per-wave bitmap (and a k-mer has the 5 bits that pick the bit)
    const int bw = exact ? std::max(4, (int)((1u << (2 * k)) >> 5)) : DB_HASH_WORDS;   // a power of two (the kernel ORs a word's offset into the bitmap's base)
    const size_t lds = (size_t)DS_WAVES * bw * 4 + (exact ? 0 : (size_t)DS_WAVES * (DB_MAXSTEPS + 1) * 8);
    // persistent grid = exactly the blocks that are resident at once (one more would run as a second round)
    int dev = 0, cus = 0, per_cu = 0;
    KMAP_CHECK_HIP(hipGetDevice(&dev));
    KMAP_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (exact) KMAP_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dedupe_bitmap_packed_kernel<true>, KMAP_WAVE * DS_WAVES, lds));
    else KMAP_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dedupe_bitmap_packed_kernel<false>, KMAP_WAVE * DS_WAVES, lds));
    per_cu = std::max(1, std::min(per_cu, (int)((size_t)(156 << 10) / lds)));   // the query says 5 x 32 KiB fit a CU; measured: 4 do (the fifth block runs as a second round, 6.3 -> 8.2 ms)
    const unsigned pgrid = (unsigned)std::min<int64_t>((n_seq + DS_WAVES - 1) / DS_WAVES, (int64_t)cus * per_cu);
    if (exact) dedupe_bitmap_packed_kernel<true><<<pgrid, KMAP_WAVE * DS_WAVES, lds, st>>>(codes_dev, inval_dev, n, borders_dev, n_seq, k, skip, bw);
    else dedupe_bitmap_packed_kernel<false><<<pgrid, KMAP_WAVE * DS_WAVES, lds, st>>>(codes_dev, inval_dev, n, borders_dev, n_seq, k, skip, bw);
    KMAP_CHECK_HIP(hipGetLastError());
    *skip_out = skip;
    return KMAP_OK;
}
}  // namespace

// fills c->bins (zeroed first) with the k-mer histogram of the packed reads; k <= 16
int kmap_counts_hist_packed_dev(kmap_counts *c, const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n,
                                const int64_t *borders_dev, int64_t n_seq, int k, int dedupe_per_read, void *stream) {
    KMAP_REQUIRE(c, "counts_hist_packed: null handle");
    KMAP_REQUIRE(k > 0 && k <= 16, "counts_hist_packed: k=%d needs the sort path (no histogram)", k);
    KMAP_REQUIRE(n >= 0 && codes_dev && inval_dev, "counts_hist_packed: bad input");
    hipStream_t st = as_stream(stream);
    uint32_t *skip = nullptr;   // per-read duplicates as one bit per position (first find_motif round), or null
    if (dedupe_per_read) {
        KMAP_REQUIRE(n_seq == 0 || borders_dev, "counts_hist_packed: dedupe needs borders");
        KMAP_TRY(dedupe_skip_bits(codes_dev, inval_dev, n, borders_dev, n_seq, k, st, &skip));
    }
    if (dedupe_per_read && !skip) {
        // reads longer than the LDS set allows: per-read dedupe on a materialised hash array (any length)
        void *hash = nullptr;
        KMAP_TRY(kmap_scratch(&hash, (size_t)(n ? n : 1) * (k < 16 ? 4 : 8), st, KMAP_SLOT_HASH));
        KMAP_TRY(kmap_hash_kmers_packed_dev(codes_dev, inval_dev, n, k, hash, stream));
        if (k < 16) KMAP_TRY(kmap_dedupe_per_read_u32_dev((uint32_t *)hash, n, borders_dev, n_seq, stream));
        else KMAP_TRY(kmap_dedupe_per_read_u64_dev((uint64_t *)hash, n, borders_dev, n_seq, stream));
        return kmap_counts_hist_hashes(c, hash, n, k, st);
    }
    if (kmap_counts_part_applies(k, n)) {
        // 10 <= k <= 16: bucket-partitioned histogram, keys hashed from the packed reads inside its count and scatter passes (no
        // 4 B / position hash array written and read back)
        return kmap_counts_part_hist_packed(c, codes_dev, inval_dev, skip, n, k, st);
    }
    KMAP_TRY(kmap_counts_prepare_bins(c, k, st));
    if (n > 0) {
        const size_t n_bins = (size_t)1 << (2 * k);
        const size_t passes = (n_bins + HP_BINS - 1) / HP_BINS;
        static const bool wide_counters = [] { const char *v = getenv("KMAP_HIST16"); return v && v[0] == '0'; }();   // A/B switch: 32-bit LDS counters
        if (n_bins >= (size_t)HP16_BINS && n_bins / HP16_BINS <= 16 && n >= (1 << 16) && !wide_counters) {
            // k = 8, 9: 16-bit LDS counters, 65 536 bins per pass
            KMAP_TRY(kmap_allow_lds((const void *)hist_packed16_kernel, (HP16_BINS / 2 + 64) * 4));
            for (size_t p = 0; p < n_bins / HP16_BINS; ++p)
                hist_packed16_kernel<<<256, HP_TPB, (HP16_BINS / 2 + 64) * 4, st>>>(codes_dev, inval_dev, n, k, (uint32_t)(p * HP16_BINS), c->bins, skip);
        } else if (passes <= 32 && n >= (1 << 16)) {   // k <= 10: 32 passes x 0.375 B/position still beat scattered device atomics
            KMAP_TRY(kmap_allow_lds((const void *)hist_packed_kernel<false, true>, (HP_BINS + 64) * 4));
            for (size_t p = 0; p < passes; ++p)
                hist_packed_kernel<false, true><<<256, HP_TPB, (HP_BINS + 64) * 4, st>>>(codes_dev, inval_dev, n, k,
                                                                                  (uint64_t)p * HP_BINS, c->bins, skip);
        } else {
            int64_t g = ((n + 15) / 16 + BLK - 1) / BLK;
            if (g > 256 * 16) g = 256 * 16;
            hist_packed_kernel<false, false><<<(unsigned)g, BLK, 16, st>>>(codes_dev, inval_dev, n, k, 0, c->bins, skip);
        }
        KMAP_CHECK_HIP(hipGetLastError());
    }
    return KMAP_OK;
}

int kmap_counts_bins(kmap_counts *c, void **bins_dev, int64_t *n_bins) {
    KMAP_REQUIRE(c && bins_dev && n_bins, "counts_bins: null");
    KMAP_TRY(kmap_counts_bins_check(c, "counts_bins"));
    *bins_dev = c->bins;
    *n_bins = (int64_t)c->bins_cap;
    return KMAP_OK;
}

int kmap_counts_finish(kmap_counts *c, int k, int merge_revcom, int64_t *n_uniq, void *stream) {
    KMAP_REQUIRE(c && c->bins && k > 0 && k <= 16 && c->bins_cap >= ((size_t)1 << (2 * k)), "counts_finish: no histogram for k=%d", k);
    KMAP_TRY(kmap_counts_bins_check(c, "counts_finish"));
    return kmap_counts_finish_hist(c, k, merge_revcom, n_uniq, as_stream(stream));
}

int kmap_counts_run_packed_dev(kmap_counts *c, const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n,
                               const int64_t *borders_dev, int64_t n_seq, int k, int dedupe_per_read, int merge_revcom,
                               int64_t *n_uniq, void *stream) {
    KMAP_REQUIRE(c, "counts_run_packed: null handle");
    KMAP_REQUIRE(k > 0 && k < 32, "counts_run_packed: k=%d out of range", k);
    KMAP_REQUIRE(n >= 0 && codes_dev && inval_dev, "counts_run_packed: bad input");
    hipStream_t st = as_stream(stream);
    if (k > 16) {   // sort path on a materialised hash array
        void *hash = nullptr;
        KMAP_TRY(kmap_scratch(&hash, (size_t)(n ? n : 1) * 8, st, KMAP_SLOT_HASH));
        KMAP_TRY(kmap_hash_kmers_packed_dev(codes_dev, inval_dev, n, k, hash, stream));
        if (dedupe_per_read) {
            KMAP_REQUIRE(n_seq == 0 || borders_dev, "counts_run_packed: dedupe needs borders");
            KMAP_TRY(kmap_dedupe_per_read_u64_dev((uint64_t *)hash, n, borders_dev, n_seq, stream));
        }
        return kmap_counts_run_hashes_dev(c, hash, n, k, merge_revcom, n_uniq, stream);
    }
    KMAP_TRY(kmap_counts_hist_packed_dev(c, codes_dev, inval_dev, n, borders_dev, n_seq, k, dedupe_per_read, stream));
    return kmap_counts_finish_hist(c, k, merge_revcom, n_uniq, st);
}

/* Key-space-sharded counting (include/kmap_hip.h; counts_internal.h: kmap_key_range): the slice [first_bin, first_bin + n_bins) -- by
 * POSITION in key order -- of the table kmap_counts_run_packed_dev would produce from the same reads, computed from the windows that
 * decide it alone.  Every rank of a multi-GPU run holds all reads and calls this with its own range; the shards, concatenated in rank
 * order, are the single-GPU table, and no table bytes are exchanged.  The histogram passes are those of a table of 2 n_bins (with
 * merge) or n_bins (without) entries over the windows that fall into it: they shrink with the number of ranks. */
int kmap_counts_run_packed_range_dev(kmap_counts *c, const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n,
                                     const int64_t *borders_dev, int64_t n_seq, int k, int dedupe_per_read, int merge_revcom,
                                     uint64_t first_bin, uint64_t n_bins, int64_t *n_uniq, void *stream) {
    KMAP_REQUIRE(c, "counts_run_packed_range: null handle");
    KMAP_REQUIRE(k >= 11 && k <= 16, "counts_run_packed_range: key ranges serve 11 <= k <= 16 (k=%d)", k);
    KMAP_REQUIRE(n >= 0 && codes_dev && inval_dev, "counts_run_packed_range: bad input");
    const uint64_t table = (uint64_t)1 << (2 * k);
    KMAP_REQUIRE(n_bins > 0 && first_bin < table && n_bins <= table - first_bin && first_bin % 8 == 0,
                 "counts_run_packed_range: range [%llu, +%llu) outside the 4^%d table or not 8-aligned", (unsigned long long)first_bin,
                 (unsigned long long)n_bins, k);
    hipStream_t st = as_stream(stream);
    // the virtual table: 4^vk bins with half = 4^vk / 2 >= n_bins (merge), or 4^vk >= n_bins (no merge); at least 4^10
    int vk = 10;
    while ((merge_revcom ? ((uint64_t)1 << (2 * vk - 1)) : ((uint64_t)1 << (2 * vk))) < n_bins) ++vk;
    uint32_t *skip = nullptr;
    if (dedupe_per_read) {
        KMAP_REQUIRE(n_seq == 0 || borders_dev, "counts_run_packed_range: dedupe needs borders");
        KMAP_TRY(dedupe_skip_bits(codes_dev, inval_dev, n, borders_dev, n_seq, k, st, &skip));
    }
    // no gain, or no room: the whole table by the usual passes, then the slice.  (vk > k: the range is more than half of the table;
    // vk == 16 with a range that reaches virtual key 0xFFFFFFFF = the invalid marker; small inputs; reads beyond the LDS dedupe's length.)
    const bool ranged = kmap_counts_part_applies(k, n) && vk <= k && !(dedupe_per_read && !skip) && n_bins <= 0xFFFFFFF0ull &&
                        !(vk == 16 && merge_revcom && n_bins > ((uint64_t)1 << 31) - 8);
    if (!ranged) {
        KMAP_TRY(kmap_counts_hist_packed_dev(c, codes_dev, inval_dev, n, borders_dev, n_seq, k, dedupe_per_read, stream));
        return kmap_counts_finish_hist_slice(c, k, merge_revcom, first_bin, n_bins, n_uniq, st);
    }
    kmap_key_range kr;
    kr.lo = (uint32_t)first_bin;
    kr.len = (uint32_t)n_bins;
    kr.half = merge_revcom ? (uint32_t)((uint64_t)1 << (2 * vk - 1)) : 0u;
    kr.sh = 32 - 2 * k;
    uint32_t *keys = nullptr;
    int64_t n_keys = 0;
    KMAP_TRY(kmap_counts_range_stage(codes_dev, inval_dev, skip, n, k, kr, &keys, &n_keys, st));
    KMAP_TRY(kmap_counts_part_hist_u32(c, keys, n_keys, vk, st));      // the table of the virtual keys: 4^vk bins, every bin written
    return kmap_counts_finish_key_range(c, k, kr, n_uniq, st);
}

// k <= 16: the bit-sliced formulation (bitslice.hip) on the reads' bit planes; the per-window kernels of this file serve k > 16
static bool bitslice_on(int k) { return k <= 16; }

namespace {
__global__ void inval_prefix_kernel(uint16_t *__restrict__ inval, int64_t m) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t left = m - 16 * g;                           // positions of this group inside the prefix
    if (left <= 0) return;
    inval[g] |= left >= 16 ? (uint16_t)0xFFFFu : (uint16_t)(0xFFFFu << (16 - (int)left));
}
}  // namespace

int kmap_inval_set_prefix_dev(uint16_t *inval_dev, int64_t m, void *stream) {
    if (m <= 0) return KMAP_OK;
    KMAP_REQUIRE(inval_dev, "inval_set_prefix: null pointer");
    const int64_t ng = (m + 15) >> 4;
    inval_prefix_kernel<<<(unsigned)((ng + 63) / 64), 64, 0, as_stream(stream)>>>(inval_dev, m);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

int kmap_mask_hamball_packed_dev(const uint32_t *codes_dev, uint16_t *inval_dev, int64_t n, int k, const uint64_t *cons,
                                 const int32_t *radius, int n_cons, const uint32_t *planes_dev, void *stream) {
    KMAP_REQUIRE(k > 0 && k < 32, "mask_hamball_packed: k=%d out of range", k);
    KMAP_REQUIRE(n_cons >= 0 && (n_cons == 0 || (cons && radius)), "mask_hamball_packed: null consensus list");
    if (n <= 0 || n_cons == 0) return KMAP_OK;
    KMAP_REQUIRE(codes_dev && inval_dev, "mask_hamball_packed: null pointer");
    KMAP_REQUIRE(k > 16 || planes_dev, "mask_hamball_packed: k <= 16 needs the bit planes (kmap_pack_planes_dev)");
    // a negative radius matches nothing (the reference's `ham_dist <= r`, kmer_count.py:594-603): such entries are dropped here --
    // the bit-sliced "count > r" test is built for r >= 0
    std::vector<uint64_t> cons_v;
    std::vector<int32_t> rad_v;
    for (int c = 0; c < n_cons; ++c)
        if (radius[c] >= 0) {
            cons_v.push_back(cons[c]);
            rad_v.push_back(radius[c]);
        }
    if (cons_v.empty()) return KMAP_OK;
    cons = cons_v.data();
    radius = rad_v.data();
    n_cons = (int)cons_v.size();
    hipStream_t st = as_stream(stream);
    const int64_t ng = (n + 15) >> 4;
    if (bitslice_on(k)) {
        // 16 consensuses per flag pass, all passes on the mask as it is on entry, then the coverage passes
        const int nb = (n_cons + 15) / 16;
        uint16_t *hitb = nullptr;
        const int64_t ngq = (ng + 9) & ~(int64_t)7;                // even (the kernel stores group pairs) and 16-byte aligned batches
        KMAP_TRY(kmap_scratch((void **)&hitb, (size_t)ngq * 2 * nb, st, KMAP_SLOT_A));
        for (int b = 0; b < nb; ++b) {
            const int m = (n_cons - 16 * b < 16) ? n_cons - 16 * b : 16;
            KMAP_TRY(kmap_bitslice_hits(planes_dev, inval_dev, n, k, cons + 16 * b, radius + 16 * b, m, 0, hitb + (size_t)b * ngq, false, st));
        }
        for (int b = 0; b < nb; ++b)
            mask_cover_packed_kernel<<<grid_for((ng + 3) / 4, BLK), BLK, 0, st>>>(hitb + (size_t)b * ngq, n, k, inval_dev);
        KMAP_CHECK_HIP(hipGetLastError());
        return KMAP_OK;
    }
    const int batches = (n_cons + 31) / 32;
    uint16_t *hit = nullptr;
    const int64_t ngp = (ng + 7) & ~(int64_t)7;                   // per-batch stride: every batch's hit array 16-byte aligned
    KMAP_TRY(kmap_scratch((void **)&hit, (size_t)ngp * 2 * batches, st, KMAP_SLOT_A));
    // all flag passes read the mask as it is on entry (the reference hashes once, kmer_count.py:605-607) ...
    for (int b = 0; b < batches; ++b) {
        ConsTabP t;
        t.n = (n_cons - 32 * b < 32) ? n_cons - 32 * b : 32;
        for (int c = 0; c < t.n; ++c) {
            t.cons[c] = cons[32 * b + c] & low_mask<uint64_t>(k);
            t.radius[c] = radius[32 * b + c];
        }
        mask_flag_packed_kernel<<<grid_for(ng, BLK), BLK, 0, st>>>(codes_dev, inval_dev, n, k, t, hit + (size_t)b * ngp);
    }
    // ... then the coverage passes OR into it
    for (int b = 0; b < batches; ++b)
        mask_cover_packed_kernel<<<grid_for((ng + 3) / 4, BLK), BLK, 0, st>>>(hit + (size_t)b * ngp, n, k, inval_dev);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

int kmap_scan_declare_uniform(kmap_scan *s, const int64_t *borders_dev, int64_t n_seq, int64_t read_len, int64_t stride, int *accepted, void *stream) {
    KMAP_REQUIRE(s && (n_seq == 0 || borders_dev), "scan_declare_uniform: null");
    return kmap_bitslice_declare_uniform(s, borders_dev, n_seq, read_len, stride, accepted, as_stream(stream));
}

int kmap_scan_run_packed_dev(kmap_scan *s, const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n,
                             const int64_t *borders_dev, int64_t n_seq, int k, uint64_t cons, int radius, int revcom,
                             int64_t *total_hits, const uint32_t *planes_dev, void *stream) {
    KMAP_REQUIRE(s, "scan_run_packed: null handle");
    KMAP_REQUIRE(k > 0 && k < 32, "scan_run_packed: k=%d out of range", k);
    KMAP_REQUIRE(n >= 0 && n_seq >= 0 && radius >= 0, "scan_run_packed: negative size");
    s->n_seq = n_seq;
    s->total = 0;
    if (total_hits) *total_hits = 0;
    if (n_seq == 0) return KMAP_OK;
    KMAP_REQUIRE(codes_dev && inval_dev && borders_dev, "scan_run_packed: null pointer");
    KMAP_REQUIRE(k > 16 || planes_dev, "scan_run_packed: k <= 16 needs the bit planes (kmap_pack_planes_dev)");
    hipStream_t st = as_stream(stream);
    KMAP_TRY(kmap_scan_reserve(s, n_seq));
    const uint64_t m = low_mask<uint64_t>(k);
    const uint64_t c = cons & m;
    uint64_t com = m - c, rcc = com & 3u;
    for (int i = 0; i < k - 1; ++i) { rcc <<= 2; com >>= 2; rcc += com & 3u; }
    if (bitslice_on(k)) {
        // hit bit per window (bit-sliced, 0.125 B per position written), then the per-read passes evaluate the few hits exactly
        const int64_t ng = (n + 15) >> 4;
        uint16_t *hit16 = nullptr;
        KMAP_TRY(kmap_scratch((void **)&hit16, (size_t)((ng + 9) & ~(int64_t)7) * 2, st, KMAP_SLOT_HASH));
        if (n > 0) KMAP_TRY(kmap_bitslice_hits(planes_dev, inval_dev, n, k, &c, &radius, 1, revcom, hit16, true, st));
        uint32_t *hit32 = reinterpret_cast<uint32_t *>(hit16);
        uint64_t total = 0;
        KMAP_TRY(kmap_bitslice_scan_reads_all(hit32, codes_dev, inval_dev, n, borders_dev, n_seq, k, c, revcom, radius, s, &total, st));
        s->total = (int64_t)total;
        if (total_hits) *total_hits = (int64_t)total;
        return KMAP_OK;
    }
    const bool flat = radius <= 14;                 // k > 16: nibble pass + thread-per-read passes; larger radii: wave per read
    const unsigned grid = (unsigned)((n_seq + SC_WAVES - 1) / SC_WAVES);
    const unsigned fgrid = (unsigned)((n_seq + FL_TPB - 1) / FL_TPB);
    uint64_t *nib = nullptr;
    uint8_t *wmin = nullptr;
    int d_inv = 0;
    if (flat) {
        const int64_t ng = (n + 15) >> 4;
        const size_t ngp = ((size_t)(ng ? ng : 1) + 15) & ~(size_t)15;
        KMAP_TRY(kmap_scratch((void **)&nib, ngp * 9, st, KMAP_SLOT_HASH));   // 8 B of nibbles + 1 B minimum per group
        wmin = reinterpret_cast<uint8_t *>(nib + ngp);
        if (ng) {
            scan_nibble_kernel<<<grid_for(ng, BLK), BLK, 0, st>>>(codes_dev, inval_dev, n, k, c, rcc, radius, revcom, nib, wmin);
        }
        // distance of an invalid window (all ones, compared like any value)
        auto pc2 = [](uint64_t x) { return __builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull); };
        d_inv = pc2((m ^ c) & m);
        if (revcom) {
            const int d2 = pc2((m ^ rcc) & m);
            d_inv = d2 < d_inv ? d2 : d_inv;
        }
        scan_reads_kernel<false><<<fgrid, FL_TPB, 0, st>>>(nib, n, borders_dev, n_seq, k, d_inv, radius, s->hits, s->mind, nullptr, nullptr, wmin);
    } else {
        scan_packed_kernel<false><<<grid, KMAP_WAVE * SC_WAVES, 0, st>>>(codes_dev, inval_dev, n, borders_dev, n_seq, k, c, rcc,
                                                                         radius, revcom, s->hits, s->mind, nullptr, nullptr);
    }
    KMAP_TRY(exclusive_scan_u32(reinterpret_cast<const uint32_t *>(s->hits), n_seq, s->offs, st));
    uint64_t total = 0;
    KMAP_CHECK_HIP(hipMemcpyAsync(&total, s->offs + n_seq, 8, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));
    KMAP_TRY(kmap_scan_reserve_pos(s, total));
    if (total) {
        if (flat)
            scan_reads_kernel<true><<<fgrid, FL_TPB, 0, st>>>(nib, n, borders_dev, n_seq, k, d_inv, radius, s->hits, s->mind, s->offs, s->pos, wmin);
        else
            scan_packed_kernel<true><<<grid, KMAP_WAVE * SC_WAVES, 0, st>>>(codes_dev, inval_dev, n, borders_dev, n_seq, k, c, rcc,
                                                                            radius, revcom, s->hits, s->mind, s->offs, s->pos);
    }
    KMAP_CHECK_HIP(hipGetLastError());
    s->total = (int64_t)total;
    if (total_hits) *total_hits = (int64_t)total;
    return KMAP_OK;
}

}  // extern "C"
