// bitslice.hip -- "is this window within Hamming distance r of the consensus" for EVERY window of the packed reads, bit-sliced
// over positions: 32 windows per thread at once instead of one funnel-shifted window at a time.
//
// Used by the occurrence scan (get_motif_occurence, motif_discovery.py:1422-1477 -- BASELINE config C5) and by masking
// (mask_input, kmer_count.py:580-610), k <= 16.  Both need, per window, only the predicate d <= r (the scan then evaluates the
// few hit windows exactly to find each read's minimum).  The per-window formulation (scan_nibble_kernel /
// mask_flag_packed_kernel in packed.hip: alignbit, shift, xor, 2-bit popcount, compare per window and strand: ~21 vector
// instructions per window, VALU-issue bound at 0.85 ms per consensus on the C3 reads) is replaced by:
//
//   * bit planes of the reads, built once per upload: planes[2w] = H, planes[2w + 1] = L for the 32 positions of word w (groups
//     2w, 2w + 1), H / L = the high / low bit of every base code, first position most significant (same order as the invalid mask);
//   * a thread takes 32 positions: H, L (and the following 32 for the window tails).  For consensus base j the mismatch plane
//     over the 32 windows is M_j = ((H << j) ^ CH_j) | ((L << j) ^ CL_j) with CH_j / CL_j = 0 or ~0 (scalar): 2 funnel shifts
//     shared by both strands + 3 logic ops per strand;
//   * the K mismatch planes are added bit-sliced by a carry-save tree of two-instruction full adders (compile-time K: ~1.6 ops
//     per plane) into a 4..5-bit counter per window, and "count > r" is the carry out of adding the constant 2^B - 1 - r (one
//     majority per counter bit);
//   * hit = (le_fwd | le_rc) for valid windows; a window that touches an invalid position has the reference's all-ones hash
//     ("compared like any value"), i.e. the same distance d_inv for all of them: hit = (d_inv <= r), a scalar.
// ~3.5 vector instructions per window at k = 8 with both strands, ~5.5 at k = 14 (r03's ripple-adder tree: 5.2 / 8.5), and the intermediate shrinks from 0.56 B
// (nibble + minimum) to 0.125 B (one bit) per position.
#include <algorithm>

#include "common.h"
#include "scan_internal.h"
#include "scan_util.h"

namespace {

constexpr int BS_TPB = 256;

// ---- planes -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t squeeze_even_bits(uint32_t x) {   // bits 30, 28, ..., 0 -> bits 15..0, order kept
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    x = (x | (x >> 8)) & 0x0000FFFFu;
    return x;
}
// thread = one 32-position word: groups 2w, 2w + 1 (the array holds an even number of groups) -> planes[2w] = H, planes[2w + 1] = L
__global__ __launch_bounds__(BS_TPB) void planes_kernel(const uint32_t *__restrict__ codes, int64_t n_words, uint32_t *__restrict__ planes) {
    const int64_t w = (int64_t)blockIdx.x * BS_TPB + threadIdx.x;
    if (w >= n_words) return;
    const uint2 c = *reinterpret_cast<const uint2 *>(codes + 2 * w);
    uint2 o;
    o.x = (squeeze_even_bits(c.x >> 1) << 16) | squeeze_even_bits(c.y >> 1);
    o.y = (squeeze_even_bits(c.x) << 16) | squeeze_even_bits(c.y);
    *reinterpret_cast<uint2 *>(planes + 2 * w) = o;
}

// ---- bit-sliced counters --------------------------------------------------------------------------------------------
// The K mismatch planes of a strand are counted by a carry-save tree of full adders, each TWO instructions on gfx950
// (v_bitop3_b32: sum = a ^ b ^ c, carry = majority): planes of equal weight are taken three at a time until one is left -- the
// count's bit of that weight -- and the carries form the next weight's planes.  K = 14: 10 full adders + 1 half adder = 22
// instructions (the r03 balanced tree of ripple adders: 46), K = 8: 13.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
constexpr int bits_for(int n) {   // bits that hold 0..n
    int b = 0;
    while ((1 << b) <= n) ++b;
    return b;
}
// N planes of one weight in cur[0 .. N) -> the count's bit of that weight (returned) and N / 2 carry planes in nxt[0 .. N / 2)
template <int N>
__device__ __forceinline__ uint32_t csa_level(uint32_t (&cur)[16], uint32_t (&nxt)[16]) {
    if constexpr (N == 0) return 0u;
    constexpr int FA = (N - 1) / 2;
#pragma unroll
    for (int it = 0; it < FA; ++it) {                                       // live planes: cur[0 .. N - 2 it)
        const int top = N - 2 * it - 1;
        const uint32_t x = cur[top], y = cur[top - 1], z = cur[top - 2];
        cur[top - 2] = xor3(x, y, z);
        nxt[it] = maj3(x, y, z);
    }
    if constexpr (N >= 2 && N % 2 == 0) {                                   // two planes left: half adder
        const uint32_t x = cur[0], y = cur[1];
        cur[0] = x ^ y;
        nxt[FA] = x & y;
    }
    return cur[0];
}
// windows whose mismatch count over the K planes exceeds r: the carry out of count + (2^B - 1 - r), r clamped to 2^B - 1
// (r >= 0: the callers drop entries with a negative radius, which match nothing).  Bit b of that constant as a scalar 0 / ~0:
// carry' = majority(count_b, carry, const_b) -- one instruction per counter bit.
template <int K>
__device__ __forceinline__ uint32_t count_greater_than(const uint32_t (&m)[K], int r) {
    constexpr int B = bits_for(K);
    uint32_t l0[16], l1[16], l2[16], l3[16], l4[16], cnt[5];
#pragma unroll
    for (int j = 0; j < K; ++j) l0[j] = m[j];
    cnt[0] = csa_level<K>(l0, l1);
    cnt[1] = csa_level<K / 2>(l1, l2);
    cnt[2] = csa_level<K / 4>(l2, l3);
    cnt[3] = csa_level<K / 8>(l3, l4);
    cnt[4] = csa_level<K / 16>(l4, l0);
    const int rr = r > (1 << B) - 1 ? (1 << B) - 1 : r;
    const uint32_t konst = (uint32_t)((1 << B) - 1 - rr);
    uint32_t carry = 0;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const uint32_t kb = 0u - ((konst >> b) & 1u);
        carry = b == 0 ? (cnt[0] & kb) : maj3(cnt[b], carry, kb);
    }
    return carry;
}

// ---- hit bits of 32 windows per thread ------------------------------------------------------------------------------
struct HitCons {
    uint32_t fwd, rc;      // consensus codes (2 bits per base, first base most significant), rc used when two != 0
    int32_t radius;
    uint32_t two;          // also test the reverse-complement strand
    uint32_t inv_hit;      // ~0 when an invalid window (all-ones hash) lies within radius of this entry, else 0
};
struct HitTab {
    HitCons c[16];
    int n;
};

template <int K>
__device__ __forceinline__ uint32_t strand_gt(const uint32_t (&hs)[K], const uint32_t (&ls)[K], uint32_t code, int radius) {
    uint32_t m[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const uint32_t base = (code >> (2 * (K - 1 - j))) & 3u;          // scalar
        const uint32_t ch = 0u - (base >> 1), cl = 0u - (base & 1u);
        m[j] = (hs[j] ^ ch) | (ls[j] ^ cl);
    }
    return count_greater_than<K>(m, radius);
}

// ---- mismatch planes through the VGPR index mode ---------------------------------------------------------------------------
// (h ^ ch) | (l ^ cl) has four inputs, two of them the consensus base: as written above a mismatch plane costs the strand two
// instructions on top of the shared funnel shifts (3 per plane and strand in all).  The base only SELECTS one of four functions of
// (h, l): E_A = h | l, E_C = h | ~l, E_G = ~h | l, E_T = ~h | ~l -- "the base here is not b" for the 64 positions a thread holds,
// 8 instructions shared by every strand and table entry.  The mismatch plane of consensus position j is then the funnel shift by
// j of E_b with b = the consensus base, and gfx9's VGPR index mode lets a SCALAR pick the register: with
// s_set_gpr_idx_on / _idx (M0[7:0] = b, SRC0 and SRC1 relative) `v_alignbit_b32 m, v40, v44, 32 - j` reads v[40 + b], v[44 + b]:
// ONE vector instruction per plane and strand.  A thread carries TWO 32-window words (E planes in v40..v47 and v48..v55, pinned by
// explicit-register constraints), so that one scalar index change serves two vector instructions and the scalar unit, which
// issues beside the vector unit, stays the smaller stream.  Blocks hold 8 / 4 / 2 / 1 planes (an asm statement takes 30
// operands); every block leaves the index mode switched off.  M0 is not on the clobber lists: the compiler reserves it (it
// sets M0 itself right before the few instructions that read it) and rejects it there.
// HAZARD (measured, not in the ISA manual's table): a vector instruction issued right behind the s_set_gpr_idx_* that changes
// M0 can still see the OLD index -- without a wait state ~1 wave in 10^4 produced hit words of the wrong planes at C5 size, a
// different set of waves every run; with `s_nop 0` (one wait state, what the manual asks between an M0 write and s_movrel /
// LDS-add-TID / interp) none in repeated full-size runs.  KMAP_IDX_WAIT is two wait states (behind every index change and
// behind s_set_gpr_idx_off, whose successor is whatever the compiler schedules next); tests/test_gpu_fullsize.py holds the
// full-size comparison against the formulation without the index mode.
#define KMAP_IDX_WAIT "s_nop 1\n\t"
#define KMAP_IDX_WAIT_END "s_nop 1"      // the same distance between switching the mode off and the compiler's next vector instruction
typedef uint32_t EPlanes __attribute__((ext_vector_type(8)));   // {A, C, G, T} of word 0 (positions P .. P + 31), of word 1 (the next 32)
#define KMAP_E_INS "{v[40:47]}"(ea), "{v[48:55]}"(eb)
__device__ __forceinline__ EPlanes make_eplanes(uint32_t H, uint32_t L, uint32_t H2, uint32_t L2) {
    EPlanes e;                           // one instruction each (left to itself the compiler spent a v_not and a 64-bit move on some)
    e[0] = H | L, e[4] = H2 | L2;
    e[1] = __builtin_amdgcn_bitop3_b32(H, L, L, 0xF3), e[5] = __builtin_amdgcn_bitop3_b32(H2, L2, L2, 0xF3);      // a | ~b
    e[2] = __builtin_amdgcn_bitop3_b32(H, L, L, 0xCF), e[6] = __builtin_amdgcn_bitop3_b32(H2, L2, L2, 0xCF);      // ~a | b
    e[3] = __builtin_amdgcn_bitop3_b32(H, L, L, 0x3F), e[7] = __builtin_amdgcn_bitop3_b32(H2, L2, L2, 0x3F);      // ~(a & b)
    return e;
}
// plane 0 needs no shift: the selected word 0 itself
__device__ __forceinline__ void idx_plane0(const EPlanes &ea, const EPlanes &eb, uint32_t c0, uint32_t &ma, uint32_t &mb) {
    asm("s_set_gpr_idx_on %[c0], gpr_idx(SRC0)\n\t" KMAP_IDX_WAIT
        "v_mov_b32 %[a0], v40\n\t"
        "v_mov_b32 %[b0], v48\n\t"
        "s_set_gpr_idx_off\n\t" KMAP_IDX_WAIT_END
        : [a0] "=&v"(ma), [b0] "=&v"(mb)
        : [c0] "s"(c0), KMAP_E_INS);
}
// planes J0 .. J0 + N - 1 (J0 >= 1) of both words: m[j] = funnel shift of E_{c[j]} by j
template <int J0>
__device__ __forceinline__ void idx_planes1(const EPlanes &ea, const EPlanes &eb, const uint32_t *c, uint32_t *ma, uint32_t *mb) {
    asm("s_set_gpr_idx_on %[c0], gpr_idx(SRC0,SRC1)\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a0], v40, v44, %[s]\n\t"
        "v_alignbit_b32 %[b0], v48, v52, %[s]\n\t"
        "s_set_gpr_idx_off\n\t" KMAP_IDX_WAIT_END
        : [a0] "=&v"(ma[J0 + 0]), [b0] "=&v"(mb[J0 + 0])
        : [c0] "s"(c[J0 + 0]), [s] "n"(32 - J0), KMAP_E_INS);
}
template <int J0>
__device__ __forceinline__ void idx_planes2(const EPlanes &ea, const EPlanes &eb, const uint32_t *c, uint32_t *ma, uint32_t *mb) {
    asm("s_set_gpr_idx_on %[c0], gpr_idx(SRC0,SRC1)\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a0], v40, v44, %[s]\n\t"
        "v_alignbit_b32 %[b0], v48, v52, %[s]\n\t"
        "s_set_gpr_idx_idx %[c1]\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a1], v40, v44, %[s]-1\n\t"
        "v_alignbit_b32 %[b1], v48, v52, %[s]-1\n\t"
        "s_set_gpr_idx_off\n\t" KMAP_IDX_WAIT_END
        : [a0] "=&v"(ma[J0 + 0]), [b0] "=&v"(mb[J0 + 0]), [a1] "=&v"(ma[J0 + 1]), [b1] "=&v"(mb[J0 + 1])
        : [c0] "s"(c[J0 + 0]), [c1] "s"(c[J0 + 1]), [s] "n"(32 - J0), KMAP_E_INS);
}
template <int J0>
__device__ __forceinline__ void idx_planes4(const EPlanes &ea, const EPlanes &eb, const uint32_t *c, uint32_t *ma, uint32_t *mb) {
    asm("s_set_gpr_idx_on %[c0], gpr_idx(SRC0,SRC1)\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a0], v40, v44, %[s]\n\t"
        "v_alignbit_b32 %[b0], v48, v52, %[s]\n\t"
        "s_set_gpr_idx_idx %[c1]\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a1], v40, v44, %[s]-1\n\t"
        "v_alignbit_b32 %[b1], v48, v52, %[s]-1\n\t"
        "s_set_gpr_idx_idx %[c2]\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a2], v40, v44, %[s]-2\n\t"
        "v_alignbit_b32 %[b2], v48, v52, %[s]-2\n\t"
        "s_set_gpr_idx_idx %[c3]\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a3], v40, v44, %[s]-3\n\t"
        "v_alignbit_b32 %[b3], v48, v52, %[s]-3\n\t"
        "s_set_gpr_idx_off\n\t" KMAP_IDX_WAIT_END
        : [a0] "=&v"(ma[J0 + 0]), [b0] "=&v"(mb[J0 + 0]), [a1] "=&v"(ma[J0 + 1]), [b1] "=&v"(mb[J0 + 1]), [a2] "=&v"(ma[J0 + 2]), [b2] "=&v"(mb[J0 + 2]), [a3] "=&v"(ma[J0 + 3]), [b3] "=&v"(mb[J0 + 3])
        : [c0] "s"(c[J0 + 0]), [c1] "s"(c[J0 + 1]), [c2] "s"(c[J0 + 2]), [c3] "s"(c[J0 + 3]), [s] "n"(32 - J0), KMAP_E_INS);
}
template <int J0>
__device__ __forceinline__ void idx_planes8(const EPlanes &ea, const EPlanes &eb, const uint32_t *c, uint32_t *ma, uint32_t *mb) {
    asm("s_set_gpr_idx_on %[c0], gpr_idx(SRC0,SRC1)\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a0], v40, v44, %[s]\n\t"
        "v_alignbit_b32 %[b0], v48, v52, %[s]\n\t"
        "s_set_gpr_idx_idx %[c1]\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a1], v40, v44, %[s]-1\n\t"
        "v_alignbit_b32 %[b1], v48, v52, %[s]-1\n\t"
        "s_set_gpr_idx_idx %[c2]\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a2], v40, v44, %[s]-2\n\t"
        "v_alignbit_b32 %[b2], v48, v52, %[s]-2\n\t"
        "s_set_gpr_idx_idx %[c3]\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a3], v40, v44, %[s]-3\n\t"
        "v_alignbit_b32 %[b3], v48, v52, %[s]-3\n\t"
        "s_set_gpr_idx_idx %[c4]\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a4], v40, v44, %[s]-4\n\t"
        "v_alignbit_b32 %[b4], v48, v52, %[s]-4\n\t"
        "s_set_gpr_idx_idx %[c5]\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a5], v40, v44, %[s]-5\n\t"
        "v_alignbit_b32 %[b5], v48, v52, %[s]-5\n\t"
        "s_set_gpr_idx_idx %[c6]\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a6], v40, v44, %[s]-6\n\t"
        "v_alignbit_b32 %[b6], v48, v52, %[s]-6\n\t"
        "s_set_gpr_idx_idx %[c7]\n\t" KMAP_IDX_WAIT
        "v_alignbit_b32 %[a7], v40, v44, %[s]-7\n\t"
        "v_alignbit_b32 %[b7], v48, v52, %[s]-7\n\t"
        "s_set_gpr_idx_off\n\t" KMAP_IDX_WAIT_END
        : [a0] "=&v"(ma[J0 + 0]), [b0] "=&v"(mb[J0 + 0]), [a1] "=&v"(ma[J0 + 1]), [b1] "=&v"(mb[J0 + 1]), [a2] "=&v"(ma[J0 + 2]), [b2] "=&v"(mb[J0 + 2]), [a3] "=&v"(ma[J0 + 3]), [b3] "=&v"(mb[J0 + 3]), [a4] "=&v"(ma[J0 + 4]), [b4] "=&v"(mb[J0 + 4]), [a5] "=&v"(ma[J0 + 5]), [b5] "=&v"(mb[J0 + 5]), [a6] "=&v"(ma[J0 + 6]), [b6] "=&v"(mb[J0 + 6]), [a7] "=&v"(ma[J0 + 7]), [b7] "=&v"(mb[J0 + 7])
        : [c0] "s"(c[J0 + 0]), [c1] "s"(c[J0 + 1]), [c2] "s"(c[J0 + 2]), [c3] "s"(c[J0 + 3]), [c4] "s"(c[J0 + 4]), [c5] "s"(c[J0 + 5]), [c6] "s"(c[J0 + 6]), [c7] "s"(c[J0 + 7]), [s] "n"(32 - J0), KMAP_E_INS);
}
#undef KMAP_E_INS
// windows of the two words whose distance to `code` exceeds the radius
template <int K>
__device__ __forceinline__ void strand_gt_idx(const EPlanes &ea, const EPlanes &eb, uint32_t code, int radius, uint32_t &gta, uint32_t &gtb) {
    uint32_t c[K], ma[K], mb[K];
#pragma unroll
    for (int j = 0; j < K; ++j) c[j] = (code >> (2 * (K - 1 - j))) & 3u;   // scalar: the consensus base of plane j
    idx_plane0(ea, eb, c[0], ma[0], mb[0]);
    constexpr int R = K - 1;                                                // planes 1 .. K - 1 in blocks of 8, 4, 2, 1
    if constexpr (R >= 8) idx_planes8<1>(ea, eb, c, ma, mb);
    constexpr int J4 = 1 + (R & 8);
    if constexpr ((R & 4) != 0) idx_planes4<J4>(ea, eb, c, ma, mb);
    constexpr int J2 = J4 + (R & 4);
    if constexpr ((R & 2) != 0) idx_planes2<J2>(ea, eb, c, ma, mb);
    constexpr int J1 = J2 + (R & 2);
    if constexpr ((R & 1) != 0) idx_planes1<J1>(ea, eb, c, ma, mb);
    gta = count_greater_than<K>(ma, radius);
    gtb = count_greater_than<K>(mb, radius);
}

// the 64 positions from group g0 (even) on as bit planes, and the windows among the first 32 that touch an invalid position.
// Groups g0 .. g0 + 3 as two aligned pairs: g0 < the number of data groups, and the array holds an even number of groups of
// which at least the last two are all-invalid halo groups (kmap_packed_groups), so g0 + 3 is inside it.
template <int K>
__device__ __forceinline__ void load_word(const uint32_t *__restrict__ planes, const uint16_t *__restrict__ inval, int64_t g0, uint32_t &H,
                                          uint32_t &L, uint32_t &H2, uint32_t &L2, uint32_t &bad) {
    const uint2 pa = *reinterpret_cast<const uint2 *>(planes + g0), pb = *reinterpret_cast<const uint2 *>(planes + g0 + 2);
    const uint32_t ia = *reinterpret_cast<const uint32_t *>(inval + g0), ib = *reinterpret_cast<const uint32_t *>(inval + g0 + 2);
    H = pa.x, L = pa.y, H2 = pb.x, L2 = pb.y;
    // OR of the invalid flags of positions p .. p + K - 1 (doubling on the 64-bit stream); a little-endian pair of flag words has
    // the first group in its low half: rotate by 16
    // (32-bit halves: funnel shift + or for the upper, shift-or for the lower word -- no 64-bit shifts; the lower word's last step is
    // not needed)
    uint32_t hi = __builtin_amdgcn_alignbit(ia, ia, 16), lo = __builtin_amdgcn_alignbit(ib, ib, 16);
#pragma unroll
    for (int have = 1; have < K;) {
        const int step = (have <= K - have) ? have : K - have;
        hi |= __builtin_amdgcn_alignbit(hi, lo, 32 - step);
        have += step;
        if (have < K) lo |= lo << step;
    }
    bad = hi;
}
template <bool WORDS>
__device__ __forceinline__ void store_word(uint16_t *__restrict__ hit16, int64_t w, uint32_t hit, int64_t n) {
    const int64_t left = n - 32 * w;                                      // positions past the end do not exist
    if (left < 32) hit &= ~((1u << (32 - (int)left)) - 1u);
    if (WORDS) reinterpret_cast<uint32_t *>(hit16)[w] = hit;
    else *reinterpret_cast<uint32_t *>(hit16 + 2 * w) = (hit << 16) | (hit >> 16);   // little-endian halves: hit16[2 w] = windows 0..15
}

// hit16[g]: bit (15 - i) set when the window at position 16 g + i is within radius of any table entry (invalid windows: the
// entry's inv_hit).  Thread = word t = groups 2t, 2t + 1.  planes / inval are read up to group 2t + 3 (inside the array: see kmap_packed_groups).
// WORDS: store the 32 hit bits as ONE word, window i in bit 31 - i (hit32[t], the scan's per-read passes); otherwise as the
// two uint16 of the mask's coverage pass.
template <int K, bool WORDS>
__global__ __launch_bounds__(BS_TPB) void hits_planes_kernel(const uint32_t *__restrict__ planes, const uint16_t *__restrict__ inval,
                                                            int64_t n, int64_t n_alloc_groups, HitTab tab,
                                                            uint16_t *__restrict__ hit16) {
    const int64_t t = (int64_t)blockIdx.x * BS_TPB + threadIdx.x;
    if (32 * t >= n) return;
    uint32_t H, L, H2, L2, bad;
    load_word<K>(planes, inval, 2 * t, H, L, H2, L2, bad);
    uint32_t hs[K], ls[K];
    hs[0] = H;
    ls[0] = L;
#pragma unroll
    for (int j = 1; j < K; ++j) {
        hs[j] = __builtin_amdgcn_alignbit(H, H2, 32 - j);                 // bit (31 - i): base at position P + i + j
        ls[j] = __builtin_amdgcn_alignbit(L, L2, 32 - j);
    }
    uint32_t hit = 0;
    for (int c = 0; c < tab.n; ++c) {
        const HitCons e = tab.c[c];
        uint32_t gt = strand_gt<K>(hs, ls, e.fwd, e.radius);
        if (e.two) gt &= strand_gt<K>(hs, ls, e.rc, e.radius);           // within radius on either strand <=> not (both exceed)
        hit |= (~gt & ~bad) | (bad & e.inv_hit);
    }
    store_word<WORDS>(hit16, t, hit, n);
}

// The same through the index mode: a block takes 2 * BS_TPB consecutive words, thread i the words i and BS_TPB + i of them.
template <int K, bool WORDS>
__global__ __launch_bounds__(BS_TPB) void hits_planes_idx_kernel(const uint32_t *__restrict__ planes, const uint16_t *__restrict__ inval,
                                                                int64_t n, int64_t n_alloc_groups, HitTab tab,
                                                                uint16_t *__restrict__ hit16) {
    const int64_t wa = (int64_t)blockIdx.x * (2 * BS_TPB) + threadIdx.x;
    if (32 * wa >= n) return;
    const int64_t wb = wa + BS_TPB;
    const bool has_b = 32 * wb < n;                                       // otherwise: word a once more, not stored
    uint32_t H, L, H2, L2, bad_a, bad_b;
    load_word<K>(planes, inval, 2 * wa, H, L, H2, L2, bad_a);
    const EPlanes ea = make_eplanes(H, L, H2, L2);
    load_word<K>(planes, inval, 2 * (has_b ? wb : wa), H, L, H2, L2, bad_b);
    const EPlanes eb = make_eplanes(H, L, H2, L2);
    uint32_t hit_a = 0, hit_b = 0;
    for (int c = 0; c < tab.n; ++c) {
        const HitCons e = tab.c[c];
        uint32_t gta, gtb;
        strand_gt_idx<K>(ea, eb, e.fwd, e.radius, gta, gtb);
        if (e.two) {
            uint32_t ra, rb;
            strand_gt_idx<K>(ea, eb, e.rc, e.radius, ra, rb);
            gta &= ra;
            gtb &= rb;
        }
        hit_a |= (~gta & ~bad_a) | (bad_a & e.inv_hit);
        hit_b |= (~gtb & ~bad_b) | (bad_b & e.inv_hit);
    }
    store_word<WORDS>(hit16, wa, hit_a, n);
    if (has_b) store_word<WORDS>(hit16, wb, hit_b, n);
}

// ---- per-read passes of the scan on the hit bits ---------------------------------------------------------------------
constexpr int HR_TPB = 256;
constexpr int HR_LONG = 1024;
__device__ __forceinline__ int64_t hr_slice_stop(int64_t L, int k) {     // python slice [0 : L - k + 1] (motif_discovery.py:1443)
    int64_t stop = L - k + 1;
    if (stop < 0) {
        stop += L;
        if (stop < 0) stop = 0;
    }
    return stop > L ? L : stop;
}
// exact min(fwd, rc) distance of the window at absolute position p.  CHECK_INVALID: a hit may be a window that touches an
// invalid position (only when the all-ones hash itself lies within the radius, d_inv <= r) -- it then has distance d_inv;
// otherwise every hit window is valid and the invalid mask need not be read.
template <bool CHECK_INVALID>
__device__ __forceinline__ int hr_dist(const uint32_t *__restrict__ codes, const uint16_t *__restrict__ inval, int64_t p, int k,
                                       uint32_t km, uint32_t cons, uint32_t rcc, int revcom) {
    const int64_t g = p >> 4;
    const int i = (int)(p & 15);
    const uint32_t hi = codes[g], lo = codes[g + 1];
    const uint32_t top = i ? __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * i) : hi;
    uint32_t h = (top >> (32 - 2 * k)) & km;
    if (CHECK_INVALID) {
        const uint64_t m = ((uint64_t)inval[g] << 32) | ((uint64_t)inval[g + 1] << 16) | inval[g + 2];
        if (((m >> (48 - i - k)) & ((1ull << k) - 1ull)) != 0) h = km;
    }
    int d = popc2(h ^ cons);
    if (revcom) {
        const int d2 = popc2(h ^ rcc);
        d = d2 < d ? d2 : d;
    }
    return d;
}
// hit word wi (positions 32 wi .. 32 wi + 31, position i in bit 31 - i) restricted to absolute positions [a, b)
__device__ __forceinline__ uint32_t hr_mask(int64_t wi, int64_t a, int64_t b) {
    uint32_t m = ~0u;
    const int64_t p0 = wi << 5;
    if (a > p0) m &= ~0u >> (int)(a - p0);
    if (b < p0 + 32) m &= ~0u << (int)(p0 + 32 - b);
    return m;
}

// The per-read work in two phases.  COUNT: the exact distance of every hit window of the read -> minimum, number of hits at
// the minimum, and whether the hits lie at more than one distance ("mixed").  WRITE: the positions at the minimum, ascending,
// to pos_out[base ...] -- for a read that is not mixed simply the set bits of its range; only mixed reads evaluate distances
// again.  (Clearing the losing bits in place instead would break reads whose borders overlap -- the API allows them.)
// Reads longer than HR_LONG positions are walked by their whole wave, 64 hit words per step, in both phases.
constexpr int HR_CHUNK = 6;              // hit words fetched together (a 150-bp read spans 5 - 6, a 300-bp read 10 - 11)
constexpr int HR_MIXED = 0x40;           // flag bit in the min_dist byte between the two kernels of the two-pass form
struct HrCtx {
    const uint32_t *__restrict__ hit32;
    const uint32_t *__restrict__ codes;
    const uint16_t *__restrict__ inval;
    int64_t n;
    int64_t n_alloc;                     // groups the code / flag arrays hold (kmap_packed_groups(n))
    int k, revcom, d_inv, radius;
    uint32_t km, cons, rcc;
    int64_t uni_len = 0, uni_stride = 0;  // stride > 0: read s = [s * stride, s * stride + len) -- the borders need not be loaded
};
struct HrRead {
    int64_t st = 0, stop = 0;
    bool quirk = false;                  // negative slice stop: every window runs off the read (all-ones hash)
    int best = 127, count = 0;
    bool mixed = false;
};
__device__ __forceinline__ void hr_setup(const HrCtx &c, const int64_t *__restrict__ borders, int64_t s, int64_t n_seq, HrRead &r) {
    if (s < n_seq) {
        int64_t en;
        if (c.uni_stride > 0) {                                          // wave-uniform: the layout was verified on the device
            r.st = s * c.uni_stride;
            en = r.st + c.uni_len;
        } else {
            r.st = borders[2 * s];
            en = borders[2 * s + 1];
        }
        if (r.st < 0) r.st = 0;
        if (en > c.n) en = c.n;
        const int64_t L = en > r.st ? en - r.st : 0;
        r.quirk = (L - c.k + 1 < 0);
        r.stop = hr_slice_stop(L, c.k);
    }
}
template <bool CHECK_INVALID>
__device__ __forceinline__ void hr_count(const HrCtx &c, HrRead &r) {
    const int lane = threadIdx.x & 63;
    int64_t stop = r.stop;
    if (r.quirk) {
        r.best = c.d_inv <= c.radius ? c.d_inv : 127;
        r.count = c.d_inv <= c.radius ? (int)stop : 0;
        stop = 0;
    }
    const bool is_long = stop > HR_LONG;
    if (stop > 0 && !is_long) {
        // 32-bit arithmetic relative to the read's first hit word (a short read spans at most 33 words)
        const uint32_t *hw = c.hit32 + (r.st >> 5);
        const int a_off = (int)(r.st & 31), end = a_off + (int)stop;       // bit range [a_off, end) of the word stream
        const int nw = (end + 31) >> 5;
        const int64_t word0 = (r.st >> 5) << 5;                           // absolute position of bit 0 of the stream
        for (int j0 = 0; j0 < nw; j0 += HR_CHUNK) {
            uint32_t xs[HR_CHUNK];                                          // the chunk's words in ONE round trip, not one per word
#pragma unroll
            for (int t = 0; t < HR_CHUNK; ++t) xs[t] = hw[min(j0 + t, nw - 1)];
            // ONE loop over the chunk's hits (three 64-bit words, earliest position in the top bit): a wave runs it as often as its
            // busiest read has hits.  One loop per 32-bit word ran each of the six as often as ITS busiest lane had hits -- about
            // twice the divergent iterations, each with the round trip of hr_dist's code loads inside.
            uint64_t q[HR_CHUNK / 2];
#pragma unroll
            for (int t = 0; t < HR_CHUNK; ++t) {
                const int j = j0 + t;
                uint32_t x = j < nw ? xs[t] : 0u;
                if (j == 0) x &= ~0u >> a_off;
                if (j == nw - 1) x &= ~0u << (32 * nw - end);
                if (t & 1) q[t >> 1] |= x;
                else q[t >> 1] = (uint64_t)x << 32;
            }
            static_assert(HR_CHUNK == 6, "three 64-bit words");
            for (;;) {
                const int sel = q[0] ? 0 : (q[1] ? 1 : 2);
                const uint64_t cur = q[0] ? q[0] : (q[1] ? q[1] : q[2]);
                if (!cur) break;
                const int tb = 63 - __builtin_clzll(cur);
                const uint64_t clr = ~(1ull << tb);
                q[0] &= sel == 0 ? clr : ~0ull;
                q[1] &= sel == 1 ? clr : ~0ull;
                q[2] &= sel == 2 ? clr : ~0ull;
                const int rel = 32 * j0 + 64 * sel + 63 - tb;
                const int d = hr_dist<CHECK_INVALID>(c.codes, c.inval, word0 + rel, c.k, c.km, c.cons, c.rcc, c.revcom);
                if (d < r.best) { r.mixed = r.mixed || r.count > 0; r.best = d; r.count = 1; }
                else if (d == r.best) ++r.count;
                else r.mixed = true;
            }
        }
    }
    unsigned long long todo = __ballot(is_long);
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int64_t a = __shfl(r.st, src), b = a + __shfl(stop, src);
        const int64_t w0 = a >> 5, w1 = (b - 1) >> 5;
        int m = 127, cc = 0, total_c = 0;
        for (int64_t wi = w0 + lane; wi <= w1; wi += 64) {
            uint32_t x = c.hit32[wi] & hr_mask(wi, a, b);
            total_c += __builtin_popcount(x);
            while (x) {
                const int tb = 31 - __builtin_clz(x);
                x &= ~(1u << tb);
                const int d = hr_dist<CHECK_INVALID>(c.codes, c.inval, (wi << 5) + (31 - tb), c.k, c.km, c.cons, c.rcc, c.revcom);
                if (d < m) { m = d; cc = 1; }
                else if (d == m) ++cc;
            }
        }
        int gm = m;
        for (int o = 32; o > 0; o >>= 1) {
            const int v = __shfl_xor(gm, o);
            gm = v < gm ? v : gm;
        }
        cc = (m == gm) ? cc : 0;
        for (int o = 32; o > 0; o >>= 1) {
            cc += __shfl_xor(cc, o);
            total_c += __shfl_xor(total_c, o);
        }
        if (lane == src) {
            r.best = gm;
            r.count = cc;
            r.mixed = total_c != cc;             // some hit of the read lies above its minimum
        }
    }
}
// cap: capacity of pos_out (writes beyond it are dropped; the caller re-runs with a larger buffer)
template <bool CHECK_INVALID>
__device__ __forceinline__ void hr_write(const HrCtx &c, const HrRead &r, uint64_t base, int32_t *__restrict__ pos_out, uint64_t cap) {
    const int lane = threadIdx.x & 63;
    int64_t stop = r.count ? r.stop : 0;                                   // nothing to write for a read without hits
    if (r.quirk) {
        for (int64_t p = 0; p < stop; ++p)
            if (base + p < cap) pos_out[base + p] = (int32_t)p;
        stop = 0;
    }
    const bool is_long = stop > HR_LONG;
    if (stop > 0 && !is_long) {
        const uint32_t *hw = c.hit32 + (r.st >> 5);
        const int a_off = (int)(r.st & 31), end = a_off + (int)stop;
        const int nw = (end + 31) >> 5;
        const int64_t word0 = (r.st >> 5) << 5;
        for (int j0 = 0; j0 < nw; j0 += HR_CHUNK) {
            uint32_t xs[HR_CHUNK];
#pragma unroll
            for (int t = 0; t < HR_CHUNK; ++t) xs[t] = hw[min(j0 + t, nw - 1)];
            uint64_t q[HR_CHUNK / 2];                                       // one loop over the chunk's hits, as in hr_count
#pragma unroll
            for (int t = 0; t < HR_CHUNK; ++t) {
                const int j = j0 + t;
                uint32_t x = j < nw ? xs[t] : 0u;
                if (j == 0) x &= ~0u >> a_off;
                if (j == nw - 1) x &= ~0u << (32 * nw - end);
                if (t & 1) q[t >> 1] |= x;
                else q[t >> 1] = (uint64_t)x << 32;
            }
            for (;;) {                           // ascending positions: most significant bit first
                const int sel = q[0] ? 0 : (q[1] ? 1 : 2);
                const uint64_t cur = q[0] ? q[0] : (q[1] ? q[1] : q[2]);
                if (!cur) break;
                const int tb = 63 - __builtin_clzll(cur);
                const uint64_t clr = ~(1ull << tb);
                q[0] &= sel == 0 ? clr : ~0ull;
                q[1] &= sel == 1 ? clr : ~0ull;
                q[2] &= sel == 2 ? clr : ~0ull;
                const int rel = 32 * j0 + 64 * sel + 63 - tb;              // position relative to the stream's bit 0
                if (r.mixed && hr_dist<CHECK_INVALID>(c.codes, c.inval, word0 + rel, c.k, c.km, c.cons, c.rcc, c.revcom) != r.best) continue;
                if (base < cap) pos_out[base] = (int32_t)(rel - a_off);
                ++base;
            }
        }
    }
    unsigned long long todo = __ballot(is_long);
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int64_t a = __shfl(r.st, src), b = a + __shfl(stop, src);
        const int64_t w0 = a >> 5, w1 = (b - 1) >> 5;
        uint64_t wbase = __shfl(base, src);
        const bool mx = __shfl((int)r.mixed, src) != 0;
        const int bst = __shfl(r.best, src);
        for (int64_t c0 = w0; c0 <= w1; c0 += 64) {
            const int64_t wi = c0 + lane;
            uint32_t keep = (wi <= w1) ? (c.hit32[wi] & hr_mask(wi, a, b)) : 0u;
            if (mx) {                            // drop the hits above the read's minimum
                uint32_t x = keep;
                while (x) {
                    const int tb = 31 - __builtin_clz(x);
                    x &= ~(1u << tb);
                    if (hr_dist<CHECK_INVALID>(c.codes, c.inval, (wi << 5) + (31 - tb), c.k, c.km, c.cons, c.rcc, c.revcom) != bst) keep &= ~(1u << tb);
                }
            }
            const int cnt = __builtin_popcount(keep);
            int inc = cnt;
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(inc, o);
                if (lane >= o) inc += v;
            }
            uint64_t at = wbase + (uint64_t)(inc - cnt);
            while (keep) {
                const int tb = 31 - __builtin_clz(keep);
                keep &= ~(1u << tb);
                if (at < cap) pos_out[at] = (int32_t)((wi << 5) + (31 - tb) - a);
                ++at;
            }
            wbase += (uint64_t)__shfl(inc, 63);
        }
    }
}

// two-pass form: count kernel (also: the block's hit total) -> exclusive scan of the 1 / 256 as many block totals (caller) ->
// write kernel (the read's offset = its block's offset + the prefix of the counts inside the block, recomputed from hits[])
__device__ __forceinline__ unsigned int hr_block_prefix(unsigned int v, unsigned int *s_wave, unsigned int &block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    unsigned int off = 0;
    block_total = 0;
    for (int w = 0; w < HR_TPB / 64; ++w) {
        if (w < wave) off += s_wave[w];
        block_total += s_wave[w];
    }
    return off + inc - v;                                                  // exclusive prefix of v inside the block
}
template <bool WRITE, bool CHECK_INVALID>
__global__ __launch_bounds__(HR_TPB) void scan_hits_reads_kernel(HrCtx c, const int64_t *__restrict__ borders, int64_t n_seq,
                                                                 int32_t *__restrict__ hits, int8_t *__restrict__ min_dist,
                                                                 uint32_t *__restrict__ block_sums, const uint64_t *__restrict__ block_offs,
                                                                 int32_t *__restrict__ pos_out) {
    __shared__ unsigned int s_wave[HR_TPB / 64];
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    HrRead r;
    hr_setup(c, borders, s, n_seq, r);
    if (!WRITE) {
        hr_count<CHECK_INVALID>(c, r);
        if (s < n_seq) {
            hits[s] = r.count;
            min_dist[s] = (int8_t)(r.best <= c.radius ? (r.best | (r.mixed ? HR_MIXED : 0)) : -1);
        }
        unsigned int total;
        (void)hr_block_prefix((unsigned int)r.count, s_wave, total);
        if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
    } else {
        if (s < n_seq) {
            r.count = hits[s];
            if (r.count) {
                const int md = min_dist[s];
                r.mixed = (md & HR_MIXED) != 0;
                r.best = md & (HR_MIXED - 1);
                if (r.mixed) min_dist[s] = (int8_t)r.best;
            }
        }
        unsigned int total;
        const unsigned int in_block = hr_block_prefix((unsigned int)r.count, s_wave, total);
        hr_write<CHECK_INVALID>(c, r, block_offs[blockIdx.x] + in_block, pos_out, ~0ull);
    }
}

// One-pass form (the default): the per-read work is done ONCE.  A block counts its 256 reads, reserves room for all their positions
// with one fetch-add on a global cursor -- blocks therefore land in the buffer in the order they happen to finish -- and writes the
// positions right away (the hit words are still in L1; reads that are not "mixed" need no second distance evaluation).  What is
// left for after the scan of the block totals is a copy of every block's contiguous segment to its place in read order: 4 B per
// HIT instead of a second walk over every read's borders, hit words and codes (C5: 1.3 ms of 6.2).
// (A decoupled look-back over published block aggregates, which would keep read order in a single kernel, was built and measured
// in r03: 5.1 ms at C3 against 0.7 ms for the three launches -- on a part with eight L2s every agent-scope release / acquire of
// the status words is an L2 write-back / invalidate, paid once per block.  The unordered reservation needs no such hand-over.)
// Short reads in the one-pass form: all hit words of the read (up to HF_WORDS = 12: 352 positions) arrive in ONE round trip of six
// 8-byte loads, the masked words stay in registers for the write phase, and the first HF_KEEP positions at the running minimum are
// kept in registers while counting -- a read whose hits all lie at one distance (nearly all: one planted hit) is written from
// them without a second walk.  (r04 counters on the general form at the C5 shape: 1213 vector + 365 scalar instructions per 64
// reads at 57 % issue utilisation -- instruction-bound as much as latency-bound: two chunks of six word loads, mask building and
// the hit loop ran once for the counts and again for the positions.)
constexpr int HF_WORDS = 12, HF_KEEP = 4, HF_CURSORS = 64;
// (r04, second pass: 1010 vector + 490 scalar instructions per wave at C3, issue utilisation 0.85.  The hit loop popped "the earliest
// hit of six 64-bit words" through a chain of compares and selects per iteration -- and the compiler turned that loop, divergent
// through its `continue`s, into three nested ones glued together by dozens of mask operations.  Now: one plain loop per 64-bit
// word (ascending order as before), the distance from a per-read code pointer + 32-bit offsets, and the wave takes the form with
// NQ = 3 words (192 positions) when all its reads fit: half the loads, masks and loops for 150-bp reads.)
typedef uint32_t hf_u32x2 __attribute__((ext_vector_type(2), aligned(4)));
template <int NQ>
struct HfRead {
    uint64_t q[NQ];                      // the read's hit bits, position p of the word stream in bit 63 - (p & 63) of q[p >> 6]
    int keep[HF_KEEP];                   // first positions (relative to the read) at the minimum, ascending
    int a_off;                           // bit offset of the read's first position in the word stream
    uint32_t row;                        // index (dwords) into the block's code rows of the group holding the stream's bit 0
    const uint16_t *ib;                  // invalid flags of that group
};
template <int NQ>
constexpr int hf_row_words() { return 4 * NQ + 1; }   // code words of 64 NQ positions + the next group (a window runs into it)
__device__ __forceinline__ int hf_words(const HrRead &r) { return (int)(((r.st & 31) + r.stop + 31) >> 5); }
__device__ __forceinline__ bool hf_applies(const HrRead &r) {
    return !r.quirk && r.stop > 0 && r.stop <= 32 * HF_WORDS && hf_words(r) <= HF_WORDS;
}
// distance of the window `rel` positions into the word stream: hr_dist on the read's code words in LDS.  (From global memory every
// iteration of the divergent hit loops began with a dependent load of two code words -- ~5 exposed round trips per wave, 75 % of
// the wave-cycles waiting.  The read's 13 / 25 code words now arrive in the same round trip as its hit words and are parked in a
// per-lane LDS row: odd row length, no bank conflicts between the lanes' rows.)
template <bool CHECK_INVALID>
__device__ __forceinline__ int hf_dist(const HrCtx &c, const uint32_t *code_rows, uint32_t row, const uint16_t *__restrict__ ib, int rel) {
    const uint32_t g = (uint32_t)rel >> 4;
    const int i = rel & 15;
    const uint32_t hi = code_rows[row + g], lo = code_rows[row + g + 1];
    const uint32_t top = i ? __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * i) : hi;
    uint32_t h = (top >> (32 - 2 * c.k)) & c.km;
    if (CHECK_INVALID) {
        const uint64_t m = ((uint64_t)ib[g] << 32) | ((uint64_t)ib[g + 1] << 16) | ib[g + 2];
        if (((m >> (48 - i - c.k)) & ((1ull << c.k) - 1ull)) != 0) h = c.km;
    }
    int d = popc2(h ^ c.cons);
    if (c.revcom) {
        const int d2 = popc2(h ^ c.rcc);
        d = d2 < d ? d2 : d;
    }
    return d;
}
template <int NQ, bool CHECK_INVALID>
__device__ __forceinline__ void hf_count(const HrCtx &c, HrRead &r, HfRead<NQ> &f, bool active, uint32_t *code_rows) {
    constexpr int NG = hf_row_words<NQ>();
    f.a_off = (int)(r.st & 31);
    f.ib = c.inval + ((r.st >> 5) << 1);
    const int end = f.a_off + (int)r.stop;
    const int nw = active ? (end + 31) >> 5 : 0;
    const uint32_t *hw = c.hit32 + (r.st >> 5);
    uint32_t x[2 * NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {          // word pairs behind the read's last word repeat its last pair (inside the array)
        const int at = (2 * t < nw) ? 2 * t : (nw > 0 ? (nw - 1) & ~1 : 0);
        const hf_u32x2 v = *reinterpret_cast<const hf_u32x2 *>(hw + at);
        x[2 * t] = v.x;
        x[2 * t + 1] = v.y;
    }
    {   // the read's code words: NG consecutive groups from the one of the stream's bit 0, moved back where they would pass the end
        // of the array (the caller guarantees n_alloc >= NG); same round trip as the hit words above
        const int64_t g0 = active ? (r.st >> 5) << 1 : 0;
        const int64_t gb = g0 + NG <= c.n_alloc ? g0 : c.n_alloc - NG;
        typedef uint32_t hf_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
        // a wave's rows start at the wave's share of the block's array (the waves of a block may run different NQ)
        const uint32_t row0 = (threadIdx.x >> 6) * (64u * hf_row_words<HF_WORDS / 2>()) + (threadIdx.x & 63u) * (uint32_t)NG;
        uint32_t *my = code_rows + row0;
        hf_u32x4 v[NQ];
#pragma unroll
        for (int t = 0; t < NQ; ++t) v[t] = *reinterpret_cast<const hf_u32x4 *>(c.codes + gb + 4 * t);
        const uint32_t last = c.codes[gb + 4 * NQ];
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            my[4 * t] = v[t].x; my[4 * t + 1] = v[t].y; my[4 * t + 2] = v[t].z; my[4 * t + 3] = v[t].w;
        }
        my[4 * NQ] = last;
        f.row = row0 + (uint32_t)(g0 - gb);
    }
#pragma unroll
    for (int j = 0; j < 2 * NQ; ++j) {
        uint32_t m = j < nw ? ~0u : 0u;
        if (j == 0) m &= ~0u >> f.a_off;
        if (j == nw - 1) m &= ~0u << (32 * nw - end);
        x[j] &= m;
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) f.q[t] = ((uint64_t)x[2 * t] << 32) | x[2 * t + 1];
#pragma unroll
    for (int t = 0; t < HF_KEEP; ++t) f.keep[t] = 0;
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        uint64_t w = f.q[t];
        while (w) {                         // a wave runs this as often as its busiest read has hits in this word
            const int lz = __builtin_clzll(w);
            w &= ~(0x8000000000000000ull >> lz);
            const int rel = 64 * t + lz;
            const int d = hf_dist<CHECK_INVALID>(c, code_rows, f.row, f.ib, rel);
            const int p = rel - f.a_off;
            if (d < r.best) {
                r.mixed = r.mixed || r.count > 0;
                r.best = d;
                r.count = 1;
                f.keep[0] = p;
            } else if (d == r.best) {
#pragma unroll
                for (int u = 1; u < HF_KEEP; ++u)
                    if (r.count == u) f.keep[u] = p;
                ++r.count;
            } else {
                r.mixed = true;
            }
        }
    }
}
template <int NQ, bool CHECK_INVALID>
__device__ __forceinline__ void hf_write(const HrCtx &c, const HrRead &r, const HfRead<NQ> &f, uint64_t base, int32_t *__restrict__ pos_out, uint64_t cap,
                                         const uint32_t *code_rows) {
    if (r.count == 0) return;
    if (r.count <= HF_KEEP) {               // the positions are in registers
#pragma unroll
        for (int t = 0; t < HF_KEEP; ++t)
            if (t < r.count && base + t < cap) pos_out[base + t] = f.keep[t];
        return;
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) {          // many hits (or several distances): walk the saved words again
        uint64_t w = f.q[t];
        while (w) {
            const int lz = __builtin_clzll(w);
            w &= ~(0x8000000000000000ull >> lz);
            const int rel = 64 * t + lz;
            if (r.mixed && hf_dist<CHECK_INVALID>(c, code_rows, f.row, f.ib, rel) != r.best) continue;
            if (base < cap) pos_out[base] = rel - f.a_off;
            ++base;
        }
    }
}

template <bool CHECK_INVALID>
__global__ __launch_bounds__(HR_TPB) void scan_hits_reads_fused_kernel(HrCtx c, const int64_t *__restrict__ borders, int64_t n_seq,
                                                                       int32_t *__restrict__ hits, int8_t *__restrict__ min_dist,
                                                                       uint32_t *__restrict__ block_sums, uint64_t *__restrict__ block_ubase,
                                                                       unsigned long long *__restrict__ cursor, int32_t *__restrict__ tmp_pos,
                                                                       uint64_t cap) {
    __shared__ unsigned int s_wave[HR_TPB / 64];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_codes[HR_TPB * hf_row_words<HF_WORDS / 2>()];     // a row of code words per thread (25.6 KB)
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    HrRead r;
    hr_setup(c, borders, s, n_seq, r);
    const bool fast = hf_applies(r);
    const bool general = !fast && (r.stop > 0 || r.quirk);                  // long reads, the negative-slice quirk
    const bool narrow = __all(!fast || hf_words(r) <= 6);                   // wave-uniform: every short read of the wave fits three 64-bit words
    HfRead<3> f3;
    HfRead<HF_WORDS / 2> f6;
    if (narrow) hf_count<3, CHECK_INVALID>(c, r, f3, fast, s_codes);
    else hf_count<HF_WORDS / 2, CHECK_INVALID>(c, r, f6, fast, s_codes);
    if (__any(general)) {                                                   // wave-uniform: the general form for the lanes that need it
        HrRead g = r;
        if (!general) { g.stop = 0; g.quirk = false; }
        hr_count<CHECK_INVALID>(c, g);
        if (general) r = g;
    }
    if (s < n_seq) {
        hits[s] = r.count;
        min_dist[s] = (int8_t)(r.best <= c.radius ? r.best : -1);
    }
    unsigned int total;
    const unsigned int in_block = hr_block_prefix((unsigned int)r.count, s_wave, total);
    if (threadIdx.x == 0) {
        // HF_CURSORS reservation counters, each with its own region of the buffer (a block takes counter blockIdx mod HF_CURSORS):
        // tens of thousands of device-scope fetch-adds on ONE address serialise at the memory side
        const unsigned cu = blockIdx.x % HF_CURSORS;
        const unsigned long long region = cap / HF_CURSORS;
        unsigned long long base = 0;
        if (total) {
            const unsigned long long at = atomicAdd(&cursor[cu], (unsigned long long)total);
            base = at + total <= region ? (unsigned long long)cu * region + at : cap;     // region full: dropped, and *overflow says so
            if (at + total > region) atomicMax(&cursor[HF_CURSORS], 1ull);
        }
        s_base = base;
        block_sums[blockIdx.x] = total;
        block_ubase[blockIdx.x] = base;
    }
    __syncthreads();
    const uint64_t base = s_base + in_block;
    if (fast) {                                                             // writes behind `cap` are dropped (the caller falls back)
        if (narrow) hf_write<3, CHECK_INVALID>(c, r, f3, base, tmp_pos, cap, s_codes);
        else hf_write<HF_WORDS / 2, CHECK_INVALID>(c, r, f6, base, tmp_pos, cap, s_codes);
    }
    if (__any(general)) {
        HrRead g = r;
        if (!general) { g.count = 0; g.stop = 0; g.quirk = false; }
        hr_write<CHECK_INVALID>(c, g, base, tmp_pos, cap);
    }
}
// segment of block b: tmp[ubase[b] .. + sums[b]) -> pos[offs[b] ..]; one wave per segment
__global__ __launch_bounds__(256) void scan_reorder_kernel(const int32_t *__restrict__ tmp, const uint64_t *__restrict__ ubase,
                                                           const uint64_t *__restrict__ offs, const uint32_t *__restrict__ sums, int64_t n_blocks,
                                                           int32_t *__restrict__ pos) {
    const int lane = threadIdx.x & 63;
    for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < n_blocks; b += (int64_t)gridDim.x * 4) {
        const uint32_t m = sums[b];
        const int32_t *src = tmp + ubase[b];
        int32_t *dst = pos + offs[b];
        for (uint32_t i = lane; i < m; i += 64) dst[i] = src[i];
    }
}

unsigned grid_of(int64_t n, int64_t per) {
    const int64_t g = (n + per - 1) / per;
    return (unsigned)(g < 1 ? 1 : g);
}

int pc2_host(uint32_t x) { return __builtin_popcount((x | (x >> 1)) & 0x55555555u); }
uint32_t rc_host(uint32_t c, int k) {
    const uint32_t m = low_mask<uint32_t>(k);
    uint32_t com = m - c, r = com & 3u;
    for (int i = 0; i < k - 1; ++i) { r <<= 2; com >>= 2; r += com & 3u; }
    return r;
}

template <int K>
void launch_hits(const uint32_t *planes, const uint16_t *inval, int64_t n, int64_t n_alloc, const HitTab &tab, uint16_t *hit16, bool words,
                 hipStream_t st) {
    const char *v = getenv("KMAP_SCAN_PLANES");                          // "plain": the formulation without the index mode (tests compare the two)
    const bool idx = !(v && !strcmp(v, "plain"));
    const int64_t n_words = (n + 31) / 32;
    if (idx && words) hits_planes_idx_kernel<K, true><<<grid_of(n_words, 2 * BS_TPB), BS_TPB, 0, st>>>(planes, inval, n, n_alloc, tab, hit16);
    else if (idx) hits_planes_idx_kernel<K, false><<<grid_of(n_words, 2 * BS_TPB), BS_TPB, 0, st>>>(planes, inval, n, n_alloc, tab, hit16);
    else if (words) hits_planes_kernel<K, true><<<grid_of(n_words, BS_TPB), BS_TPB, 0, st>>>(planes, inval, n, n_alloc, tab, hit16);
    else hits_planes_kernel<K, false><<<grid_of(n_words, BS_TPB), BS_TPB, 0, st>>>(planes, inval, n, n_alloc, tab, hit16);
}

}  // namespace

// hit bits of all windows for up to 16 table entries (k <= 16); hit16: uint16[(n + 15) / 16 rounded up to even] in group order,
// or (words) uint32[(n + 31) / 32] with window i of word t in bit 31 - i
int kmap_bitslice_hits(const uint32_t *planes, const uint16_t *inval, int64_t n, int k, const uint64_t *cons, const int32_t *radius,
                       int n_cons, int revcom_pairs, uint16_t *hit16, bool words, hipStream_t st) {
    KMAP_REQUIRE(k >= 1 && k <= 16 && n_cons >= 1 && n_cons <= 16, "bitslice_hits: k / table size out of range");
    HitTab tab;
    memset(&tab, 0, sizeof tab);
    tab.n = n_cons;
    const uint32_t km = low_mask<uint32_t>(k);
    for (int c = 0; c < n_cons; ++c) {
        HitCons &e = tab.c[c];
        e.fwd = (uint32_t)cons[c] & km;
        e.rc = rc_host(e.fwd, k);
        e.two = revcom_pairs ? 1u : 0u;
        e.radius = radius[c];
        int d_inv = pc2_host((km ^ e.fwd) & km);
        if (e.two) {
            const int d2 = pc2_host((km ^ e.rc) & km);
            d_inv = d2 < d_inv ? d2 : d_inv;
        }
        e.inv_hit = d_inv <= e.radius ? ~0u : 0u;
    }
    const int64_t n_alloc = kmap_packed_groups(n);
    switch (k) {
#define KMAP_BS_CASE(KK) case KK: launch_hits<KK>(planes, inval, n, n_alloc, tab, hit16, words, st); break;
        KMAP_BS_CASE(1) KMAP_BS_CASE(2) KMAP_BS_CASE(3) KMAP_BS_CASE(4) KMAP_BS_CASE(5) KMAP_BS_CASE(6) KMAP_BS_CASE(7) KMAP_BS_CASE(8)
        KMAP_BS_CASE(9) KMAP_BS_CASE(10) KMAP_BS_CASE(11) KMAP_BS_CASE(12) KMAP_BS_CASE(13) KMAP_BS_CASE(14) KMAP_BS_CASE(15) KMAP_BS_CASE(16)
#undef KMAP_BS_CASE
    }
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

// the scan's per-read passes on the hit bits (counts + minimum, then -- after the caller's scan of the counts -- the positions)
namespace {
// flag[0] |= 1 when some read s is not [s * stride, s * stride + len)
__global__ __launch_bounds__(256) void borders_uniform_kernel(const int64_t *__restrict__ borders, int64_t n_seq, int64_t len, int64_t stride,
                                                              unsigned int *__restrict__ flag) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_seq) return;
    const bool bad = borders[2 * s] != s * stride || borders[2 * s + 1] != s * stride + len;
    if (bad && *reinterpret_cast<volatile unsigned int *>(flag) == 0u) atomicOr(flag, 1u);
}
}  // namespace
// The caller DECLARES that read s of `borders` is [s * stride, s * stride + len) (fixed-length reads: every BASELINE configuration);
// the declaration is verified on the device (one pass over the borders + a 4-byte read-back) and, if true, kept in the handle: the
// per-read kernels of later runs on the same border array take the borders from s instead of loading 16 bytes per read (0.8 GB of the
// pass's 6.7 GB at C5).  It holds until the handle sees other borders -- and must be renewed by the caller if the CONTENT of the array
// at that address changes (kmap_hip.h).  No automatic detection: a cache keyed by a device address could outlive the array it described.
int kmap_bitslice_declare_uniform(kmap_scan *s, const int64_t *borders, int64_t n_seq, int64_t len, int64_t stride, int *accepted, hipStream_t st) {
    s->geo_borders = nullptr; s->geo_n_seq = -1; s->geo_len = 0; s->geo_stride = 0;
    if (accepted) *accepted = 0;
    static const bool off = [] { const char *v = getenv("KMAP_SCAN_UNIFORM"); return v && v[0] == '0'; }();   // A/B switch: always load the borders
    if (off || n_seq < 1 || len < 0 || stride <= 0 || stride < len) return KMAP_OK;
    unsigned int *flag = nullptr;
    KMAP_TRY(kmap_scratch((void **)&flag, 64, st, KMAP_SLOT_D));
    KMAP_CHECK_HIP(hipMemsetAsync(flag, 0, 4, st));
    borders_uniform_kernel<<<(unsigned)((n_seq + 255) / 256), 256, 0, st>>>(borders, n_seq, len, stride, flag);
    unsigned int h = 1;
    KMAP_CHECK_HIP(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));
    if (h == 0) {
        s->geo_borders = borders; s->geo_n_seq = n_seq; s->geo_len = len; s->geo_stride = stride;
        if (accepted) *accepted = 1;
    }
    return KMAP_OK;
}
// the declared layout applies to this run's borders?
static void scan_geometry(const kmap_scan *s, const int64_t *borders, int64_t n_seq, HrCtx &c) {
    const bool same = s->geo_stride > 0 && s->geo_borders == borders && s->geo_n_seq == n_seq;
    c.uni_len = same ? s->geo_len : 0;
    c.uni_stride = same ? s->geo_stride : 0;
}

static HrCtx make_ctx(const uint32_t *hit32, const uint32_t *codes, const uint16_t *inval, int64_t n, int k, uint64_t cons, int revcom,
                      int radius) {
    HrCtx c;
    c.hit32 = hit32; c.codes = codes; c.inval = inval; c.n = n; c.n_alloc = kmap_packed_groups(n); c.k = k; c.revcom = revcom; c.radius = radius;
    c.km = low_mask<uint32_t>(k);
    c.cons = (uint32_t)cons & c.km;
    c.rcc = rc_host(c.cons, k);
    c.d_inv = pc2_host((c.km ^ c.cons) & c.km);
    if (revcom) {
        const int d2 = pc2_host((c.km ^ c.rcc) & c.km);
        c.d_inv = d2 < c.d_inv ? d2 : c.d_inv;
    }
    return c;
}

int kmap_bitslice_scan_reads(bool write, const uint32_t *hit32, const uint32_t *codes, const uint16_t *inval, int64_t n,
                             const int64_t *borders, int64_t n_seq, int k, uint64_t cons, int revcom, int radius, kmap_scan *s,
                             hipStream_t st) {
    HrCtx c = make_ctx(hit32, codes, inval, n, k, cons, revcom, radius);
    scan_geometry(s, borders, n_seq, c);
    const unsigned grid = grid_of(n_seq, HR_TPB);
    const bool chk = c.d_inv <= radius;            // only then can a hit be a window that touches an invalid position
    // s->offs doubles as [block offsets uint64 (n_blocks + 1) | block sums uint32 (n_blocks)]: n_seq + 1 uint64 are allocated
    uint32_t *bsums = reinterpret_cast<uint32_t *>(s->offs + (size_t)grid + 1);
#define KMAP_HR(W, C)                                                                                                      \
    scan_hits_reads_kernel<W, C><<<grid, HR_TPB, 0, st>>>(c, borders, n_seq, s->hits, s->mind, bsums, (const uint64_t *)s->offs, \
                                                          W ? s->pos : nullptr)
    if (!write) { if (chk) KMAP_HR(false, true); else KMAP_HR(false, false); }
    else { if (chk) KMAP_HR(true, true); else KMAP_HR(true, false); }
#undef KMAP_HR
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

// The whole per-read part of the scan: hits[] / min_dist[] per read, positions in read order in s->pos, *total_out hits.
// One pass + a segment copy when the hits fit the temporary buffer (2 n_seq entries, at least 2^20), else the two-pass form.
int kmap_bitslice_scan_reads_all(const uint32_t *hit32, const uint32_t *codes, const uint16_t *inval, int64_t n, const int64_t *borders,
                                 int64_t n_seq, int k, uint64_t cons, int revcom, int radius, kmap_scan *s, uint64_t *total_out,
                                 hipStream_t st) {
    HrCtx c = make_ctx(hit32, codes, inval, n, k, cons, revcom, radius);
    scan_geometry(s, borders, n_seq, c);
    const int64_t nblk = (n_seq + HR_TPB - 1) / HR_TPB;
    const bool chk = c.d_inv <= radius;
    if (c.n_alloc < hf_row_words<HF_WORDS / 2>()) {                          // an input shorter than one row of code words: the two-pass form
        KMAP_TRY(kmap_bitslice_scan_reads(false, hit32, codes, inval, n, borders, n_seq, k, cons, revcom, radius, s, st));
        KMAP_TRY(exclusive_scan_u32(reinterpret_cast<uint32_t *>(s->offs + nblk + 1), nblk, s->offs, st));
        uint64_t tot = 0;
        KMAP_CHECK_HIP(hipMemcpyAsync(&tot, s->offs + nblk, 8, hipMemcpyDeviceToHost, st));
        KMAP_CHECK_HIP(hipStreamSynchronize(st));
        KMAP_TRY(kmap_scan_reserve_pos(s, tot));
        *total_out = tot;
        if (tot == 0) return KMAP_OK;
        return kmap_bitslice_scan_reads(true, hit32, codes, inval, n, borders, n_seq, k, cons, revcom, radius, s, st);
    }
    // s->offs ((n_seq + 1) uint64 + 128 B) holds: block offsets uint64[nblk + 1] | block sums uint32[nblk] (padded to 8 B) |
    // unordered block bases uint64[nblk] | cursor uint64
    uint64_t *boffs = s->offs;
    uint32_t *bsums = reinterpret_cast<uint32_t *>(s->offs + nblk + 1);
    uint64_t *ubase = s->offs + nblk + 1 + (nblk + 1) / 2;
    // the reservation counters (+ overflow flag) live in the scratch arena behind the temporary positions
    const size_t book = (size_t)(nblk + 1 + (nblk + 1) / 2 + nblk) * 8;
    KMAP_REQUIRE(book <= ((size_t)n_seq + 1) * 8 + 128, "scan: block bookkeeping does not fit");
    const uint64_t cap = ((uint64_t)std::max<int64_t>(2 * n_seq, (int64_t)1 << 20) + 2 * HF_CURSORS - 1) / (2 * HF_CURSORS) * (2 * HF_CURSORS);   // HF_CURSORS even regions
    int32_t *tmp = nullptr;
    KMAP_TRY(kmap_scratch((void **)&tmp, cap * 4 + (HF_CURSORS + 1) * 8, st, KMAP_SLOT_PART));
    unsigned long long *cursor = reinterpret_cast<unsigned long long *>(tmp + cap);       // cap is even: 8-byte aligned
    KMAP_CHECK_HIP(hipMemsetAsync(cursor, 0, (HF_CURSORS + 1) * 8, st));
    if (chk) scan_hits_reads_fused_kernel<true><<<(unsigned)nblk, HR_TPB, 0, st>>>(c, borders, n_seq, s->hits, s->mind, bsums, ubase, cursor, tmp, cap);
    else scan_hits_reads_fused_kernel<false><<<(unsigned)nblk, HR_TPB, 0, st>>>(c, borders, n_seq, s->hits, s->mind, bsums, ubase, cursor, tmp, cap);
    KMAP_CHECK_HIP(hipGetLastError());
    KMAP_TRY(exclusive_scan_u32(bsums, nblk, boffs, st));
    uint64_t total = 0;
    unsigned long long overflow = 0;
    KMAP_CHECK_HIP(hipMemcpyAsync(&total, boffs + nblk, 8, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipMemcpyAsync(&overflow, cursor + HF_CURSORS, 8, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));
    KMAP_TRY(kmap_scan_reserve_pos(s, total));
    *total_out = total;
    if (total == 0) return KMAP_OK;
    if (!overflow) {
        const unsigned grid = (unsigned)std::min<int64_t>((nblk + 3) / 4, 8192);
        scan_reorder_kernel<<<grid, 256, 0, st>>>(tmp, ubase, boffs, bsums, nblk, s->pos);
        KMAP_CHECK_HIP(hipGetLastError());
        return KMAP_OK;
    }
    // more hits than a region of the temporary buffer holds (> 2 per read on average): count again with the mixed flags kept, then write in order
    KMAP_TRY(kmap_bitslice_scan_reads(false, hit32, codes, inval, n, borders, n_seq, k, cons, revcom, radius, s, st));
    KMAP_TRY(exclusive_scan_u32(bsums, nblk, boffs, st));
    return kmap_bitslice_scan_reads(true, hit32, codes, inval, n, borders, n_seq, k, cons, revcom, radius, s, st);
}

extern "C" {

int kmap_pack_planes_dev(const uint32_t *codes_dev, int64_t n, uint32_t *planes_dev, void *stream) {
    KMAP_REQUIRE(n >= 0, "pack_planes: negative size");
    const int64_t ng = kmap_packed_groups(n);
    KMAP_REQUIRE(codes_dev && planes_dev, "pack_planes: null pointer");
    KMAP_REQUIRE((((uintptr_t)codes_dev | (uintptr_t)planes_dev) & 7u) == 0u, "pack_planes: codes / planes must be 8-byte aligned (whole arrays, not offsets into them)");
    planes_kernel<<<grid_of(ng / 2, BS_TPB), BS_TPB, 0, as_stream(stream)>>>(codes_dev, ng / 2, planes_dev);   // ng is even
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

}  // extern "C"
