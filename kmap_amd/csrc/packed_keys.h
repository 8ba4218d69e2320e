// packed_keys.h -- the 16 uint32 k-mer keys of one 16-position group of the 2-bit packed reads (k <= 16), shared by the hash
// kernels of packed.hip and the partitioned histogram of counts_part.hip (which takes its keys straight from the packed reads
// instead of from a materialised 4 B/position hash array).
#pragma once
#include "common.h"

// keys[i] = hash of the window starting at position 16 g + i, or 0xFFFFFFFF when the window touches an invalid position, is a
// per-read duplicate (skip bit) or starts at / behind position n.  k = 16: the all-T 16-mer's hash IS 0xFFFFFFFF -- its valid
// windows are reported in n_ones (and leave as invalid keys).  32-bit windows: v_alignbit + shift; the 16 validity flags from
// one doubling pass over the 48-bit invalid stream.
__device__ __forceinline__ void packed_group_keys(const uint32_t *__restrict__ codes, const uint16_t *__restrict__ inval,
                                                  const uint32_t *__restrict__ skip, int64_t n, int k, int64_t g, uint32_t keys[16],
                                                  uint32_t &n_ones) {
    if (16 * g >= n) {                                                     // group behind the array (last tile of a scatter pass): no loads
#pragma unroll
        for (int i = 0; i < 16; ++i) keys[i] = 0xFFFFFFFFu;
        n_ones = 0;
        return;
    }
    const uint32_t hi = codes[g], lo = codes[g + 1];
    uint64_t bad = ((uint64_t)inval[g] << 32) | ((uint64_t)inval[g + 1] << 16) | inval[g + 2];
    for (int have = 1; have < k;) {
        const int step = (have <= k - have) ? have : k - have;
        bad |= bad << step;
        have += step;
    }
    uint32_t drop16 = (uint32_t)(bad >> 32) & 0xFFFFu;                     // windows 0..15 in bits 15..0
    if (skip) drop16 |= (skip[g >> 1] >> ((g & 1) ? 0 : 16)) & 0xFFFFu;
    const int64_t left = n - 16 * g;                                       // windows that start inside the array
    if (left < 16) drop16 |= left <= 0 ? 0xFFFFu : ((1u << (16 - (int)left)) - 1u);
    const int sh = 32 - 2 * k;
    n_ones = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t top = (i == 0) ? hi : __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * i);
        const uint32_t h = top >> sh;
        const uint32_t d = (uint32_t)__builtin_amdgcn_sbfe((int)drop16, 15 - i, 1);   // all ones when dropped
        n_ones += (~d & (uint32_t)(h == 0xFFFFFFFFu));
        keys[i] = h | d;
    }
}
