// hamdist_matrix.hip -- all-pairs Hamming matrix over sampled k-mers (the headline kernel).
//
// Replaces cal_samp_kmer_hamdist_mat (reference motif_discovery.py:759-808): n_uniq launches of
// cal_ham_dist_kernel (taichi_core.py:63-104) + per-label override (:789-800) + the Python block
// expansion (:705-730) become ONE launch that writes the expanded N x N uint8 matrix.
//
// Roofline: HBM-write bound.  Algorithmic bytes per launch = nrows*N (u8 out) + N*sizeof(hash)
// (+N gid bytes).
//
// Two kernels: hamdist_tile_kernel (k <= 16, 4-KiB pitch with an odd chunk count per row, N >= 4096: XCD-affine 4-KiB
// chunk tiles + one-hot codes, see its header below) and hamdist_matrix_kernel, the general path described here: per
// output byte the VALU does xor / lshr / or3 / bcnt / lshl_or.
// Mapping (wave64): a lane owns 16 consecutive columns -> one 16-byte store per row; a wave owns
// 1024 consecutive columns x 8 rows (short-lived waves keep the chip's stores close to memory order:
// 8 rows/wave 5.83 TB/s, 16: 5.45, 64: 4.95 at N = 50k); the row's hash is wave-uniform (v_readlane from a
// register holding the wave's row hashes), the 16 column hashes live in VGPRs for all rows of the wave.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int COLS_PER_LANE = 16;
constexpr int COLS_PER_WAVE = COLS_PER_LANE * KMAP_WAVE;   // 1024
constexpr int ROWS_PER_WAVE = 8;    // default (<= 64: row hashes live in one register per lane); see sweep in DESIGN.md
constexpr int WAVES_PER_BLOCK = 8;   // max; launch uses wpb <= this

// gid[i] = 0 for "compare all k bases"; g > 0 = index+1 of a consensus shorter than k:
// pairs with equal non-zero gid are compared on the first clen bases, i.e. (a^b) >> gshift[g].
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct ByteTab {   // passed by value in the kernarg segment (no H2D copy, no sync)
    uint8_t v[256];
};

__global__ void build_gid_kernel(const int32_t *__restrict__ label, int64_t n, ByteTab lab2gid, int n_lab,
                                 uint8_t *__restrict__ gid) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t l = label[i];
    gid[i] = (l >= 0 && l < n_lab) ? lab2gid.v[l] : 0;
}

__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint64_t rl(uint64_t v, int lane) {
    uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane);
    uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// popcount of non-zero 2-bit groups, written so that hipcc emits lshr / or3 / bcnt:
// popc((x | x>>1) | ~M) = popc2(x) + popc(~M)  -> the constant is folded into `bias`.
__device__ __forceinline__ uint32_t p2(uint32_t x) {   // = popc2(x) + 16
    return (uint32_t)__builtin_popcount(x | (x >> 1) | 0xAAAAAAAAu);
}
__device__ __forceinline__ uint32_t p2(uint64_t x) {   // = popc2(x) + 32
    return (uint32_t)__builtin_popcountll(x | (x >> 1) | 0xAAAAAAAAAAAAAAAAull);
}
template <typename H>
constexpr uint32_t pack_bias() {   // removes the +16/+32 of four packed p2() results
    return 0u - (uint32_t)(4 * sizeof(H)) * 0x01010101u;
}

// the row loop of one wave; VEC = the lane's 16 columns are in range and 16-byte aligned
template <typename H, bool VEC>
__device__ __forceinline__ void run_rows(const H (&b)[COLS_PER_LANE], const uint32_t (&gcol)[COLS_PER_LANE / 4], H arow,
                                         uint32_t grow, uint32_t srow, int rcount, uint8_t *orow, int64_t ld,
                                         int64_t col0, int64_t n) {
    for (int r = 0; r < rcount; ++r, orow += ld) {
        const H a = rl(arow, r);
        const uint32_t g = rl(grow, r);
        uint32_t w[COLS_PER_LANE / 4];
        if (g == 0) {   // wave-uniform branch: all k bases for every pair of this row
#pragma unroll
            for (int v = 0; v < COLS_PER_LANE / 4; ++v) {
                uint32_t acc = p2((H)(a ^ b[4 * v])) + pack_bias<H>();
                acc += p2((H)(a ^ b[4 * v + 1])) << 8;
                acc += p2((H)(a ^ b[4 * v + 2])) << 16;
                acc += p2((H)(a ^ b[4 * v + 3])) << 24;
                w[v] = acc;
            }
        } else {        // row belongs to a short consensus: same-group pairs use the first clen bases
            const uint32_t sh = rl(srow, r);
#pragma unroll
            for (int v = 0; v < COLS_PER_LANE / 4; ++v) {
                uint32_t acc = pack_bias<H>();
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    H x = a ^ b[4 * v + c];
                    bool same = ((gcol[v] >> (8 * c)) & 0xFFu) == g;
                    x = same ? (H)(x >> sh) : x;
                    acc += p2(x) << (8 * c);
                }
                w[v] = acc;
            }
        }
        if constexpr (VEC) {
            u32x4 o = {w[0], w[1], w[2], w[3]};
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(orow));   // write-once streaming output (5.02 vs 4.77 TB/s at N = 50 k)
        } else {
#pragma unroll
            for (int c = 0; c < COLS_PER_LANE; ++c)
                if (col0 + c < n) orow[c] = (uint8_t)(w[c >> 2] >> (8 * (c & 3)));
        }
    }
}

template <typename H>
__global__ __launch_bounds__(KMAP_WAVE *WAVES_PER_BLOCK) void hamdist_matrix_kernel(
    const H *__restrict__ kh, const uint8_t *__restrict__ gid, ByteTab gshift, int64_t n, H mask,
    int64_t row0, int64_t nrows, uint8_t *__restrict__ out, int64_t ld, int vec_ok, int rpw, int row_major, int wpb) {
    const int lane = threadIdx.x & (KMAP_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned bx = row_major ? blockIdx.y : blockIdx.x, by = row_major ? blockIdx.x : blockIdx.y;
    const int64_t col0 = (int64_t)bx * COLS_PER_WAVE + (int64_t)lane * COLS_PER_LANE;
    const int64_t rbase = ((int64_t)by * wpb + wave) * rpw;
    if (rbase >= nrows) return;   // wave-uniform

    // ---- the wave's 64 row hashes: one coalesced load, then v_readlane per row ----
    const int64_t myrow = row0 + rbase + lane;
    const bool rvalid = (rbase + lane < nrows);
    const H arow = rvalid ? (H)(kh[myrow] & mask) : (H)0;
    const uint32_t grow = rvalid ? (uint32_t)gid[myrow] : 0u;
    const uint32_t srow = gshift.v[grow];
    const int rcount = (int)((nrows - rbase < rpw) ? (nrows - rbase) : rpw);
    uint8_t *orow = out + rbase * ld + col0;

    // ---- the lane's 16 column hashes + group ids ----
    H b[COLS_PER_LANE];
    uint32_t gcol[COLS_PER_LANE / 4];
    // whole wave on the aligned interior? (wave-uniform: the last lane's columns are in range)
    const int64_t wave_col_end = (int64_t)bx * COLS_PER_WAVE + COLS_PER_WAVE;
    if (vec_ok && wave_col_end <= n) {
        constexpr int HV = 16 / sizeof(H);   // hashes per 16-byte load
        const u32x4 *src = reinterpret_cast<const u32x4 *>(kh + col0);
#pragma unroll
        for (int v = 0; v < COLS_PER_LANE / HV; ++v) {
            u32x4 t = src[v];
            if constexpr (sizeof(H) == 4) {
                b[4 * v + 0] = t.x; b[4 * v + 1] = t.y; b[4 * v + 2] = t.z; b[4 * v + 3] = t.w;
            } else {
                b[2 * v + 0] = ((uint64_t)t.y << 32) | t.x;
                b[2 * v + 1] = ((uint64_t)t.w << 32) | t.z;
            }
        }
        u32x4 g = *reinterpret_cast<const u32x4 *>(gid + col0);
        gcol[0] = g.x; gcol[1] = g.y; gcol[2] = g.z; gcol[3] = g.w;
#pragma unroll
        for (int c = 0; c < COLS_PER_LANE; ++c) b[c] &= mask;
        run_rows<H, true>(b, gcol, arow, grow, srow, rcount, orow, ld, col0, n);
    } else {
#pragma unroll
        for (int c = 0; c < COLS_PER_LANE; ++c) b[c] = (col0 + c < n) ? (H)(kh[col0 + c] & mask) : (H)0;
#pragma unroll
        for (int v = 0; v < COLS_PER_LANE / 4; ++v) {
            uint32_t w = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (col0 + 4 * v + c < n) w |= (uint32_t)gid[col0 + 4 * v + c] << (8 * c);
            gcol[v] = w;
        }
        run_rows<H, false>(b, gcol, arow, grow, srow, rcount, orow, ld, col0, n);
    }
}


// ---- tiled one-hot path (k <= 16) -----------------------------------------------------------------------------------
// Two measured facts about MI355X stores (tools/probes/write_bw4..8.hip) shape this kernel:
//  * the chip sustains 6.5-6.9 TB/s of stores only when (a) every 4-KiB-aligned chunk is written by one short-lived
//    workgroup, (b) each XCD keeps writing the same residue class of (chunk index mod 8) -- workgroups are dispatched
//    round-robin over the 8 XCDs, so block b serves residue b % 8 -- and (c) few chunks are in flight (R = 4 rows per
//    block x 4 resident blocks per CU, enforced with a 40-KiB dynamic-LDS reservation; sweep in DESIGN.md).  Row-strided tiles without these rules
//    stay at 5.3-5.8 TB/s, persistent blocks at 5.5.
//  * at 5.25 VALU ops per output byte the 2-bit formulation is itself within 10 % of the VALU ceiling (0.334 ms at
//    N = 50 k).  With one-hot codes (4 bits per base, built once per launch) a distance is popcount(a & ~b): and + bcnt +
//    pack = 2.75 ops per byte for k <= 8, 4.75 for k <= 16.
// Tile = one 4-KiB column block x R rows spaced 8 apart (all its chunks share one residue); requires ld % 4096 == 0 and
// ld / 4096 odd (kmap_amd.hamdist.pitch_for provides it).
constexpr int T_TPB = 256;
constexpr int T_LDS_THROTTLE = 40 * 1024;   // dynamic LDS reserved per block: four resident blocks per CU (rule (c) above)

template <typename H>
__global__ void build_codes_kernel(const H *__restrict__ kh, int64_t n, int k, H mask, int onehot, uint32_t *__restrict__ c0,
                                   uint32_t *__restrict__ c1, const int32_t *__restrict__ label, ByteTab lab2gid, int n_lab,
                                   uint8_t *__restrict__ gid) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t l = label[i];                       // group ids in the same pass (build_gid_kernel's job on the general path)
    gid[i] = (l >= 0 && l < n_lab) ? lab2gid.v[l] : 0;
    const uint64_t h = (uint64_t)(kh[i] & mask);
    if (onehot == 0) {          // 9 <= k <= 15: the tile kernel compares the 2-bit hashes themselves
        c0[i] = (uint32_t)h;
        return;
    }
    uint64_t code = 0;
    for (int b = 0; b < k; ++b) code |= 1ull << (4 * b + (int)((h >> (2 * (k - 1 - b))) & 3));   // base b -> nibble b
    c0[i] = (uint32_t)code;
    if (c1) c1[i] = (uint32_t)(code >> 32);
}

__device__ __forceinline__ uint32_t bcnt(uint32_t x, uint32_t acc) { return (uint32_t)__builtin_popcount(x) + acc; }

template <int CW, int R>
__global__ __launch_bounds__(T_TPB) void hamdist_tile_kernel(const uint32_t *__restrict__ c0, const uint32_t *__restrict__ c1,
                                                             const uint8_t *__restrict__ gid, ByteTab gshift, int k, int64_t n,
                                                             int64_t row0, int64_t nrows, uint8_t *__restrict__ out, int64_t ld,
                                                             int cb, int inv, int shift) {
    // grid = (8 * cb, groups): the linear workgroup id is y * 8cb + x, so (blockIdx.x & 7) is the XCD the block lands on
    const int x = (int)(blockIdx.x & 7);
    const int c = (int)(blockIdx.x >> 3);
    const int64_t g = blockIdx.y;
    const int rho = ((((x - c - shift) % 8 + 8) % 8) * inv) & 7;      // (ld/4096 * row + c + shift) % 8 == x
    // the tile's R row codes / group ids: one vector load per wave up front (lane j holds row j), v_readlane per row.
    // Nothing is loaded inside the row loop, so no s_waitcnt vmcnt ever waits on the stores already issued.  Loaded BEFORE
    // lanes past the last column leave: v_readlane must find lanes 0..R-1 written even in a wave that keeps only lane 0.
    const int lane = threadIdx.x & (KMAP_WAVE - 1);
    const int64_t my_rl = g * (8 * R) + rho + 8 * (lane < R ? lane : 0);
    const bool rv = lane < R && my_rl < nrows;
    const uint32_t ra0 = rv ? c0[row0 + my_rl] : 0u;
    const uint32_t ra1 = (CW == 2 && rv) ? c1[row0 + my_rl] : 0u;
    const uint32_t rgid = rv ? (uint32_t)gid[row0 + my_rl] : 0u;

    const int64_t col0 = (int64_t)c * 4096 + (int64_t)threadIdx.x * COLS_PER_LANE;
    // only whole waves leave (wave-uniform); lanes past the last column stay and are predicated off, so that every lane the
    // v_readlane's below read from is live
    if ((int64_t)c * 4096 + (int64_t)(threadIdx.x & ~(KMAP_WAVE - 1)) * COLS_PER_LANE >= n) return;
    const bool full = col0 + COLS_PER_LANE <= n;

    uint32_t nb0[COLS_PER_LANE], nb1[CW == 2 ? COLS_PER_LANE : 1], gcol[COLS_PER_LANE / 4];
    if (full) {
#pragma unroll
        for (int v = 0; v < COLS_PER_LANE / 4; ++v) {
            const u32x4 t = *reinterpret_cast<const u32x4 *>(c0 + col0 + 4 * v);
            nb0[4 * v] = t.x; nb0[4 * v + 1] = t.y; nb0[4 * v + 2] = t.z; nb0[4 * v + 3] = t.w;
            if constexpr (CW != 0) {   // one-hot: keep the complement
                nb0[4 * v] = ~t.x; nb0[4 * v + 1] = ~t.y; nb0[4 * v + 2] = ~t.z; nb0[4 * v + 3] = ~t.w;
            }
            if constexpr (CW == 2) {
                const u32x4 u = *reinterpret_cast<const u32x4 *>(c1 + col0 + 4 * v);
                nb1[4 * v] = ~u.x; nb1[4 * v + 1] = ~u.y; nb1[4 * v + 2] = ~u.z; nb1[4 * v + 3] = ~u.w;
            }
        }
        const u32x4 gg = *reinterpret_cast<const u32x4 *>(gid + col0);
        gcol[0] = gg.x; gcol[1] = gg.y; gcol[2] = gg.z; gcol[3] = gg.w;
    } else {
#pragma unroll
        for (int cc = 0; cc < COLS_PER_LANE; ++cc) {
            nb0[cc] = (col0 + cc < n) ? (CW != 0 ? ~c0[col0 + cc] : c0[col0 + cc]) : 0u;
            if constexpr (CW == 2) nb1[cc] = (col0 + cc < n) ? ~c1[col0 + cc] : 0u;
        }
#pragma unroll
        for (int v = 0; v < COLS_PER_LANE / 4; ++v) {
            uint32_t w = 0;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                if (col0 + 4 * v + cc < n) w |= (uint32_t)gid[col0 + 4 * v + cc] << (8 * cc);
            gcol[v] = w;
        }
    }

    // do this lane's 16 columns carry one group id (lanes past the last column: no)
    const uint32_t lane_g = gcol[0] & 0xFFu;
    const bool lane_uniform = full && gcol[0] == lane_g * 0x01010101u && gcol[1] == gcol[0] && gcol[2] == gcol[0] && gcol[3] == gcol[0];

#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int64_t rloc = g * (8 * R) + rho + 8 * j;          // row inside this call's output (wave-uniform)
        if (rloc >= nrows) break;
        const uint32_t a0 = rl(ra0, j);
        const uint32_t a1 = (CW == 2) ? rl(ra1, j) : 0u;
        const uint32_t rg = rl(rgid, j);
        uint32_t w[COLS_PER_LANE / 4];
        if constexpr (CW == 0) {   // 2-bit hashes: xor / lshr / or3 / bcnt / pack, as in hamdist_matrix_kernel
            if (rg == 0) {
#pragma unroll
                for (int v = 0; v < COLS_PER_LANE / 4; ++v) {
                    uint32_t acc = p2(a0 ^ nb0[4 * v]) + pack_bias<uint32_t>();
                    acc += p2(a0 ^ nb0[4 * v + 1]) << 8;
                    acc += p2(a0 ^ nb0[4 * v + 2]) << 16;
                    acc += p2(a0 ^ nb0[4 * v + 3]) << 24;
                    w[v] = acc;
                }
            } else {
                const uint32_t sh = (uint32_t)gshift.v[rg];
                const bool w_same = __all(lane_uniform && lane_g == rg), w_none = __all(lane_uniform && lane_g != rg);
                if (w_same || w_none) {   // whole wave inside / outside the row's group (see the one-hot branch below)
                    const uint32_t s2 = w_same ? sh : 0u;
#pragma unroll
                    for (int v = 0; v < COLS_PER_LANE / 4; ++v) {
                        uint32_t acc = p2((a0 ^ nb0[4 * v]) >> s2) + pack_bias<uint32_t>();
                        acc += p2((a0 ^ nb0[4 * v + 1]) >> s2) << 8;
                        acc += p2((a0 ^ nb0[4 * v + 2]) >> s2) << 16;
                        acc += p2((a0 ^ nb0[4 * v + 3]) >> s2) << 24;
                        w[v] = acc;
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < COLS_PER_LANE / 4; ++v) {
                        uint32_t acc = pack_bias<uint32_t>();
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            uint32_t xx = a0 ^ nb0[4 * v + cc];
                            if (((gcol[v] >> (8 * cc)) & 0xFFu) == rg) xx >>= sh;
                            acc += p2(xx) << (8 * cc);
                        }
                        w[v] = acc;
                    }
                }
            }
        } else if (rg == 0) {
#pragma unroll
            for (int v = 0; v < COLS_PER_LANE / 4; ++v) {
                uint32_t acc = 0;
#pragma unroll
                for (int cc = 3; cc >= 0; --cc) {
                    uint32_t d = bcnt(a0 & nb0[4 * v + cc], 0u);
                    if constexpr (CW == 2) d = bcnt(a1 & nb1[4 * v + cc], d);
                    acc = (acc << 8) | d;
                }
                w[v] = acc;
            }
        } else {   // row of a short consensus: same-group pairs are compared on the first clen bases (= low 4*clen code bits)
            const int clen = k - (int)(gshift.v[rg] >> 1);
            const uint64_t pm = (clen >= 16) ? ~0ull : ((1ull << (4 * clen)) - 1ull);
            const uint32_t p0 = a0 & (uint32_t)pm, p1 = a1 & (uint32_t)(pm >> 32);
            // labels arrive grouped (sample_disp_kmer emits label after label), so nearly every wave's 1024 columns lie entirely
            // inside the row's group or entirely outside it: a wave-uniform test picks the row code once (prefix-masked or whole)
            // and the row runs at the plain rows' 2.75 ops per byte; only the waves on a group boundary select per byte
            const bool w_same = __all(lane_uniform && lane_g == rg), w_none = __all(lane_uniform && lane_g != rg);
            if (w_same || w_none) {
                const uint32_t e0 = w_same ? p0 : a0, e1 = w_same ? p1 : a1;
#pragma unroll
                for (int v = 0; v < COLS_PER_LANE / 4; ++v) {
                    uint32_t acc = 0;
#pragma unroll
                    for (int cc = 3; cc >= 0; --cc) {
                        uint32_t d = bcnt(e0 & nb0[4 * v + cc], 0u);
                        if constexpr (CW == 2) d = bcnt(e1 & nb1[4 * v + cc], d);
                        acc = (acc << 8) | d;
                    }
                    w[v] = acc;
                }
            } else {
#pragma unroll
                for (int v = 0; v < COLS_PER_LANE / 4; ++v) {
                    uint32_t acc = 0;
#pragma unroll
                    for (int cc = 3; cc >= 0; --cc) {
                        const bool same = ((gcol[v] >> (8 * cc)) & 0xFFu) == rg;
                        uint32_t d = bcnt((same ? p0 : a0) & nb0[4 * v + cc], 0u);
                        if constexpr (CW == 2) d = bcnt((same ? p1 : a1) & nb1[4 * v + cc], d);
                        acc = (acc << 8) | d;
                    }
                    w[v] = acc;
                }
            }
        }
        uint8_t *orow = out + rloc * ld + col0;
        if (full) {
            u32x4 o = {w[0], w[1], w[2], w[3]};
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(orow));   // write-once streaming output (5.02 vs 4.77 TB/s at N = 50 k)
        } else {
#pragma unroll
            for (int cc = 0; cc < COLS_PER_LANE; ++cc)
                if (col0 + cc < n) orow[cc] = (uint8_t)(w[cc >> 2] >> (8 * (cc & 3)));
        }
    }
}

template <int CW, int R>
int launch_tile(const uint32_t *c0, const uint32_t *c1, const uint8_t *gid, const ByteTab &gshift, int k, int64_t n, int64_t row0,
                int64_t nrows, uint8_t *out, int64_t ld, hipStream_t st) {
    const int cb = (int)((n + 4095) / 4096);
    const int cpr = (int)(ld >> 12);
    int inv = 1;
    for (int t = 1; t < 8; t += 2)
        if (((cpr * t) & 7) == 1) inv = t;
    const int shift = (int)(((uintptr_t)out >> 12) & 7);
    const int64_t groups = (nrows + 8 * R - 1) / (8 * R);
    KMAP_REQUIRE(groups <= 65535, "hamdist_matrix: nrows too large for one launch (%lld)", (long long)nrows);
    const dim3 blocks((unsigned)(8 * cb), (unsigned)groups);
    KMAP_TRY(kmap_allow_lds((const void *)hamdist_tile_kernel<CW, R>, 160 * 1024));
    hamdist_tile_kernel<CW, R><<<blocks, T_TPB, T_LDS_THROTTLE, st>>>(c0, c1, gid, gshift, k, n, row0, nrows, out, ld, cb, inv, shift);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

template <typename H>
int launch_matrix(const H *kh_dev, const int32_t *label_dev, int64_t n, int k, const int32_t *clen, int n_lab,
                  int64_t row0, int64_t nrows, uint8_t *out_dev, int64_t ld, void *stream) {
    KMAP_REQUIRE(k > 0 && 2 * k <= (int)(8 * sizeof(H)) && k < 32, "hamdist_matrix: k=%d out of range for %zu-byte hash",
                 k, sizeof(H));
    KMAP_REQUIRE(n >= 0 && nrows >= 0 && row0 >= 0 && row0 + nrows <= n, "hamdist_matrix: bad row range");
    KMAP_REQUIRE(ld >= n, "hamdist_matrix: ld < n");
    KMAP_REQUIRE(n_lab >= 0 && (n_lab == 0 || clen), "hamdist_matrix: clen missing");
    if (n == 0 || nrows == 0) return KMAP_OK;
    KMAP_REQUIRE(kh_dev && label_dev && out_dev, "hamdist_matrix: null pointer");
    hipStream_t st = as_stream(stream);

    // label -> group id table (host, tiny)
    ByteTab lab2gid, gshift;
    memset(&lab2gid, 0, sizeof lab2gid);
    memset(&gshift, 0, sizeof gshift);
    int n_short = 0;
    KMAP_REQUIRE(n_lab <= 255, "hamdist_matrix: more than 255 consensus labels");
    for (int l = 0; l < n_lab; ++l) {
        KMAP_REQUIRE(clen[l] > 0 && clen[l] <= k, "hamdist_matrix: clen[%d]=%d not in (0,k]", l, clen[l]);
        if (clen[l] < k) {
            lab2gid.v[l] = (uint8_t)(++n_short);
            gshift.v[n_short] = (uint8_t)(2 * (k - clen[l]));
        }
    }
    // scratch: gid[n] (cached arena; the tables travel in the kernarg segment)
    uint8_t *gid = nullptr;
    KMAP_TRY(kmap_scratch((void **)&gid, ((size_t)n + 15) & ~(size_t)15, st, KMAP_SLOT_A));

    // tiled one-hot path: k <= 16, 4-KiB row pitch with an odd number of chunks per row, at least one full column block.
    // Rows per tile: 4 for the one-word one-hot compare (2.75 ops per byte), 8 for the heavier compares (k = 12: 0.428 vs 0.464 ms)
    if (k <= 16 && (ld % 4096) == 0 && ((ld >> 12) & 1) && n >= 4096 && ((uintptr_t)out_dev % 4096) == 0) {
        uint32_t *codes = nullptr;
        const size_t npad = ((size_t)n + 63) & ~(size_t)63;
        KMAP_TRY(kmap_scratch((void **)&codes, npad * 8, st, KMAP_SLOT_B));
        const int onehot = (k <= 8) ? 1 : (k == 16) ? 2 : 0;      // 9..15: 2-bit compare (one register per column)
        uint32_t *c0 = codes, *c1 = (onehot == 2) ? codes + npad : nullptr;
        build_codes_kernel<H><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(kh_dev, n, k, low_mask<H>(k), onehot, c0, c1,
                                                                                       label_dev, lab2gid, n_lab, gid);
        if (onehot == 1) return launch_tile<1, 4>(c0, c1, gid, gshift, k, n, row0, nrows, out_dev, ld, st);
        if (onehot == 0) return launch_tile<0, 8>(c0, c1, gid, gshift, k, n, row0, nrows, out_dev, ld, st);
        return launch_tile<2, 8>(c0, c1, gid, gshift, k, n, row0, nrows, out_dev, ld, st);
    }
    // general kernel: k > 16, small or unaligned outputs
    build_gid_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(label_dev, n, lab2gid, n_lab, gid);
    const int vec_ok = ((uintptr_t)kh_dev % 16 == 0) && ((uintptr_t)out_dev % 16 == 0) && (ld % 16 == 0);
    constexpr int rpw = ROWS_PER_WAVE, wpb = 2;
    const unsigned gx = (unsigned)((n + COLS_PER_WAVE - 1) / COLS_PER_WAVE);
    const unsigned gy = (unsigned)((nrows + (int64_t)rpw * wpb - 1) / ((int64_t)rpw * wpb));
    KMAP_REQUIRE(gy <= 65535u, "hamdist_matrix: nrows too large for one launch (%lld)", (long long)nrows);
    hamdist_matrix_kernel<H><<<dim3(gx, gy), dim3(KMAP_WAVE * wpb), 0, st>>>(kh_dev, gid, gshift, n, low_mask<H>(k), row0, nrows, out_dev, ld,
                                                                             vec_ok, rpw, 0, wpb);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

}  // namespace

extern "C" {

int64_t kmap_hamdist_pitch(int64_t n) {
    if (n < 4096) return (n + 255) & ~(int64_t)255;
    return (((n + 4095) >> 12) | 1) << 12;
}

int kmap_hamdist_matrix_u32_dev(const uint32_t *kh_dev, const int32_t *label_dev, int64_t n, int k, const int32_t *clen,
                                int n_lab, int64_t row0, int64_t nrows, uint8_t *out_dev, int64_t ld, void *stream) {
    KMAP_REQUIRE(k < 16, "hamdist_matrix_u32: k=%d needs the u64 entry point", k);
    return launch_matrix<uint32_t>(kh_dev, label_dev, n, k, clen, n_lab, row0, nrows, out_dev, ld, stream);
}
int kmap_hamdist_matrix_u64_dev(const uint64_t *kh_dev, const int32_t *label_dev, int64_t n, int k, const int32_t *clen,
                                int n_lab, int64_t row0, int64_t nrows, uint8_t *out_dev, int64_t ld, void *stream) {
    return launch_matrix<uint64_t>(kh_dev, label_dev, n, k, clen, n_lab, row0, nrows, out_dev, ld, stream);
}

int kmap_hamdist_matrix_u8(const uint64_t *kh, const int32_t *label, int64_t n, int k, const int32_t *clen, int n_lab,
                           uint8_t *out) {
    KMAP_REQUIRE(n >= 0 && k > 0 && k < 32, "hamdist_matrix_u8: bad n/k");
    if (n == 0) return KMAP_OK;
    KMAP_REQUIRE(kh && label && out, "hamdist_matrix_u8: null pointer");
    const int64_t ld = kmap_hamdist_pitch(n);
    DevBuf dkh, dlab, dout;
    KMAP_TRY(dkh.alloc((size_t)n * 8));
    KMAP_TRY(dlab.alloc((size_t)n * 4));
    KMAP_TRY(dout.alloc((size_t)n * ld));
    KMAP_CHECK_HIP(hipMemcpy(dkh.p, kh, (size_t)n * 8, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(dlab.p, label, (size_t)n * 4, hipMemcpyHostToDevice));
    KMAP_TRY(launch_matrix<uint64_t>(dkh.as<uint64_t>(), dlab.as<int32_t>(), n, k, clen, n_lab, 0, n,
                                     dout.as<uint8_t>(), ld, nullptr));
    KMAP_CHECK_HIP(hipMemcpy2D(out, (size_t)n, dout.p, (size_t)ld, (size_t)n, (size_t)n, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

}  // extern "C"
