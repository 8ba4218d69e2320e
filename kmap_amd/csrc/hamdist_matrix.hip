// hamdist_matrix.hip -- all-pairs Hamming matrix over sampled k-mers (the headline kernel).
//
// Replaces cal_samp_kmer_hamdist_mat (reference motif_discovery.py:759-808): n_uniq launches of
// cal_ham_dist_kernel (taichi_core.py:63-104) + per-label override (:789-800) + the Python block
// expansion (:705-730) become ONE launch that writes the expanded N x N uint8 matrix.
//
// Roofline: HBM-write bound.  Algorithmic bytes per launch = nrows*N (u8 out) + N*sizeof(hash)
// (+N gid bytes).  Per output byte the VALU does xor / lshr / or3 / bcnt / lshl_or.
//
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
template <typename H, bool NT, bool VEC>
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
            if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(orow));
            else *reinterpret_cast<u32x4 *>(orow) = o;
        } else {
#pragma unroll
            for (int c = 0; c < COLS_PER_LANE; ++c)
                if (col0 + c < n) orow[c] = (uint8_t)(w[c >> 2] >> (8 * (c & 3)));
        }
    }
}

template <typename H, bool NT>
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
        run_rows<H, NT, true>(b, gcol, arow, grow, srow, rcount, orow, ld, col0, n);
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
        run_rows<H, NT, false>(b, gcol, arow, grow, srow, rcount, orow, ld, col0, n);
    }
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
    build_gid_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(label_dev, n, lab2gid, n_lab, gid);

    const int vec_ok = ((uintptr_t)kh_dev % 16 == 0) && ((uintptr_t)out_dev % 16 == 0) && (ld % 16 == 0);
    static const int rpw = [] { const char *e = getenv("KMAP_HAMDIST_RPW"); int v = e ? atoi(e) : ROWS_PER_WAVE; return (v >= 1 && v <= 64) ? v : ROWS_PER_WAVE; }();
    static const int row_major = [] { const char *e = getenv("KMAP_HAMDIST_ROWMAJOR"); return e ? atoi(e) : 0; }();
    const unsigned gx = (unsigned)((n + COLS_PER_WAVE - 1) / COLS_PER_WAVE);
    static const int wpb = [] { const char *e = getenv("KMAP_HAMDIST_WPB"); int v = e ? atoi(e) : 2; return (v >= 1 && v <= WAVES_PER_BLOCK) ? v : 2; }();
    const unsigned gy = (unsigned)((nrows + (int64_t)rpw * wpb - 1) / ((int64_t)rpw * wpb));
    dim3 grid = row_major ? dim3(gy, gx) : dim3(gx, gy);
    KMAP_REQUIRE(grid.y <= 65535u, "hamdist_matrix: nrows too large for one launch (%lld)", (long long)nrows);
    // non-temporal stores by default (write-once streaming output: 5.02 vs 4.77 TB/s measured at N=50k);
    // KMAP_HAMDIST_NT=0 switches back to default-policy stores for A/B runs
    static const bool nt = !(getenv("KMAP_HAMDIST_NT") && getenv("KMAP_HAMDIST_NT")[0] == '0');
    if (nt)
        hamdist_matrix_kernel<H, true><<<grid, dim3(KMAP_WAVE * wpb), 0, st>>>(
            kh_dev, gid, gshift, n, low_mask<H>(k), row0, nrows, out_dev, ld, vec_ok, rpw, row_major, wpb);
    else
        hamdist_matrix_kernel<H, false><<<grid, dim3(KMAP_WAVE * wpb), 0, st>>>(
            kh_dev, gid, gshift, n, low_mask<H>(k), row0, nrows, out_dev, ld, vec_ok, rpw, row_major, wpb);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

}  // namespace

extern "C" {

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
    const int64_t ld = (n + 255) & ~(int64_t)255;
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
