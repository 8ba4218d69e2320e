// embed.hip -- kNN smoothing and the 2-D embedding loop (reference visualization.py:90-326,
// kernels taichi_core.py:227-326), device resident.
//
//  * knn sums   : sums[i,j] = sum_{a in nb[i], b in nb[j]} D[a,b] as exact integers (u16), computed as
//                 A*D*A^T in two factored steps through LDS (20+20 reads per pair instead of 400).
//  * forces     : one pass over the rows a GPU owns: q_ij, clip, cross-entropy partial (j > i),
//                 T_ij = q/(1-q)*(p-q), g_i = sum_j T_ij (y_i - y_j).  p comes either from an f32 matrix
//                 or from LUT[sums[i,j]] (the LUT holds the reference's numpy-evaluated
//                 exp(-sigmoid(s/n_nb/n_nb)/0.5) for every possible integer sum).
//       FAST mode: a wave sweeps 4 rows at a time, lanes over columns, DPP/shuffle reduction.
//       SEQ  mode: one lane per row, j ascending, IEEE f32 without FMA == the reference's arithmetic.
//  * apply      : loss -> best-list insert (bisect.insort_right) -> early-stop test -> y += -(4 g) lr
//                 -> add_jitter (as written in the reference: only points 0 and 1 are ever touched),
//                 all on device; the host only pre-draws the jitter normals from numpy's RNG stream.
//
// Everything float here is compiled with -ffp-contract=off and no fast-math: the per-pair values are
// bit-identical to numpy's f32 scalar arithmetic; FAST mode differs from the reference only in the
// order of the row sums.
#include <math.h>
#include <stdlib.h>

#include <vector>

#include <type_traits>

#include "common.h"
#include "seq_div.h"

namespace {

constexpr int BLK = 256;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// =================================================================================================
// kNN sums
// =================================================================================================
constexpr int KNN_TPB = 1024;
constexpr int KNN_CHUNK_MAX = 65536;   // u16 entries of one row staged in LDS (128 KiB of the 160 KiB)

__global__ void transpose_nb_kernel(const int32_t *__restrict__ nb, int64_t n, int n_nb, int32_t *__restrict__ nbT) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n_nb) return;
    const int64_t i = t / n_nb;
    const int a = (int)(t % n_nb);
    nbT[(int64_t)a * n + i] = nb[t];
}

// one block per output row i (grid-stride); dynamic LDS: chunk u16 entries
__global__ __launch_bounds__(KNN_TPB) void knn_sums_kernel(const uint8_t *__restrict__ D, int64_t ldd,
                                                            const int32_t *__restrict__ nb,
                                                            const int32_t *__restrict__ nbT, int64_t n, int n_nb,
                                                            int64_t row0, int64_t nrows, uint16_t *__restrict__ T,
                                                            int64_t ldt, int chunk) {
    extern __shared__ __attribute__((aligned(16))) uint16_t M[];
    const int tid = threadIdx.x;
    for (int64_t lr = blockIdx.x; lr < nrows; lr += gridDim.x) {
        const int64_t i = row0 + lr;
        uint16_t *Trow = T + lr * ldt;
        for (int64_t c0 = 0; c0 < n; c0 += chunk) {
            const int64_t cend = (c0 + chunk < n) ? c0 + chunk : n;
            // ---- step 1: M[b - c0] = sum_a D[nb[i][a], b] ----
            for (int64_t b = c0 + (int64_t)tid * 16; b < cend; b += (int64_t)KNN_TPB * 16) {
                uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};   // 16-bit fields: bytes 0,2 / 1,3 of each dword
                if (b + 16 <= n) {
                    for (int a = 0; a < n_nb; ++a) {
                        const int64_t r = nb[i * n_nb + a];   // block-uniform -> scalar load
                        const u32x4 w = *reinterpret_cast<const u32x4 *>(D + r * ldd + b);
                        lo[0] += w.x & 0x00FF00FFu; hi[0] += (w.x >> 8) & 0x00FF00FFu;
                        lo[1] += w.y & 0x00FF00FFu; hi[1] += (w.y >> 8) & 0x00FF00FFu;
                        lo[2] += w.z & 0x00FF00FFu; hi[2] += (w.z >> 8) & 0x00FF00FFu;
                        lo[3] += w.w & 0x00FF00FFu; hi[3] += (w.w >> 8) & 0x00FF00FFu;
                    }
                } else {   // ragged right edge: byte loads
                    for (int a = 0; a < n_nb; ++a) {
                        const int64_t r = nb[i * n_nb + a];
                        for (int c = 0; c < 16 && b + c < n; ++c) {
                            const uint32_t v = D[r * ldd + b + c];
                            const int d = c >> 2, f = c & 3;
                            if (f & 1) hi[d] += v << (8 * (f & 2));
                            else lo[d] += v << (8 * (f & 2));
                        }
                    }
                }
                uint32_t o[8];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    o[2 * d] = (lo[d] & 0xFFFFu) | (hi[d] << 16);
                    o[2 * d + 1] = (lo[d] >> 16) | (hi[d] & 0xFFFF0000u);
                }
                u32x4 *dst = reinterpret_cast<u32x4 *>(M + (b - c0));
                dst[0] = u32x4{o[0], o[1], o[2], o[3]};
                dst[1] = u32x4{o[4], o[5], o[6], o[7]};
            }
            __syncthreads();
            // ---- step 2: T[i,j] (+)= sum_{b in nb[j] within chunk} M[b - c0] ----
            for (int64_t j = tid; j < n; j += KNN_TPB) {
                uint32_t s = 0;
                for (int a = 0; a < n_nb; ++a) {
                    const int64_t b = nbT[(int64_t)a * n + j];
                    if (b >= c0 && b < cend) s += M[b - c0];
                }
                if (c0 > 0) s += Trow[j];
                if (cend == n && j == i) s = 0;   // diagonal forced to 0 (visualization.py:103,107)
                Trow[j] = (uint16_t)s;
            }
            __syncthreads();
        }
    }
}

// R output rows per block: the transposed neighbour table (n x n_nb indices) is the dominant traffic of the one-row kernel
// above -- it is re-read for every output row (N x 4 MB = 200 GB at N = 50 k) -- so R rows share one pass over it.  The R
// neighbour-sum rows live in LDS as uint8 when n_nb * max(D) <= 255 (k <= 12 with 20 neighbours; 50 KB per row at N = 50 k),
// else uint16, and the indices are uint16 when n <= 65536.
template <typename IT>
__global__ void transpose_nb_t_kernel(const int32_t *__restrict__ nb, int64_t n, int n_nb, IT *__restrict__ nbT) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n_nb) return;
    nbT[(t % n_nb) * n + t / n_nb] = (IT)nb[t];
}
__global__ __launch_bounds__(256) void max_u8_kernel(const uint8_t *__restrict__ D, int64_t ldd, int64_t n, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    const int64_t total16 = n * (ldd / 16);
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total16; q += (int64_t)gridDim.x * 256) {
        const int64_t r = q / (ldd / 16), c = (q % (ldd / 16)) * 16;
        if (c >= n) continue;                                      // pitch padding is not part of the matrix
        const u32x4 w = *reinterpret_cast<const u32x4 *>(D + r * ldd + c);
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                if (c + 4 * d + f < n) m = max(m, (ws[d] >> (8 * f)) & 0xFFu);
    }
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_down(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
template <int R, typename MT, typename IT>
__global__ __launch_bounds__(KNN_TPB) void knn_sums_rows_kernel(const uint8_t *__restrict__ D, int64_t ldd,
                                                                 const int32_t *__restrict__ nb, const IT *__restrict__ nbT,
                                                                 int64_t n, int n_nb, int64_t row0, int64_t nrows,
                                                                 uint16_t *__restrict__ T, int64_t ldt, int64_t mpitch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t Mraw[];
    MT *Ms = reinterpret_cast<MT *>(Mraw);
    const int tid = threadIdx.x;
    const int64_t n_groups = (nrows + R - 1) / R;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        // ---- step 1: Ms[r][b] = sum_a D[nb[i_r][a], b] ----
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t lr = grp * R + r;
            if (lr >= nrows) break;
            const int64_t i = row0 + lr;
            MT *Mr = Ms + (int64_t)r * mpitch;
            for (int64_t b = (int64_t)tid * 16; b < n; b += (int64_t)KNN_TPB * 16) {
                uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};   // 16-bit fields: bytes 0,2 / 1,3 of each dword
                if (b + 16 <= n) {
                    for (int a = 0; a < n_nb; ++a) {
                        const int64_t rr = nb[i * n_nb + a];   // block-uniform -> scalar load
                        const u32x4 w = *reinterpret_cast<const u32x4 *>(D + rr * ldd + b);
                        lo[0] += w.x & 0x00FF00FFu; hi[0] += (w.x >> 8) & 0x00FF00FFu;
                        lo[1] += w.y & 0x00FF00FFu; hi[1] += (w.y >> 8) & 0x00FF00FFu;
                        lo[2] += w.z & 0x00FF00FFu; hi[2] += (w.z >> 8) & 0x00FF00FFu;
                        lo[3] += w.w & 0x00FF00FFu; hi[3] += (w.w >> 8) & 0x00FF00FFu;
                    }
                } else {   // ragged right edge: byte loads
                    for (int a = 0; a < n_nb; ++a) {
                        const int64_t rr = nb[i * n_nb + a];
                        for (int c = 0; c < 16 && b + c < n; ++c) {
                            const uint32_t v = D[rr * ldd + b + c];
                            const int d = c >> 2, f = c & 3;
                            if (f & 1) hi[d] += v << (8 * (f & 2));
                            else lo[d] += v << (8 * (f & 2));
                        }
                    }
                }
                if constexpr (sizeof(MT) == 1) {   // sums fit a byte: bytes 0,2 from lo, bytes 1,3 from hi
                    u32x4 o;
                    o.x = (lo[0] & 0x00FF00FFu) | ((hi[0] & 0x00FF00FFu) << 8);
                    o.y = (lo[1] & 0x00FF00FFu) | ((hi[1] & 0x00FF00FFu) << 8);
                    o.z = (lo[2] & 0x00FF00FFu) | ((hi[2] & 0x00FF00FFu) << 8);
                    o.w = (lo[3] & 0x00FF00FFu) | ((hi[3] & 0x00FF00FFu) << 8);
                    *reinterpret_cast<u32x4 *>(Mr + b) = o;
                } else {
                    uint32_t o[8];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        o[2 * d] = (lo[d] & 0xFFFFu) | (hi[d] << 16);
                        o[2 * d + 1] = (lo[d] >> 16) | (hi[d] & 0xFFFF0000u);
                    }
                    u32x4 *dst = reinterpret_cast<u32x4 *>(Mr + b);
                    dst[0] = u32x4{o[0], o[1], o[2], o[3]};
                    dst[1] = u32x4{o[4], o[5], o[6], o[7]};
                }
            }
        }
        __syncthreads();
        // ---- step 2: T[i_r, j] = sum_{b in nb[j]} Ms[r][b], one pass over the neighbour table for all R rows ----
        for (int64_t j = tid; j < n; j += KNN_TPB) {
            uint32_t sacc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) sacc[r] = 0;
            for (int a = 0; a < n_nb; ++a) {
                const int64_t b = (int64_t)nbT[(int64_t)a * n + j];
#pragma unroll
                for (int r = 0; r < R; ++r) sacc[r] += Ms[(int64_t)r * mpitch + b];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t lr = grp * R + r;
                if (lr < nrows) T[lr * ldt + j] = (j == row0 + lr) ? (uint16_t)0 : (uint16_t)sacc[r];   // diagonal forced to 0
            }
        }
        __syncthreads();
    }
}
template <int R, typename MT, typename IT>
int launch_knn_rows(const uint8_t *D_dev, int64_t ldd, const int32_t *nb_dev, const IT *nbT, int64_t n, int n_nb, int64_t row0,
                    int64_t nrows, uint16_t *sums_dev, int64_t lds, int64_t mpitch, hipStream_t st) {
    const size_t bytes = (size_t)R * mpitch * sizeof(MT);
    KMAP_TRY(kmap_allow_lds((const void *)knn_sums_rows_kernel<R, MT, IT>, 150 * 1024));
    const int64_t groups = (nrows + R - 1) / R;
    const int64_t grid = groups < 2048 ? groups : 2048;
    knn_sums_rows_kernel<R, MT, IT><<<(unsigned)grid, KNN_TPB, bytes, st>>>(D_dev, ldd, nb_dev, nbT, n, n_nb, row0, nrows, sums_dev,
                                                                          lds, mpitch);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

// k-NN selection on uint8 rows: wave per row.  The threshold value t is where the cumulative count of the row's values reaches
// n_nb; the row's entries < t are taken, plus the first (n_nb - count_lt) entries == t in index order.
// Generic rows (any byte values; the fallback): pass 1 histograms the row in LDS (256 bins per wave, atomics), pass 2 walks the
// row 64 entries at a time with ballot-ordered compaction.  That was the only kernel in r01 / early r02 and took 6.4 ms at
// N = 50 000: a Hamming row holds ~9 distinct values, so all 64 lanes hit the same few LDS words, and same-address LDS atomics
// run at ~0.1 lane per clock (tools/probes/lds_atomic_rate.hip: 8-10 lanes per clock on distinct addresses).
// Fast rows (16-byte aligned pitch, every value < 32 -- Hamming distances of k < 32 always are): lane-private counters
// bins[value][lane] (plain ds_add on 64 different words, 33 x 64 counters per wave), 16 bytes per lane and load; pass 2 tests 16
// bytes per lane with SWAR compares (bytes < t, bytes == t) and only the rare steps that hold a selected entry (20 of 50 000)
// leave the wave-uniform fast path.
constexpr int SEL_WAVES = 4;
constexpr int SEL_VALS = 33;                    // values 0..31 + one bin for "32 and above" (such a row takes the generic path)
__device__ __forceinline__ void knn_select_row_generic(const uint8_t *__restrict__ row, int64_t n, int n_nb, uint32_t *h, int lane,
                                                       int32_t *__restrict__ out) {
    for (int b = lane; b < 256; b += 64) h[b] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int64_t j = lane; j < n; j += 64) atomicAdd(&h[row[j]], 1u);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    // threshold (every lane computes the same scalar walk)
    uint32_t below = 0;
    int t = 0;
    for (; t < 256; ++t) {
        const uint32_t c = h[t];
        if (below + c >= (uint32_t)n_nb) break;
        below += c;
    }
    uint32_t quota_eq = (uint32_t)n_nb - below;   // entries == t still to take (n >= n_nb guaranteed by the host)
    uint32_t written = 0;
    for (int64_t j0 = 0; j0 < n && written < (uint32_t)n_nb; j0 += 64) {
        const int64_t j = j0 + lane;
        const int v = (j < n) ? (int)row[j] : 256;
        const unsigned long long m_lt = __ballot(v < t);
        const unsigned long long m_eq = __ballot(v == t);
        const unsigned long long lanebit = 1ull << lane, lower = lanebit - 1ull;
        const uint32_t n_lt = (uint32_t)__popcll(m_lt);
        if (v < t) out[written + (uint32_t)__popcll(m_lt & lower)] = (int32_t)j;
        const uint32_t rank_eq = (uint32_t)__popcll(m_eq & lower);
        if (v == t && rank_eq < quota_eq) out[written + n_lt + rank_eq] = (int32_t)j;   // lt and eq slots interleave per chunk
        const uint32_t take_eq = (uint32_t)__popcll(m_eq) < quota_eq ? (uint32_t)__popcll(m_eq) : quota_eq;
        quota_eq -= take_eq;
        written += n_lt + take_eq;
    }
}
// bit 7 of every byte of x that is < t / == t (bytes and t below 128; T = t in every byte)
__device__ __forceinline__ uint32_t swar_lt(uint32_t x, uint32_t T) { return ~((x | 0x80808080u) - T) & 0x80808080u; }
__device__ __forceinline__ uint32_t swar_eq(uint32_t x, uint32_t T) { return ~(((x ^ T) | 0x80808080u) - 0x01010101u) & 0x80808080u; }
// bits 7, 15, 23, 31 of m -> bits 0..3
__device__ __forceinline__ uint32_t swar_pack4(uint32_t m) { return (((m >> 7) * 0x00204081u) >> 21) & 0xFu; }

__global__ __launch_bounds__(KMAP_WAVE *SEL_WAVES) void knn_select_kernel(const uint8_t *__restrict__ D, int64_t ldd,
                                                                           int64_t n, int n_nb, int64_t row0, int64_t nrows,
                                                                           int32_t *__restrict__ nb, int aligned) {
    __shared__ uint32_t bins[SEL_WAVES][SEL_VALS * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t lr = (int64_t)blockIdx.x * SEL_WAVES + wave;
    if (lr >= nrows) return;
    uint32_t *h = bins[wave];
    const uint8_t *row = D + (row0 + lr) * ldd;
    int32_t *out = nb + lr * n_nb;
    if (!aligned) {
        knn_select_row_generic(row, n, n_nb, h, lane, out);
        return;
    }
#pragma unroll
    for (int v = 0; v < SEL_VALS; ++v) h[v * 64 + lane] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint4 *row4 = reinterpret_cast<const uint4 *>(row);
    const int nsteps = (int)((n + 1023) >> 10);                             // 1024 bytes per wave and step
    const int nfull = (int)(n >> 10);
    const int lane_chunks = (int)(ldd >> 4);                                // 16-byte chunks inside the row's pitch
    for (int s = 0; s < nsteps; ++s) {
        const int chunk = s * 64 + lane;
        uint4 w = make_uint4(~0u, ~0u, ~0u, ~0u);
        if (chunk < lane_chunks) w = row4[chunk];
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        if (s < nfull) {                                                    // wave-uniform: all 1024 entries exist
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                uint32_t v = (ws[b >> 2] >> (8 * (b & 3))) & 0xFFu;
                v = v < 32u ? v : 32u;
                atomicAdd(&h[v * 64 + lane], 1u);
            }
        } else {
            const int64_t j0 = (int64_t)chunk * 16;
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                uint32_t v = (ws[b >> 2] >> (8 * (b & 3))) & 0xFFu;
                v = v < 32u ? v : 32u;
                if (j0 + b < n) atomicAdd(&h[v * 64 + lane], 1u);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    // lane v (< 33) adds the 64 private counters of value v; an inclusive scan over values 0..31 finds the threshold
    uint32_t tot = 0;
    if (lane < SEL_VALS) {
        const uint4 *p = reinterpret_cast<const uint4 *>(h + lane * 64);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint4 c = p[q];
            tot += (c.x + c.y) + (c.z + c.w);
        }
    }
    const uint32_t big = (uint32_t)__builtin_amdgcn_readlane((int)tot, 32);
    uint32_t cum = lane < 32 ? tot : 0u;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const uint32_t up = __shfl_up(cum, o);
        if (lane >= o) cum += up;
    }
    const unsigned long long reach = __ballot(lane < 32 && cum >= (uint32_t)n_nb);
    if (big != 0u || reach == 0ull) {                                       // a value >= 32 in the row: exact generic path
        __builtin_amdgcn_wave_barrier();
        knn_select_row_generic(row, n, n_nb, h, lane, out);
        return;
    }
    const int t = __builtin_ctzll(reach);
    const uint32_t below = t ? (uint32_t)__builtin_amdgcn_readlane((int)cum, t - 1) : 0u;
    uint32_t quota_eq = (uint32_t)n_nb - below;                             // entries == t still to take
    uint32_t need_lt = below;                                               // entries < t still to find
    uint32_t written = 0;
    const uint32_t T = (uint32_t)t * 0x01010101u;
    for (int s = 0; s < nsteps && written < (uint32_t)n_nb; ++s) {
        const int chunk = s * 64 + lane;
        uint4 w = make_uint4(~0u, ~0u, ~0u, ~0u);                           // 0xFF bytes: neither < t nor == t (t < 32)
        if (chunk < lane_chunks) w = row4[chunk];
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        uint32_t lt[4], eq[4], any = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            // 0xFF bytes (>= 128) would break the SWAR borrow argument only by reading as (x & 0x7F) = 0x7F >= t: still "not below"
            lt[d] = need_lt ? swar_lt(ws[d], T) : 0u;
            eq[d] = quota_eq ? swar_eq(ws[d], T) : 0u;
            any |= lt[d] | eq[d];
        }
        unsigned long long cand = __ballot(any != 0u);
        if (cand == 0ull) continue;                                         // wave-uniform: nothing selectable in these 1024 entries
        uint32_t lt16 = 0, eq16 = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            lt16 |= swar_pack4(lt[d]) << (4 * d);
            eq16 |= swar_pack4(eq[d]) << (4 * d);
        }
        const int64_t j0 = (int64_t)chunk * 16;
        if (s >= nfull) {                                                   // entries behind the row's end do not count
            const int64_t left = n - j0;
            const uint32_t ok = left >= 16 ? 0xFFFFu : left <= 0 ? 0u : (1u << (int)left) - 1u;
            lt16 &= ok;
            eq16 &= ok;
        }
        cand = __ballot((lt16 | eq16) != 0u);
        while (cand && written < (uint32_t)n_nb) {                          // scalar: lanes in order, bytes in order = index order
            const int L = __builtin_ctzll(cand);
            cand &= cand - 1;
            uint32_t ltL = (uint32_t)__builtin_amdgcn_readlane((int)lt16, L);
            uint32_t eqL = (uint32_t)__builtin_amdgcn_readlane((int)eq16, L);
            uint32_t both = ltL | eqL;
            const int32_t base = (int32_t)(((int64_t)s * 64 + L) * 16);
            while (both && written < (uint32_t)n_nb) {
                const int b = __builtin_ctz(both);
                both &= both - 1;
                const bool is_lt = (ltL >> b) & 1u;
                if (is_lt || quota_eq) {
                    if (lane == 0) out[written] = base + b;
                    ++written;
                    if (is_lt) --need_lt;
                    else --quota_eq;
                }
            }
        }
    }
}

// generic float smoothing in the reference's summation order (taichi_core.py:227-249):
// thread per (i,j), i<j: 400 gathers ii-outer/jj-inner, /n_nb twice; mirrored; diagonal 0
__global__ __launch_bounds__(BLK) void knn_smooth_f32_kernel(const float *__restrict__ D, const int32_t *__restrict__ nb,
                                                             int64_t n, int n_nb, float *__restrict__ S) {
    const int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= n * n) return;
    const int64_t i = t / n, j = t % n;
    if (i == j) { S[t] = 0.0f; return; }
    if (i > j) return;
    float s = 0.0f;
    for (int ii = 0; ii < n_nb; ++ii) {
        const int64_t r = nb[i * n_nb + ii];
        for (int jj = 0; jj < n_nb; ++jj) s += D[r * n + nb[j * n_nb + jj]];
    }
    s = s / (float)n_nb;
    s = s / (float)n_nb;
    S[i * n + j] = s;
    S[j * n + i] = s;
}

// =================================================================================================
// per-pair arithmetic shared by all force kernels (IEEE f32, numpy scalar order)
// =================================================================================================
struct Pair {
    float q, t, ce;
};
__device__ __forceinline__ float q_of(float dx, float dy) {
    const float d2 = dx * dx + dy * dy;              // (dx*dx) + (dy*dy), no FMA (taichi_core.py:254)
    float q = 1.0f / (1.0f + d2);                    // :255
    q = fminf(q, 0.999f);                            // np.minimum(prob, 1 - 1e-3)   visualization.py:254
    q = fmaxf(q, 0.001f);                            // np.maximum(prob, 1e-3)       visualization.py:255
    return q;
}
__device__ __forceinline__ float t_of(float p, float q) {
    return (q / (1.0f - q)) * (p - q);               // visualization.py:132-134
}
template <bool EXACT_LOG>
__device__ __forceinline__ float ce_of(float p, float q) {
    // taichi_core.py:279-303: eps = 1e-10 branches (q is already clipped to [1e-3, 1-1e-3]); branch-free.
    // !EXACT_LOG: v_log_f32 (log2, 1 ulp) * ln2 -- arguments lie in [1e-3, 0.999], no denormal handling needed.
    const float eps = 1e-10f;
    const float lq = EXACT_LOG ? logf(q) : __builtin_amdgcn_logf(q) * 0.69314718f;
    const float l1q = EXACT_LOG ? logf(1.0f - q) : __builtin_amdgcn_logf(1.0f - q) * 0.69314718f;
    const float full = -p * lq - (1.0f - p) * l1q;
    const float hi = (p > 1.0f - eps) ? -lq : full;
    return (p < eps) ? -l1q : hi;
}

// probability source: f32 rows, or u16 sums + LUT (LUT copy in LDS)
struct ProbSrc {
    const float *pf;        // [nrows x ld] or null
    const uint16_t *ps;     // [nrows x ld] or null
    const float *lut;       // device LUT
    int64_t ld;
    int lut_len;
};

// =================================================================================================
// FAST forces: a wave owns F_RPW rows at a time and sweeps the columns, 8 columns per lane per step.
// Not bit-pinned (row sums are reduced wavefront-parallel), so the per-pair math uses v_rcp_f32 /
// v_log_f32 and explicit FMAs: ~25 VALU + 4 transcendental issues per pair instead of ~96.
//   q   = clamp(1/(1+d2)),  t = q/(1-q)*(p-q),  g += t*(y_i-y_j)
//   ce  = -(p*ln q + (1-p)*ln(1-q)) = -ln2 * (log2(1-q) + p*(log2 q - log2(1-q)))   (eps branches of the
//         reference change ce by < 1e-9 relative and are dropped here; SEQ mode keeps them)
// The diagonal needs no predicate for the gradient (dx = dy = 0 -> t*0 = 0); the loss takes j > i only.
// =================================================================================================
constexpr int F_RPW = 2;          // rows per wave
constexpr int F_WAVES = 8;        // waves per block
constexpr int F_CPL = 8;          // columns per lane per step (16 B of u16 sums / 32 B of f32)
constexpr int F_LUT_LDS = 12416;  // floats of LUT cached in LDS (n_nb^2*k+1 <= 400*31+1 = 12401)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// FAST per-pair core.  q = clip(1/(1+d2), 1e-3, 1-1e-3) is obtained by clamping d2 to [1/999, 999] (the same interval),
// which turns q, 1-q = d2/(1+d2) and q/(1-q) = 1/d2 into products of ONE reciprocal: r = 1/(d2 (1+d2)), q = r d2,
// q/(1-q) = r (1+d2).  The cross-entropy term -(p ln q + (1-p) ln(1-q)) = -ln2 (log2(1-q) - p log2 d2) needs one log per
// pair plus one log of the product of the (1-q) of a lane's 8 columns (each in [1e-3, 0.999]: the product stays normal in
// f32, and for far pairs -- q = 1e-3, the bulk of the sum -- the product form has a smaller systematic error than 8 logs).
constexpr float FAST_D2_MIN = 1.0f / 999.0f, FAST_D2_MAX = 999.0f;
__device__ __forceinline__ void fast_core(float dx, float dy, float p, float &t, float &omq, float &d2c) {
    d2c = __builtin_amdgcn_fmed3f(__builtin_fmaf(dx, dx, dy * dy), FAST_D2_MIN, FAST_D2_MAX);
    const float s1 = 1.0f + d2c;
    const float r = __builtin_amdgcn_rcpf(d2c * s1);
    const float q = r * d2c;
    omq = 1.0f - q;
    t = (r * s1) * (p - q);
}
enum { PL_NONE = 0, PL_ALL = 1, PL_MASK = 2 };   // loss terms: none of the 8 pairs / all of them / only j > i (and j < n)
template <int PL, bool GUARD>
__device__ __forceinline__ void fast_pairs(const float (&p)[F_CPL], const float (&xj)[F_CPL], const float (&yj)[F_CPL],
                                           float xi, float yi, int64_t gi, int64_t j0, int64_t n, float &gx, float &gy,
                                           float &ce2) {
    float esum = 0.0f, prod = 1.0f;
#pragma unroll
    for (int c = 0; c < F_CPL; ++c) {
        const float dx = xi - xj[c], dy = yi - yj[c];
        float t, omq, d2c;
        fast_core(dx, dy, p[c], t, omq, d2c);
        const int64_t j = j0 + c;
        if (GUARD) t = (j < n) ? t : 0.0f;
        gx = __builtin_fmaf(t, dx, gx);
        gy = __builtin_fmaf(t, dy, gy);
        if (PL == PL_ALL) {
            esum = __builtin_fmaf(p[c], __builtin_amdgcn_logf(d2c), esum);
            prod *= omq;
        } else if (PL == PL_MASK) {
            const bool live = (j > gi) && (j < n);
            esum += live ? p[c] * __builtin_amdgcn_logf(d2c) : 0.0f;
            prod *= live ? omq : 1.0f;
        }
    }
    if (PL != PL_NONE) ce2 += __builtin_amdgcn_logf(prod) - esum;
}

template <bool LUTSRC>
__global__ __launch_bounds__(KMAP_WAVE *F_WAVES) void forces_fast_kernel(ProbSrc src, const float *__restrict__ Y,
                                                                          int64_t n, int64_t row0, int64_t nrows,
                                                                          float *__restrict__ G,
                                                                          double *__restrict__ loss_part) {
    extern __shared__ __attribute__((aligned(16))) float lut_s[];   // lut_len floats (dynamic: sized by the launch)
    __shared__ double wloss[F_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (LUTSRC) {
        for (int t = threadIdx.x; t < src.lut_len && t < F_LUT_LDS; t += blockDim.x) lut_s[t] = src.lut[t];
        __syncthreads();
    }
    const float *X = Y, *Yy = Y + n;
    double wave_loss = 0.0;
    const int64_t rbase = ((int64_t)blockIdx.x * F_WAVES + wave) * F_RPW;
    if (rbase < nrows) {
        float xi[F_RPW], yi[F_RPW], gx[F_RPW], gy[F_RPW];
        int64_t gi[F_RPW], lrow[F_RPW];
#pragma unroll
        for (int r = 0; r < F_RPW; ++r) {
            lrow[r] = (rbase + r < nrows) ? rbase + r : nrows - 1;   // clamped duplicate rows are discarded below
            gi[r] = row0 + lrow[r];
            xi[r] = X[gi[r]];
            yi[r] = Yy[gi[r]];
            gx[r] = gy[r] = 0.0f;
        }
        const bool vec_ok = (src.ld % 8 == 0) && (n % 4 == 0 || true);
        float ce2 = 0.0f;   // log2 units, f32 partial flushed into f64 every 16 steps
        int step = 0;
        for (int64_t j0 = (int64_t)lane * F_CPL; j0 < n; j0 += (int64_t)KMAP_WAVE * F_CPL, ++step) {
            const bool full = (j0 + F_CPL <= n);
            float xj[F_CPL], yj[F_CPL];
            if (full && (n % 4 == 0)) {   // 16-byte aligned coordinate rows
                const f32x4 a0 = *reinterpret_cast<const f32x4 *>(X + j0), a1 = *reinterpret_cast<const f32x4 *>(X + j0 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(Yy + j0), b1 = *reinterpret_cast<const f32x4 *>(Yy + j0 + 4);
                xj[0] = a0.x; xj[1] = a0.y; xj[2] = a0.z; xj[3] = a0.w; xj[4] = a1.x; xj[5] = a1.y; xj[6] = a1.z; xj[7] = a1.w;
                yj[0] = b0.x; yj[1] = b0.y; yj[2] = b0.z; yj[3] = b0.w; yj[4] = b1.x; yj[5] = b1.y; yj[6] = b1.z; yj[7] = b1.w;
            } else {
#pragma unroll
                for (int c = 0; c < F_CPL; ++c) {
                    const int64_t j = (j0 + c < n) ? j0 + c : n - 1;
                    xj[c] = X[j];
                    yj[c] = Yy[j];
                }
            }
#pragma unroll
            for (int r = 0; r < F_RPW; ++r) {
                float p[F_CPL];
                if (LUTSRC) {
                    const uint16_t *row = src.ps + lrow[r] * src.ld + j0;
                    if (full && vec_ok) {
                        const u32x4 w = *reinterpret_cast<const u32x4 *>(row);
                        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                        for (int c = 0; c < F_CPL; ++c) p[c] = lut_s[(ws[c >> 1] >> (16 * (c & 1))) & 0xFFFFu];
                    } else {
#pragma unroll
                        for (int c = 0; c < F_CPL; ++c) p[c] = (j0 + c < n) ? lut_s[row[c]] : 0.0f;
                    }
                } else {
                    const float *row = src.pf + lrow[r] * src.ld + j0;
#pragma unroll
                    for (int c = 0; c < F_CPL; ++c) p[c] = (j0 + c < n) ? row[c] : 0.0f;
                }
                float e = 0.0f;
                // wave-uniform choice: the step's 512 columns lie right of the diagonal (all loss terms), left of it (none:
                // each unordered pair is charged once, to its j > i side) or straddle it / the end of the row (masked)
                const int64_t sj0 = (int64_t)step * (KMAP_WAVE * F_CPL);
                const bool wfull = sj0 + KMAP_WAVE * F_CPL <= n;
                if (wfull && sj0 > gi[r]) fast_pairs<PL_ALL, false>(p, xj, yj, xi[r], yi[r], gi[r], j0, n, gx[r], gy[r], e);
                else if (wfull && sj0 + KMAP_WAVE * F_CPL - 1 <= gi[r]) fast_pairs<PL_NONE, false>(p, xj, yj, xi[r], yi[r], gi[r], j0, n, gx[r], gy[r], e);
                else fast_pairs<PL_MASK, true>(p, xj, yj, xi[r], yi[r], gi[r], j0, n, gx[r], gy[r], e);
                ce2 += (rbase + r < nrows) ? e : 0.0f;
            }
            if ((step & 15) == 15) {
                wave_loss += (double)ce2;
                ce2 = 0.0f;
            }
        }
        wave_loss += (double)ce2;
        wave_loss *= -0.6931471805599453;   // log2 -> -ln
#pragma unroll
        for (int r = 0; r < F_RPW; ++r) {
            for (int o = 32; o > 0; o >>= 1) {
                gx[r] += __shfl_down(gx[r], o);
                gy[r] += __shfl_down(gy[r], o);
            }
            if (lane == 0 && rbase + r < nrows) {
                G[gi[r]] = gx[r];
                G[n + gi[r]] = gy[r];
            }
        }
        for (int o = 32; o > 0; o >>= 1) wave_loss += __shfl_down(wave_loss, o);
    }
    if (lane == 0) wloss[wave] = wave_loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < F_WAVES; ++w) s += wloss[w];
        loss_part[blockIdx.x] = s;
    }
}

// =================================================================================================
// FAST forces, symmetric form (single-GPU, all rows local): every unordered pair {i<j} is evaluated once.
// A wave owns a tile of SY_R rows x 512 columns (lane = 8 consecutive columns); t*(y_i-y_j) goes to the row's sum
// (wave reduction per row) and, negated, to the lane's column accumulators (registers, over the tile's rows).
// Tiles write disjoint slices of two partial buffers -- rowpart[J][2][N] and colpart[I][2][N] -- and a second kernel
// adds the partials in a fixed order: deterministic, no atomics.  Tiles entirely below the diagonal are skipped.
// =================================================================================================
constexpr int SY_RB = 64;                // rows per row block (one lane of the wave holds one row's coordinates)
constexpr int SY_NRB = 4;                // row blocks a wave walks through with its column accumulators live
constexpr int SY_R = SY_RB * SY_NRB;     // rows per tile
constexpr int SY_C = KMAP_WAVE * F_CPL;  // 512 columns per tile
constexpr int SY_WAVES = 4;              // tiles (consecutive column chunks) per block

__device__ __forceinline__ bool sy_tile_live(int64_t I, int64_t J) {   // some pair of the tile has j > i
    return (J + 1) * SY_C - 1 > I * SY_R;
}
// wave-wide sum by DPP (no LDS): inclusive row scan (row_shr 1,2,4,8), then row_bcast15 / row_bcast31; lane 63 = total
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0x111, 0xF>(v);   // row_shr:1
    v = dpp_add<0x112, 0xF>(v);   // row_shr:2
    v = dpp_add<0x114, 0xF>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF>(v);   // row_shr:8   -> lane 15 of every row holds the row's sum
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
    return v;
}

template <bool LUTSRC>
__global__ __launch_bounds__(KMAP_WAVE *SY_WAVES) void forces_sym_kernel(ProbSrc src, const float *__restrict__ Y, int64_t n,
                                                                         float *__restrict__ rowpart,
                                                                         float *__restrict__ colpart,
                                                                         double *__restrict__ loss_part, int64_t nJ, int world,
                                                                         int rank) {
    extern __shared__ __attribute__((aligned(16))) float lut_s[];
    if (LUTSRC) {
        for (int t = threadIdx.x; t < src.lut_len && t < F_LUT_LDS; t += blockDim.x) lut_s[t] = src.lut[t];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blockIdx.y = local row block; global row block I (cyclic over the ranks).  Probability rows and column partials are
    // indexed by the local block, row partials by the global row.
    const int64_t Il = blockIdx.y, I = (int64_t)rank + (int64_t)world * Il, J = (int64_t)blockIdx.x * SY_WAVES + wave;
    const int64_t part_idx = Il * (gridDim.x * SY_WAVES) + J;
    double wave_loss = 0.0;
    if (J < nJ && sy_tile_live(I, J)) {
        const float *X = Y, *Yy = Y + n;
        const int64_t j0 = J * SY_C + (int64_t)lane * F_CPL;
        float xj[F_CPL], yj[F_CPL], cgx[F_CPL], cgy[F_CPL];
        const bool full = (j0 + F_CPL <= n);
#pragma unroll
        for (int c = 0; c < F_CPL; ++c) {
            const int64_t j = (j0 + c < n) ? j0 + c : n - 1;
            xj[c] = X[j];
            yj[c] = Yy[j];
            cgx[c] = cgy[c] = 0.0f;
        }
        const bool vec_ok = LUTSRC && (src.ld % 8 == 0) && full;
        const bool full_tile = (J + 1) * SY_C <= n;   // wave-uniform: every lane's 8 columns exist
        float ce2 = 0.0f;
        for (int rb = 0; rb < SY_NRB; ++rb) {
            const int64_t r0 = I * SY_R + (int64_t)rb * SY_RB;
            if (r0 >= n || !((J + 1) * SY_C - 1 > r0)) break;   // row blocks further down lie entirely below the diagonal
            const int nr = (int)((n - r0 < SY_RB) ? n - r0 : SY_RB);
            const int64_t myrow = (r0 + lane < n) ? r0 + lane : n - 1;
            const float xr = X[myrow], yr = Yy[myrow];
            // row blocks whose rows all lie left of the tile's first column (and full-width tiles) need no j > i / j < n tests
            const bool interior = full_tile && (J * SY_C > r0 + nr - 1);
            auto run_rows = [&](auto check_tag) {
                constexpr bool CHECK = decltype(check_tag)::value;
                for (int r = 0; r < nr; ++r) {
                    const int64_t gi = r0 + r;
                    const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xr), r));
                    const float yi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yr), r));
                    float gx = 0.0f, gy = 0.0f;
                    if (!CHECK || (j0 + F_CPL - 1 > gi && j0 < n)) {   // this lane has at least one column right of the diagonal
                        float p[F_CPL];
                        if (LUTSRC) {
                            const uint16_t *row = src.ps + (Il * SY_R + (gi - I * SY_R)) * src.ld + j0;
                            if (vec_ok) {
                                const u32x4 w = *reinterpret_cast<const u32x4 *>(row);
                                const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                                for (int c = 0; c < F_CPL; ++c) p[c] = lut_s[(ws[c >> 1] >> (16 * (c & 1))) & 0xFFFFu];
                            } else {
#pragma unroll
                                for (int c = 0; c < F_CPL; ++c) p[c] = (j0 + c < n) ? lut_s[row[c]] : 0.0f;
                            }
                        } else {
                            const float *row = src.pf + (Il * SY_R + (gi - I * SY_R)) * src.ld + j0;
#pragma unroll
                            for (int c = 0; c < F_CPL; ++c) p[c] = (j0 + c < n) ? row[c] : 0.0f;
                        }
                        float esum = 0.0f, prod = 1.0f;
#pragma unroll
                        for (int c = 0; c < F_CPL; ++c) {
                            const int64_t j = j0 + c;
                            const float dx = xi - xj[c], dy = yi - yj[c];
                            float t, omq, d2c;
                            fast_core(dx, dy, p[c], t, omq, d2c);
                            const float lterm = p[c] * __builtin_amdgcn_logf(d2c);
                            if (CHECK) {
                                const bool live = (j > gi) && (j < n);
                                t = live ? t : 0.0f;
                                esum += live ? lterm : 0.0f;
                                prod *= live ? omq : 1.0f;
                            } else {
                                esum += lterm;
                                prod *= omq;
                            }
                            gx = __builtin_fmaf(t, dx, gx);            // row side: + t (y_i - y_j)
                            gy = __builtin_fmaf(t, dy, gy);
                            cgx[c] = __builtin_fmaf(-t, dx, cgx[c]);   // column side: - t (y_i - y_j)  (negation is an operand modifier)
                            cgy[c] = __builtin_fmaf(-t, dy, cgy[c]);
                        }
                        ce2 += __builtin_amdgcn_logf(prod) - esum;
                    }
                    gx = wave_sum_to_lane63(gx);
                    gy = wave_sum_to_lane63(gy);
                    if (lane == 63) {
                        rowpart[(J * 2 + 0) * n + gi] = gx;
                        rowpart[(J * 2 + 1) * n + gi] = gy;
                    }
                    if ((r & 15) == 15) {
                        wave_loss += (double)ce2;
                        ce2 = 0.0f;
                    }
                }
            };
            if (interior) run_rows(std::false_type{});
            else run_rows(std::true_type{});
        }
        wave_loss += (double)ce2;
        wave_loss *= -0.6931471805599453;
#pragma unroll
        for (int c = 0; c < F_CPL; ++c) {
            if (j0 + c < n) {
                colpart[(Il * 2 + 0) * n + j0 + c] = cgx[c];
                colpart[(Il * 2 + 1) * n + j0 + c] = cgy[c];
            }
        }
        for (int o = 32; o > 0; o >>= 1) wave_loss += __shfl_down(wave_loss, o);
    }
    if (lane == 0) loss_part[part_idx] = wave_loss;
}

// =================================================================================================
// FAST forces, symmetric form, second generation (u16 sums + LUT source).  Same partial buffers and loss layout as
// forces_sym_kernel, other work split and instruction stream.  PMC passes of the first kernel (profiles/r02_pmc*.json) showed
// 75 % VALU issue utilisation at ~23 VALU instructions per pair, 55 % of the wave-cycles parked on loads, and only ~2.4 waves
// per wave slot over the whole launch (9.8 k waves of 256 rows x 512 columns on 4096 slots: a long tail).  Here:
//   * a block = one 256-row x 512-column tile, its four waves take one 64-row block each (4x finer work units, 39 k waves);
//     the waves' column-side sums meet in LDS, so the column partials stay one slice per 256-row block;
//   * rows go in groups of S2_G; the 16-byte sums loads of group g+1 are in flight while group g is evaluated (the old kernel
//     loaded, waited, gathered, waited, computed -- per row);
//   * all per-pair arithmetic on column PAIRS as v_pk_{add,mul,fma}_f32 (two pairs per issue slot): with m = d2c (1 + d2c),
//     r = 1/m:  q = r d2c,  q/(1-q) = r + q  (~16 issue slots per pair incl. the two 8-cycle transcendentals, down from ~25);
//   * LUT byte offsets by one SDWA shift per pair (v_lshlrev_b32_sdwa picks the 16-bit half and scales it by 4), the LUT at LDS
//     address 0 so that the shift result is the ds_read address;
//   * the row-side partial sums of a group (gx, gy of S2_G rows) are reduced together: transposed through a per-wave LDS
//     scratch (ds_write per value, one ds_read_b128 per lane, a few adds, DPP row shifts) instead of 2 x 6 DPP steps per row.
// Row blocks that touch the diagonal or the right / bottom edge take the masked generic path (the first kernel's arithmetic).
// =================================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int S2_G = 2;                        // rows per group
constexpr int S2_TSTRIDE = 68;                 // dwords between value types in the transpose scratch (64 lanes + 4: conflict-free)
constexpr int S2_SCRATCH = 2 * S2_G * S2_TSTRIDE;   // dwords per wave (2 S2_G value types)
constexpr int S2_CS = 16 * 64;                 // dwords per wave of the column-sum exchange: 16 components x 64 lanes

__device__ __forceinline__ uint32_t lut_off_lo(uint32_t w, uint32_t two) {   // (w & 0xFFFF) << 2
    uint32_t a;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(a) : "s"(two), "v"(w));
    return a;
}
__device__ __forceinline__ uint32_t lut_off_hi(uint32_t w, uint32_t two) {   // (w >> 16) << 2
    uint32_t a;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(a) : "s"(two), "v"(w));
    return a;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// float at an absolute LDS byte address.  The LUT is the kernel's first LDS object and starts at address 0 (checked on entry), so
// the SDWA result IS the ds_read address: no per-pair add of the (relocated, zero) base of the extern __shared__ symbol.
__device__ __forceinline__ float lds_f32_at(uint32_t byte_addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) float *>(byte_addr);
}

__global__ __launch_bounds__(KMAP_WAVE *SY_WAVES, 4) void forces_sym2_kernel(ProbSrc src, const float *__restrict__ Y, int64_t n,
                                                                             float *__restrict__ rowpart, float *__restrict__ colpart,
                                                                             double *__restrict__ loss_part, int64_t nJ, int64_t part_ld,
                                                                             int world, int rank, int lut_pad) {
    static_assert(SY_WAVES == SY_NRB, "one wave per 64-row block of the tile");
    extern __shared__ __attribute__((aligned(16))) float lut_s[];   // [lut_pad] LUT | SY_WAVES transpose scratches | column-sum exchange
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blockIdx.x = column tile J, blockIdx.y = local 256-row block; global row block I (cyclic over the ranks)
    const int64_t Il = blockIdx.y, I = (int64_t)rank + (int64_t)world * Il, J = blockIdx.x;
    if (!sy_tile_live(I, J)) {                                      // tile entirely below the diagonal (block-uniform)
        if (threadIdx.x == 0) loss_part[Il * part_ld + J] = 0.0;
        return;
    }
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)lut_s != 0u) __builtin_trap();   // see lds_f32_at
    for (int t = threadIdx.x; t < src.lut_len && t < F_LUT_LDS; t += blockDim.x) lut_s[t] = src.lut[t];
    __syncthreads();
    float *scratch = lut_s + lut_pad + wave * S2_SCRATCH;
    float *colx = lut_s + lut_pad + SY_WAVES * S2_SCRATCH;          // [wave][component 0..15][lane]
    const float *X = Y, *Yy = Y + n;
    const int64_t j0 = J * SY_C + (int64_t)lane * F_CPL;
    f32x2 cgx[4], cgy[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) cgx[d] = cgy[d] = f32x2{0.0f, 0.0f};
    double wave_loss = 0.0;
    const int64_t r0 = I * SY_R + (int64_t)wave * SY_RB;            // this wave's 64 rows
    if (r0 < n && (J + 1) * SY_C - 1 > r0) {
        f32x2 xj[4], yj[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int64_t ja = (j0 + 2 * d < n) ? j0 + 2 * d : n - 1, jb = (j0 + 2 * d + 1 < n) ? j0 + 2 * d + 1 : n - 1;
            xj[d] = f32x2{X[ja], X[jb]};
            yj[d] = f32x2{Yy[ja], Yy[jb]};
        }
        const int nr = (int)((n - r0 < SY_RB) ? n - r0 : SY_RB);
        const int64_t myrow = (r0 + lane < n) ? r0 + lane : n - 1;
        float xrv = X[myrow], yrv = Yy[myrow];                      // lane r holds row r's coordinates
        const uint16_t *rows = src.ps + (Il * SY_R + (int64_t)wave * SY_RB) * src.ld + j0;   // row r of the block: rows + r * ld
        float ce2 = 0.0f;
        // interior: 64 rows, all left of the tile's first column, all 512 columns exist -> no j > i / j < n tests
        const bool interior = ((J + 1) * SY_C <= n) && (src.ld % 8 == 0) && nr == SY_RB && (J * SY_C > r0 + nr - 1);
        if (interior) {
            const uint32_t two = 2;
            u32x4 wn[S2_G];
#pragma unroll
            for (int a = 0; a < S2_G; ++a) wn[a] = *reinterpret_cast<const u32x4 *>(rows + (int64_t)a * src.ld);
#pragma unroll 1
            for (int g = 0; g < SY_RB / S2_G; ++g) {
                u32x4 wc[S2_G];
#pragma unroll
                for (int a = 0; a < S2_G; ++a) wc[a] = wn[a];
                if (g + 1 < SY_RB / S2_G) {
#pragma unroll
                    for (int a = 0; a < S2_G; ++a)
                        wn[a] = *reinterpret_cast<const u32x4 *>(rows + (int64_t)((g + 1) * S2_G + a) * src.ld);
                }
                float part[2 * S2_G];
                f32x2 es2 = f32x2{0.0f, 0.0f};
#pragma unroll
                for (int a = 0; a < S2_G; ++a) {
                    const int r = g * S2_G + a;
                    float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xrv), r));
                    float yi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yrv), r));
                    const uint32_t ws[4] = {wc[a].x, wc[a].y, wc[a].z, wc[a].w};
                    f32x2 p2[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d)   // LUT gather: one SDWA shift (16-bit half -> byte offset) + one ds_read_b32 per pair
                        p2[d] = f32x2{lds_f32_at(lut_off_lo(ws[d], two)), lds_f32_at(lut_off_hi(ws[d], two))};
                    f32x2 gx2 = f32x2{0.0f, 0.0f}, gy2 = f32x2{0.0f, 0.0f}, pr2 = f32x2{1.0f, 1.0f};
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const f32x2 xi2 = f32x2{xi, xi}, yi2 = f32x2{yi, yi};
                        const f32x2 dx = xi2 - xj[d], dy = yi2 - yj[d];
                        const f32x2 d2 = pk_fma(dx, dx, dy * dy);
                        const f32x2 d2c = f32x2{__builtin_amdgcn_fmed3f(d2.x, FAST_D2_MIN, FAST_D2_MAX),
                                                __builtin_amdgcn_fmed3f(d2.y, FAST_D2_MIN, FAST_D2_MAX)};
                        const f32x2 m = pk_fma(d2c, d2c, d2c);                       // d2c (1 + d2c)
                        const f32x2 rr = f32x2{__builtin_amdgcn_rcpf(m.x), __builtin_amdgcn_rcpf(m.y)};
                        const f32x2 q = rr * d2c;                                    // 1 / (1 + d2c), clipped through d2c
                        const f32x2 u = rr + q;                                      // q / (1 - q) = 1 / d2c
                        const f32x2 t = u * (p2[d] - q);
                        const f32x2 omq = f32x2{1.0f, 1.0f} - q;
                        gx2 = pk_fma(t, dx, gx2);                                    // row side: + t (y_i - y_j)
                        gy2 = pk_fma(t, dy, gy2);
                        cgx[d] = pk_fma(-t, dx, cgx[d]);                             // column side (negation = operand modifier)
                        cgy[d] = pk_fma(-t, dy, cgy[d]);
                        const f32x2 lg = f32x2{__builtin_amdgcn_logf(d2c.x), __builtin_amdgcn_logf(d2c.y)};
                        es2 = pk_fma(p2[d], lg, es2);
                        pr2 = pr2 * omq;
                        // two column pairs at a time: the empty asm ties their results to the inputs of the next two (row
                        // coordinates), so only two of the group's independent chains are interleaved -- enough to fill most
                        // trans / packed-op wait states; left alone the scheduler overlaps all of them and needs > 200 VGPRs.
                        // The LUT gathers above stay free to issue early.
                        if (d & 1)
                            asm volatile("" : "+s"(xi), "+s"(yi), "+v"(xrv), "+v"(yrv), "+v"(gx2), "+v"(gy2), "+v"(es2), "+v"(pr2),
                                         "+v"(cgx[d]), "+v"(cgy[d]), "+v"(cgx[d - 1]), "+v"(cgy[d - 1]));
                    }
                    part[2 * a] = gx2.x + gx2.y;
                    part[2 * a + 1] = gy2.x + gy2.y;
                    ce2 += __builtin_amdgcn_logf(pr2.x * pr2.y);
                }
                ce2 -= es2.x + es2.y;
                // the 2 S2_G partial sums of the group, summed over the wave's 64 lanes: value type t goes to scratch[t][lane],
                // lane L then adds the S2_E entries [L / S2_L][S2_E (L % S2_L) ..] and DPP row shifts finish the groups of S2_L lanes
                constexpr int S2_T = 2 * S2_G, S2_L = 64 / S2_T, S2_E = 64 / S2_L;   // value types; lanes per type; entries per lane
#pragma unroll
                for (int t = 0; t < S2_T; ++t) scratch[t * S2_TSTRIDE + lane] = part[t];
                __builtin_amdgcn_wave_barrier();
                const float *mine = scratch + (lane / S2_L) * S2_TSTRIDE + (lane % S2_L) * S2_E;
                float sred;
                if constexpr (S2_E == 8) {
                    const f32x4 va = *reinterpret_cast<const f32x4 *>(mine), vb = *reinterpret_cast<const f32x4 *>(mine + 4);
                    sred = ((va.x + va.y) + (va.z + va.w)) + ((vb.x + vb.y) + (vb.z + vb.w));
                } else {
                    static_assert(S2_E == 4 || S2_E == 8, "rows per group: 2 or 4");
                    const f32x4 va = *reinterpret_cast<const f32x4 *>(mine);
                    sred = (va.x + va.y) + (va.z + va.w);
                }
                __builtin_amdgcn_wave_barrier();
                sred = dpp_add<0x111, 0xF>(sred);   // row_shr:1
                sred = dpp_add<0x112, 0xF>(sred);   // row_shr:2
                sred = dpp_add<0x114, 0xF>(sred);   // row_shr:4
                if constexpr (S2_L == 16) sred = dpp_add<0x118, 0xF>(sred);   // row_shr:8
                if ((lane % S2_L) == S2_L - 1) {   // the last lane of each group holds the total of its value type
                    const int t = lane / S2_L;                               // row t >> 1 of the group, x / y
                    rowpart[(J * 2 + (t & 1)) * n + r0 + g * S2_G + (t >> 1)] = sred;
                }
                if (((g + 1) * S2_G) % 16 == 0) {
                    wave_loss += (double)ce2;
                    ce2 = 0.0f;
                }
            }
        } else {
            for (int r = 0; r < nr; ++r) {
                const int64_t gi = r0 + r;
                const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xrv), r));
                const float yi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yrv), r));
                float gx = 0.0f, gy = 0.0f;
                if (j0 + F_CPL - 1 > gi && j0 < n) {   // this lane has at least one column right of the diagonal
                    float esum = 0.0f, prod = 1.0f;
#pragma unroll
                    for (int c = 0; c < F_CPL; ++c) {
                        const int64_t j = j0 + c;
                        const float p = (j < n) ? lut_s[rows[(int64_t)r * src.ld + c]] : 0.0f;
                        const float xc = (c & 1) ? xj[c >> 1].y : xj[c >> 1].x, yc = (c & 1) ? yj[c >> 1].y : yj[c >> 1].x;
                        const float dx = xi - xc, dy = yi - yc;
                        float t, omq, d2c;
                        fast_core(dx, dy, p, t, omq, d2c);
                        const float lterm = p * __builtin_amdgcn_logf(d2c);
                        const bool live = (j > gi) && (j < n);
                        t = live ? t : 0.0f;
                        esum += live ? lterm : 0.0f;
                        prod *= live ? omq : 1.0f;
                        gx = __builtin_fmaf(t, dx, gx);
                        gy = __builtin_fmaf(t, dy, gy);
                        if (c & 1) {
                            cgx[c >> 1].y = __builtin_fmaf(-t, dx, cgx[c >> 1].y);
                            cgy[c >> 1].y = __builtin_fmaf(-t, dy, cgy[c >> 1].y);
                        } else {
                            cgx[c >> 1].x = __builtin_fmaf(-t, dx, cgx[c >> 1].x);
                            cgy[c >> 1].x = __builtin_fmaf(-t, dy, cgy[c >> 1].x);
                        }
                    }
                    ce2 += __builtin_amdgcn_logf(prod) - esum;
                }
                gx = wave_sum_to_lane63(gx);
                gy = wave_sum_to_lane63(gy);
                if (lane == 63) {
                    rowpart[(J * 2 + 0) * n + gi] = gx;
                    rowpart[(J * 2 + 1) * n + gi] = gy;
                }
                if ((r & 15) == 15) {
                    wave_loss += (double)ce2;
                    ce2 = 0.0f;
                }
            }
        }
        wave_loss += (double)ce2;
        wave_loss *= -0.6931471805599453;
        for (int o = 32; o > 0; o >>= 1) wave_loss += __shfl_down(wave_loss, o);
    }
    // column side: the four waves' sums over their 64 rows meet in LDS; one slice per 256-row block leaves the CU
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        colx[(wave * 16 + 2 * d) * 64 + lane] = cgx[d].x;
        colx[(wave * 16 + 2 * d + 1) * 64 + lane] = cgx[d].y;
        colx[(wave * 16 + 8 + 2 * d) * 64 + lane] = cgy[d].x;
        colx[(wave * 16 + 8 + 2 * d + 1) * 64 + lane] = cgy[d].y;
    }
    double *wl = reinterpret_cast<double *>(colx + SY_WAVES * S2_CS);
    if (lane == 0) wl[wave] = wave_loss;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = q * 256 + threadIdx.x;              // output element: x / y component, column of the tile
        const int cxy = e >> 9, col = e & 511, comp = cxy * 8 + (col & 7), l = col >> 3;
        float s = colx[(0 * 16 + comp) * 64 + l];
        s += colx[(1 * 16 + comp) * 64 + l];
        s += colx[(2 * 16 + comp) * 64 + l];
        s += colx[(3 * 16 + comp) * 64 + l];
        const int64_t j = J * SY_C + col;
        if (j < n) colpart[(Il * 2 + cxy) * n + j] = s;
    }
    if (threadIdx.x == 0) loss_part[Il * part_ld + J] = ((wl[0] + wl[1]) + (wl[2] + wl[3]));
}

// G[c][i] = sum_J rowpart[J][c][i] (tiles right of i's row block) + sum_I colpart[I][c][i] (row blocks above / at i)
__global__ __launch_bounds__(BLK) void sym_reduce_kernel(const float *__restrict__ rowpart, const float *__restrict__ colpart,
                                                         int64_t n, int64_t n_lblocks, int64_t nJ, int world, int rank,
                                                         float *__restrict__ G) {
    // 8 lanes per output element (a one-lane walk over ~300 partials is latency-bound: 113 us at N = 50 k): lane l adds the
    // partials l, l+8, ... in order, then the 8 lane sums are combined in a fixed butterfly -- deterministic, no atomics.
    // Sharded: row partials exist only for the rows of this rank's blocks, column partials for its local blocks; the ranks'
    // G buffers are then summed by the all-reduce.
    constexpr int SPLIT = 8;
    const int64_t t = ((int64_t)blockIdx.x * BLK + threadIdx.x) / SPLIT;
    const int l = threadIdx.x & (SPLIT - 1);
    float g = 0.0f;
    if (t < 2 * n) {
        const int c = (int)(t / n);
        const int64_t i = t % n;
        const int64_t Ii = i / SY_R, Ji = i / SY_C;
        if (Ii % world == rank)
            for (int64_t J = l; J < nJ; J += SPLIT)
                if (sy_tile_live(Ii, J)) g += rowpart[(J * 2 + c) * n + i];
        for (int64_t b = l; b < n_lblocks; b += SPLIT)
            if (sy_tile_live((int64_t)rank + (int64_t)world * b, Ji)) g += colpart[(b * 2 + c) * n + i];
    }
    g += __shfl_xor(g, 1);
    g += __shfl_xor(g, 2);
    g += __shfl_xor(g, 4);
    if (t < 2 * n && l == 0) G[t] = g;
}

// =================================================================================================
// SEQ forces: the reference's summation order (taichi_core.py:305-326: ret_val += diff[i,j] * (y[k,i] - y[k,j]),
// j ascending, j != i), IEEE f32, no FMA.  A row is owned by the 4 lanes of a quad: for a group of 4 columns
// each sub-lane evaluates one term (q, t, t*dx, t*dy -- independent work), then ALL lanes of the row add the
// 4 terms in column order (DPP quad broadcasts), so they carry identical accumulators and the sum order is exactly
// j = 0, 1, 2, ...  The loss needs no order (f64 accumulation of f32 terms), each sub-lane keeps its own.
// =================================================================================================
constexpr int SQ_SUB = 4;                       // sub-lanes per row = one DPP quad
constexpr int SQ_ROWS = KMAP_WAVE / SQ_SUB;     // rows per wave
// acc + (value of `v` in lane S of the caller's quad) as ONE v_add_f32 with a DPP quad_perm source.
// hipcc does not fold __builtin_amdgcn_update_dpp into the add, so the instruction is written out; the DPP operand
// `v` is always produced more than 2 VALU instructions earlier (the DPP read-after-VALU-write hazard, cdna_hip 5.7).
template <int S>
__device__ __forceinline__ float add_quad_bcast(float acc, float v) {
    float r;
    if constexpr (S == 0) asm("v_add_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(acc));
    if constexpr (S == 1) asm("v_add_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(acc));
    if constexpr (S == 2) asm("v_add_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(acc));
    if constexpr (S == 3) asm("v_add_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(acc));
    return r;
}
// the eight columns of quad lane S, x and y chains alternating, as ONE asm statement: between single-instruction asm statements the
// compiler puts an s_nop behind every other pair of adds (24 per 64-add batch; it guards the accumulator, written two instructions
// earlier, as if it were the DPP source -- only src0 goes through the DPP network, and the terms are written long before)
template <int S>
__device__ __forceinline__ void add_quad_block(float &gx, float &gy, const float (&tx)[8], const float (&ty)[8]) {
#define KMAP_QP8(SS, P)                                                                                                            \
    if constexpr (S == SS)                                                                                                        \
        asm("v_add_f32_dpp %0, %2, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %10, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %3, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %11, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %4, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %12, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %5, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %13, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %6, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %14, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %7, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %15, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %8, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %16, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %9, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %17, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf"        \
            : "+v"(gx), "+v"(gy)                                                                                                  \
            : "v"(tx[0]), "v"(tx[1]), "v"(tx[2]), "v"(tx[3]), "v"(tx[4]), "v"(tx[5]), "v"(tx[6]), "v"(tx[7]), "v"(ty[0]), "v"(ty[1]), "v"(ty[2]),  \
              "v"(ty[3]), "v"(ty[4]), "v"(ty[5]), "v"(ty[6]), "v"(ty[7]));
    KMAP_QP8(0, "[0,0,0,0]") KMAP_QP8(1, "[1,1,1,1]") KMAP_QP8(2, "[2,2,2,2]") KMAP_QP8(3, "[3,3,3,3]")
    KMAP_QP8(4, "[0,0,2,2]") KMAP_QP8(5, "[1,1,3,3]")        // two sub-lanes per row (pair form): lane 0 / 1 of each lane pair
#undef KMAP_QP8
}
constexpr int SQ_CPL = 8;                       // consecutive columns per lane per batch (one 16-byte load of u16 sums)

struct SeqBatch {                                // raw operands of one batch of one lane
    uint32_t w[4];                               // 8 u16 sums (LUT source) ...
    float pf[SQ_CPL];                            // ... or 8 f32 probabilities
    float x[SQ_CPL], y[SQ_CPL];
};
template <bool LUTSRC>
__device__ __forceinline__ void seq_load(SeqBatch &b, const ProbSrc &src, const float *__restrict__ X,
                                         const float *__restrict__ Yy, int64_t lrc, int64_t jl, int64_t n, bool vec) {
    if (vec && jl + SQ_CPL <= n) {
        if (LUTSRC) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(src.ps + lrc * src.ld + jl);
            b.w[0] = v.x; b.w[1] = v.y; b.w[2] = v.z; b.w[3] = v.w;
        } else {
#pragma unroll
            for (int c = 0; c < SQ_CPL; ++c) b.pf[c] = src.pf[lrc * src.ld + jl + c];
        }
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(X + jl), a1 = *reinterpret_cast<const f32x4 *>(X + jl + 4);
        const f32x4 c0 = *reinterpret_cast<const f32x4 *>(Yy + jl), c1 = *reinterpret_cast<const f32x4 *>(Yy + jl + 4);
        b.x[0] = a0.x; b.x[1] = a0.y; b.x[2] = a0.z; b.x[3] = a0.w; b.x[4] = a1.x; b.x[5] = a1.y; b.x[6] = a1.z; b.x[7] = a1.w;
        b.y[0] = c0.x; b.y[1] = c0.y; b.y[2] = c0.z; b.y[3] = c0.w; b.y[4] = c1.x; b.y[5] = c1.y; b.y[6] = c1.z; b.y[7] = c1.w;
    } else {
#pragma unroll
        for (int c = 0; c < SQ_CPL; ++c) {
            const int64_t j = (jl + c < n) ? jl + c : n - 1;
            if (LUTSRC) {
                const uint32_t v = src.ps[lrc * src.ld + j];
                if (c & 1) b.w[c >> 1] |= v << 16;
                else b.w[c >> 1] = v;
            } else {
                b.pf[c] = src.pf[lrc * src.ld + j];
            }
            b.x[c] = X[j];
            b.y[c] = Yy[j];
        }
    }
}

constexpr int SQ_WAVES = 4;                     // waves per block (the LUT is staged once per block)

// The 8 terms of one lane's batch: t * dx, t * dy of columns jl32 .. jl32 + 7 against point (xi, yi) = row i32, and the batch's
// cross-entropy contribution in log2 units.  SLOW: generic IEEE divisions (some squared distance beyond 1e30); LOSS: the batch has
// columns right of the wave's rows; MASK: per-term predicates (the batch reaches past column n - 1 or contains the diagonal of
// one of the wave's rows).  The two divisions are the exhaustively verified short sequences of seq_div.h.
template <bool LUTSRC, bool SLOW, bool LOSS, bool MASK>
__device__ __forceinline__ void seq_terms(const SeqBatch &cur, const float *__restrict__ lut_s, float xi, float yi, int i32, int n32,
                                          int jl32, float (&tx)[SQ_CPL], float (&ty)[SQ_CPL], float &ce2) {
    float prod = 1.0f, esum = 0.0f;
#pragma unroll
    for (int c = 0; c < SQ_CPL; ++c) {                               // 8 independent terms
        const int j = jl32 + c;
        const float p = LUTSRC ? lut_s[(cur.w[c >> 1] >> (16 * (c & 1))) & 0xFFFFu] : cur.pf[c];
        const float dx = xi - cur.x[c], dy = yi - cur.y[c];
        const float d2 = dx * dx + dy * dy;                          // (dx*dx) + (dy*dy), no FMA (taichi_core.py:254)
        float q = SLOW ? 1.0f / (1.0f + d2) : seq_rcp<KMAP_SEQ_RCP_STEPS>(1.0f + d2);        // :255
        q = __builtin_amdgcn_fmed3f(q, 0.001f, 0.999f);              // np.minimum(.., 1 - 1e-3), np.maximum(.., 1e-3)
        const float omq = 1.0f - q;
        const float u = SLOW ? q / omq : seq_quo<KMAP_SEQ_QUO_RSTEPS, KMAP_SEQ_QUO_STEPS>(q, omq);   // visualization.py:132-134
        const float t = u * (p - q);
        const bool use = !MASK || ((j < n32) && (j != i32));
        tx[c] = use ? t * dx : 0.0f;                                 // products rounded on their own (-ffp-contract=off)
        ty[c] = use ? t * dy : 0.0f;
        if (LOSS) {
            // -(p ln q + (1-p) ln(1-q)) = -ln2 (log2(1-q) + p log2(q/(1-q))): one log per pair + one log of the product
            // of the eight (1-q); the reference's eps branches change a term by < 1e-9 relative (p < 1e-10) or not at
            // all (p = 1), and the loss is not part of the bit-pinned path
            const bool live = !MASK || ((j < n32) && (j > i32));     // a plain batch with loss lies right of all the wave's rows
            esum += live ? p * __builtin_amdgcn_logf(u) : 0.0f;
            prod *= live ? omq : 1.0f;
        }
    }
    if (LOSS) ce2 = __builtin_amdgcn_logf(prod) + esum;
}
// wave-uniform dispatch over the variants.  rows_in_wave consecutive rows from wave_row_min; batch = columns [j0, j0 + batch_cols)
template <bool LUTSRC>
__device__ __forceinline__ void seq_terms_dispatch(const SeqBatch &cur, const float *__restrict__ lut_s, float xi, float yi, int i32,
                                                   int64_t n, int64_t j0, int batch_cols, int64_t wave_row_min, int rows_in_wave,
                                                   int jl32, float (&tx)[SQ_CPL], float (&ty)[SQ_CPL], float &ce2) {
    // (a) no column of the batch lies right of any of the wave's rows -> no loss terms (each unordered pair is charged to its
    // j > i side); (b) some squared distance is too large for the short divisions -> generic division
    const bool want_loss = (j0 + batch_cols - 1) > wave_row_min;
    float d2max = 0.0f;
#pragma unroll
    for (int c = 0; c < SQ_CPL; ++c) {
        const float dx = xi - cur.x[c], dy = yi - cur.y[c];
        d2max = fmaxf(d2max, dx * dx + dy * dy);
    }
    const bool slow = __any(!(d2max < 1e30f));
    // (c) the batch neither reaches past column n-1 nor contains the diagonal of any of the wave's rows -> no per-term masks
    const bool plain = (j0 + batch_cols <= n) && (j0 + batch_cols - 1 < wave_row_min || j0 > wave_row_min + rows_in_wave - 1);
    const int n32 = (int)n;
    ce2 = 0.0f;
    if (slow) {   // rare (coordinates beyond 1e15): one generic instantiation
        seq_terms<LUTSRC, true, true, true>(cur, lut_s, xi, yi, i32, n32, jl32, tx, ty, ce2);
    } else if (plain) {
        if (want_loss) seq_terms<LUTSRC, false, true, false>(cur, lut_s, xi, yi, i32, n32, jl32, tx, ty, ce2);
        else seq_terms<LUTSRC, false, false, false>(cur, lut_s, xi, yi, i32, n32, jl32, tx, ty, ce2);
    } else {
        seq_terms<LUTSRC, false, true, true>(cur, lut_s, xi, yi, i32, n32, jl32, tx, ty, ce2);
    }
}

// quad form, block `bid` of the rows [0, nrows) (lut_s: the block's LUT copy in LDS, already filled; wl: SQ_WAVES doubles of LDS)
// SUB = 4: the quad form.  SUB = 2 (pair form): two sub-lanes per row, 32 rows per wave -- every ordered add then serves 32 rows instead
// of 16 (4 instead of 8 add instructions per pair), at twice the columns per lane; one round of pair waves replaces two rounds of quad
// waves (12.6 N against 2 x 7.3 N instructions per SIMD).  Rows [lrow0, lrow0 + ...) of the local range, bounded by `nrows`.
template <bool LUTSRC, int SUB>
__device__ __forceinline__ void seq_quad_body(const ProbSrc &src, const float *__restrict__ Y, int64_t n, int64_t row0, int64_t lrow0,
                                              int64_t nrows, float *__restrict__ G, double *__restrict__ loss_part, int64_t bid,
                                              const float *__restrict__ lut_s, double *wl) {
    constexpr int SQ_SUB = SUB, SQ_ROWS = KMAP_WAVE / SUB, SQ_BATCH = SUB * SQ_CPL;   // shadow the quad form's constants
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (SQ_SUB - 1);
    const int64_t lr = lrow0 + (bid * SQ_WAVES + wave) * SQ_ROWS + (lane / SQ_SUB);
    const bool valid = lr < nrows;
    const int64_t lrc = valid ? lr : nrows - 1;
    const int64_t i = row0 + lrc;
    const float *X = Y, *Yy = Y + n;
    const float xi = X[i], yi = Yy[i];
    const bool vec = ((n & 3) == 0) && (!LUTSRC || (src.ld % 8 == 0));
    const int i32 = (int)i;                      // n < 2^31 (checked by the host)
    const int64_t wave_row_min = row0 + lrow0 + (bid * SQ_WAVES + wave) * SQ_ROWS;   // smallest global row of the wave
    float gx = 0.0f, gy = 0.0f, ce_acc = 0.0f;
    double loss = 0.0;
    SeqBatch cur, nxt;
    seq_load<LUTSRC>(cur, src, X, Yy, lrc, (int64_t)sub * SQ_CPL, n, vec);
    for (int64_t j0 = 0; j0 < n; j0 += SQ_BATCH) {
        const int64_t jl = j0 + (int64_t)sub * SQ_CPL;                       // this lane's first column of the batch
        if (j0 + SQ_BATCH < n) seq_load<LUTSRC>(nxt, src, X, Yy, lrc, jl + SQ_BATCH, n, vec);   // prefetch
        float tx[SQ_CPL], ty[SQ_CPL];
        float ce2;                                                           // loss terms in log2 units (order-free)
        seq_terms_dispatch<LUTSRC>(cur, lut_s, xi, yi, i32, n, j0, SQ_BATCH, wave_row_min, SQ_ROWS, (int)jl, tx, ty, ce2);
        ce_acc += ce2;
        // ordered accumulation over the batch's 32 columns: column j0 + 8*s2 + c lives in sub-lane s2, slot c
        asm volatile("s_nop 1");
        if constexpr (SUB == 4) {
            add_quad_block<0>(gx, gy, tx, ty);
            add_quad_block<1>(gx, gy, tx, ty);
            add_quad_block<2>(gx, gy, tx, ty);
            add_quad_block<3>(gx, gy, tx, ty);
        } else {
            add_quad_block<4>(gx, gy, tx, ty);
            add_quad_block<5>(gx, gy, tx, ty);
        }
        if (((j0 / SQ_BATCH) & 7) == 7) {
            loss += (double)ce_acc;
            ce_acc = 0.0f;
        }
        cur = nxt;
    }
    loss += (double)ce_acc;
    loss *= -0.6931471805599453;   // log2 units -> -ln
    if (valid && sub == 0) {
        G[i] = gx;
        G[n + i] = gy;
    }
    if (!valid) loss = 0.0;
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_down(loss, o);
    if (lane == 0) wl[wave] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < SQ_WAVES; ++w) t += wl[w];
        loss_part[bid] = t;
    }
}

// SEQ forces, WIDE form: a row is owned by GW lanes (GW = 8 .. 64; 64 / GW rows per wave).  Same terms, same order of the row
// sum; what differs is how the terms reach the accumulator: every lane writes its 8 (t dx, t dy) pairs to a per-wave LDS strip,
// and the row's first lane then adds the strip's 8 GW pairs in column order (one packed f32 add per pair).
// Per column and wave that costs 288 / (8 GW) + 1 VALU instructions instead of the quad form's 11 -- but GW / 4 times the lanes
// per row, i.e. more total work: it is for the rows that do NOT fill the machine.  With 16 rows per quad wave and 3 waves per SIMD
// (145 VGPRs), 49 152 rows fill an MI355X exactly; the 848 remaining rows of N = 50 000 were a fourth round of 14 blocks that ran
// alone for a full millisecond (N = 49 152: 3.00 ms, N = 49 216: 3.98 ms).  As 848 one-row waves they take ~0.16 ms.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// wide form, block `bid` of the local rows [lrow0, nrows); xch_all: SQ_WAVES x 64 x 8 (t dx, t dy) pairs of LDS
// acc + (value of `v` in lane K of the caller's 16-lane DPP row), valid in lane 0 of the row: row_ror:(16 - K) makes lane L read lane
// (L + K) mod 16.
template <int K>
__device__ __forceinline__ float add_row16_lane(float acc, float v) {
    float r;
    if constexpr (K == 0) { r = acc + v; return r; }
#define KMAP_ROR(KK, N) if constexpr (K == KK) asm("v_add_f32_dpp %0, %1, %2 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(acc));
    KMAP_ROR(1, 15) KMAP_ROR(2, 14) KMAP_ROR(3, 13) KMAP_ROR(4, 12) KMAP_ROR(5, 11) KMAP_ROR(6, 10) KMAP_ROR(7, 9) KMAP_ROR(8, 8)
    KMAP_ROR(9, 7) KMAP_ROR(10, 6) KMAP_ROR(11, 5) KMAP_ROR(12, 4) KMAP_ROR(13, 3) KMAP_ROR(14, 2) KMAP_ROR(15, 1)
#undef KMAP_ROR
    return r;
}
// the eight columns of source lane K (K >= 1) of a batch, x and y chains alternating, as ONE asm statement: left to itself the
// compiler puts an s_nop behind every pair of these adds (it treats the accumulator, written two instructions earlier, as if it
// were the DPP source: 112 s_nop for the 240 adds of a batch).  Only src0 goes through the DPP network, and tx / ty are written
// long before.
template <int K>
__device__ __forceinline__ void add_row16_block(float &gx, float &gy, const float (&tx)[SQ_CPL], const float (&ty)[SQ_CPL]) {
#define KMAP_ROR8(KK, N)                                                                                                          \
    if constexpr (K == KK)                                                                                                        \
        asm("v_add_f32_dpp %0, %2, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %10, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %3, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %11, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %4, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %12, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %5, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %13, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %6, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %14, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %7, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %15, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %8, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %16, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %9, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %17, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf"        \
            : "+v"(gx), "+v"(gy)                                                                                                  \
            : "v"(tx[0]), "v"(tx[1]), "v"(tx[2]), "v"(tx[3]), "v"(tx[4]), "v"(tx[5]), "v"(tx[6]), "v"(tx[7]), "v"(ty[0]), "v"(ty[1]), "v"(ty[2]),  \
              "v"(ty[3]), "v"(ty[4]), "v"(ty[5]), "v"(ty[6]), "v"(ty[7]));
    KMAP_ROR8(1, 15) KMAP_ROR8(2, 14) KMAP_ROR8(3, 13) KMAP_ROR8(4, 12) KMAP_ROR8(5, 11) KMAP_ROR8(6, 10) KMAP_ROR8(7, 9) KMAP_ROR8(8, 8)
    KMAP_ROR8(9, 7) KMAP_ROR8(10, 6) KMAP_ROR8(11, 5) KMAP_ROR8(12, 4) KMAP_ROR8(13, 3) KMAP_ROR8(14, 2) KMAP_ROR8(15, 1)
#undef KMAP_ROR8
}
// Sixteen lanes per row (one DPP row), four rows per wave, no LDS: for sessions with fewer rows than one round of quad waves, where
// a wave is a chain of dependent adds and not a share of issue slots.  Lane s of a row computes the terms of columns
// j0 + 8 s .. + 7 of a 128-column batch; lane 0 of the row adds them in column order through row_ror sources (the other lanes
// execute the same adds on rotated operands and are ignored).  The x and y chains alternate, so consecutive adds of one chain are
// two instructions apart -- their latency -- and nothing waits for an LDS round trip as in the strip-exchange form.
constexpr int SR_SUB = 16, SR_ROWS = KMAP_WAVE / SR_SUB, SR_BATCH = SR_SUB * SQ_CPL;
template <bool LUTSRC>
__device__ __forceinline__ void seq_row16_body(const ProbSrc &src, const float *__restrict__ Y, int64_t n, int64_t row0, int64_t lrow0,
                                               int64_t nrows, float *__restrict__ G, double *__restrict__ loss_part, int64_t bid,
                                               const float *__restrict__ lut_s, double *wl) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (SR_SUB - 1);
    const int64_t wave_lr = lrow0 + (bid * SQ_WAVES + wave) * SR_ROWS;    // first local row of the wave
    const int64_t lr = wave_lr + (lane / SR_SUB);
    const bool valid = lr < nrows;
    const int64_t lrc = valid ? lr : nrows - 1;
    const int64_t i = row0 + lrc;
    const float *X = Y, *Yy = Y + n;
    const float xi = X[i], yi = Yy[i];
    const bool vec = ((n & 3) == 0) && (!LUTSRC || (src.ld % 8 == 0));
    const int i32 = (int)i;
    const int64_t wave_row_min = row0 + wave_lr;
    float gx = 0.0f, gy = 0.0f, ce_acc = 0.0f;
    double loss = 0.0;
    // (computing the terms of batch b + 1 in the same loop body as the adds of batch b -- a software pipeline for the scheduler to
    // interleave -- measured slower: 0.066 vs 0.058 ms at N = 4000)
    SeqBatch cur, nxt;
    seq_load<LUTSRC>(cur, src, X, Yy, lrc, (int64_t)sub * SQ_CPL, n, vec);
    for (int64_t j0 = 0; j0 < n; j0 += SR_BATCH) {
        const int64_t jl = j0 + (int64_t)sub * SQ_CPL;
        if (j0 + SR_BATCH < n) seq_load<LUTSRC>(nxt, src, X, Yy, lrc, jl + SR_BATCH, n, vec);   // prefetch
        float tx[SQ_CPL], ty[SQ_CPL];
        float ce2;
        seq_terms_dispatch<LUTSRC>(cur, lut_s, xi, yi, i32, n, j0, SR_BATCH, wave_row_min, SR_ROWS, (int)jl, tx, ty, ce2);
        ce_acc += ce2;
        asm volatile("s_nop 1");
#define SEQ_ADD16(K)                                                                                  \
        _Pragma("unroll") for (int c = 0; c < SQ_CPL; ++c) {                                          \
            gx = add_row16_lane<K>(gx, tx[c]);                                                        \
            gy = add_row16_lane<K>(gy, ty[c]);                                                        \
        }
        SEQ_ADD16(0)
#undef SEQ_ADD16
        add_row16_block<1>(gx, gy, tx, ty); add_row16_block<2>(gx, gy, tx, ty); add_row16_block<3>(gx, gy, tx, ty);
        add_row16_block<4>(gx, gy, tx, ty); add_row16_block<5>(gx, gy, tx, ty); add_row16_block<6>(gx, gy, tx, ty);
        add_row16_block<7>(gx, gy, tx, ty); add_row16_block<8>(gx, gy, tx, ty); add_row16_block<9>(gx, gy, tx, ty);
        add_row16_block<10>(gx, gy, tx, ty); add_row16_block<11>(gx, gy, tx, ty); add_row16_block<12>(gx, gy, tx, ty);
        add_row16_block<13>(gx, gy, tx, ty); add_row16_block<14>(gx, gy, tx, ty); add_row16_block<15>(gx, gy, tx, ty);
        if (((j0 / SR_BATCH) & 7) == 7) {
            loss += (double)ce_acc;
            ce_acc = 0.0f;
        }
        cur = nxt;
    }
    loss += (double)ce_acc;
    loss *= -0.6931471805599453;   // log2 units -> -ln
    if (valid && sub == 0) {
        G[i] = gx;
        G[n + i] = gy;
    }
    if (!valid) loss = 0.0;
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_down(loss, o);
    if (lane == 0) wl[wave] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < SQ_WAVES; ++w) t += wl[w];
        loss_part[bid] = t;
    }
}

template <bool LUTSRC, int GW>
__device__ __forceinline__ void seq_wide_body(const ProbSrc &src, const float *__restrict__ Y, int64_t n, int64_t row0, int64_t lrow0,
                                              int64_t nrows, float *__restrict__ G, double *__restrict__ loss_part, int64_t bid,
                                              const float *__restrict__ lut_s, double *wl, f32x2 *xch_all) {
    constexpr int RW = KMAP_WAVE / GW;                       // rows per wave
    constexpr int BC = GW * SQ_CPL;                          // columns per batch
    f32x2 *xch = xch_all + (size_t)(threadIdx.x >> 6) * (KMAP_WAVE * SQ_CPL);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (GW - 1), grp = lane / GW;
    const int64_t wave_lr = lrow0 + (bid * SQ_WAVES + wave) * RW;    // first local row of the wave
    const int64_t lr = wave_lr + grp;
    const bool valid = lr < nrows;
    const int64_t lrc = valid ? lr : nrows - 1;
    const int64_t i = row0 + lrc;
    const float *X = Y, *Yy = Y + n;
    const float xi = X[i], yi = Yy[i];
    const bool vec = ((n & 3) == 0) && (!LUTSRC || (src.ld % 8 == 0));
    const int i32 = (int)i;
    const int64_t wave_row_min = row0 + wave_lr;
    f32x2 acc = {0.0f, 0.0f};
    float ce_acc = 0.0f;
    double loss = 0.0;
    SeqBatch cur, nxt;
    seq_load<LUTSRC>(cur, src, X, Yy, lrc, (int64_t)sub * SQ_CPL, n, vec);
    f32x2 *mine = xch + (size_t)lane * SQ_CPL;                         // = strip of row grp, columns sub * 8 .. + 7
    const f32x4 *strip = reinterpret_cast<const f32x4 *>(xch + (size_t)grp * BC);
    int batch = 0;
    for (int64_t j0 = 0; j0 < n; j0 += BC, ++batch) {
        const int64_t jl = j0 + (int64_t)sub * SQ_CPL;
        if (j0 + BC < n) seq_load<LUTSRC>(nxt, src, X, Yy, lrc, jl + BC, n, vec);          // prefetch
        float tx[SQ_CPL], ty[SQ_CPL];
        float ce2;
        seq_terms_dispatch<LUTSRC>(cur, lut_s, xi, yi, i32, n, j0, BC, wave_row_min, RW, (int)jl, tx, ty, ce2);
        ce_acc += ce2;
#pragma unroll
        for (int c = 0; c < SQ_CPL; c += 2) {
            const f32x4 v = {tx[c], ty[c], tx[c + 1], ty[c + 1]};
            *reinterpret_cast<f32x4 *>(mine + c) = v;
        }
        // same wave writes and reads: LDS operations of a wave execute in order; the compiler must not move the reads up
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (sub == 0) {   // ONE lane per row carries the sum: 64 lanes reading the same 16 bytes made the strip reads LDS-bandwidth bound
#pragma unroll 16
            for (int m = 0; m < BC / 2; ++m) {                               // ordered: columns j0 + 2m, j0 + 2m + 1
                const f32x4 v = strip[m];
                acc += f32x2{v.x, v.y};
                acc += f32x2{v.z, v.w};
            }
        }
        asm volatile("" ::: "memory");                                       // the next batch's writes stay behind these reads
        if ((batch & 7) == 7) {
            loss += (double)ce_acc;
            ce_acc = 0.0f;
        }
        cur = nxt;
    }
    loss += (double)ce_acc;
    loss *= -0.6931471805599453;
    if (valid && sub == 0) {
        G[i] = acc.x;
        G[n + i] = acc.y;
    }
    if (!valid) loss = 0.0;
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_down(loss, o);
    if (lane == 0) wl[wave] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < SQ_WAVES; ++w) t += wl[w];
        loss_part[bid] = t;
    }
}

// One launch for both forms: blocks [0, nb_tail) take the left-over rows in the wide form (GW lanes per row; GW = 0: none), the
// blocks behind them the whole rounds in the quad form.  Launched together the wide waves share their SIMDs with three quad
// waves each, which hides the latency of their N-step dependent add chain (alone on the machine -- as a second launch -- the
// 848 one-row waves of N = 50 000 took 0.5 - 0.6 ms; as a fourth wave per SIMD they cost their ~8 % of issue slots).  Wide
// blocks come first in the grid so that they are placed before the CUs fill up.
template <bool LUTSRC, int GW>
__global__ __launch_bounds__(KMAP_WAVE *SQ_WAVES) void forces_seq_kernel(ProbSrc src, const float *__restrict__ Y, int64_t n,
                                                               int64_t row0, int64_t pair_rows, int64_t main_rows, int64_t nrows,
                                                               int nb_tail, int nb_pair, int nb_main, float *__restrict__ G,
                                                               double *__restrict__ loss_part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr size_t XCH_FLOATS = GW ? (size_t)SQ_WAVES * KMAP_WAVE * SQ_CPL * 2 : 0;   // exchange strips first, the LUT behind them
    float *lut_s = smem + XCH_FLOATS;
    if (LUTSRC) {
        for (int t = threadIdx.x; t < src.lut_len && t < F_LUT_LDS; t += blockDim.x) lut_s[t] = src.lut[t];
        __syncthreads();
    }
    __shared__ double wl[SQ_WAVES];
    if constexpr (GW != 0) {
        if ((int)blockIdx.x < nb_tail) {   // block-uniform
            if constexpr (GW == 16)   // one DPP row per matrix row: no exchange through LDS
                seq_row16_body<LUTSRC>(src, Y, n, row0, main_rows, nrows, G, loss_part + nb_main, (int64_t)blockIdx.x, lut_s, wl);
            else
                seq_wide_body<LUTSRC, GW>(src, Y, n, row0, main_rows, nrows, G, loss_part + nb_main, (int64_t)blockIdx.x, lut_s, wl,
                                          reinterpret_cast<f32x2 *>(smem));
            return;
        }
    }
    // grid order: wide blocks, pair blocks (rows [0, pair_rows)), quad blocks (rows [pair_rows, main_rows)); loss partials: quad |
    // pair | wide (nb_main = quad + pair blocks)
    const int b = (int)blockIdx.x - nb_tail;
    if (b < nb_pair) seq_quad_body<LUTSRC, 2>(src, Y, n, row0, 0, pair_rows, G, loss_part + (nb_main - nb_pair), (int64_t)b, lut_s, wl);
    else seq_quad_body<LUTSRC, 4>(src, Y, n, row0, pair_rows, main_rows, G, loss_part, (int64_t)(b - nb_pair), lut_s, wl);
}

// deterministic reduction of the per-block loss partials inside the fused apply kernel (every block computes the same total):
// thread t adds part[t], part[t + 256], ... in order, then a fixed tree over the 256 thread sums
__device__ __forceinline__ double loss_total_256(const double *__restrict__ part, int n_part, double *sh /* [256] LDS */) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n_part; i += BLK) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = BLK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    const double total = sh[0];
    __syncthreads();
    return total;
}
// standalone reduction (symmetric kernels: ~2 10^4 partials; multi-GPU: a rank's partials before the all-reduce): 1024 threads,
// thread t adds part[t], part[t + 1024], ... in order, then a fixed tree (a 256-thread block took 30 us on 19 600 partials)
constexpr int RL_TPB = 1024;
__global__ __launch_bounds__(RL_TPB) void reduce_loss_kernel(const double *__restrict__ part, int n_part,
                                                             double *__restrict__ loss_out) {
    __shared__ double sh[RL_TPB];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_part; i += RL_TPB) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = RL_TPB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = sh[0];
}

// ---- multi-GPU one-message protocol ---------------------------------------------------------------
// A rank's contribution to an iteration is ONE float buffer: [0, 2N) its gradient entries (own rows, or partial sums for all
// points in the cyclic layout) and MSG_EXTRA floats behind them that carry its loss partial as an exact integer: the f64 value
// in 48.48 fixed point, cut into six 16-bit limbs, each stored as a float.  A float32 SUM all-reduce adds limbs of up to 256
// ranks without rounding (6 x < 2^24), in any order, so every rank decodes the same total whatever algorithm the collective
// library picks -- one collective per iteration instead of a float32 and a float64 one, and the stop / snapshot decisions
// (which come from the loss alone) cannot diverge between ranks.  Limb 6 flags a non-finite or out-of-range partial (-> NaN).
constexpr int MSG_EXTRA = 8;   // six limbs, flag, pad
__device__ __forceinline__ void loss_to_limbs(double v, float *__restrict__ tail) {
    const bool bad = !(v >= 0.0) || !(v < 140737488355328.0);            // NaN, negative or >= 2^47
    unsigned long long hi = 0, lo = 0;
    if (!bad) {
        hi = (unsigned long long)v;                                       // floor (v >= 0)
        lo = (unsigned long long)((v - (double)hi) * 281474976710656.0);  // exact difference, truncated at 2^-48
    }
    for (int i = 0; i < 3; ++i) tail[i] = (float)((lo >> (16 * i)) & 0xFFFFull);
    for (int i = 0; i < 3; ++i) tail[3 + i] = (float)((hi >> (16 * i)) & 0xFFFFull);
    tail[6] = bad ? 1.0f : 0.0f;
    tail[7] = 0.0f;
}
__device__ __forceinline__ double loss_from_limbs(const float *__restrict__ tail) {
    if (tail[6] != 0.0f) return (double)NAN;
    unsigned long long lo = 0, hi = 0;                                    // summed limbs carry past 16 bits: integer Horner
    for (int i = 2; i >= 0; --i) lo = (lo << 16) + (unsigned long long)tail[i];
    for (int i = 2; i >= 0; --i) hi = (hi << 16) + (unsigned long long)tail[3 + i];
    hi += lo >> 48;
    lo &= 0xFFFFFFFFFFFFull;
    return (double)hi + (double)lo * (1.0 / 281474976710656.0);
}
// reduce_loss_kernel with the total written as limbs behind the gradient (n_part = 0: a rank without rows sends zero)
__global__ __launch_bounds__(RL_TPB) void reduce_loss_limbs_kernel(const double *__restrict__ part, int n_part,
                                                                   float *__restrict__ tail) {
    __shared__ double sh[RL_TPB];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_part; i += RL_TPB) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = RL_TPB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_to_limbs(sh[0], tail);
}

// =================================================================================================
// loop state + apply
// =================================================================================================
constexpr int MAX_BEST = 64;
struct LoopState {
    long long iters;        // reference iterations executed (loss evaluations)
    int stopped;            // early stop reached (visualization.py:310-311)
    int jitter_used;        // normals consumed from the pre-drawn stream
    int n_best;
    float prev_loss;        // `loss` of the reference loop (inf before the first iteration)
    float last_loss;
    float worst_loss;       // = best_loss[n_best - 1], worst_slot = best_slot[n_best - 1]: with the fields above, all a non-leader
    int worst_slot;         //   thread reads (40 bytes instead of the whole 560-byte record)
    float best_loss[MAX_BEST];   // ascending (bisect.insort_right order)
    int best_slot[MAX_BEST];     // snapshot buffer holding that entry
};

// what one reference iteration decides from the loss (visualization.py:303-311), identical in every thread
struct StepDecision {
    float loss;
    int slot;               // snapshot buffer that takes the iterate if `insert`
    bool halted, insert, stop;
};
__device__ __forceinline__ StepDecision step_decide(const LoopState *__restrict__ st, double loss_total) {
    StepDecision d;
    d.halted = st->stopped != 0;
    d.loss = (float)(2.0 * loss_total);                                  // np.sum(ce) * 2 (visualization.py:176)
    d.insert = d.loss < st->worst_loss;                                  // :303 (false for NaN)
    d.stop = fabsf(st->prev_loss - d.loss) < 1e-7f * fabsf(d.loss);      // :310
    d.slot = st->worst_slot;
    return d;
}
// one coordinate (not point 0 / 1, which the leader owns): snapshot, then the gradient step
__device__ __forceinline__ void step_element(const StepDecision &d, int64_t idx, float g_raw, float *__restrict__ Y,
                                             float *__restrict__ snaps, int64_t n, float lr) {
    const float y = Y[idx];
    if (d.insert) snaps[(int64_t)d.slot * 2 * n + idx] = y;             // snapshot of the iterate that produced `loss`
    if (!d.stop) {
        const float g = 4.0f * g_raw;                                    // gradient_loss_taichi returns 4.0 * ret (:145)
        Y[idx] = y + (-g * lr);                                          // ld_data += (-grad_loss * learning_rate) (:316)
    }
}
// the leader thread: loop record (best list, stop flag, loss log) and points 0 / 1 incl. add_jitter.  gsp = raw gradient of
// x0, x1, y0, y1
__device__ void step_leader(LoopState *__restrict__ states, int cur, const StepDecision &d, float *__restrict__ Y, const float (&gsp)[4],
                            float *__restrict__ snaps, int64_t n, float lr, const double *__restrict__ normals,
                            const int *__restrict__ n_normals_dev, float *__restrict__ loss_log, int64_t loss_log_cap) {
    const LoopState st = states[cur];
    if (d.halted) {
        states[cur ^ 1] = st;
        return;
    }
    const int nb = st.n_best, slot = d.slot;
    const float loss = d.loss;
    LoopState ns = st;
    ns.iters = st.iters + 1;
    ns.last_loss = loss;
    if (st.iters < loss_log_cap) loss_log[st.iters] = loss;
    if (d.insert) {   // best_res_list[:-1] then bisect.insort_right by loss (:304-308)
        int pos = 0;
        while (pos < nb - 1 && !(st.best_loss[pos] > loss)) ++pos;
        for (int t = nb - 1; t > pos; --t) {
            ns.best_loss[t] = st.best_loss[t - 1];
            ns.best_slot[t] = st.best_slot[t - 1];
        }
        ns.best_loss[pos] = loss;
        ns.best_slot[pos] = slot;
        ns.worst_loss = ns.best_loss[nb - 1];
        ns.worst_slot = ns.best_slot[nb - 1];
        for (int p = 0; p < 2 && p < n; ++p)
            for (int c = 0; c < 2; ++c) snaps[(int64_t)slot * 2 * n + (int64_t)c * n + p] = Y[(int64_t)c * n + p];
    }
    if (d.stop) {
        ns.stopped = 1;
    } else {
        ns.prev_loss = loss;
        const int n_normals = *n_normals_dev;
        // update + add_jitter for points 0 and 1 (visualization.py:179-196 indexes the 2 x N array as N x 2,
        // so `ld_data[:, p]` is the (x_p, y_p) pair of point p)
        for (int p = 0; p < 2 && p < n; ++p) {
            float v[2];
            for (int c = 0; c < 2; ++c) {
                const float g = 4.0f * gsp[2 * c + p];
                v[c] = Y[(int64_t)c * n + p] + (-g * lr);
            }
            const int lo_i = (v[1] < v[0]) ? 1 : 0;                     // argsort of two values (stable)
            const float diff = v[1 - lo_i] - v[lo_i];                   // np.diff of the sorted pair
            if (diff < 0.1f) {
                const double nrm = (ns.jitter_used < n_normals) ? normals[ns.jitter_used] : 0.0;
                ns.jitter_used += 1;
                v[lo_i] = (float)((double)v[lo_i] + nrm);               // f32 array element += f64 draw
            }
            for (int c = 0; c < 2; ++c) Y[(int64_t)c * n + p] = v[c];
        }
    }
    states[cur ^ 1] = ns;
}

// apply with the gradient in memory.  FUSED_LOSS: the block first reduces the force kernel's loss partials itself (single-GPU
// step: forces -> this kernel); otherwise the total comes from loss_in[0] (multi-GPU: after the all-reduce)
template <bool FUSED_LOSS>
__global__ __launch_bounds__(BLK) void apply_kernel(LoopState *__restrict__ states, int cur, float *__restrict__ Y,
                                                    const float *__restrict__ G, const double *__restrict__ loss_in, int n_part,
                                                    float *__restrict__ snaps, int64_t n, float lr,
                                                    const double *__restrict__ normals, const int *__restrict__ n_normals_dev,
                                                    float *__restrict__ loss_log, int64_t loss_log_cap) {
    __shared__ double sh[FUSED_LOSS ? BLK : 1];
    const double total = FUSED_LOSS ? loss_total_256(loss_in, n_part, sh) : loss_in[0];
    const StepDecision d = step_decide(&states[cur], total);   // every block reads the same, already complete record
    const int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const bool special = (idx == 0 || idx == 1 || idx == n || idx == n + 1);   // owned by the leader (jitter)
    if (!d.halted && idx < 2 * n && !special) step_element(d, idx, G[idx], Y, snaps, n, lr);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float gsp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int c = 0; c < 2; ++c)
            for (int p = 0; p < 2 && p < n; ++p) gsp[2 * c + p] = G[(int64_t)c * n + p];
        step_leader(states, cur, d, Y, gsp, snaps, n, lr, normals, n_normals_dev, loss_log, loss_log_cap);
    }
}

// apply for the one-message protocol: the summed message M = [gradient 2N | loss limbs].  CLEAR (row-sharded sessions, whose
// force kernel fills only the rank's own rows): every entry is zeroed once it has been read, so the next iteration's message
// starts from x + 0 + ... + 0 without a memset launch (the cyclic layout overwrites all 2N entries itself).
template <bool CLEAR>
__global__ __launch_bounds__(BLK) void apply_msg_kernel(LoopState *__restrict__ states, int cur, float *__restrict__ Y,
                                                        float *__restrict__ M, float *__restrict__ snaps, int64_t n, float lr,
                                                        const double *__restrict__ normals, const int *__restrict__ n_normals_dev,
                                                        float *__restrict__ loss_log, int64_t loss_log_cap) {
    const double total = loss_from_limbs(M + 2 * n);
    const StepDecision d = step_decide(&states[cur], total);
    const int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const bool special = (idx == 0 || idx == 1 || idx == n || idx == n + 1);   // owned by the leader (jitter)
    if (idx < 2 * n && !special) {
        const float g = M[idx];
        if (CLEAR) M[idx] = 0.0f;
        if (!d.halted) step_element(d, idx, g, Y, snaps, n, lr);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float gsp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int c = 0; c < 2; ++c)
            for (int p = 0; p < 2 && p < n; ++p) {
                gsp[2 * c + p] = M[(int64_t)c * n + p];
                if (CLEAR) M[(int64_t)c * n + p] = 0.0f;
            }
        step_leader(states, cur, d, Y, gsp, snaps, n, lr, normals, n_normals_dev, loss_log, loss_log_cap);
    }
}

// symmetric FAST kernel, single GPU: the sum of the row / column partials (sym_reduce_kernel's order: 8 lanes per element, lane l
// adds partials l, l + 8, ..., fixed butterfly) fused with apply -- the gradient never goes to memory.  The first four 8-lane
// groups of block 0 take x0, x1, y0, y1 and hand them to the leader through LDS; the other groups take the remaining 2N - 4
// coordinates in order.
__global__ __launch_bounds__(BLK) void sym_apply_kernel(LoopState *__restrict__ states, int cur, float *__restrict__ Y,
                                                        const float *__restrict__ rowpart, const float *__restrict__ colpart,
                                                        int64_t n_lblocks, int64_t nJ, const double *__restrict__ loss_sum,
                                                        float *__restrict__ snaps, int64_t n, float lr,
                                                        const double *__restrict__ normals, const int *__restrict__ n_normals_dev,
                                                        float *__restrict__ loss_log, int64_t loss_log_cap) {
    constexpr int SPLIT = 8;
    __shared__ float gsp_s[4];
    const StepDecision d = step_decide(&states[cur], loss_sum[0]);
    const int64_t grp = ((int64_t)blockIdx.x * BLK + threadIdx.x) / SPLIT;
    const int l = threadIdx.x & (SPLIT - 1);
    // group -> coordinate index: 0..3 -> x0, x1, y0, y1; 4.. -> the others in order
    int64_t idx;
    if (grp < 4) idx = (grp >> 1) * n + (grp & 1);
    else idx = (grp - 4 < n - 2) ? grp - 4 + 2 : grp - 4 + 4;
    const bool live = idx < 2 * n && (grp >= 4 || (grp & 1) < n);
    float g = 0.0f;
    if (live && !d.halted) {
        const int c = (int)(idx / n);
        const int64_t i = idx % n;
        const int64_t Ii = i / SY_R, Ji = i / SY_C;
        for (int64_t J = l; J < nJ; J += SPLIT)
            if (sy_tile_live(Ii, J)) g += rowpart[(J * 2 + c) * n + i];
        for (int64_t b = l; b < n_lblocks; b += SPLIT)
            if (sy_tile_live(b, Ji)) g += colpart[(b * 2 + c) * n + i];
    }
    g += __shfl_xor(g, 1);
    g += __shfl_xor(g, 2);
    g += __shfl_xor(g, 4);
    if (blockIdx.x == 0) {   // block-uniform
        if (grp < 4 && l == 0) gsp_s[grp] = g;
        __syncthreads();
    }
    if (live && !d.halted && grp >= 4 && l == 0) step_element(d, idx, g, Y, snaps, n, lr);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float gsp[4] = {gsp_s[0], gsp_s[1], gsp_s[2], gsp_s[3]};
        step_leader(states, cur, d, Y, gsp, snaps, n, lr, normals, n_normals_dev, loss_log, loss_log_cap);
    }
}

}  // namespace

// =================================================================================================
// session object
// =================================================================================================
struct kmap_embed {
    int64_t n = 0, row0 = 0, nrows = 0;
    int n_best = 10, mode = KMAP_EMBED_FAST;
    float lr = 0.01f;
    ProbSrc src{};
    float *lut_dev = nullptr;
    float *Y = nullptr, *G = nullptr, *snaps = nullptr, *loss_log = nullptr;
    double *loss_part = nullptr, *loss_sum = nullptr, *normals = nullptr;
    int n_normals = 0, n_part = 0;
    int64_t loss_log_cap = 1 << 16;
    LoopState *states = nullptr;
    int cur = 0;
    bool have_prob = false, have_coords = false;
    // SEQ: rows [0, seq_main_rows) of the local range go to the quad kernel (16 rows per wave), the rest -- the rows that would
    // form a last, nearly empty round of blocks -- to the wide kernel with seq_tail_g lanes per row (0: no tail)
    int64_t seq_main_rows = 0;
    int seq_tail_g = 0;
    int64_t seq_pair_rows = 0;      // of the main rows, [0, seq_pair_rows) run in the pair form (32 rows per wave); multiple of 128
    // symmetric FAST path (all rows local): partial buffers
    float *rowpart = nullptr, *colpart = nullptr;
    int64_t symI = 0, symJ = 0;
    bool sym = false;
    // symmetric FAST path sharded over ranks: rank r owns the 256-row blocks I = r, r + world, ... (cyclic: the upper-triangle
    // work per block shrinks with I); its probability rows are stored block after block (local block b = I / world)
    int world = 1, rank = 0;
    int64_t n_lblocks = 0;
    // jitter normals: fixed-capacity device buffer + device-resident count, so that the kernel arguments of an iteration never
    // change between launches (a captured hipGraph stays valid when the host refills the pool)
    int *n_normals_dev = nullptr;
    int normals_cap = 0;
    // two iterations (both parities of the double-buffered loop record) captured as one hipGraph and replayed by kmap_embed_step
    hipGraphExec_t gexec = nullptr;
    hipStream_t gstream = nullptr;
    int graph_cur = 0;
    bool graph_failed = false;
};

namespace {
void drop_graph(kmap_embed *e) {   // kernel arguments changed: the captured iterations are stale
    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    e->gexec = nullptr;
}
int seq_pair_blocks(const kmap_embed *e) { return (int)(e->seq_pair_rows / (2 * SQ_ROWS * SQ_WAVES)); }
// quad + pair blocks
int seq_main_blocks(const kmap_embed *e) {
    return seq_pair_blocks(e) + (int)((e->seq_main_rows - e->seq_pair_rows + SQ_ROWS * SQ_WAVES - 1) / (SQ_ROWS * SQ_WAVES));
}
int seq_tail_blocks(const kmap_embed *e) {
    if (!e->seq_tail_g) return 0;
    const int64_t rows_per_block = (int64_t)SQ_WAVES * (KMAP_WAVE / e->seq_tail_g);
    return (int)((e->nrows - e->seq_main_rows + rows_per_block - 1) / rows_per_block);
}
// How the SEQ rows are split between the quad kernel and the wide kernel.  The kernels are VALU-issue bound and every wave of a
// SIMD shares its issue slots, so the cost of a set of waves is (waves on the fullest SIMD) x (instructions per wave); per
// column a quad wave issues ~9.4 instructions (8 terms x 27 + 64 adds + ~20 per 32 columns), a wide wave with g lanes per row
// ~(30 / g + 1.2).  Whole rounds of quad waves (one wave on every SIMD) are the cheapest way to do rows; what is left over is
// given to whichever form finishes it soonest.
void seq_split(kmap_embed *e) {
    e->seq_main_rows = e->nrows;
    e->seq_tail_g = 0;
    e->seq_pair_rows = 0;
    static const int off = [] { const char *v = getenv("KMAP_SEQ_TAIL"); return v && v[0] == '0'; }();   // A/B switch
    if (off || e->nrows <= 0) return;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int64_t simds = 4 * (int64_t)cus;
    const int64_t round_rows = simds * SQ_ROWS;                       // rows of one wave on every SIMD
    const int64_t main_rows = (e->nrows / round_rows) * round_rows;
    const int64_t rem = e->nrows - main_rows;
    // two rounds of quad waves -> one round of pair waves (every ordered add serves 32 rows: 12.6 N against 14.6 N instructions per
    // SIMD); an odd round stays in the quad form and shares the SIMDs with the pair round
    static const int pair_on = [] { const char *v = getenv("KMAP_SEQ_PAIR"); return !(v && v[0] == '0'); }();   // A/B switch
    // ... but only next to quad waves: a pair round alone on the SIMDs (one wave each) exposes its add chain (N = 33 000: 1.52 against
    // 1.29 ms for two quad rounds), so at least one quad round stays -- R rounds of quad rows become (R - 1) / 2 pair rounds + the rest
    const int64_t rounds_q = main_rows / round_rows;
    static const int force_pr = [] { const char *v = getenv("KMAP_SEQ_PAIR_ROUNDS"); return v ? atoi(v) : -1; }();   // measurements
    int64_t pair_rounds = rounds_q >= 3 ? (rounds_q - 1) / 2 : 0;
    if (force_pr >= 0 && 2 * (int64_t)force_pr <= rounds_q) pair_rounds = force_pr;
    e->seq_pair_rows = pair_on ? pair_rounds * 2 * round_rows : 0;
    if (rem == 0) return;
    auto rounds = [&](int64_t waves) { return (double)((waves + simds - 1) / simds); };
    double best = rounds((rem + SQ_ROWS - 1) / SQ_ROWS) * 9.4;       // the remainder as quad waves
    int best_g = 0;
    for (int g : {8, 16, 32, 64}) {
        const int64_t waves = (rem + (KMAP_WAVE / g) - 1) / (KMAP_WAVE / g);
        const double cost = rounds(waves) * (30.0 / g + 1.2);
        if (cost < best) { best = cost; best_g = g; }
    }
    if (main_rows == 0) {
        // fewer rows than one round of quad waves (N < 16 384 on this part): the waves are dependent-add chains, not issue slots, and
        // whole-round counting misjudges 1.2 waves per SIMD as two rounds.  Measured (tools/bench_embed.py --modes seq, ms per
        // iteration, g = quad / 8 / 16 / 32 / 64): N = 1000: .036 .035 .026 .022 .021; 3000: .085 .073 .049 .051 .060;
        // 5000: .138 .115 .105 .105 .133; 8000: .214 .174 .155 .199 .307; 10 000: .265 .293 .249 .297 .462; 14 000: .372 .412 .429 ...
        best_g = rem <= 1500 ? 64 : rem <= 2500 ? 32 : rem <= 12000 ? 16 : 0;
    }
    static const int force_g = [] { const char *v = getenv("KMAP_SEQ_G"); return v ? atoi(v) : -1; }();   // measurements: 0 = quad, 8 .. 64
    if (force_g >= 0) best_g = force_g;
    if (best_g) {
        e->seq_main_rows = main_rows;
        e->seq_tail_g = best_g;
    }
}
int n_force_blocks(const kmap_embed *e) {
    if (e->sym) return (int)(e->n_lblocks * (((e->symJ + SY_WAVES - 1) / SY_WAVES) * SY_WAVES));
    if (e->mode == KMAP_EMBED_SEQ) return seq_main_blocks(e) + seq_tail_blocks(e);
    return (int)((e->nrows + F_RPW * F_WAVES - 1) / (F_RPW * F_WAVES));
}
}  // namespace

extern "C" {

int kmap_knn_sums_u8_dev(const uint8_t *D_dev, int64_t ldd, const int32_t *nb_dev, int64_t n, int n_nb, int64_t row0,
                         int64_t nrows, uint16_t *sums_dev, int64_t lds, void *stream) {
    KMAP_REQUIRE(n >= 0 && nrows >= 0 && row0 >= 0 && row0 + nrows <= n, "knn_sums: bad row range");
    KMAP_REQUIRE(n_nb > 0 && n_nb <= 256, "knn_sums: n_nb=%d out of range", n_nb);
    KMAP_REQUIRE(ldd >= n && lds >= n, "knn_sums: leading dimension < n");
    KMAP_REQUIRE(ldd % 16 == 0 && ((uintptr_t)D_dev % 16) == 0, "knn_sums: D must be 16-byte aligned with ldd %% 16 == 0");
    if (n == 0 || nrows == 0) return KMAP_OK;
    KMAP_REQUIRE(D_dev && nb_dev && sums_dev, "knn_sums: null pointer");
    hipStream_t st = as_stream(stream);
    int32_t *nbT = nullptr;
    KMAP_TRY(kmap_scratch((void **)&nbT, (size_t)n * n_nb * 4, st, KMAP_SLOT_D));
    {   // several output rows per block when their neighbour-sum rows fit LDS together
        static const int rows_on = [] { const char *e = getenv("KMAP_KNN_ROWS"); return e ? atoi(e) : 1; }();
        uint32_t *dmax_dev = nullptr, dmax = 255;
        KMAP_TRY(kmap_scratch((void **)&dmax_dev, 64, st, KMAP_SLOT_C));
        KMAP_CHECK_HIP(hipMemsetAsync(dmax_dev, 0, 4, st));
        max_u8_kernel<<<2048, 256, 0, st>>>(D_dev, ldd, n, dmax_dev);
        KMAP_CHECK_HIP(hipMemcpyAsync(&dmax, dmax_dev, 4, hipMemcpyDeviceToHost, st));
        KMAP_CHECK_HIP(hipStreamSynchronize(st));
        const bool m8 = (uint64_t)dmax * (uint64_t)n_nb <= 255u;
        const int64_t mpitch = (n + 15) & ~(int64_t)15;
        const int64_t row_bytes = mpitch * (m8 ? 1 : 2);
        int R = (int)((150 * 1024) / row_bytes);
        if (R > 4) R = 4;
        if (rows_on && R >= 2) {
            const bool i16 = n <= 65536;
            const unsigned tb = (unsigned)((n * n_nb + 255) / 256);
            if (i16) transpose_nb_t_kernel<uint16_t><<<tb, 256, 0, st>>>(nb_dev, n, n_nb, (uint16_t *)nbT);
            else transpose_nb_t_kernel<int32_t><<<tb, 256, 0, st>>>(nb_dev, n, n_nb, nbT);
#define KMAP_KNN(RR, MT, IT) launch_knn_rows<RR, MT, IT>(D_dev, ldd, nb_dev, (const IT *)nbT, n, n_nb, row0, nrows, sums_dev, lds, mpitch, st)
#define KMAP_KNN_R(MT, IT) (R == 2 ? KMAP_KNN(2, MT, IT) : R == 3 ? KMAP_KNN(3, MT, IT) : KMAP_KNN(4, MT, IT))
            if (m8) return i16 ? KMAP_KNN_R(uint8_t, uint16_t) : KMAP_KNN_R(uint8_t, int32_t);
            return i16 ? KMAP_KNN_R(uint16_t, uint16_t) : KMAP_KNN_R(uint16_t, int32_t);
#undef KMAP_KNN_R
#undef KMAP_KNN
        }
    }
    transpose_nb_kernel<<<(unsigned)((n * n_nb + 255) / 256), 256, 0, st>>>(nb_dev, n, n_nb, nbT);
    int64_t chunk = (n + 15) & ~(int64_t)15;
    if (chunk > KNN_CHUNK_MAX) chunk = KNN_CHUNK_MAX;
    const size_t lds_bytes = (size_t)chunk * 2;
    KMAP_TRY(kmap_allow_lds((const void *)knn_sums_kernel, KNN_CHUNK_MAX * 2));
    int64_t grid = nrows < 2048 ? nrows : 2048;
    knn_sums_kernel<<<(unsigned)grid, KNN_TPB, lds_bytes, st>>>(D_dev, ldd, nb_dev, nbT, n, n_nb, row0, nrows, sums_dev,
                                                                lds, (int)chunk);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

int kmap_knn_select_u8_dev(const uint8_t *D_dev, int64_t ldd, int64_t n, int n_nb, int64_t row0, int64_t nrows,
                           int32_t *nb_out_dev, void *stream) {
    KMAP_REQUIRE(n >= 0 && nrows >= 0 && row0 >= 0 && row0 + nrows <= n && ldd >= n, "knn_select: bad sizes");
    KMAP_REQUIRE(n_nb > 0 && n_nb <= n, "knn_select: n_nb=%d must be in [1, n]", n_nb);
    if (nrows == 0) return KMAP_OK;
    KMAP_REQUIRE(D_dev && nb_out_dev, "knn_select: null pointer");
    const int aligned = (ldd % 16 == 0) && ((uintptr_t)D_dev % 16 == 0);   // 16-byte row loads (always true for kmap_hamdist_pitch)
    knn_select_kernel<<<(unsigned)((nrows + SEL_WAVES - 1) / SEL_WAVES), KMAP_WAVE * SEL_WAVES, 0, as_stream(stream)>>>(
        D_dev, ldd, n, n_nb, row0, nrows, nb_out_dev, aligned);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

int kmap_knn_smooth_f32(const float *D, const int32_t *nb, int64_t n, int n_nb, float *S_out) {
    KMAP_REQUIRE(n >= 0 && n_nb > 0, "knn_smooth_f32: bad sizes");
    if (n == 0) return KMAP_OK;
    KMAP_REQUIRE(D && nb && S_out, "knn_smooth_f32: null pointer");
    DevBuf dD, dnb, dS;
    KMAP_TRY(dD.alloc((size_t)n * n * 4));
    KMAP_TRY(dnb.alloc((size_t)n * n_nb * 4));
    KMAP_TRY(dS.alloc((size_t)n * n * 4));
    KMAP_CHECK_HIP(hipMemcpy(dD.p, D, (size_t)n * n * 4, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(dnb.p, nb, (size_t)n * n_nb * 4, hipMemcpyHostToDevice));
    knn_smooth_f32_kernel<<<(unsigned)((n * n + BLK - 1) / BLK), BLK>>>(dD.as<float>(), dnb.as<int32_t>(), n, n_nb,
                                                                         dS.as<float>());
    KMAP_CHECK_HIP(hipGetLastError());
    KMAP_CHECK_HIP(hipMemcpy(S_out, dS.p, (size_t)n * n * 4, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

// ---- session ---------------------------------------------------------------------------------
static int embed_create_impl(kmap_embed **out, int64_t n, int64_t row0, int64_t nrows, int n_best, float learning_rate,
                             int mode, int world, int rank);

int kmap_embed_create(kmap_embed **out, int64_t n, int64_t row0, int64_t nrows, int n_best, float learning_rate,
                      int mode) {
    return embed_create_impl(out, n, row0, nrows, n_best, learning_rate, mode, 1, 0);
}

int64_t kmap_embed_cyclic_blocks(int64_t n, int world, int rank) {
    if (n <= 0 || world <= 0 || rank < 0 || rank >= world) return 0;
    const int64_t nI = (n + SY_R - 1) / SY_R;
    return nI > rank ? (nI - rank + world - 1) / world : 0;
}

int kmap_embed_create_cyclic(kmap_embed **out, int64_t n, int world, int rank, int n_best, float learning_rate) {
    KMAP_REQUIRE(world >= 1 && rank >= 0 && rank < world, "embed_create_cyclic: bad world / rank");
    return embed_create_impl(out, n, 0, n, n_best, learning_rate, KMAP_EMBED_FAST, world, rank);
}

static int embed_create_impl(kmap_embed **out, int64_t n, int64_t row0, int64_t nrows, int n_best, float learning_rate,
                             int mode, int world, int rank) {
    KMAP_REQUIRE(out, "embed_create: null");
    KMAP_REQUIRE(n > 0 && row0 >= 0 && nrows >= 0 && row0 + nrows <= n, "embed_create: bad row range");
    KMAP_REQUIRE(n < ((int64_t)1 << 31) - 64, "embed_create: n too large");
    KMAP_REQUIRE(n_best > 0 && n_best <= MAX_BEST, "embed_create: n_best must be in [1,%d]", MAX_BEST);
    KMAP_REQUIRE(mode == KMAP_EMBED_FAST || mode == KMAP_EMBED_SEQ, "embed_create: unknown mode %d", mode);
    kmap_embed *e = new kmap_embed();
    e->n = n; e->row0 = row0; e->nrows = nrows; e->n_best = n_best; e->lr = learning_rate; e->mode = mode;
    {   // symmetric FAST kernel: single-GPU all-rows sessions (KMAP_EMBED_SYM=0 switches it off for A/B runs)
        const char *env = getenv("KMAP_EMBED_SYM");
        e->sym = (mode == KMAP_EMBED_FAST) && row0 == 0 && nrows == n && n >= 16384 && !(env && env[0] == '0');   // below ~16k the tall tiles leave CUs idle
        if (world > 1) e->sym = true;                       // the cyclic creator always runs the symmetric kernel
        e->symI = (n + SY_R - 1) / SY_R;
        e->symJ = (n + SY_C - 1) / SY_C;
        e->world = world;
        e->rank = rank;
        e->n_lblocks = kmap_embed_cyclic_blocks(n, world, rank);
    }
    if (mode == KMAP_EMBED_SEQ) seq_split(e);
    e->n_part = n_force_blocks(e) > 0 ? n_force_blocks(e) : 1;
    hipError_t err = hipSuccess;
    auto A = [&](void **p, size_t b) { if (err == hipSuccess) err = hipMalloc(p, b ? b : 16); };
    A((void **)&e->Y, (size_t)2 * n * 4);
    A((void **)&e->G, (size_t)2 * n * 4);
    A((void **)&e->snaps, (size_t)n_best * 2 * n * 4);
    A((void **)&e->loss_log, (size_t)e->loss_log_cap * 4);
    A((void **)&e->loss_part, (size_t)e->n_part * 8);
    A((void **)&e->loss_sum, 8);
    A((void **)&e->states, 2 * sizeof(LoopState));
    A((void **)&e->lut_dev, F_LUT_LDS * 4);
    A((void **)&e->n_normals_dev, 16);
    if (e->sym) {
        A((void **)&e->rowpart, (size_t)e->symJ * 2 * n * 4);
        A((void **)&e->colpart, (size_t)(e->n_lblocks ? e->n_lblocks : 1) * 2 * n * 4);
    }
    if (err != hipSuccess) {
        kmap_set_error("embed_create: %s", hipGetErrorString(err));
        kmap_embed_destroy(e);
        return KMAP_E_NOMEM;
    }
    LoopState s0;
    memset(&s0, 0, sizeof s0);
    s0.n_best = n_best;
    s0.prev_loss = INFINITY;
    s0.last_loss = INFINITY;
    for (int b = 0; b < MAX_BEST; ++b) {
        s0.best_loss[b] = INFINITY;
        s0.best_slot[b] = b;
    }
    s0.worst_loss = INFINITY;
    s0.worst_slot = n_best - 1;
    KMAP_CHECK_HIP(hipMemcpy(&e->states[0], &s0, sizeof s0, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(&e->states[1], &s0, sizeof s0, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemset(e->G, 0, (size_t)2 * n * 4));
    KMAP_CHECK_HIP(hipMemset(e->loss_part, 0, (size_t)e->n_part * 8));
    KMAP_CHECK_HIP(hipMemset(e->n_normals_dev, 0, 16));
    *out = e;
    return KMAP_OK;
}

int kmap_embed_destroy(kmap_embed *e) {
    if (!e) return KMAP_OK;
    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    if (e->gstream) (void)hipStreamDestroy(e->gstream);
    void *ptrs[] = {e->Y, e->G, e->snaps, e->loss_log, e->loss_part, e->loss_sum, e->states, e->lut_dev, e->normals,
                    e->rowpart, e->colpart, e->n_normals_dev};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete e;
    return KMAP_OK;
}

int kmap_embed_set_prob_f32(kmap_embed *e, const float *p_rows_dev, int64_t ld) {
    KMAP_REQUIRE(e && p_rows_dev && ld >= e->n, "embed_set_prob_f32: bad arguments");
    e->src = ProbSrc{p_rows_dev, nullptr, nullptr, ld, 0};
    e->have_prob = true;
    drop_graph(e);
    return KMAP_OK;
}

int kmap_embed_set_prob_lut(kmap_embed *e, const uint16_t *sums_rows_dev, int64_t ld, const float *lut, int lut_len) {
    KMAP_REQUIRE(e && sums_rows_dev && lut && ld >= e->n, "embed_set_prob_lut: bad arguments");
    KMAP_REQUIRE(lut_len > 0 && lut_len <= F_LUT_LDS, "embed_set_prob_lut: lut_len=%d exceeds %d", lut_len, F_LUT_LDS);
    KMAP_CHECK_HIP(hipMemcpy(e->lut_dev, lut, (size_t)lut_len * 4, hipMemcpyHostToDevice));
    e->src = ProbSrc{nullptr, sums_rows_dev, e->lut_dev, ld, lut_len};
    e->have_prob = true;
    drop_graph(e);
    return KMAP_OK;
}

int kmap_embed_set_coords(kmap_embed *e, const float *coords_2xn, const float *placeholders) {
    KMAP_REQUIRE(e && coords_2xn, "embed_set_coords: null");
    KMAP_CHECK_HIP(hipMemcpy(e->Y, coords_2xn, (size_t)2 * e->n * 4, hipMemcpyHostToDevice));
    if (placeholders)
        KMAP_CHECK_HIP(hipMemcpy(e->snaps, placeholders, (size_t)e->n_best * 2 * e->n * 4, hipMemcpyHostToDevice));
    else
        KMAP_CHECK_HIP(hipMemset(e->snaps, 0, (size_t)e->n_best * 2 * e->n * 4));
    e->have_coords = true;
    return KMAP_OK;
}

int kmap_embed_set_jitter(kmap_embed *e, const double *normals, int n_normals) {
    KMAP_REQUIRE(e && n_normals >= 0 && (n_normals == 0 || normals), "embed_set_jitter: bad arguments");
    KMAP_CHECK_HIP(hipDeviceSynchronize());
    if (n_normals > e->normals_cap) {           // grow geometrically: the pointer (a kernel argument) rarely changes
        int cap = e->normals_cap ? e->normals_cap : 8192;
        while (cap < n_normals) cap *= 2;
        if (e->normals) KMAP_CHECK_HIP(hipFree(e->normals));
        e->normals = nullptr;
        e->normals_cap = 0;
        KMAP_CHECK_HIP(hipMalloc((void **)&e->normals, (size_t)cap * 8));
        e->normals_cap = cap;
        drop_graph(e);
    }
    if (n_normals) KMAP_CHECK_HIP(hipMemcpy(e->normals, normals, (size_t)n_normals * 8, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(e->n_normals_dev, &n_normals, 4, hipMemcpyHostToDevice));
    e->n_normals = n_normals;
    return KMAP_OK;
}

namespace {
// the force kernel of the session (+ the row / column partial sum into G for the symmetric kernels when `reduce_sym`);
// loss partials go to e->loss_part
int launch_forces(kmap_embed *e, float *G, bool reduce_sym, hipStream_t st) {
    const int nblk = n_force_blocks(e);
    const bool lut = e->src.ps != nullptr;
    const size_t lds = lut ? (((size_t)e->src.lut_len * 4 + 15) & ~(size_t)15) : 16;
    if (e->sym) {
        dim3 grid((unsigned)((e->symJ + SY_WAVES - 1) / SY_WAVES), (unsigned)e->n_lblocks);
        static const bool sym2 = [] { const char *v = getenv("KMAP_EMBED_SYM2"); return !(v && v[0] == '0'); }();   // A/B switch
        if (lut && sym2) {
            const int lut_pad = (int)(lds / 4);
            const size_t lds2 = lds + ((size_t)SY_WAVES * S2_SCRATCH + (size_t)SY_WAVES * S2_CS) * 4 + SY_WAVES * 8;
            const int64_t part_ld = ((e->symJ + SY_WAVES - 1) / SY_WAVES) * SY_WAVES;   // loss partials keep the first kernel's layout
            forces_sym2_kernel<<<dim3((unsigned)e->symJ, (unsigned)e->n_lblocks), KMAP_WAVE * SY_WAVES, lds2, st>>>(
                e->src, e->Y, e->n, e->rowpart, e->colpart, e->loss_part, e->symJ, part_ld, e->world, e->rank, lut_pad);
        } else if (lut) forces_sym_kernel<true><<<grid, KMAP_WAVE * SY_WAVES, lds, st>>>(e->src, e->Y, e->n, e->rowpart, e->colpart, e->loss_part, e->symJ, e->world, e->rank);
        else forces_sym_kernel<false><<<grid, KMAP_WAVE * SY_WAVES, lds, st>>>(e->src, e->Y, e->n, e->rowpart, e->colpart, e->loss_part, e->symJ, e->world, e->rank);
        if (reduce_sym)
            sym_reduce_kernel<<<(unsigned)((2 * e->n * 8 + BLK - 1) / BLK), BLK, 0, st>>>(e->rowpart, e->colpart, e->n, e->n_lblocks, e->symJ, e->world, e->rank, G);
    } else if (e->mode == KMAP_EMBED_SEQ) {
        const int nb_main = seq_main_blocks(e), nb_tail = seq_tail_blocks(e);
        const size_t lds_w = (nb_tail ? (size_t)SQ_WAVES * KMAP_WAVE * SQ_CPL * 8 : 0) + lds;
#define KMAP_SEQ(LUT, GW)                                                                                                          \
        do {                                                                                                                       \
            KMAP_TRY(kmap_allow_lds((const void *)forces_seq_kernel<LUT, GW>, (int)lds_w));                                        \
            forces_seq_kernel<LUT, GW><<<nb_tail + nb_main, KMAP_WAVE * SQ_WAVES, lds_w, st>>>(e->src, e->Y, e->n, e->row0, e->seq_pair_rows, \
                                                                                              e->seq_main_rows, e->nrows, nb_tail,      \
                                                                                              seq_pair_blocks(e), nb_main, G, e->loss_part); \
        } while (0)
#define KMAP_SEQ_G(LUT)                                                                                       \
        do {                                                                                                  \
            const int g = nb_tail ? e->seq_tail_g : 0;                                                        \
            if (g == 0) KMAP_SEQ(LUT, 0); else if (g == 8) KMAP_SEQ(LUT, 8); else if (g == 16) KMAP_SEQ(LUT, 16); \
            else if (g == 32) KMAP_SEQ(LUT, 32); else KMAP_SEQ(LUT, 64);                                        \
        } while (0)
        if (lut) KMAP_SEQ_G(true); else KMAP_SEQ_G(false);
#undef KMAP_SEQ_G
#undef KMAP_SEQ
    } else {
        if (lut) forces_fast_kernel<true><<<nblk, KMAP_WAVE * F_WAVES, lds, st>>>(e->src, e->Y, e->n, e->row0, e->nrows, G, e->loss_part);
        else forces_fast_kernel<false><<<nblk, KMAP_WAVE * F_WAVES, lds, st>>>(e->src, e->Y, e->n, e->row0, e->nrows, G, e->loss_part);
    }
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

// one single-GPU iteration with the fused tails: forces -> apply<loss reduction fused> (2 launches), or for the symmetric
// kernels forces -> loss reduction -> partial sums + apply (3 launches instead of 4; the gradient never goes to memory)
int launch_iteration(kmap_embed *e, hipStream_t st) {
    KMAP_TRY(launch_forces(e, e->G, false, st));
    const int nblk = n_force_blocks(e);
    if (e->sym) {
        reduce_loss_kernel<<<1, RL_TPB, 0, st>>>(e->loss_part, nblk, e->loss_sum);
        sym_apply_kernel<<<(unsigned)(((2 * e->n + 4) * 8 + BLK - 1) / BLK), BLK, 0, st>>>(
            e->states, e->cur, e->Y, e->rowpart, e->colpart, e->n_lblocks, e->symJ, e->loss_sum, e->snaps, e->n, e->lr, e->normals,
            e->n_normals_dev, e->loss_log, e->loss_log_cap);
    } else {
        apply_kernel<true><<<(unsigned)((2 * e->n + BLK - 1) / BLK), BLK, 0, st>>>(e->states, e->cur, e->Y, e->G, e->loss_part, nblk, e->snaps,
                                                                                  e->n, e->lr, e->normals, e->n_normals_dev,
                                                                                  e->loss_log, e->loss_log_cap);
    }
    KMAP_CHECK_HIP(hipGetLastError());
    e->cur ^= 1;
    return KMAP_OK;
}

// capture two iterations (parities cur, cur ^ 1) into a graph; any failure switches graph replay off for the session
bool ensure_graph(kmap_embed *e) {
    if (e->graph_failed) return false;
    if (e->gexec) return true;
    // opt-in (KMAP_EMBED_GRAPH=1): measured on ROCm 7.2 / MI355X, replaying the captured pair of iterations is SLOWER than
    // launching the same 2-3 kernels directly (N = 50 k FAST 0.958 vs 0.882 ms / iteration, N = 5 k SEQ 0.175 vs 0.163, FAST 0.047 vs
    // 0.041): the launches are already queued ahead of the GPU, and the graph adds per-node dispatch cost
    static const bool on = [] { const char *v = getenv("KMAP_EMBED_GRAPH"); return v && v[0] == '1'; }();
    if (!on) { e->graph_failed = true; return false; }
    if (!e->gstream && hipStreamCreateWithFlags(&e->gstream, hipStreamNonBlocking) != hipSuccess) { e->graph_failed = true; return false; }
    hipGraph_t graph = nullptr;
    const int cur0 = e->cur;
    bool ok = hipStreamBeginCapture(e->gstream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
        ok = launch_iteration(e, e->gstream) == KMAP_OK && launch_iteration(e, e->gstream) == KMAP_OK;
        ok = (hipStreamEndCapture(e->gstream, &graph) == hipSuccess) && ok && graph;
    }
    e->cur = cur0;                                   // captured, not executed
    if (ok) ok = hipGraphInstantiate(&e->gexec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
    if (!ok) {
        (void)hipGetLastError();
        e->gexec = nullptr;
        e->graph_failed = true;
        return false;
    }
    e->graph_cur = cur0;
    return true;
}
}  // namespace

int kmap_embed_forces(kmap_embed *e, float *grad_dev_2xn, double *loss_dev, void *stream) {
    KMAP_REQUIRE(e && e->have_prob && e->have_coords, "embed_forces: probabilities/coordinates not set");
    hipStream_t st = as_stream(stream);
    float *G = grad_dev_2xn ? grad_dev_2xn : e->G;
    double *L = loss_dev ? loss_dev : e->loss_sum;
    const int nblk = n_force_blocks(e);
    if (nblk == 0) {
        KMAP_CHECK_HIP(hipMemsetAsync(L, 0, 8, st));
        return KMAP_OK;
    }
    KMAP_TRY(launch_forces(e, G, true, st));
    reduce_loss_kernel<<<1, RL_TPB, 0, st>>>(e->loss_part, nblk, L);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

int kmap_embed_apply(kmap_embed *e, const float *grad_dev_2xn, const double *loss_dev, void *stream) {
    KMAP_REQUIRE(e && e->have_coords, "embed_apply: coordinates not set");
    hipStream_t st = as_stream(stream);
    const float *G = grad_dev_2xn ? grad_dev_2xn : e->G;
    const double *L = loss_dev ? loss_dev : e->loss_sum;
    const unsigned grid = (unsigned)((2 * e->n + BLK - 1) / BLK);
    apply_kernel<false><<<grid, BLK, 0, st>>>(e->states, e->cur, e->Y, G, L, 0, e->snaps, e->n, e->lr, e->normals, e->n_normals_dev,
                                              e->loss_log, e->loss_log_cap);
    KMAP_CHECK_HIP(hipGetLastError());
    e->cur ^= 1;
    return KMAP_OK;
}

int64_t kmap_embed_msg_floats(int64_t n) { return n > 0 ? 2 * n + MSG_EXTRA : 0; }

int kmap_embed_forces_msg(kmap_embed *e, float *msg_dev, void *stream) {
    KMAP_REQUIRE(e && e->have_prob && e->have_coords, "embed_forces_msg: probabilities/coordinates not set");
    KMAP_REQUIRE(msg_dev, "embed_forces_msg: null message buffer");
    hipStream_t st = as_stream(stream);
    const int nblk = n_force_blocks(e);
    if (nblk > 0) KMAP_TRY(launch_forces(e, msg_dev, true, st));
    reduce_loss_limbs_kernel<<<1, RL_TPB, 0, st>>>(e->loss_part, nblk, msg_dev + 2 * e->n);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

int kmap_embed_apply_msg(kmap_embed *e, float *msg_dev, void *stream) {
    KMAP_REQUIRE(e && e->have_coords, "embed_apply_msg: coordinates not set");
    KMAP_REQUIRE(msg_dev, "embed_apply_msg: null message buffer");
    hipStream_t st = as_stream(stream);
    const unsigned grid = (unsigned)((2 * e->n + BLK - 1) / BLK);
    const bool clear = !e->sym && !(e->row0 == 0 && e->nrows == e->n);      // the force kernel leaves other ranks' rows alone
    if (clear)
        apply_msg_kernel<true><<<grid, BLK, 0, st>>>(e->states, e->cur, e->Y, msg_dev, e->snaps, e->n, e->lr, e->normals, e->n_normals_dev,
                                                     e->loss_log, e->loss_log_cap);
    else
        apply_msg_kernel<false><<<grid, BLK, 0, st>>>(e->states, e->cur, e->Y, msg_dev, e->snaps, e->n, e->lr, e->normals, e->n_normals_dev,
                                                      e->loss_log, e->loss_log_cap);
    KMAP_CHECK_HIP(hipGetLastError());
    e->cur ^= 1;
    return KMAP_OK;
}

int kmap_embed_step(kmap_embed *e, int n_iter, void *stream) {
    KMAP_REQUIRE(e && e->row0 == 0 && e->nrows == e->n && e->world == 1, "embed_step: single-GPU convenience needs all rows local");
    KMAP_REQUIRE(e->have_prob && e->have_coords, "embed_step: probabilities/coordinates not set");
    hipStream_t st = as_stream(stream);
    int it = 0;
    if (n_iter >= 4 && ensure_graph(e)) {
        if (e->cur != e->graph_cur && it < n_iter) {   // realign the parity the graph was captured at
            KMAP_TRY(launch_iteration(e, st));
            ++it;
        }
        for (; it + 2 <= n_iter; it += 2) {
            if (hipGraphLaunch(e->gexec, st) != hipSuccess) {   // e.g. a stream the runtime cannot launch graphs into
                (void)hipGetLastError();
                drop_graph(e);
                e->graph_failed = true;
                break;
            }
        }
    }
    for (; it < n_iter; ++it) KMAP_TRY(launch_iteration(e, st));
    return KMAP_OK;
}

int kmap_embed_state(kmap_embed *e, int64_t *iters, int *stopped, float *last_loss, float *best_loss, int *jitter_used,
                     void *stream) {
    KMAP_REQUIRE(e, "embed_state: null");
    KMAP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    LoopState s;
    KMAP_CHECK_HIP(hipMemcpy(&s, &e->states[e->cur], sizeof s, hipMemcpyDeviceToHost));
    if (iters) *iters = s.iters;
    if (stopped) *stopped = s.stopped;
    if (last_loss) *last_loss = s.last_loss;
    if (best_loss) *best_loss = s.best_loss[0];
    if (jitter_used) *jitter_used = s.jitter_used;
    return KMAP_OK;
}

int kmap_embed_get_coords(kmap_embed *e, float *coords_2xn, void *stream) {
    KMAP_REQUIRE(e && coords_2xn, "embed_get_coords: null");
    KMAP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    KMAP_CHECK_HIP(hipMemcpy(coords_2xn, e->Y, (size_t)2 * e->n * 4, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

int kmap_embed_get_best(kmap_embed *e, float *coords_2xn, void *stream) {
    KMAP_REQUIRE(e && coords_2xn, "embed_get_best: null");
    KMAP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    LoopState s;
    KMAP_CHECK_HIP(hipMemcpy(&s, &e->states[e->cur], sizeof s, hipMemcpyDeviceToHost));
    KMAP_CHECK_HIP(hipMemcpy(coords_2xn, e->snaps + (size_t)s.best_slot[0] * 2 * e->n, (size_t)2 * e->n * 4,
                             hipMemcpyDeviceToHost));   // best_res_list[0][1] (visualization.py:325)
    return KMAP_OK;
}

int kmap_embed_get_losses(kmap_embed *e, float *losses, int64_t max_n, int64_t *n_out, void *stream) {
    KMAP_REQUIRE(e && n_out, "embed_get_losses: null");
    KMAP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    LoopState s;
    KMAP_CHECK_HIP(hipMemcpy(&s, &e->states[e->cur], sizeof s, hipMemcpyDeviceToHost));
    int64_t m = s.iters < e->loss_log_cap ? s.iters : e->loss_log_cap;
    if (m > max_n) m = max_n;
    if (m > 0 && losses) KMAP_CHECK_HIP(hipMemcpy(losses, e->loss_log, (size_t)m * 4, hipMemcpyDeviceToHost));
    *n_out = m;
    return KMAP_OK;
}

void *kmap_embed_coords_dev(kmap_embed *e) { return e ? (void *)e->Y : nullptr; }

// ---- drop-in L3 float operators (host pointers, blocking) -----------------------------------------
}  // extern "C"

namespace {
__global__ __launch_bounds__(BLK) void ld_prob_kernel(const float *__restrict__ Y, int64_t n, float *__restrict__ Q) {
    const int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= n * n) return;
    const int64_t i = t / n, j = t % n;
    // np.ones then the kernel fills i != j; clip applies to the diagonal's 1.0 too (visualization.py:251-255)
    Q[t] = (i == j) ? 0.999f : q_of(Y[i] - Y[j], Y[n + i] - Y[n + j]);
}
__global__ __launch_bounds__(BLK) void ce_rows_kernel(const float *__restrict__ P, const float *__restrict__ Q, int64_t n,
                                                      double *__restrict__ part) {
    __shared__ double sh[BLK];
    double s = 0.0;
    const int64_t i = blockIdx.x;
    for (int64_t j = i + 1 + threadIdx.x; j < n; j += BLK) {
        float q = Q[i * n + j];
        const float eps = 1e-10f;
        q = (q < eps) ? eps : ((q > 1.0f - eps) ? 1.0f - eps : q);   // taichi_core.py:283-288
        s += (double)ce_of<true>(P[i * n + j], q);
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = BLK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[i] = sh[0];
}
__global__ __launch_bounds__(KMAP_WAVE) void grad_rows_kernel(const float *__restrict__ P, const float *__restrict__ Q,
                                                              const float *__restrict__ Y, int64_t n,
                                                              float *__restrict__ G) {
    const int64_t i = (int64_t)blockIdx.x * KMAP_WAVE + threadIdx.x;
    if (i >= n) return;
    const float xi = Y[i], yi = Y[n + i];
    float gx = 0.0f, gy = 0.0f;
    for (int64_t j = 0; j < n; ++j) {
        if (j == i) continue;
        const float t = t_of(P[i * n + j], Q[i * n + j]);
        gx = gx + t * (xi - Y[j]);
        gy = gy + t * (yi - Y[n + j]);
    }
    G[i] = 4.0f * gx;
    G[n + i] = 4.0f * gy;
}
}  // namespace

extern "C" {

int kmap_ld_prob_mat_f32(const float *ld_2xn, int64_t n, float *q_out_nxn) {
    KMAP_REQUIRE(n >= 0, "ld_prob_mat: n<0");
    if (n == 0) return KMAP_OK;
    KMAP_REQUIRE(ld_2xn && q_out_nxn, "ld_prob_mat: null pointer");
    DevBuf dy, dq;
    KMAP_TRY(dy.alloc((size_t)2 * n * 4));
    KMAP_TRY(dq.alloc((size_t)n * n * 4));
    KMAP_CHECK_HIP(hipMemcpy(dy.p, ld_2xn, (size_t)2 * n * 4, hipMemcpyHostToDevice));
    ld_prob_kernel<<<(unsigned)((n * n + BLK - 1) / BLK), BLK>>>(dy.as<float>(), n, dq.as<float>());
    KMAP_CHECK_HIP(hipGetLastError());
    KMAP_CHECK_HIP(hipMemcpy(q_out_nxn, dq.p, (size_t)n * n * 4, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

int kmap_cross_entropy_f32(const float *p_nxn, const float *q_nxn, int64_t n, float *loss_out) {
    KMAP_REQUIRE(n >= 0 && loss_out, "cross_entropy: bad arguments");
    *loss_out = 0.0f;
    if (n == 0) return KMAP_OK;
    KMAP_REQUIRE(p_nxn && q_nxn, "cross_entropy: null pointer");
    DevBuf dp, dq, dpart, dsum;
    KMAP_TRY(dp.alloc((size_t)n * n * 4));
    KMAP_TRY(dq.alloc((size_t)n * n * 4));
    KMAP_TRY(dpart.alloc((size_t)n * 8));
    KMAP_TRY(dsum.alloc(8));
    KMAP_CHECK_HIP(hipMemcpy(dp.p, p_nxn, (size_t)n * n * 4, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(dq.p, q_nxn, (size_t)n * n * 4, hipMemcpyHostToDevice));
    ce_rows_kernel<<<(unsigned)n, BLK>>>(dp.as<float>(), dq.as<float>(), n, dpart.as<double>());
    reduce_loss_kernel<<<1, RL_TPB>>>(dpart.as<double>(), (int)n, dsum.as<double>());
    KMAP_CHECK_HIP(hipGetLastError());
    double s = 0.0;
    KMAP_CHECK_HIP(hipMemcpy(&s, dsum.p, 8, hipMemcpyDeviceToHost));
    *loss_out = (float)(2.0 * s);
    return KMAP_OK;
}

int kmap_gradient_loss_f32(const float *p_nxn, const float *q_nxn, const float *ld_2xn, int64_t n, float *grad_out_2xn) {
    KMAP_REQUIRE(n >= 0, "gradient_loss: n<0");
    if (n == 0) return KMAP_OK;
    KMAP_REQUIRE(p_nxn && q_nxn && ld_2xn && grad_out_2xn, "gradient_loss: null pointer");
    DevBuf dp, dq, dy, dg;
    KMAP_TRY(dp.alloc((size_t)n * n * 4));
    KMAP_TRY(dq.alloc((size_t)n * n * 4));
    KMAP_TRY(dy.alloc((size_t)2 * n * 4));
    KMAP_TRY(dg.alloc((size_t)2 * n * 4));
    KMAP_CHECK_HIP(hipMemcpy(dp.p, p_nxn, (size_t)n * n * 4, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(dq.p, q_nxn, (size_t)n * n * 4, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(dy.p, ld_2xn, (size_t)2 * n * 4, hipMemcpyHostToDevice));
    grad_rows_kernel<<<(unsigned)((n + KMAP_WAVE - 1) / KMAP_WAVE), KMAP_WAVE>>>(dp.as<float>(), dq.as<float>(),
                                                                                 dy.as<float>(), n, dg.as<float>());
    KMAP_CHECK_HIP(hipGetLastError());
    KMAP_CHECK_HIP(hipMemcpy(grad_out_2xn, dg.p, (size_t)2 * n * 4, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

}  // extern "C"
