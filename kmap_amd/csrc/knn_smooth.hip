// knn_smooth.hip -- neighbour selection and neighbour sums of the smoothing step from a GIVEN matrix (knn_smooth, reference
// visualization.py:90-109 + taichi_core.py:227-249); the pipeline itself takes the sums from the k-mers (knn_profile.hip).
//
//  * knn select : 20 smallest entries per uint8 row (threshold value + index order), wave per row.
//  * knn sums   : sums[i,j] = sum_{a in nb[i], b in nb[j]} D[a,b] as exact integers (u16), computed as
//                 A*D*A^T in two factored steps through LDS (20+20 reads per pair instead of 400).
//  * knn_smooth_f32: the generic float operator in the reference's summation order.
#include <stdlib.h>

#include "embed_internal.h"

namespace {
constexpr int BLK = EMB_BLK;

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
    // Both passes keep the chunks of the next TWO steps in flight (unconditional loads on clamped chunk indices; r04: one
    // dependent 16-byte load per step and wave, 49 exposed round trips per pass, was the whole kernel -- 71 % of the wave-cycles
    // waiting at 35 % issue utilisation).
    const int last_chunk = lane_chunks - 1;
    auto chunk_at = [&](int s) { const int c = s * 64 + lane; return row4[c < last_chunk ? c : last_chunk]; };
    uint4 pf0 = chunk_at(0), pf1 = chunk_at(1);
    for (int s = 0; s < nsteps; ++s) {
        const int chunk = s * 64 + lane;
        uint4 w = pf0;
        pf0 = pf1;
        pf1 = chunk_at(s + 2);
        if (chunk >= lane_chunks) w = make_uint4(~0u, ~0u, ~0u, ~0u);
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
    pf0 = chunk_at(0);
    pf1 = chunk_at(1);
    for (int s = 0; s < nsteps && written < (uint32_t)n_nb; ++s) {
        const int chunk = s * 64 + lane;
        uint4 w = pf0;
        pf0 = pf1;
        pf1 = chunk_at(s + 2);
        if (chunk >= lane_chunks) w = make_uint4(~0u, ~0u, ~0u, ~0u);       // 0xFF bytes: neither < t nor == t (t < 32)
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

// ---- one pass over the row (n_nb <= SEL1_CAP, 16-byte aligned pitch, values < 32: every Hamming row of the product) -------------
// The two-pass kernel above reads the row once for the value histogram and again, up to the last selected entry, for the indices
// (r04: 1.54 x N^2 bytes, 16 lane-private LDS atomics per 16 bytes in pass 1).  Here the wave keeps the selection of the entries
// SEEN SO FAR while it reads the row once:
//   * state: a bound T (an upper bound of the value of the row's n_nb-th smallest entry), per value v <= T a list of the indices
//     taken (LDS, <= n_nb each, in index order) and its length cnt[v] (lane v of a vector register; cnt[T] in a scalar),
//     below = sum of cnt[v < T].  Always below < n_nb and below + cnt[T] <= n_nb, with equality once n_nb entries <= T were seen;
//   * an entry == T enters while below + cnt[T] < n_nb (a later one never does: the ones held have lower indices); an entry < T
//     enters, and if the selection was full the last entry of T's list leaves; when that list is empty and below == n_nb the bound
//     drops to the largest value that has entries;
//   * so once the selection is full only entries BELOW the bound matter: per 1024 entries four SWAR compares and a ballot;
//   * the first bound comes from the value counters of SEL1_BOOT steps of the row (8192 entries around the diagonal, re-read from
//     L2 later): a sample of the row that holds n_nb entries <= T0 proves the row does.  The tighter the first bound, the fewer
//     entries ever take the scalar insertion path (N = 50 000, k = 8: T0 = 2 against a final 1 or 2).
// A sample without n_nb entries below 32 (never a Hamming row of k < 32) sends the row to the generic two-pass path; larger bytes
// elsewhere in the row are just never candidates.
constexpr int SEL1_CAP = 32;
constexpr int SEL1_BOOT = 8;
constexpr int SEL1_WAVES = 4;
__global__ __launch_bounds__(KMAP_WAVE *SEL1_WAVES) void knn_select1_kernel(const uint8_t *__restrict__ D, int64_t ldd, int64_t n, int n_nb,
                                                                             int64_t row0, int64_t nrows, int32_t *__restrict__ nb) {
    __shared__ uint32_t bins[SEL1_WAVES][SEL_VALS * 64];     // the lists (32 x SEL1_CAP words) re-use a wave's counters after the bound is known
    static_assert(32 * SEL1_CAP <= SEL_VALS * 64, "lists alias the counters");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t lr = (int64_t)blockIdx.x * SEL1_WAVES + wave;
    if (lr >= nrows) return;
    uint32_t *h = bins[wave];
    int32_t *lists = reinterpret_cast<int32_t *>(bins[wave]);
    const uint8_t *row = D + (row0 + lr) * ldd;
    int32_t *out = nb + lr * n_nb;
    const uint4 *row4 = reinterpret_cast<const uint4 *>(row);
    const int nsteps = (int)((n + 1023) >> 10), nfull = (int)(n >> 10);
    const int lane_chunks = (int)(ldd >> 4), last_chunk = lane_chunks - 1;
    auto chunk_at = [&](int s) { const int c = s * 64 + lane; return row4[c < last_chunk ? c : last_chunk]; };
    auto valid16 = [&](int s, int chunk) -> uint32_t {                       // bytes of the lane's chunk that lie inside the row
        if (s < nfull) return 0xFFFFu;
        const int64_t left = n - (int64_t)chunk * 16;
        return left >= 16 ? 0xFFFFu : left <= 0 ? 0u : (1u << (int)left) - 1u;
    };
    // ---- the first bound: value counters of the first steps
#pragma unroll
    for (int v = 0; v < SEL_VALS; ++v) h[v * 64 + lane] = 0;
    __builtin_amdgcn_wave_barrier();
    // the sample: SEL1_BOOT steps around the row's own column -- in the pipeline's matrices (k-mers grouped by label, copies of a
    // sampled k-mer adjacent) a row's nearest entries sit around its diagonal, so the first bound is close to the final one; any
    // window of the row would be valid
    const int nboot = nsteps < SEL1_BOOT ? nsteps : SEL1_BOOT;
    int s_first = (int)((row0 + lr) >> 10) - SEL1_BOOT / 2;
    s_first = s_first < 0 ? 0 : (s_first + nboot > nsteps ? nsteps - nboot : s_first);
    for (int s = s_first; s < s_first + nboot; ++s) {
        const int chunk = s * 64 + lane;
        uint4 w = chunk_at(s);
        if (chunk >= lane_chunks) w = make_uint4(~0u, ~0u, ~0u, ~0u);
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        const uint32_t ok = valid16(s, chunk);
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            uint32_t v = (ws[b >> 2] >> (8 * (b & 3))) & 0xFFu;
            v = v < 32u ? v : 32u;
            if ((ok >> b) & 1u) atomicAdd(&h[v * 64 + lane], 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    uint32_t tot = 0;
    if (lane < SEL_VALS) {
        const uint4 *p = reinterpret_cast<const uint4 *>(h + lane * 64);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint4 c = p[q];
            tot += (c.x + c.y) + (c.z + c.w);
        }
    }
    uint32_t bigflag = (uint32_t)__builtin_amdgcn_readlane((int)tot, 32);   // some byte >= 32
    uint32_t cum = lane < 32 ? tot : 0u;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const uint32_t up = __shfl_up(cum, o);
        if (lane >= o) cum += up;
    }
    const unsigned long long reach = __ballot(lane < 32 && cum >= (uint32_t)n_nb);
    if (reach == 0ull) bigflag = 1u;             // fewer than n_nb entries below 32 in the sample
    __builtin_amdgcn_wave_barrier();             // the counters are dead: their words become the lists
    // ---- state
    int T = bigflag ? 0 : __builtin_ctzll(reach);
    uint32_t below = 0, cntT = 0;                // entries held with value < T / == T
    uint32_t cntv = 0;                           // lane v < T: entries held with value v
    // one candidate entry (index j, value v), in index order
    auto take = [&](int v, int32_t j) {
        if (v > T) return;                                                   // the bound dropped inside this step
        if (v == T) {
            if (below + cntT < (uint32_t)n_nb) {
                if (lane == 0) lists[T * SEL1_CAP + (int)cntT] = j;
                ++cntT;
            }
            return;
        }
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cntv, v);
        if (lane == 0) lists[v * SEL1_CAP + (int)c] = j;                     // c <= below < n_nb <= SEL1_CAP
        cntv = lane == v ? c + 1u : cntv;
        ++below;
        if (below + cntT > (uint32_t)n_nb) --cntT;                           // the selection was full: the last entry == T leaves (cntT > 0 here)
        if (cntT == 0 && below == (uint32_t)n_nb) {                          // all n_nb below the bound: it drops to the largest value held
            const unsigned long long held = __ballot(cntv != 0u && lane < T);
            const int Tn = 63 - __builtin_clzll(held);
            cntT = (uint32_t)__builtin_amdgcn_readlane((int)cntv, Tn);
            cntv = lane == Tn ? 0u : cntv;
            below -= cntT;
            T = Tn;
        }
    };
    uint4 pf0 = chunk_at(0), pf1 = chunk_at(1), pf2 = chunk_at(2);
    for (int s = 0; s < nsteps && bigflag == 0u; ++s) {
        const int chunk = s * 64 + lane;
        uint4 w = pf0;
        pf0 = pf1; pf1 = pf2; pf2 = chunk_at(s + 3);
        if (chunk >= lane_chunks) w = make_uint4(~0u, ~0u, ~0u, ~0u);       // 0xFF bytes: never candidates
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        // candidates: bytes <= T while the selection is not full, bytes < T afterwards
        const uint32_t TT = (uint32_t)(T + (below + cntT < (uint32_t)n_nb ? 1 : 0)) * 0x01010101u;
        // (bytes >= 32 behind the sample are simply never candidates: the bound is below 32; `& ~x` keeps bytes >= 128, whose
        // borrow would read as "below", out -- one v_bitop3 with the complement and the mask)
        uint32_t lt[4], any = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            lt[d] = ~((ws[d] | 0x80808080u) - TT) & ~ws[d] & 0x80808080u;
            any |= lt[d];
        }
        const uint32_t ok = s < nfull ? 0xFFFFu : valid16(s, chunk);         // the row's last step: bytes behind the row do not count
        if (__ballot(any != 0u) == 0ull) continue;                           // wave-uniform: no candidate in these 1024 entries
        uint32_t m16 = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) m16 |= swar_pack4(lt[d]) << (4 * d);
        m16 &= ok;
        unsigned long long cand = __ballot(m16 != 0u);
        while (cand) {                                                       // scalar: lanes in order, bytes in order = index order
            const int L = __builtin_ctzll(cand);
            cand &= cand - 1;
            uint32_t mL = (uint32_t)__builtin_amdgcn_readlane((int)m16, L);
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)ws[0], L), w1 = (uint32_t)__builtin_amdgcn_readlane((int)ws[1], L);
            const uint32_t w2 = (uint32_t)__builtin_amdgcn_readlane((int)ws[2], L), w3 = (uint32_t)__builtin_amdgcn_readlane((int)ws[3], L);
            const int32_t base = (int32_t)(((int64_t)s * 64 + L) * 16);
            while (mL) {
                const int b = __builtin_ctz(mL);
                mL &= mL - 1;
                const uint32_t word = (b < 8) ? (b < 4 ? w0 : w1) : (b < 12 ? w2 : w3);
                take((int)((word >> (8 * (b & 3))) & 0xFFu), base + b);
            }
        }
    }
    if (bigflag) {
        __builtin_amdgcn_wave_barrier();
        knn_select_row_generic(row, n, n_nb, h, lane, out);
        return;
    }
    // ---- write the lists: lane v copies the entries of value v (lane T: cntT of them) behind those of the smaller values
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const uint32_t mine = lane < T ? cntv : (lane == T ? cntT : 0u);
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
    }
    const uint32_t at = inc - mine;
    for (uint32_t i = 0; i < (uint32_t)n_nb; ++i)
        if (i < mine && lane < 32) out[at + i] = lists[lane * SEL1_CAP + (int)i];
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

// ---- repeated rows: the host-side neighbour choice (np.argpartition) partitions a repeated row once -------------------------------
// fresh[lr] = 1 if row row0 + lr differs from the row above it somewhere in its n bytes (lr = 0: always 1); one block per row
__global__ __launch_bounds__(256) void rows_fresh_kernel(const uint8_t *__restrict__ D, int64_t ldd, int64_t n, int64_t row0,
                                                         uint8_t *__restrict__ fresh) {
    const int64_t lr = blockIdx.x;
    int diff = lr == 0;
    if (lr > 0) {
        const uint8_t *a = D + (row0 + lr) * ldd, *b = a - ldd;
        int64_t done = 0;
        if ((((uintptr_t)a | (uintptr_t)b) & 15) == 0) {
            const int64_t nv = n / 16;
            const u32x4 *av = reinterpret_cast<const u32x4 *>(a), *bv = reinterpret_cast<const u32x4 *>(b);
            for (int64_t t = threadIdx.x; t < nv; t += blockDim.x) {
                const u32x4 x = av[t] ^ bv[t];
                diff |= (x.x | x.y | x.z | x.w) != 0u;
            }
            done = nv * 16;
        }
        for (int64_t t = done + threadIdx.x; t < n; t += blockDim.x) diff |= a[t] != b[t];
    }
    const int any = __syncthreads_or(diff);
    if (threadIdx.x == 0) fresh[lr] = any ? 1 : 0;
}
// out row o = row idx[o] of D (n bytes each); one block per output row
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint8_t *__restrict__ D, int64_t ldd, int64_t n, const int32_t *__restrict__ idx,
                                                          uint8_t *__restrict__ out, int64_t ldo) {
    const int64_t o = blockIdx.x;
    const uint8_t *a = D + (int64_t)idx[o] * ldd;
    uint8_t *b = out + o * ldo;
    int64_t done = 0;
    if ((((uintptr_t)a | (uintptr_t)b) & 15) == 0) {
        const int64_t nv = n / 16;
        for (int64_t t = threadIdx.x; t < nv; t += blockDim.x) reinterpret_cast<u32x4 *>(b)[t] = reinterpret_cast<const u32x4 *>(a)[t];
        done = nv * 16;
    }
    for (int64_t t = done + threadIdx.x; t < n; t += blockDim.x) b[t] = a[t];
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
        if (R >= 2) {
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
    // rows too long for two of them in LDS (n > 75 k at byte sums, 37 k at 16-bit sums): one row per block, in column chunks
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
    static const bool two_pass = [] { const char *v = getenv("KMAP_KNN_SELECT"); return v && v[0] == '2'; }();   // A/B switch: the two-pass kernel
    static const bool one_pass = [] { const char *v = getenv("KMAP_KNN_SELECT"); return v && v[0] == '1'; }();
    // r05, N = 50 000, ms per launch (tools/probes/knn_select_only.py), one pass / two passes: unique sorted k-mers 0.60 / 0.74, a
    // synthetic count-expanded sample 0.61 / 0.68, the C3 hand-over itself 0.74 / 0.66 -- half of its rows are motif k-mers whose 20
    // neighbours are label mates at the start of the row, where the two-pass kernel's second pass ends after a few steps.  The
    // product's matrices are hand-overs: the two-pass kernel stays the default, the one-pass kernel is KMAP_KNN_SELECT=1.
    if (aligned && n_nb <= SEL1_CAP && !two_pass && one_pass) {
        knn_select1_kernel<<<(unsigned)((nrows + SEL1_WAVES - 1) / SEL1_WAVES), KMAP_WAVE * SEL1_WAVES, 0, as_stream(stream)>>>(
            D_dev, ldd, n, n_nb, row0, nrows, nb_out_dev);
        KMAP_CHECK_HIP(hipGetLastError());
        return KMAP_OK;
    }
    knn_select_kernel<<<(unsigned)((nrows + SEL_WAVES - 1) / SEL_WAVES), KMAP_WAVE * SEL_WAVES, 0, as_stream(stream)>>>(
        D_dev, ldd, n, n_nb, row0, nrows, nb_out_dev, aligned);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

int kmap_rows_fresh_u8_dev(const uint8_t *D_dev, int64_t ldd, int64_t n, int64_t row0, int64_t nrows, uint8_t *fresh_dev, void *stream) {
    KMAP_REQUIRE(n >= 0 && nrows >= 0 && row0 >= 0 && ldd >= n, "rows_fresh: bad sizes");
    if (nrows == 0) return KMAP_OK;
    KMAP_REQUIRE(D_dev && fresh_dev, "rows_fresh: null pointer");
    KMAP_REQUIRE(nrows < ((int64_t)1 << 31), "rows_fresh: too many rows for one launch");
    rows_fresh_kernel<<<(unsigned)nrows, 256, 0, as_stream(stream)>>>(D_dev, ldd, n, row0, fresh_dev);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}
int kmap_gather_rows_u8_dev(const uint8_t *D_dev, int64_t ldd, int64_t n, const int32_t *idx_dev, int64_t n_idx, uint8_t *out_dev,
                            int64_t ldo, void *stream) {
    KMAP_REQUIRE(n >= 0 && n_idx >= 0 && ldd >= n && ldo >= n, "gather_rows: bad sizes");
    if (n_idx == 0) return KMAP_OK;
    KMAP_REQUIRE(D_dev && idx_dev && out_dev, "gather_rows: null pointer");
    KMAP_REQUIRE(n_idx < ((int64_t)1 << 31), "gather_rows: too many rows for one launch");
    gather_rows_kernel<<<(unsigned)n_idx, 256, 0, as_stream(stream)>>>(D_dev, ldd, n, idx_dev, out_dev, ldo);
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

}  // extern "C"
