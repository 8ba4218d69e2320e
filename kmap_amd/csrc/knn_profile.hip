// knn_profile.hip -- neighbour sums of the smoothing step straight from the k-mers (knn_smooth, reference
// visualization.py:90-109 + taichi_core.py:227-249: 400 gathers of D per pair).
//
//   sums[i,j] = sum_{a in nb[i]} sum_{b in nb[j]} D[a,b],  D[a,b] = mismatching bases of k-mers a and b
//             = n_nb^2 k - sum_{position p, base x} cnt_i[p][x] cnt_j[p][x]
// where cnt_i[p][x] counts the neighbours of i that have base x at position p: the double sum over neighbour pairs is a dot
// product of two 4k-entry count profiles (one dword per position: four byte counters), v_dot4_u32_u8 does four entries per
// instruction.  The short-consensus rule of the matrix (pairs sharing a label whose consensus is shorter than k are compared
// on its first clen bases only, reference motif_discovery.py:789-800) subtracts, per such label g, the tail mismatches of
// the neighbour pairs that both carry g: tail_g c_i c_j - <tailprofile_i, tailprofile_j>.  Exact integers throughout; the
// result equals kmap_knn_sums_u8_dev on the matrix kmap_hamdist_matrix_* writes -- without reading the matrix.
#include <stdlib.h>

#include "common.h"

namespace {
constexpr int KP_NG = 4;          // short-consensus labels handled here (more: the matrix-based kernel)
// columns per lane: 8 at k <= 8 (one 16-byte non-temporal store of uint16 sums per row; 64 registers of column profiles), 4 at
// k <= 16 (8-byte store; the 16-dword profiles of 8 columns would take 128 registers)
constexpr int KP_ROWS = 32;       // rows per wave (8: the wave's 64 profile dwords per lane were loaded for 720 instructions of work -- 23 % VALU utilisation)
constexpr int KP_WAVES = 4;

struct GroupTab {
    int32_t clen[KP_NG];
    int n;
};

// profile dword (i, p): byte x = number of neighbours of i with base x at position p; group profiles only count neighbours
// with that group id and only tail positions p >= clen_g; cg[i][g] = neighbours of i in group g
template <typename H>
__global__ __launch_bounds__(256) void knn_profile_kernel(const H *__restrict__ kh, const uint8_t *__restrict__ gid,
                                                          const int32_t *__restrict__ nb, int64_t n, int k, int kd, int n_nb,
                                                          GroupTab gt, uint32_t *__restrict__ V, uint32_t *__restrict__ Vg,
                                                          uint8_t *__restrict__ cg, uint32_t *__restrict__ GA, uint32_t *__restrict__ GB) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * kd) return;
    const int64_t i = t / kd;
    const int p = (int)(t % kd);
    uint32_t v = 0, vg[KP_NG] = {0, 0, 0, 0}, c[KP_NG] = {0, 0, 0, 0};
    if (p < k) {
        for (int a = 0; a < n_nb; ++a) {
            const int64_t r = nb[i * n_nb + a];
            const uint32_t x = (uint32_t)((kh[r] >> (2 * (k - 1 - p))) & 3);
            const uint32_t one = 1u << (8 * x);
            v += one;
            const int g = gid[r];
            if (g > 0 && g <= gt.n) {
                if (p >= gt.clen[g - 1]) vg[g - 1] += one;
                c[g - 1] += 1;
            }
        }
    }
    V[t] = v;
#pragma unroll
    for (int g = 0; g < KP_NG; ++g) {
        if (g < gt.n) {
            Vg[((int64_t)g * n + i) * kd + p] = vg[g];
            if (p == 0) cg[(int64_t)g * n + i] = (uint8_t)c[g];
            if (GA) {   // MFMA form (knn_sums_mfma_kernel): signed bytes, the rank-one term tail * c_i * c_j folded into unused slots
                uint32_t ga = 0, gb = vg[g];
                if (p < gt.clen[g]) {                               // positions in front of the tail carry no group counts
                    const int tail = k - gt.clen[g];
                    for (int x = 0; x < 4; ++x)
                        if (4 * p + x < tail) ga |= (c[g] & 255u) << (8 * x);
                    gb = ga;
                } else {
                    for (int x = 0; x < 4; ++x) ga |= ((0u - ((vg[g] >> (8 * x)) & 255u)) & 255u) << (8 * x);
                }
                GA[((int64_t)g * n + i) * kd + p] = ga;
                GB[((int64_t)g * n + i) * kd + p] = gb;
            }
        }
    }
}

// Block = 64 * KP_CPL columns x KP_WAVES * KP_ROWS rows; a wave keeps its columns' profiles in registers and walks its rows.
// The short-consensus correction needs, per group, the columns' group profiles and neighbour counts: staged once per block in LDS
// ([group][column of the lane][16-byte half][lane]: conflict-free 16-byte reads) and in two packed registers per group -- read
// from global memory inside the row loop (a byte and, where non-zero, eight dwords per column and row with neighbours in the
// group) they made those rows five times as expensive as the others.  A column without neighbours in the group has an all-zero
// group profile and count, so the correction is applied to every column of such a row without a per-lane test.
template <int KD, int KP_CPL>
__global__ __launch_bounds__(KMAP_WAVE *KP_WAVES) void knn_sums_profile_kernel(const uint32_t *__restrict__ V,
                                                                              const uint32_t *__restrict__ Vg,
                                                                              const uint8_t *__restrict__ cg, int64_t n, int k,
                                                                              int n_nb, GroupTab gt, int64_t row0, int64_t nrows,
                                                                              uint16_t *__restrict__ T, int64_t ldt, int zero_diag) {
    extern __shared__ uint4 kp_lds[];                           // [g][c][h][lane]
    constexpr int QH = KD / 4;                                  // 16-byte pieces per profile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t jw0 = (int64_t)blockIdx.x * KMAP_WAVE * KP_CPL;   // first column of the block
    const int64_t j0 = jw0 + (int64_t)lane * KP_CPL;
    const int64_t r0 = ((int64_t)blockIdx.y * KP_WAVES + wave) * KP_ROWS;
    for (int t = threadIdx.x; t < gt.n * KP_CPL * QH * 64; t += KMAP_WAVE * KP_WAVES) {
        const int l = t & 63, h = (t >> 6) % QH, c = (t >> 6) / QH % KP_CPL, g = (t >> 6) / QH / KP_CPL;
        int64_t j = jw0 + (int64_t)l * KP_CPL + c;
        j = j < n ? j : n - 1;
        kp_lds[t] = *reinterpret_cast<const uint4 *>(Vg + ((int64_t)g * n + j) * KD + 4 * h);
    }
    uint32_t cjp[KP_NG][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};  // the columns' neighbour counts per group, a byte each
#pragma unroll
    for (int g = 0; g < KP_NG; ++g)
        if (g < gt.n) {
#pragma unroll
            for (int c = 0; c < KP_CPL; ++c) {
                const int64_t j = (j0 + c < n) ? j0 + c : n - 1;
                cjp[g][c >> 2] |= (uint32_t)cg[(int64_t)g * n + j] << (8 * (c & 3));
            }
        }
    __syncthreads();
    if (r0 >= nrows) return;                                   // wave-uniform
    const bool full = j0 + KP_CPL <= n;
    uint32_t vj[KP_CPL][KD];
#pragma unroll
    for (int c = 0; c < KP_CPL; ++c) {
        const int64_t j = (j0 + c < n) ? j0 + c : n - 1;
#pragma unroll
        for (int p = 0; p < KD; ++p) vj[c][p] = V[j * KD + p];
    }
    const uint32_t base = (uint32_t)(n_nb * n_nb * k);
    for (int r = 0; r < KP_ROWS; ++r) {
        const int64_t lr = r0 + r;
        if (lr >= nrows) break;
        const int64_t i = row0 + lr;
        uint32_t vi[KD];
#pragma unroll
        for (int p = 0; p < KD; ++p) vi[p] = V[i * KD + p];    // wave-uniform address -> scalar loads
        uint32_t s[KP_CPL];
#pragma unroll
        for (int c = 0; c < KP_CPL; ++c) {
            uint32_t d = 0;
#pragma unroll
            for (int p = 0; p < KD; ++p) d = __builtin_amdgcn_udot4(vi[p], vj[c][p], d, false);
            s[c] = base - d;
        }
#pragma unroll
        for (int g = 0; g < KP_NG; ++g) {                      // rows with neighbours in a short-consensus group
            if (g >= gt.n) break;
            const uint32_t ci = cg[(int64_t)g * n + i];
            if (ci == 0) continue;                             // wave-uniform
            const uint32_t tc = (uint32_t)(k - gt.clen[g]) * ci;
            const uint32_t *Vgi = Vg + ((int64_t)g * n + i) * KD;
            uint32_t gi[KD];
#pragma unroll
            for (int p = 0; p < KD; ++p) gi[p] = Vgi[p];       // wave-uniform
#pragma unroll
            for (int c = 0; c < KP_CPL; ++c) {
                uint32_t d = 0;
#pragma unroll
                for (int h = 0; h < QH; ++h) {
                    const uint4 q = kp_lds[((g * KP_CPL + c) * QH + h) * 64 + lane];
                    d = __builtin_amdgcn_udot4(gi[4 * h], q.x, d, false);
                    d = __builtin_amdgcn_udot4(gi[4 * h + 1], q.y, d, false);
                    d = __builtin_amdgcn_udot4(gi[4 * h + 2], q.z, d, false);
                    d = __builtin_amdgcn_udot4(gi[4 * h + 3], q.w, d, false);
                }
                const uint32_t cj = (cjp[g][c >> 2] >> (8 * (c & 3))) & 255u;
                s[c] -= tc * cj - d;                           // cj = 0: the column's group profile is all zero, d = 0
            }
        }
        if (zero_diag && i >= jw0 && i < jw0 + KMAP_WAVE * KP_CPL) {   // the diagonal crosses this block's columns (wave-uniform)
#pragma unroll
            for (int c = 0; c < KP_CPL; ++c)
                if (j0 + c == i) s[c] = 0;                     // diagonal forced to 0 (visualization.py:103,107)
        }
        uint16_t *dst = T + lr * ldt + j0;
        if (full && ((ldt & (KP_CPL - 1)) == 0)) {
            if constexpr (KP_CPL == 8) {       // write-once streaming output: 16 bytes per lane, non-temporal
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 o = {(s[0] & 0xFFFFu) | (s[1] << 16), (s[2] & 0xFFFFu) | (s[3] << 16), (s[4] & 0xFFFFu) | (s[5] << 16),
                                 (s[6] & 0xFFFFu) | (s[7] << 16)};
                __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(dst));
            } else {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2 *>(dst) = u32x2{(s[0] & 0xFFFFu) | (s[1] << 16), (s[2] & 0xFFFFu) | (s[3] << 16)};
            }
        } else {
#pragma unroll
            for (int c = 0; c < KP_CPL; ++c)
                if (j0 + c < n) dst[c] = (uint16_t)s[c];
        }
    }
}

// ---- the same sums on the matrix cores --------------------------------------------------------------------------------------------
// <profile_i, profile_j> is a dot product of 4 k byte counters (<= n_nb <= 127: signed bytes): one v_mfma_i32_32x32x32_i8 per 32 x 32
// tile of pairs at k <= 8 (K = 32 bytes = the whole profile), two at k <= 16.  A lane holds 16 bytes of a row k-mer's profile (A) and
// 16 of a column k-mer's (B), the same byte range [16 h, 16 h + 16) of the MFMA's K in both (h = lane >> 5), so the pairing of
// bytes is the dot product's whatever order the hardware walks K in.  The short-consensus correction rides along as one more
// MFMA per group on group profiles prepared by knn_profile_kernel: tail bytes negated on the A side (- <tail_i, tail_j>) and the
// group's neighbour counts c_i / c_j written into `tail` unused byte slots in front of the tail (+ tail c_i c_j).  i32 accumulators:
// exact.  The 32 x 32 results have their column on the lane; a wave collects a 32-row x 256-column strip in LDS (uint16, row
// pitch 528 B: rows r and r + 4 of an accumulator register on different banks) and stores it as 16 bytes per lane, 512 B per row.
// 1.0 ms at N = 50 000 against 1.7 (3.2 before the LDS-staged corrections) for the v_dot4 kernel, whose 64 dot instructions per
// row and lane are the bound; this one is a store stream.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
constexpr int KM_COLS = 256, KM_PITCH = KM_COLS * 2 + 16, KM_WAVES = 4, KM_STRIPS = 4;
template <int KD>
__global__ __launch_bounds__(KMAP_WAVE *KM_WAVES) void knn_sums_mfma_kernel(const uint32_t *__restrict__ V, const uint32_t *__restrict__ GA,
                                                                           const uint32_t *__restrict__ GB, int n_groups, int64_t n,
                                                                           uint32_t base, int64_t row0, int64_t nrows,
                                                                           uint16_t *__restrict__ T, int64_t ldt, int zero_diag) {
    extern __shared__ uint4 km_lds[];
    constexpr int NK = KD / 8, NT = KM_COLS / 32;               // MFMAs per profile (32 bytes of K each), tiles per strip
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int64_t lr0 = ((int64_t)blockIdx.y * KM_WAVES + wave) * 32;   // the wave's 32 rows (local)
    if (lr0 >= nrows) return;                                   // wave-uniform; no block-wide barrier below
    char *strip = reinterpret_cast<char *>(km_lds) + (size_t)wave * 32 * KM_PITCH;
    int64_t i = row0 + lr0 + r;
    i = i < n ? i : n - 1;
    i32x4 a[NK], ag[KP_NG][NK];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) a[kk] = *reinterpret_cast<const i32x4 *>(V + i * KD + kk * 8 + 4 * h);
#pragma unroll
    for (int g = 0; g < KP_NG; ++g)
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
            ag[g][kk] = g < n_groups ? *reinterpret_cast<const i32x4 *>(GA + ((int64_t)g * n + i) * KD + kk * 8 + 4 * h) : i32x4{0, 0, 0, 0};
    const bool wide = (ldt & 7) == 0 && (reinterpret_cast<uintptr_t>(T) & 15) == 0;
    const int64_t i0 = row0 + lr0;
    // the wave walks KM_STRIPS strips of 256 columns with the same rows; the column operands of strip s + 1 are requested before
    // strip s is written out (one strip per wave: 1.11 ms at N = 50 000, the operand round trip exposed at 8 waves per CU)
    i32x4 b[NT][NK], bn[NT][NK];
    auto load_strip = [&](int64_t cs, i32x4 (&dst)[NT][NK]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int64_t j = cs + 32 * t + r;                        // tiles behind column n compute on clamped columns; never written out
            j = j < n ? j : n - 1;
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) dst[t][kk] = *reinterpret_cast<const i32x4 *>(V + j * KD + kk * 8 + 4 * h);
        }
    };
    const int64_t c_first = (int64_t)blockIdx.x * KM_STRIPS * KM_COLS;
    load_strip(c_first, b);
    for (int s = 0; s < KM_STRIPS; ++s) {
        const int64_t c0 = c_first + (int64_t)s * KM_COLS;
        if (c0 >= n) break;                                     // wave-uniform
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int64_t jt = c0 + 32 * t;
            i32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[kk], b[t][kk], acc, 0, 0, 0);
            if (n_groups) {                                     // short-consensus groups: rare enough to fetch their operands here
                int64_t j = jt + r;
                j = j < n ? j : n - 1;
#pragma unroll
                for (int g = 0; g < KP_NG; ++g) {
                    if (g >= n_groups) break;
#pragma unroll
                    for (int kk = 0; kk < NK; ++kk) {
                        const i32x4 bg = *reinterpret_cast<const i32x4 *>(GB + ((int64_t)g * n + j) * KD + kk * 8 + 4 * h);
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ag[g][kk], bg, acc, 0, 0, 0);
                    }
                }
            }
            const bool on_diag = zero_diag && i0 < jt + 32 && jt < i0 + 32;  // wave-uniform
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * h;  // C / D layout of the 32 x 32 forms: column = lane & 31
                uint32_t val = base - (uint32_t)acc[q];
                if (on_diag && i0 + row == jt + r) val = 0;     // diagonal forced to 0 (visualization.py:103,107)
                *reinterpret_cast<uint16_t *>(strip + row * KM_PITCH + (32 * t + r) * 2) = (uint16_t)val;
            }
        }
        if (s + 1 < KM_STRIPS) load_strip(c0 + KM_COLS, bn);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the strip is this wave's own: LDS executes a wave's operations in order
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = 2 * it + h;
            const int64_t lr = lr0 + row, j = c0 + 8 * r;
            if (lr >= nrows || j >= n) continue;
            const uint4 q = *reinterpret_cast<const uint4 *>(strip + row * KM_PITCH + r * 16);
            uint16_t *dst = T + lr * ldt + j;
            if (wide && j + 8 <= n) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(u32x4{q.x, q.y, q.z, q.w}, reinterpret_cast<u32x4 *>(dst));
            } else {
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
                for (int e = 0; e < 8; ++e)
                    if (j + e < n) dst[e] = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) b[t][kk] = bn[t][kk];
    }
}

__global__ void kp_gid_kernel(const int32_t *__restrict__ label, int64_t n, const uint8_t *__restrict__ lab2gid, int n_lab,
                              uint8_t *__restrict__ gid) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t l = label[i];
    gid[i] = (l >= 0 && l < n_lab) ? lab2gid[l] : 0;
}

template <typename H>
int knn_sums_kmers(const H *kh_dev, const int32_t *label_dev, int64_t n, int k, const int32_t *clen, int n_lab,
                   const int32_t *nb_dev, int n_nb, int64_t row0, int64_t nrows, uint16_t *sums_dev, int64_t lds, void *stream) {
    // KMAP_KNN_NATURAL_DIAG (kmap_hip.h): S[i][i] keeps the value the formula gives it instead of the reference's 0
    const int zero_diag = (n_nb & KMAP_KNN_NATURAL_DIAG) ? 0 : 1;
    n_nb &= ~KMAP_KNN_NATURAL_DIAG;
    KMAP_REQUIRE(n >= 0 && nrows >= 0 && row0 >= 0 && row0 + nrows <= n, "knn_sums_kmers: bad row range");
    KMAP_REQUIRE(n_nb > 0 && n_nb <= 255 && lds >= n, "knn_sums_kmers: bad n_nb / leading dimension");
    KMAP_REQUIRE(n_lab >= 0 && n_lab <= 255 && (n_lab == 0 || clen), "knn_sums_kmers: bad label table");
    if (k < 1 || k > 16 || (int64_t)n_nb * n_nb * k > 65535) {
        kmap_set_error("knn_sums_kmers: k=%d, n_nb=%d outside the profile kernel's range (use the matrix-based entry point)", k, n_nb);
        return KMAP_E_UNSUP;
    }
    uint8_t lab2gid[256];
    memset(lab2gid, 0, sizeof lab2gid);
    GroupTab gt;
    gt.n = 0;
    for (int l = 0; l < n_lab; ++l) {
        KMAP_REQUIRE(clen[l] > 0 && clen[l] <= k, "knn_sums_kmers: clen[%d]=%d not in (0,k]", l, clen[l]);
        if (clen[l] < k) {
            if (gt.n == KP_NG) {
                kmap_set_error("knn_sums_kmers: more than %d short consensuses (use the matrix-based entry point)", KP_NG);
                return KMAP_E_UNSUP;
            }
            gt.clen[gt.n] = clen[l];
            lab2gid[l] = (uint8_t)(++gt.n);
        }
    }
    if (n == 0 || nrows == 0) return KMAP_OK;
    KMAP_REQUIRE(kh_dev && label_dev && nb_dev && sums_dev, "knn_sums_kmers: null pointer");
    hipStream_t st = as_stream(stream);
    const int kd = (k <= 8) ? 8 : 16;
    // matrix-core form: byte counters must be signed bytes, and the rank-one term of every short group needs `tail` free byte slots
    // (otherwise -- more than 127 neighbours, or a consensus so short that its tail does not fit -- the v_dot4 kernel above)
    bool mfma = n_nb <= 127;
    for (int g = 0; g < gt.n; ++g) mfma = mfma && (k - gt.clen[g]) <= 4 * gt.clen[g];
    // scratch: V [n][kd] u32 | Vg [g][n][kd] u32 | GA, GB [g][n][kd] u32 (MFMA form) | cg [g][n] u8 | gid [n] u8 | lab2gid table
    const size_t nV = (size_t)n * kd, ng = (size_t)(gt.n ? gt.n : 1);
    void *buf = nullptr;
    KMAP_TRY(kmap_scratch(&buf, nV * 4 + 3 * ng * nV * 4 + ng * (size_t)n + (size_t)n + 256 + 64, st, KMAP_SLOT_D));
    uint32_t *V = (uint32_t *)buf, *Vg = V + nV, *GA = Vg + ng * nV, *GB = GA + ng * nV;
    uint8_t *cg = (uint8_t *)(GB + ng * nV), *gid = cg + ng * (size_t)n, *tab = gid + n;
    KMAP_CHECK_HIP(hipMemcpyAsync(tab, lab2gid, 256, hipMemcpyHostToDevice, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));                  // lab2gid is a stack buffer
    kp_gid_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(label_dev, n, tab, n_lab, gid);
    knn_profile_kernel<H><<<(unsigned)((nV + 255) / 256), 256, 0, st>>>(kh_dev, gid, nb_dev, n, k, kd, n_nb, gt, V, Vg, cg, mfma ? GA : nullptr,
                                                                        mfma ? GB : nullptr);
    if (mfma) {
        const dim3 grid((unsigned)((n + KM_STRIPS * KM_COLS - 1) / (KM_STRIPS * KM_COLS)), (unsigned)((nrows + 32 * KM_WAVES - 1) / (32 * KM_WAVES)));
        KMAP_REQUIRE(grid.y <= 65535u, "knn_sums_kmers: nrows too large for one launch");
        const size_t lds_b = (size_t)KM_WAVES * 32 * KM_PITCH;
        const uint32_t base = (uint32_t)(n_nb * n_nb * k);
        if (kd == 8) {
            KMAP_TRY(kmap_allow_lds((const void *)knn_sums_mfma_kernel<8>, (int)lds_b));
            knn_sums_mfma_kernel<8><<<grid, KMAP_WAVE * KM_WAVES, lds_b, st>>>(V, GA, GB, gt.n, n, base, row0, nrows, sums_dev, lds, zero_diag);
        } else {
            KMAP_TRY(kmap_allow_lds((const void *)knn_sums_mfma_kernel<16>, (int)lds_b));
            knn_sums_mfma_kernel<16><<<grid, KMAP_WAVE * KM_WAVES, lds_b, st>>>(V, GA, GB, gt.n, n, base, row0, nrows, sums_dev, lds, zero_diag);
        }
        KMAP_CHECK_HIP(hipGetLastError());
        return KMAP_OK;
    }
    const int cpl = (kd == 8 && ((uintptr_t)sums_dev % 16) == 0) ? 8 : 4;
    const dim3 grid((unsigned)((n + KMAP_WAVE * cpl - 1) / (KMAP_WAVE * cpl)),
                    (unsigned)((nrows + KP_ROWS * KP_WAVES - 1) / (KP_ROWS * KP_WAVES)));
    KMAP_REQUIRE(grid.y <= 65535u, "knn_sums_kmers: nrows too large for one launch");
    const size_t lds_b = (size_t)gt.n * cpl * (kd / 4) * 64 * 16;   // <= 64 KiB (four groups)
    if (kd == 8 && cpl == 8) knn_sums_profile_kernel<8, 8><<<grid, KMAP_WAVE * KP_WAVES, lds_b, st>>>(V, Vg, cg, n, k, n_nb, gt, row0, nrows, sums_dev, lds, zero_diag);
    else if (kd == 8) knn_sums_profile_kernel<8, 4><<<grid, KMAP_WAVE * KP_WAVES, lds_b, st>>>(V, Vg, cg, n, k, n_nb, gt, row0, nrows, sums_dev, lds, zero_diag);
    else knn_sums_profile_kernel<16, 4><<<grid, KMAP_WAVE * KP_WAVES, lds_b, st>>>(V, Vg, cg, n, k, n_nb, gt, row0, nrows, sums_dev, lds, zero_diag);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}
}  // namespace

extern "C" {
int kmap_knn_sums_kmers_u32_dev(const uint32_t *kh_dev, const int32_t *label_dev, int64_t n, int k, const int32_t *clen, int n_lab,
                                const int32_t *nb_dev, int n_nb, int64_t row0, int64_t nrows, uint16_t *sums_dev, int64_t lds,
                                void *stream) {
    return knn_sums_kmers<uint32_t>(kh_dev, label_dev, n, k, clen, n_lab, nb_dev, n_nb, row0, nrows, sums_dev, lds, stream);
}
int kmap_knn_sums_kmers_u64_dev(const uint64_t *kh_dev, const int32_t *label_dev, int64_t n, int k, const int32_t *clen, int n_lab,
                                const int32_t *nb_dev, int n_nb, int64_t row0, int64_t nrows, uint16_t *sums_dev, int64_t lds,
                                void *stream) {
    return knn_sums_kmers<uint64_t>(kh_dev, label_dev, n, k, clen, n_lab, nb_dev, n_nb, row0, nrows, sums_dev, lds, stream);
}
}
