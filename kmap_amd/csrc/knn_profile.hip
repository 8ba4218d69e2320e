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
#include "common.h"

namespace {
constexpr int KP_NG = 4;          // short-consensus labels handled here (more: the matrix-based kernel)
// columns per lane: 8 at k <= 8 (one 16-byte non-temporal store of uint16 sums per row; 64 registers of column profiles), 4 at
// k <= 16 (8-byte store; the 16-dword profiles of 8 columns would take 128 registers)
constexpr int KP_ROWS = 8;        // rows per wave
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
                                                          uint8_t *__restrict__ cg) {
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
        }
    }
}

template <int KD, int KP_CPL>
__global__ __launch_bounds__(KMAP_WAVE *KP_WAVES) void knn_sums_profile_kernel(const uint32_t *__restrict__ V,
                                                                              const uint32_t *__restrict__ Vg,
                                                                              const uint8_t *__restrict__ cg, int64_t n, int k,
                                                                              int n_nb, GroupTab gt, int64_t row0, int64_t nrows,
                                                                              uint16_t *__restrict__ T, int64_t ldt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j0 = ((int64_t)blockIdx.x * KMAP_WAVE + lane) * KP_CPL;
    const int64_t r0 = ((int64_t)blockIdx.y * KP_WAVES + wave) * KP_ROWS;
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
        for (int g = 0; g < gt.n; ++g) {                       // rows with neighbours in a short-consensus group (rare)
            const uint32_t ci = cg[(int64_t)g * n + i];
            if (ci == 0) continue;                             // wave-uniform
            const uint32_t tail = (uint32_t)(k - gt.clen[g]);
            const uint32_t *Vgi = Vg + ((int64_t)g * n + i) * KD;
#pragma unroll
            for (int c = 0; c < KP_CPL; ++c) {
                const int64_t j = (j0 + c < n) ? j0 + c : n - 1;
                const uint32_t cj = cg[(int64_t)g * n + j];
                if (cj == 0) continue;
                const uint32_t *Vgj = Vg + ((int64_t)g * n + j) * KD;
                uint32_t d = 0;
#pragma unroll
                for (int p = 0; p < KD; ++p) d = __builtin_amdgcn_udot4(Vgi[p], Vgj[p], d, false);
                s[c] -= tail * ci * cj - d;
            }
        }
#pragma unroll
        for (int c = 0; c < KP_CPL; ++c)
            if (j0 + c == i) s[c] = 0;                         // diagonal forced to 0 (visualization.py:103,107)
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
    // scratch: V [n][kd] u32 | Vg [g][n][kd] u32 | cg [g][n] u8 | gid [n] u8 | lab2gid table
    const size_t nV = (size_t)n * kd, ng = (size_t)(gt.n ? gt.n : 1);
    void *buf = nullptr;
    KMAP_TRY(kmap_scratch(&buf, nV * 4 + ng * nV * 4 + ng * (size_t)n + (size_t)n + 256 + 64, st, KMAP_SLOT_D));
    uint32_t *V = (uint32_t *)buf, *Vg = V + nV;
    uint8_t *cg = (uint8_t *)(Vg + ng * nV), *gid = cg + ng * (size_t)n, *tab = gid + n;
    KMAP_CHECK_HIP(hipMemcpyAsync(tab, lab2gid, 256, hipMemcpyHostToDevice, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));                  // lab2gid is a stack buffer
    kp_gid_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(label_dev, n, tab, n_lab, gid);
    knn_profile_kernel<H><<<(unsigned)((nV + 255) / 256), 256, 0, st>>>(kh_dev, gid, nb_dev, n, k, kd, n_nb, gt, V, Vg, cg);
    const int cpl = (kd == 8 && ((uintptr_t)sums_dev % 16) == 0) ? 8 : 4;
    const dim3 grid((unsigned)((n + KMAP_WAVE * cpl - 1) / (KMAP_WAVE * cpl)),
                    (unsigned)((nrows + KP_ROWS * KP_WAVES - 1) / (KP_ROWS * KP_WAVES)));
    KMAP_REQUIRE(grid.y <= 65535u, "knn_sums_kmers: nrows too large for one launch");
    if (kd == 8 && cpl == 8) knn_sums_profile_kernel<8, 8><<<grid, KMAP_WAVE * KP_WAVES, 0, st>>>(V, Vg, cg, n, k, n_nb, gt, row0, nrows, sums_dev, lds);
    else if (kd == 8) knn_sums_profile_kernel<8, 4><<<grid, KMAP_WAVE * KP_WAVES, 0, st>>>(V, Vg, cg, n, k, n_nb, gt, row0, nrows, sums_dev, lds);
    else knn_sums_profile_kernel<16, 4><<<grid, KMAP_WAVE * KP_WAVES, 0, st>>>(V, Vg, cg, n, k, n_nb, gt, row0, nrows, sums_dev, lds);
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
