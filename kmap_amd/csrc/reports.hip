// reports.hip -- consumers of the counted k-mers / the occurrence hit list:
//   * Hamming-ball extraction + position count matrix (reference motif_discovery.py:924-986: ex_hamball_kh_arr, cal_cnt_mat)
//   * motif position density (reference motif_discovery.py:1255-1327: get_motif_pos_density)
// Integer results are exact; the density is f64 with the reference's operation order per term (no FMA: the library is
// built with -ffp-contract=off) and a fixed, chunked summation order over reads.
#include <vector>

#include "common.h"
#include "counts_internal.h"
#include "scan_util.h"

namespace {
constexpr int HB_TPB = 256;
constexpr int HB_ITEMS = 4;

// bit0: inside the ball, bit1: the reverse complement is strictly closer (the member is re-oriented)
template <typename H>
__device__ __forceinline__ uint32_t ball_flags(H h, H c, H rc, H mask, int radius, int revcom) {
    int d = popc2((H)((h ^ c) & mask));
    uint32_t f = 0;
    if (revcom) {
        const int drc = popc2((H)((h ^ rc) & mask));
        if (drc < d) {
            f = 2;
            d = drc;
        }
    }
    return (d <= radius) ? (f | 1u) : 0u;
}

template <typename H>
__global__ __launch_bounds__(HB_TPB) void hamball_count_kernel(const H *__restrict__ uniq, int64_t n, H c, H rc, int k,
                                                               int radius, int revcom, uint32_t *__restrict__ block_counts) {
    __shared__ uint32_t wsum[HB_TPB / 64];
    const H mask = low_mask<H>(k);
    const int64_t x0 = ((int64_t)blockIdx.x * HB_TPB + threadIdx.x) * HB_ITEMS;
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < HB_ITEMS; ++j)
        if (x0 + j < n) m += ball_flags<H>(uniq[x0 + j], c, rc, mask, radius, revcom) & 1u;
    for (int o = 32; o > 0; o >>= 1) m += __shfl_down(m, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ordered write of the ball members (re-oriented where flagged) + per-position base counts weighted by the k-mer counts
template <typename H, typename CT>
__global__ __launch_bounds__(HB_TPB) void hamball_write_kernel(const H *__restrict__ uniq, const CT *__restrict__ cnt, int64_t n,
                                                               H c, H rc, int k, int radius, int revcom,
                                                               const uint64_t *__restrict__ block_off, H *__restrict__ out_kh,
                                                               CT *__restrict__ out_cnt, unsigned long long *__restrict__ cnt_mat) {
    __shared__ uint32_t wsum[HB_TPB / 64];
    __shared__ unsigned long long mat[4 * 32];
    if (threadIdx.x < 128) mat[threadIdx.x] = 0;
    const H mask = low_mask<H>(k);
    const int64_t x0 = ((int64_t)blockIdx.x * HB_TPB + threadIdx.x) * HB_ITEMS;
    H keys[HB_ITEMS];
    CT cnts[HB_ITEMS];
    uint32_t flags = 0, m = 0;
#pragma unroll
    for (int j = 0; j < HB_ITEMS; ++j) {
        keys[j] = 0;
        cnts[j] = 0;
        if (x0 + j < n) {
            const H h = uniq[x0 + j];
            const uint32_t f = ball_flags<H>(h, c, rc, mask, radius, revcom);
            if (f & 1u) {
                keys[j] = (f & 2u) ? revcom_hash(h, k) : h;
                cnts[j] = cnt[x0 + j];
                flags |= 1u << j;
                ++m;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = m;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    uint64_t pos = block_off[blockIdx.x] + woff + (inc - m);
#pragma unroll
    for (int j = 0; j < HB_ITEMS; ++j) {
        if (flags & (1u << j)) {
            out_kh[pos] = keys[j];
            out_cnt[pos] = cnts[j];
            ++pos;
            for (int p = 0; p < k; ++p) {   // base at position p (first base most significant)
                const int b = (int)((keys[j] >> (2 * (k - 1 - p))) & 3);
                atomicAdd(&mat[b * 32 + p], (unsigned long long)(long long)cnts[j]);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 128 && (threadIdx.x & 31) < k && mat[threadIdx.x])
        atomicAdd(&cnt_mat[(threadIdx.x >> 5) * k + (threadIdx.x & 31)], mat[threadIdx.x]);
}

// ball of `conseq_kh` over n (k-mer, count) entries that are ON THE DEVICE: members (re-oriented where flagged) in table order ->
// ou / oc (device), base counts weighted by the k-mer counts -> mat (device, 4 x 32 slots), number of members -> total
template <typename H, typename CT>
int hamball_core(const H *u_dev, const CT *c_dev, int64_t n, int k, uint64_t conseq_kh, int radius, int revcom, DevBuf &ou, DevBuf &oc,
                 DevBuf &mat, uint64_t *total_out) {
    hipStream_t st = nullptr;
    KMAP_TRY(mat.alloc(4 * 32 * 8));
    KMAP_CHECK_HIP(hipMemset(mat.p, 0, 4 * 32 * 8));
    const unsigned nb = (unsigned)((n + HB_TPB * HB_ITEMS - 1) / (HB_TPB * HB_ITEMS));
    uint32_t *bc = nullptr;
    uint64_t *boff = nullptr;
    KMAP_TRY(kmap_scratch((void **)&bc, (size_t)nb * 4, st, KMAP_SLOT_A));
    KMAP_TRY(kmap_scratch((void **)&boff, ((size_t)nb + 1) * 8, st, KMAP_SLOT_B));
    const H ch = (H)conseq_kh;
    // host-side reverse complement of the consensus (same arithmetic as the device helper)
    H rc = 0;
    {
        const H com = (H)(low_mask<H>(k) - ch);
        for (int p = 0; p < k; ++p) rc = (H)((rc << 2) | ((com >> (2 * p)) & 3));
    }
    hamball_count_kernel<H><<<nb, HB_TPB, 0, st>>>(u_dev, n, ch, rc, k, radius, revcom, bc);
    KMAP_TRY(exclusive_scan_u32(bc, nb, boff, st));
    uint64_t total = 0;
    KMAP_CHECK_HIP(hipMemcpyAsync(&total, boff + nb, 8, hipMemcpyDeviceToHost, st));
    KMAP_CHECK_HIP(hipStreamSynchronize(st));
    KMAP_TRY(ou.alloc((size_t)total * sizeof(H)));
    KMAP_TRY(oc.alloc((size_t)total * sizeof(CT)));
    hamball_write_kernel<H, CT><<<nb, HB_TPB, 0, st>>>(u_dev, c_dev, n, ch, rc, k, radius, revcom, boff, ou.as<H>(), oc.as<CT>(),
                                                       mat.as<unsigned long long>());
    KMAP_CHECK_HIP(hipGetLastError());
    *total_out = total;
    return KMAP_OK;
}

template <typename H, typename CT>
int hamball_extract(const void *uniq_host, const void *cnt_host, int64_t n, int k, uint64_t conseq_kh, int radius, int revcom,
                    void *out_kh, void *out_cnt, int64_t *n_out, int64_t *cnt_mat) {
    DevBuf u, c, ou, oc, mat;
    KMAP_TRY(u.alloc((size_t)n * sizeof(H)));
    KMAP_TRY(c.alloc((size_t)n * sizeof(CT)));
    KMAP_CHECK_HIP(hipMemcpy(u.p, uniq_host, (size_t)n * sizeof(H), hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(c.p, cnt_host, (size_t)n * sizeof(CT), hipMemcpyHostToDevice));
    uint64_t total = 0;
    KMAP_TRY((hamball_core<H, CT>(u.as<H>(), c.as<CT>(), n, k, conseq_kh, radius, revcom, ou, oc, mat, &total)));
    KMAP_CHECK_HIP(hipMemcpy(out_kh, ou.p, (size_t)total * sizeof(H), hipMemcpyDeviceToHost));
    KMAP_CHECK_HIP(hipMemcpy(out_cnt, oc.p, (size_t)total * sizeof(CT), hipMemcpyDeviceToHost));
    if (cnt_mat) KMAP_CHECK_HIP(hipMemcpy(cnt_mat, mat.p, (size_t)4 * k * 8, hipMemcpyDeviceToHost));
    *n_out = (int64_t)total;
    return KMAP_OK;
}

// the same over the table a counts handle still holds in HBM (uint32 counts whatever k); the members' counts reach the host in the
// reference's dtype (int32 for k < 16, int64 otherwise).  cap < members: nothing is written, *n_out says how many there are.
template <typename H, typename CT>
int hamball_resident(kmap_counts *c, uint64_t conseq_kh, int radius, int revcom, int64_t cap, void *out_kh, void *out_cnt,
                     int64_t *n_out, int64_t *cnt_mat) {
    DevBuf ou, oc, mat;
    uint64_t total = 0;
    KMAP_TRY((hamball_core<H, uint32_t>((const H *)c->uniq, c->cnt, c->n_uniq, c->k, conseq_kh, radius, revcom, ou, oc, mat, &total)));
    *n_out = (int64_t)total;
    if ((int64_t)total > cap) return KMAP_OK;
    if (total) {
        KMAP_CHECK_HIP(hipMemcpy(out_kh, ou.p, (size_t)total * sizeof(H), hipMemcpyDeviceToHost));
        // counts: uint32 on the device -> the caller's int32 (same bits) / int64 (widened in place, from the back: the uint32 values
        // are first copied into the low half of the caller's own buffer -- no host temporary, whatever the radius)
        char *bytes = (char *)out_cnt;
        KMAP_CHECK_HIP(hipMemcpy(bytes, oc.p, (size_t)total * 4, hipMemcpyDeviceToHost));
        if (sizeof(CT) == 8) {
            for (size_t i = (size_t)total; i-- > 0;) {                       // entry i's 8 bytes cover the uint32 entries 2 i, 2 i + 1 >= i
                uint32_t v;
                memcpy(&v, bytes + 4 * i, 4);
                const int64_t w = (int64_t)v;
                memcpy(bytes + 8 * i, &w, 8);
            }
        }
    }
    if (cnt_mat) KMAP_CHECK_HIP(hipMemcpy(cnt_mat, mat.p, (size_t)4 * c->k * 8, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

// ---- position density ------------------------------------------------------------------------------------------
constexpr int PD_TPB = 128;
constexpr int PD_READS = 256;   // reads per block: one partial density row per block, summed in block order afterwards

__global__ __launch_bounds__(PD_TPB) void pos_density_kernel(const int32_t *__restrict__ hits, const int64_t *__restrict__ offs,
                                                             const int32_t *__restrict__ pos, const int64_t *__restrict__ seq_len,
                                                             int64_t n_seq, int kmer_len, const double *__restrict__ x_arr, int nx,
                                                             double scale, double *__restrict__ partial) {
    const int64_t r0 = (int64_t)blockIdx.x * PD_READS;
    const int64_t r1 = (r0 + PD_READS < n_seq) ? r0 + PD_READS : n_seq;
    for (int xi = threadIdx.x; xi < nx; xi += PD_TPB) {
        const double x = x_arr[xi];
        double acc = 0.0;
        for (int64_t r = r0; r < r1; ++r) {
            const int m = hits[r];
            if (m <= 0) continue;
            const double denom = (double)seq_len[r] - (double)kmer_len + 1.0;   // float(seq_len) - kmer_len + 1
            const int64_t o = offs[r];
            double s = 0.0;
            for (int i = 0; i < m; ++i) {
                const double rel = ((double)pos[o + i] + 0.0) / denom;
                const double z = (x - rel) / scale;
                s += exp(-(z * z) / 2.0) / 2.5066282746310002 / scale;     // scipy norm.pdf: exp(-z**2/2)/sqrt(2*pi), /scale
            }
            acc += s / (double)m;
        }
        partial[(size_t)blockIdx.x * nx + xi] = acc;
    }
}

__global__ void pos_density_reduce_kernel(const double *__restrict__ partial, int64_t n_blocks, int nx, double *__restrict__ out) {
    const int xi = blockIdx.x * blockDim.x + threadIdx.x;
    if (xi >= nx) return;
    double acc = 0.0;
    for (int64_t b = 0; b < n_blocks; ++b) acc += partial[(size_t)b * nx + xi];
    out[xi] = acc;
}
}  // namespace

extern "C" {

int kmap_hamball_extract(const void *uniq_host, const void *cnt_host, int64_t n, int k, uint64_t conseq_kh, int max_ham_dist,
                         int revcom_mode, void *out_kh, void *out_cnt, int64_t *n_out, int64_t *cnt_mat) {
    KMAP_REQUIRE(k > 0 && k < 32, "hamball_extract: kmer_len %d outside 1..31", k);
    KMAP_REQUIRE(n >= 0 && n_out && (n == 0 || (uniq_host && cnt_host && out_kh && out_cnt)), "hamball_extract: null argument");
    *n_out = 0;
    if (cnt_mat) memset(cnt_mat, 0, (size_t)4 * k * 8);
    if (n == 0) return KMAP_OK;
    if (k < 16)
        return hamball_extract<uint32_t, int32_t>(uniq_host, cnt_host, n, k, conseq_kh, max_ham_dist, revcom_mode, out_kh, out_cnt,
                                                  n_out, cnt_mat);
    return hamball_extract<uint64_t, int64_t>(uniq_host, cnt_host, n, k, conseq_kh, max_ham_dist, revcom_mode, out_kh, out_cnt,
                                              n_out, cnt_mat);
}

int kmap_counts_hamball_extract(kmap_counts *c, uint64_t conseq_kh, int max_ham_dist, int revcom_mode, int64_t cap, void *out_kh,
                                void *out_cnt, int64_t *n_out, int64_t *cnt_mat) {
    KMAP_REQUIRE(c && c->k > 0 && n_out, "counts_hamball_extract: nothing counted yet / null output");
    KMAP_REQUIRE(cap >= 0 && (cap == 0 || (out_kh && out_cnt)), "counts_hamball_extract: null output arrays");
    *n_out = 0;
    if (cnt_mat) memset(cnt_mat, 0, (size_t)4 * c->k * 8);
    if (c->n_uniq == 0) return KMAP_OK;
    if (c->narrow) return hamball_resident<uint32_t, int32_t>(c, conseq_kh, max_ham_dist, revcom_mode, cap, out_kh, out_cnt, n_out, cnt_mat);
    return hamball_resident<uint64_t, int64_t>(c, conseq_kh, max_ham_dist, revcom_mode, cap, out_kh, out_cnt, n_out, cnt_mat);
}

int kmap_pos_density(const int32_t *hits, const int64_t *offs, const int32_t *pos, const int64_t *seq_len, int64_t n_seq,
                     int kmer_len, const double *x_arr, int nx, double x_step, double *density) {
    KMAP_REQUIRE(n_seq >= 0 && nx > 0 && x_arr && density && x_step > 0.0, "pos_density: bad arguments");
    KMAP_REQUIRE(n_seq == 0 || (hits && offs && seq_len), "pos_density: null argument");
    for (int i = 0; i < nx; ++i) density[i] = 0.0;
    if (n_seq == 0) return KMAP_OK;
    const int64_t total = offs[n_seq];
    KMAP_REQUIRE(total == 0 || pos, "pos_density: null positions");
    for (int64_t r = 0; r < n_seq; ++r)   // shapes are checked on the host before any launch
        KMAP_REQUIRE(hits[r] >= 0 && offs[r + 1] - offs[r] == hits[r], "pos_density: offs[%lld] does not match hits", (long long)r);
    DevBuf dh, doff, dpos, dlen, dx, dpart, dout;
    const int64_t nb = (n_seq + PD_READS - 1) / PD_READS;
    KMAP_TRY(dh.alloc((size_t)n_seq * 4));
    KMAP_TRY(doff.alloc((size_t)(n_seq + 1) * 8));
    KMAP_TRY(dpos.alloc((size_t)total * 4));
    KMAP_TRY(dlen.alloc((size_t)n_seq * 8));
    KMAP_TRY(dx.alloc((size_t)nx * 8));
    KMAP_TRY(dpart.alloc((size_t)nb * nx * 8));
    KMAP_TRY(dout.alloc((size_t)nx * 8));
    KMAP_CHECK_HIP(hipMemcpy(dh.p, hits, (size_t)n_seq * 4, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(doff.p, offs, (size_t)(n_seq + 1) * 8, hipMemcpyHostToDevice));
    if (total) KMAP_CHECK_HIP(hipMemcpy(dpos.p, pos, (size_t)total * 4, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(dlen.p, seq_len, (size_t)n_seq * 8, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(dx.p, x_arr, (size_t)nx * 8, hipMemcpyHostToDevice));
    pos_density_kernel<<<(unsigned)nb, PD_TPB>>>(dh.as<int32_t>(), doff.as<int64_t>(), dpos.as<int32_t>(), dlen.as<int64_t>(), n_seq,
                                                 kmer_len, dx.as<double>(), nx, x_step, dpart.as<double>());
    pos_density_reduce_kernel<<<(nx + 127) / 128, 128>>>(dpart.as<double>(), nb, nx, dout.as<double>());
    KMAP_CHECK_HIP(hipGetLastError());
    KMAP_CHECK_HIP(hipMemcpy(density, dout.p, (size_t)nx * 8, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

}  // extern "C"

// ---- labelled sampling on the device (sample_disp_kmer, reference motif_discovery.py:812-921) --------------------------
// For the k-mer counts of a long final k (10^8..10^9 unique k-mers) the reference's numpy formulation builds n_conseq x n_uniq
// matrices on the host.  Here the unique k-mers stay on the device: one kernel labels them (and re-orients reverse-complement
// members), per-label weight sums / prefix sums / searches run on the device, and only the sampled entries come back.
namespace {
struct LabTab {
    uint64_t cons[63], rccons[63];
    int32_t clen[63], radius[63];
    int n;
};

template <typename H>
__global__ __launch_bounds__(256) void label_kernel(H *__restrict__ uniq, int64_t n, int k, LabTab t, int radius_k, int revcom,
                                                    uint8_t *__restrict__ label) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const H h = uniq[i];
    int best = 1 << 20, lab = t.n;
    bool best_rc = false;
    for (int c = 0; c < t.n; ++c) {
        const int cl = t.clen[c];
        const H cm = low_mask<H>(cl);
        int d = popc2((H)(((h >> (2 * (k - cl))) ^ (H)t.cons[c]) & cm));      // head: first cl bases vs the consensus
        bool rc = false;
        if (revcom) {
            const int d2 = popc2((H)((h ^ (H)t.rccons[c]) & cm));             // tail: last cl bases vs its reverse complement
            rc = d2 < d;
            d = rc ? d2 : d;
        }
        if (d > t.radius[c]) d = k;                                           // outside this consensus' ball (:869-870)
        if (d < best) {                                                       // np.argmin: first minimum
            best = d;
            lab = c;
            best_rc = rc;
        }
    }
    if (best > radius_k) lab = t.n;                                           // noise label (:873)
    label[i] = (uint8_t)lab;
    if (revcom && lab < t.n && best_rc) uniq[i] = revcom_hash(h, k);          // align members with their consensus (:876-883)
}

template <typename CT>
__global__ __launch_bounds__(256) void label_sums_kernel(const uint8_t *__restrict__ label, const CT *__restrict__ cnt, int64_t n,
                                                         int n_labels, unsigned long long *__restrict__ wsum,
                                                         unsigned long long *__restrict__ members) {
    __shared__ unsigned long long sw[64], sm[64];
    if (threadIdx.x < 64) sw[threadIdx.x] = sm[threadIdx.x] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int l = label[i];
        if (l < n_labels) {
            atomicAdd(&sw[l], (unsigned long long)(long long)cnt[i]);
            atomicAdd(&sm[l], 1ull);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < n_labels) {
        if (sw[threadIdx.x]) atomicAdd(&wsum[threadIdx.x], sw[threadIdx.x]);
        if (sm[threadIdx.x]) atomicAdd(&members[threadIdx.x], sm[threadIdx.x]);
    }
}

// w[i] = weight of entry i if it carries label c (its count, or 1 when cnt == nullptr), else 0
template <typename CT>
__global__ __launch_bounds__(256) void label_weight_kernel(const uint8_t *__restrict__ label, const CT *__restrict__ cnt, int64_t n,
                                                           int c, uint32_t *__restrict__ w) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    w[i] = (label[i] == c) ? (cnt ? (uint32_t)cnt[i] : 1u) : 0u;
}

// first index whose inclusive prefix sum exceeds the target: np.searchsorted(cdf, x, side="right") on integer weights
__global__ __launch_bounds__(256) void prefix_search_kernel(const uint64_t *__restrict__ excl, int64_t n, const int64_t *__restrict__ target,
                                                            int64_t m, int64_t *__restrict__ idx) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const uint64_t t = (uint64_t)target[j];
    int64_t lo = 0, hi = n;                       // smallest i in [0, n) with excl[i + 1] > t; n if none
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (excl[mid + 1] > t) hi = mid;
        else lo = mid + 1;
    }
    idx[j] = lo;
}

__global__ __launch_bounds__(256) void compact_label_kernel(const uint8_t *__restrict__ label, const uint64_t *__restrict__ excl,
                                                            int64_t n, int c, int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && label[i] == c) out[excl[i]] = i;
}

template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const T *__restrict__ src, const int64_t *__restrict__ idx, int64_t m,
                                                     T *__restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < m) out[j] = src[idx[j]];
}
}  // namespace

extern "C" {

int kmap_label_kmers_dev(void *uniq_dev, int64_t n, int k, int n_cons, const uint64_t *cons_kh, const int32_t *cons_len,
                         const int32_t *cons_radius, int radius_k, int revcom_mode, uint8_t *label_dev, void *stream) {
    KMAP_REQUIRE(k > 0 && k < 32 && n >= 0, "label_kmers: bad k / n");
    KMAP_REQUIRE(n_cons > 0 && n_cons <= 63 && cons_kh && cons_len && cons_radius, "label_kmers: 1..63 consensuses");
    if (n == 0) return KMAP_OK;
    KMAP_REQUIRE(uniq_dev && label_dev, "label_kmers: null pointer");
    LabTab t;
    t.n = n_cons;
    for (int c = 0; c < n_cons; ++c) {
        const int cl = cons_len[c];
        KMAP_REQUIRE(cl > 0 && cl <= k, "label_kmers: consensus length %d not in (0, k]", cl);
        const uint64_t m = low_mask<uint64_t>(cl), ch = cons_kh[c] & m;
        uint64_t com = m - ch, rc = 0;
        for (int p = 0; p < cl; ++p) rc = (rc << 2) | ((com >> (2 * p)) & 3);
        t.cons[c] = ch;
        t.rccons[c] = rc;
        t.clen[c] = cl;
        t.radius[c] = cons_radius[c];
    }
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (k < 16) label_kernel<uint32_t><<<nb, 256, 0, as_stream(stream)>>>((uint32_t *)uniq_dev, n, k, t, radius_k, revcom_mode, label_dev);
    else label_kernel<uint64_t><<<nb, 256, 0, as_stream(stream)>>>((uint64_t *)uniq_dev, n, k, t, radius_k, revcom_mode, label_dev);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

/* per label l < n_labels (<= 64): sum of the counts and number of members; cnt is int32 (cnt64 = 0) or int64 */
int kmap_label_sums_dev(const uint8_t *label_dev, const void *cnt_dev, int cnt64, int64_t n, int n_labels, int64_t *weight_sums,
                        int64_t *member_counts) {
    KMAP_REQUIRE(n >= 0 && n_labels > 0 && n_labels <= 64 && weight_sums && member_counts, "label_sums: bad arguments");
    for (int l = 0; l < n_labels; ++l) weight_sums[l] = member_counts[l] = 0;
    if (n == 0) return KMAP_OK;
    KMAP_REQUIRE(label_dev && cnt_dev, "label_sums: null pointer");
    DevBuf acc;
    KMAP_TRY(acc.alloc(128 * 8));
    KMAP_CHECK_HIP(hipMemset(acc.p, 0, 128 * 8));
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    unsigned long long *w = acc.as<unsigned long long>();
    if (cnt64) label_sums_kernel<int64_t><<<(unsigned)g, 256>>>(label_dev, (const int64_t *)cnt_dev, n, n_labels, w, w + 64);
    else label_sums_kernel<int32_t><<<(unsigned)g, 256>>>(label_dev, (const int32_t *)cnt_dev, n, n_labels, w, w + 64);
    KMAP_CHECK_HIP(hipGetLastError());
    unsigned long long host[128];
    KMAP_CHECK_HIP(hipMemcpy(host, acc.p, sizeof host, hipMemcpyDeviceToHost));
    for (int l = 0; l < n_labels; ++l) {
        weight_sums[l] = (int64_t)host[l];
        member_counts[l] = (int64_t)host[64 + l];
    }
    return KMAP_OK;
}

/* excl_dev[0..n] (uint64) = exclusive prefix sums of the label-c weights (counts, or 1 per member when cnt_dev is NULL);
 * excl_dev[n] = total.  scratch_dev: n uint32. */
int kmap_label_prefix_dev(const uint8_t *label_dev, const void *cnt_dev, int cnt64, int64_t n, int c, uint32_t *scratch_dev,
                          uint64_t *excl_dev, void *stream) {
    KMAP_REQUIRE(n > 0 && label_dev && scratch_dev && excl_dev, "label_prefix: bad arguments");
    hipStream_t st = as_stream(stream);
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (cnt_dev && cnt64) label_weight_kernel<int64_t><<<nb, 256, 0, st>>>(label_dev, (const int64_t *)cnt_dev, n, c, scratch_dev);
    else label_weight_kernel<int32_t><<<nb, 256, 0, st>>>(label_dev, (const int32_t *)cnt_dev, n, c, scratch_dev);
    KMAP_TRY(exclusive_scan_u32(scratch_dev, n, excl_dev, st));
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

/* idx_out[j] = first i with (inclusive prefix)[i] > targets[j]  (n if none); targets / idx_out are host arrays */
int kmap_prefix_search_dev(const uint64_t *excl_dev, int64_t n, const int64_t *targets, int64_t m, int64_t *idx_out) {
    KMAP_REQUIRE(n > 0 && excl_dev && m >= 0 && (m == 0 || (targets && idx_out)), "prefix_search: bad arguments");
    if (m == 0) return KMAP_OK;
    DevBuf t, o;
    KMAP_TRY(t.alloc((size_t)m * 8));
    KMAP_TRY(o.alloc((size_t)m * 8));
    KMAP_CHECK_HIP(hipMemcpy(t.p, targets, (size_t)m * 8, hipMemcpyHostToDevice));
    prefix_search_kernel<<<(unsigned)((m + 255) / 256), 256>>>(excl_dev, n, t.as<int64_t>(), m, o.as<int64_t>());
    KMAP_CHECK_HIP(hipGetLastError());
    KMAP_CHECK_HIP(hipMemcpy(idx_out, o.p, (size_t)m * 8, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

/* indices (ascending) of the entries carrying label c, using the member-count prefix from kmap_label_prefix_dev(cnt = NULL);
 * idx_out: host int64[excl[n]] */
int kmap_label_members_dev(const uint8_t *label_dev, const uint64_t *excl_dev, int64_t n, int c, int64_t m, int64_t *idx_out) {
    KMAP_REQUIRE(n > 0 && label_dev && excl_dev && m >= 0 && (m == 0 || idx_out), "label_members: bad arguments");
    if (m == 0) return KMAP_OK;
    DevBuf o;
    KMAP_TRY(o.alloc((size_t)m * 8));
    compact_label_kernel<<<(unsigned)((n + 255) / 256), 256>>>(label_dev, excl_dev, n, c, o.as<int64_t>());
    KMAP_CHECK_HIP(hipGetLastError());
    KMAP_CHECK_HIP(hipMemcpy(idx_out, o.p, (size_t)m * 8, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

/* out[j] = src_dev[idx[j]] for elem_bytes in {1, 4, 8}; idx / out are host arrays */
int kmap_gather_dev(const void *src_dev, int elem_bytes, const int64_t *idx, int64_t m, void *out) {
    KMAP_REQUIRE(elem_bytes == 1 || elem_bytes == 4 || elem_bytes == 8, "gather: element size %d", elem_bytes);
    KMAP_REQUIRE(m >= 0 && (m == 0 || (src_dev && idx && out)), "gather: bad arguments");
    if (m == 0) return KMAP_OK;
    DevBuf di, dout;
    KMAP_TRY(di.alloc((size_t)m * 8));
    KMAP_TRY(dout.alloc((size_t)m * elem_bytes));
    KMAP_CHECK_HIP(hipMemcpy(di.p, idx, (size_t)m * 8, hipMemcpyHostToDevice));
    const unsigned nb = (unsigned)((m + 255) / 256);
    if (elem_bytes == 1) gather_kernel<uint8_t><<<nb, 256>>>((const uint8_t *)src_dev, di.as<int64_t>(), m, dout.as<uint8_t>());
    else if (elem_bytes == 4) gather_kernel<uint32_t><<<nb, 256>>>((const uint32_t *)src_dev, di.as<int64_t>(), m, dout.as<uint32_t>());
    else gather_kernel<uint64_t><<<nb, 256>>>((const uint64_t *)src_dev, di.as<int64_t>(), m, dout.as<uint64_t>());
    KMAP_CHECK_HIP(hipGetLastError());
    KMAP_CHECK_HIP(hipMemcpy(out, dout.p, (size_t)m * elem_bytes, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

}  // extern "C"
