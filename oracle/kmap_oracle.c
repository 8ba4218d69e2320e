/*
 * kmap_oracle.c -- CPU restatement of the kmap hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the *checker* for the HIP implementation in kmap_amd/csrc/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product path
 * (kmap_amd/) never does.  Parity status: PINNED -- every function below is checked
 * against golden vectors produced by running the reference itself (tests/golden/
 * gen_golden.py, see tests/test_oracle_golden.py).
 *
 * Each function restates, loop by loop, what the reference computes; citations are
 * file:line into /root/reference/src/kmap/.  Nothing here is copied from the reference
 * (which is Python/Taichi); this is plain C written from its behaviour.
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -ffp-contract=off, no fast-math).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KO_MISSING 255u

/* ---- hashing: taichi_core.py:3-61, kmer_count.py:449-473 -------------------------------
 * hash at EVERY array index; big-endian base-4; invalid (= dtype max) if the window
 * touches a 255 byte or runs past the end of the array. */
void ko_hash_u32(const uint8_t *seq, int64_t n, int k, uint32_t *out) {
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        int bad = (p + k > n);
        uint32_t h = 0;
        for (int i = 0; i < k && p + i < n; ++i) {
            uint8_t b = seq[p + i];
            if (b == KO_MISSING) bad = 1;
            h = (h << 2) + b;
        }
        out[p] = bad ? UINT32_MAX : h;
    }
}
void ko_hash_u64(const uint8_t *seq, int64_t n, int k, uint64_t *out) {
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        int bad = (p + k > n);
        uint64_t h = 0;
        for (int i = 0; i < k && p + i < n; ++i) {
            uint8_t b = seq[p + i];
            if (b == KO_MISSING) bad = 1;
            h = (h << 2) + b;
        }
        out[p] = bad ? UINT64_MAX : h;
    }
}

/* ---- per-read de-duplication: kmer_count.py:743-760 -------------------------------------
 * inside every [st,en) keep the FIRST occurrence of each value, overwrite later ones with
 * the invalid hash (np.unique(return_index) gives first occurrences). */
void ko_dedupe_u32(uint32_t *h, const int64_t *borders, int64_t n_seq) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t s = 0; s < n_seq; ++s) {
        int64_t st = borders[2 * s], en = borders[2 * s + 1];
        for (int64_t i = st; i < en; ++i) {
            uint32_t v = h[i];
            if (v == UINT32_MAX) continue;
            for (int64_t j = st; j < i; ++j)
                if (h[j] == v) { h[i] = UINT32_MAX; break; }
        }
    }
}
void ko_dedupe_u64(uint64_t *h, const int64_t *borders, int64_t n_seq) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t s = 0; s < n_seq; ++s) {
        int64_t st = borders[2 * s], en = borders[2 * s + 1];
        for (int64_t i = st; i < en; ++i) {
            uint64_t v = h[i];
            if (v == UINT64_MAX) continue;
            for (int64_t j = st; j < i; ++j)
                if (h[j] == v) { h[i] = UINT64_MAX; break; }
        }
    }
}

/* ---- unique + counts: kmer_count.py:476-491 (np.unique, invalid dropped) ---------------- */
static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}
/* uniq/cnt must hold n entries; returns the number of unique valid hashes */
int64_t ko_count_u32(const uint32_t *h, int64_t n, uint32_t *uniq, int64_t *cnt) {
    uint32_t *t = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof *t);
    memcpy(t, h, (size_t)n * sizeof *t);
    qsort(t, (size_t)n, sizeof *t, cmp_u32);
    int64_t m = 0;
    for (int64_t i = 0; i < n;) {
        int64_t j = i;
        while (j < n && t[j] == t[i]) ++j;
        if (t[i] != UINT32_MAX) { uniq[m] = t[i]; cnt[m] = j - i; ++m; }
        i = j;
    }
    free(t);
    return m;
}
int64_t ko_count_u64(const uint64_t *h, int64_t n, uint64_t *uniq, int64_t *cnt) {
    uint64_t *t = (uint64_t *)malloc((size_t)(n ? n : 1) * sizeof *t);
    memcpy(t, h, (size_t)n * sizeof *t);
    qsort(t, (size_t)n, sizeof *t, cmp_u64);
    int64_t m = 0;
    for (int64_t i = 0; i < n;) {
        int64_t j = i;
        while (j < n && t[j] == t[i]) ++j;
        if (t[i] != UINT64_MAX) { uniq[m] = t[i]; cnt[m] = j - i; ++m; }
        i = j;
    }
    free(t);
    return m;
}

/* ---- reverse complement: taichi_core.py:181-224, kmer_count.py:613-640 -------------------
 * com = (4^k - 1) - h ; then reverse the k 2-bit groups. */
static inline uint64_t rc64(uint64_t h, int k) {
    uint64_t mask = (k >= 32) ? UINT64_MAX : ((1ull << (2 * k)) - 1);
    uint64_t com = mask - h, r = com & 3u;
    for (int i = 0; i < k - 1; ++i) { r <<= 2; com >>= 2; r += com & 3u; }
    return r;
}
static inline uint32_t rc32(uint32_t h, int k) {
    uint32_t mask = (uint32_t)((1ull << (2 * k)) - 1);
    uint32_t com = mask - h, r = com & 3u;           /* u32 wrap-around as in the kernel */
    for (int i = 0; i < k - 1; ++i) { r <<= 2; com >>= 2; r += com & 3u; }
    return r;
}
void ko_revcom_u32(const uint32_t *in, int64_t n, int k, uint32_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = rc32(in[i], k);
}
void ko_revcom_u64(const uint64_t *in, int64_t n, int k, uint64_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = rc64(in[i], k);
}

/* ---- merge_revcom: kmer_count.py:643-685 (keep_lower_hash_flag = True) --------------------
 * uniq is ascending.  cnt[x] += cnt[rc(x)] for every x whose revcom is present (a palindrome
 * is its own partner, so it doubles); the higher member of a present pair is deleted; a
 * remaining x > rc(x) (partner absent) is REPLACED by rc(x) in place -- no re-sort.
 * Counts are added in the count dtype of the caller (wrap-around is the caller's cast). */
static int64_t bs64(const uint64_t *a, int64_t n, uint64_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
    return (lo < n && a[lo] == v) ? lo : -1;
}
int64_t ko_merge_revcom_u64(const uint64_t *uniq, const int64_t *cnt, int64_t n, int k, int narrow,
                            uint64_t *ouniq, int64_t *ocnt) {
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t x = uniq[i];
        uint64_t r = narrow ? (uint64_t)rc32((uint32_t)x, k) : rc64(x, k);
        int64_t j = bs64(uniq, n, r);
        if (j >= 0 && x > r) continue;                 /* higher member of a present pair */
        ouniq[m] = (x > r) ? r : x;
        ocnt[m] = cnt[i] + (j >= 0 ? cnt[j] : 0);
        ++m;
    }
    return m;
}

/* ---- Hamming distance over the low 2k bits: taichi_core.py:63-104 ------------------------
 * invalid hashes are compared like any other value (kmer_count.py:494-515). */
static inline int ham64(uint64_t a, uint64_t b, int k) {
    uint64_t x = a ^ b;
    int d = 0;
    for (int i = 0; i < k; ++i) { d += (x & 3u) != 0; x >>= 2; }
    return d;
}
void ko_ham_u32(const uint32_t *h, int64_t n, uint32_t c, int k, uint8_t *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)ham64(h[i], c, k);
}
void ko_ham_u64(const uint64_t *h, int64_t n, uint64_t c, int k, uint8_t *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)ham64(h[i], c, k);
}
/* head: taichi_core.py:108-124,144-160 -- consensus vs the FIRST clen bases of the k-mer */
void ko_ham_head_u32(const uint32_t *h, int64_t n, uint32_t c, int k, int clen, uint8_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)ham64(h[i] >> (2 * (k - clen)), c, clen);
}
void ko_ham_head_u64(const uint64_t *h, int64_t n, uint64_t c, int k, int clen, uint8_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)ham64(h[i] >> (2 * (k - clen)), c, clen);
}
/* tail: taichi_core.py:127-141,163-177 -- consensus vs the LAST clen bases */
void ko_ham_tail_u32(const uint32_t *h, int64_t n, uint32_t c, int k, int clen, uint8_t *out) {
    (void)k;
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)ham64(h[i], c, clen);
}
void ko_ham_tail_u64(const uint64_t *h, int64_t n, uint64_t c, int k, int clen, uint8_t *out) {
    (void)k;
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)ham64(h[i], c, clen);
}

/* ---- mask_input: kmer_count.py:580-610 ---------------------------------------------------
 * hashes are computed ONCE from the incoming array; for each consensus in turn, every
 * position with distance <= r gets [i, min(i+k, n)) overwritten with 255.  Invalid hashes
 * (all ones) take part in the comparison (so a poly-T-like consensus masks across
 * separators -- verified reference behaviour). */
void ko_mask_input(uint8_t *seq, int64_t n, int k, const uint64_t *cons, const int64_t *r, int n_cons) {
    uint64_t *h = (uint64_t *)malloc((size_t)(n ? n : 1) * sizeof *h);
    if (k < 16) {
        uint32_t *h32 = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof *h32);
        ko_hash_u32(seq, n, k, h32);
        for (int64_t i = 0; i < n; ++i) h[i] = h32[i];   /* invalid stays 0xFFFFFFFF (u32 compare) */
        free(h32);
    } else {
        ko_hash_u64(seq, n, k, h);
    }
    for (int c = 0; c < n_cons; ++c) {
        for (int64_t i = 0; i < n; ++i) {
            if (ham64(h[i], cons[c], k) <= r[c]) {
                int64_t j = (i + k < n) ? i + k : n;
                memset(seq + i, 255, (size_t)(j - i));
            }
        }
    }
    free(h);
}

/* ---- Hamming-ball mass: motif_discovery.py:666-673 ---------------------------------------
 * sum of counts of all unique k-mers within r of cand (or of its revcom when revcom != 0) */
void ko_hamball_mass(const uint64_t *uniq, const int64_t *cnt, int64_t n, int k, int narrow,
                     const uint64_t *cand, int n_cand, int r, int revcom, double *out) {
    for (int c = 0; c < n_cand; ++c) {
        uint64_t a = cand[c];
        uint64_t b = narrow ? (uint64_t)rc32((uint32_t)a, k) : rc64(a, k);
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            int d = ham64(uniq[i], a, k);
            if (revcom) { int d2 = ham64(uniq[i], b, k); if (d2 < d) d = d2; }
            if (d <= r) s += (double)cnt[i];
        }
        out[c] = s;
    }
}

/* ---- motif occurrence scan of one read: motif_discovery.py:1422-1477 ---------------------
 * read = seq[st, st+len) WITHOUT separator.  Candidate positions are the first
 * slice_stop = python-slice [0 : len-k+1] entries of the per-read hash array (a negative
 * stop wraps, as Python does); dist = min(fwd, rc); hits = positions with dist <= r that
 * are at the read's minimum hit distance.  Writes hit positions (ascending) and returns
 * their number; the >20 random subsample is the caller's (host RNG). */
int64_t ko_scan_read(const uint8_t *read, int64_t len, int k, uint64_t cons, int r, int revcom,
                     int32_t *pos_out, int *min_dist_out) {
    int64_t stop = len - k + 1;
    if (stop < 0) { stop += len; if (stop < 0) stop = 0; }
    if (stop > len) stop = len;
    int narrow = (k < 16);
    uint64_t rcc = narrow ? (uint64_t)rc32((uint32_t)cons, k) : rc64(cons, k);
    uint64_t inval = narrow ? (uint64_t)UINT32_MAX : UINT64_MAX;
    int best = 1 << 30;
    int64_t m = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int64_t p = 0; p < stop; ++p) {
            int bad = (p + k > len);
            uint64_t h = 0;
            for (int i = 0; i < k && p + i < len; ++i) {
                if (read[p + i] == KO_MISSING) bad = 1;
                h = (h << 2) + read[p + i];
            }
            if (narrow) h &= 0xFFFFFFFFull;
            if (bad) h = inval;
            int d = ham64(h, cons, k);
            if (revcom) { int d2 = ham64(h, rcc, k); if (d2 < d) d = d2; }
            if (d > r) continue;
            if (pass == 0) { if (d < best) best = d; }
            else if (d == best) pos_out[m++] = (int32_t)p;
        }
        if (pass == 0 && best == (1 << 30)) break;
    }
    *min_dist_out = (best == (1 << 30)) ? -1 : best;
    return m;
}

/* ---- sampled-k-mer Hamming matrix: motif_discovery.py:759-808 ----------------------------
 * n_uniq x n_uniq distances over k bases; pairs that share label l with len(conseq_l) < k are
 * recomputed on the first len(conseq_l) bases (`clen[l]`; labels >= n_lab are never
 * overridden).  out is row-major u8 with leading dimension n (the reference stores int64;
 * values are identical). */
static inline int ham_pop64(uint64_t x, int k) {
    uint64_t m = (k >= 32) ? UINT64_MAX : ((1ull << (2 * k)) - 1);
    x &= m;
    uint64_t y = (x | (x >> 1)) & 0x5555555555555555ull;
    return __builtin_popcountll(y);
}
void ko_hamdist_matrix(const uint64_t *kh, const int32_t *label, int64_t n, int k,
                       const int32_t *clen, int n_lab, uint8_t *out) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < n; ++i) {
        uint64_t a = kh[i];
        int li = label[i];
        int ci = (li >= 0 && li < n_lab) ? clen[li] : k;
        for (int64_t j = 0; j < n; ++j) {
            uint64_t x = a ^ kh[j];
            int d;
            if (label[j] == li && ci < k) d = ham_pop64(x >> (2 * (k - ci)), ci);
            else d = ham_pop64(x, k);
            out[i * n + j] = (uint8_t)d;
        }
    }
}
/* a rows-subset variant for timing the CPU baseline on a bounded sample */
void ko_hamdist_rows(const uint64_t *kh, const int32_t *label, int64_t n, int k, const int32_t *clen,
                     int n_lab, int64_t row0, int64_t nrows, uint8_t *out) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = row0; i < row0 + nrows; ++i) {
        uint64_t a = kh[i];
        int li = label[i];
        int ci = (li >= 0 && li < n_lab) ? clen[li] : k;
        uint8_t *o = out + (i - row0) * n;
        for (int64_t j = 0; j < n; ++j) {
            uint64_t x = a ^ kh[j];
            int d;
            if (label[j] == li && ci < k) d = ham_pop64(x >> (2 * (k - ci)), ci);
            else d = ham_pop64(x, k);
            o[j] = (uint8_t)d;
        }
    }
}

/* ---- kNN smoothing: visualization.py:90-109, taichi_core.py:227-249 -----------------------
 * S[i,j] = (sum_{ii,jj} D[nb[i,ii], nb[j,jj]]) / n_nb / n_nb for i<j in f32 (sequential sum
 * in the reference's ii-outer / jj-inner order), mirrored, diagonal 0. */
void ko_knn_smooth_f32(const float *D, const int32_t *nb, int64_t n, int n_nb, float *S) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t i = 0; i < n; ++i) {
        S[i * n + i] = 0.0f;
        for (int64_t j = i + 1; j < n; ++j) {
            float s = 0.0f;
            for (int ii = 0; ii < n_nb; ++ii)
                for (int jj = 0; jj < n_nb; ++jj)
                    s += D[(int64_t)nb[i * n_nb + ii] * n + nb[j * n_nb + jj]];
            s = s / (float)n_nb;
            s = s / (float)n_nb;
            S[i * n + j] = s;
        }
    }
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < i; ++j) S[i * n + j] = S[j * n + i];
}

/* ---- gradient row sums: taichi_core.py:305-326, visualization.py:131-145 ------------------
 * g[k,i] = sum_{j != i} T[i,j] * (y[k,i] - y[k,j]), f32, j ascending, no FMA (build flag
 * -ffp-contract=off).  The caller multiplies by 4.0f (visualization.py:145). */
void ko_gradient_rows(const float *T, const float *y, int64_t n, float *g) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        for (int kk = 0; kk < 2; ++kk) {
            float s = 0.0f;
            float yi = y[kk * n + i];
            for (int64_t j = 0; j < n; ++j) {
                if (j == i) continue;
                float d = yi - y[kk * n + j];
                float p = T[i * n + j] * d;
                s = s + p;
            }
            g[kk * n + i] = s;
        }
    }
}
