/*
 * kmap_cpu_baseline.c -- compiled OpenMP port of the kmap hot path, built to be TIMED on the host cores of the GPU box
 * (bench.py's `cpu_baseline` leg: SURVEY.md 8(d) "CPU baseline").  TEST / BENCH INFRASTRUCTURE ONLY: nothing
 * under kmap_amd/ loads it.
 *
 * Why a second CPU file next to kmap_oracle.c: the oracle restates the reference loop by loop for clarity (qsort, serial
 * masking, one Hamming pass per candidate) and is the CHECKER; a baseline should be what a competent CPU implementation of
 * the same algorithm costs -- per-thread histograms instead of np.unique's sort, a hashed per-read set instead of the
 * reference's Python loop, one fused pass per embedding iteration instead of ten N x N temporaries.  Results are the
 * reference's (tests/test_oracle_golden.py::test_cpu_baseline_equals_oracle compares every function below with the oracle);
 * the summation order of the embedding gradient is the reference's (j ascending, f32, no FMA), so the gradient is bit-equal.
 * Citations are file:line into /root/reference/src/kmap/.
 *
 * Build: oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off).  Every entry point takes `threads` (0 = OpenMP default).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KB_MISSING 255u

static inline uint32_t kb_rc(uint32_t h, int k) {           /* kmer_count.py:626-640 */
    uint32_t mask = (k >= 16) ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    uint32_t c = mask - h, r = 0;
    for (int i = 0; i < k; ++i) { r = (r << 2) | (c & 3u); c >>= 2; }
    return r;
}
static inline int kb_ham(uint32_t a, uint32_t b) {           /* taichi_core.py:63-72 on the low 2k bits (both < 4^k) */
    uint32_t x = a ^ b;
    return __builtin_popcount((x | (x >> 1)) & 0x55555555u);
}
static void kb_threads(int threads) { if (threads > 0) omp_set_num_threads(threads); }

/* ---- counting: comp_kmer_hash + remove_duplicate_hash_per_seq + count_uniq_hash (kmer_count.py:449-491,743-760), k <= 13.
 * hist: uint32[4^k], zeroed here.  Windows never cross a 255.  dedupe: a k-mer counts once per read (first occurrence).
 * borders (n_seq,2) int64 [start,end).  Positions outside every read (none in the array contract) are ignored. */
void kb_count(const uint8_t *seq, int64_t n, const int64_t *borders, int64_t n_seq, int k, int dedupe, uint32_t *hist,
              int threads) {
    kb_threads(threads);
    const size_t n_bins = (size_t)1 << (2 * k);
    const uint32_t mask = (uint32_t)(n_bins - 1);
    memset(hist, 0, n_bins * 4);
    (void)n;
#pragma omp parallel
    {
        uint32_t *mine = (uint32_t *)calloc(n_bins, 4);
        /* per-read set: open addressing, 4x the longest plausible read; generation stamps instead of clearing */
        enum { SET = 4096 };
        uint32_t *keys = (uint32_t *)malloc(SET * 4), *gen = (uint32_t *)calloc(SET, 4);
        uint32_t g = 0;
#pragma omp for schedule(dynamic, 1024)
        for (int64_t s = 0; s < n_seq; ++s) {
            const int64_t st = borders[2 * s], en = borders[2 * s + 1];
            ++g;
            uint32_t h = 0;
            int run = 0;                                      /* valid bases in the current window run */
            const int use_set = dedupe && (en - st) * 2 <= SET;
            for (int64_t p = st; p < en; ++p) {
                const uint8_t b = seq[p];
                if (b == KB_MISSING) { run = 0; h = 0; continue; }
                h = ((h << 2) | b) & mask;
                if (++run < k) continue;
                if (!dedupe) { ++mine[h]; continue; }
                if (use_set) {
                    uint32_t slot = (h * 2654435761u) >> 20;  /* 12 bits */
                    int dup = 0;
                    while (gen[slot] == g) {
                        if (keys[slot] == h) { dup = 1; break; }
                        slot = (slot + 1) & (SET - 1);
                    }
                    if (dup) continue;
                    gen[slot] = g;
                    keys[slot] = h;
                    ++mine[h];
                } else {                                      /* very long read: quadratic look-back like the oracle */
                    int dup = 0;
                    uint32_t h2 = 0;
                    int run2 = 0;
                    for (int64_t q = st; q < p && !dup; ++q) {
                        const uint8_t b2 = seq[q];
                        if (b2 == KB_MISSING) { run2 = 0; h2 = 0; continue; }
                        h2 = ((h2 << 2) | b2) & mask;
                        if (++run2 >= k && h2 == h) dup = 1;
                    }
                    if (!dup) ++mine[h];
                }
            }
        }
#pragma omp critical
        for (size_t i = 0; i < n_bins; ++i) hist[i] += mine[i];
        free(mine);
        free(keys);
        free(gen);
    }
}

/* ---- merge_revcom on the histogram (kmer_count.py:643-685): the kept (lower) member of a pair takes both counts, palindromes
 * double, the higher member is dropped; a k-mer whose partner is absent is stored under min(x, rc(x)).  Output: compacted
 * (uniq ascending by the stored key's ORIGINAL position, cnt) as a set -- the tests compare it with the oracle as a dict. */
int64_t kb_merge_compact(const uint32_t *hist, int k, int merge, uint32_t *uniq, int64_t *cnt) {
    const size_t n_bins = (size_t)1 << (2 * k);
    int64_t m = 0;
    for (size_t x = 0; x < n_bins; ++x) {
        const uint32_t c = hist[x];
        if (!c) continue;
        if (!merge) { uniq[m] = (uint32_t)x; cnt[m++] = c; continue; }
        const uint32_t r = kb_rc((uint32_t)x, k);
        const uint32_t cr = hist[r];
        if (cr) {                                             /* partner present (a palindrome is its own partner) */
            if (x <= r) { uniq[m] = (uint32_t)x; cnt[m++] = (int64_t)c + cr; }
        } else {
            uniq[m] = (uint32_t)(x < r ? x : r);
            cnt[m++] = c;
        }
    }
    return m;
}

/* ---- Hamming-ball mass of candidates over the counted k-mers (motif_discovery.py:666-673) */
void kb_hamball_mass(const uint32_t *uniq, const int64_t *cnt, int64_t n, int k, const uint32_t *cand, int n_cand, int r,
                     int revcom, double *out, int threads) {
    kb_threads(threads);
    for (int c = 0; c < n_cand; ++c) {
        const uint32_t a = cand[c], b = kb_rc(a, k);
        double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            int d = kb_ham(uniq[i], a);
            if (revcom) { const int d2 = kb_ham(uniq[i], b); if (d2 < d) d = d2; }
            if (d <= r) s += (double)cnt[i];
        }
        out[c] = s;
    }
}

/* ---- mask_input (kmer_count.py:580-610): every window within r[c] of cons[c] (hashes of the INCOMING array; an invalid window
 * compares as the all-ones hash, which lets a poly-T-like consensus mask across separators -- the reference's behaviour)
 * overwrites [i, min(i+k, n)) with 255.  Two passes: flags from the incoming array, then the cover. */
void kb_mask(uint8_t *seq, int64_t n, int k, const uint32_t *cons, const int32_t *r, int n_cons, int threads) {
    kb_threads(threads);
    uint8_t *flag = (uint8_t *)calloc((size_t)(n ? n : 1), 1);
    const uint32_t mask = (k >= 16) ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        int bad = (i + k > n);
        uint32_t h = 0;
        for (int t = 0; t < k && i + t < n; ++t) {
            if (seq[i + t] == KB_MISSING) bad = 1;
            h = (h << 2) + seq[i + t];
        }
        for (int c = 0; c < n_cons; ++c) {
            /* the reference compares the low 2k bits of the hash array entry; an invalid entry is all ones */
            const uint32_t hv = bad ? mask : (h & mask);
            if (kb_ham(hv, cons[c] & mask) <= r[c]) { flag[i] = 1; break; }
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t lo = i - k + 1 > 0 ? i - k + 1 : 0;
        for (int64_t j = i; j >= lo; --j)
            if (flag[j]) { seq[i] = 255; break; }
    }
    free(flag);
}

/* ---- find_motif for one k (motif_discovery.py:594-702), k <= 13: count (first round with per-read dedupe unless rep_mode) ->
 * top_k candidates (largest count, lowest hash on ties: np.argpartition's tie order is numpy-specific) -> Hamming-ball mass ->
 * ratio test -> mask consensus + revcom -> recount without dedupe.  seq is masked in place.  Returns the number of consensus
 * sequences found; cons_out[i], prop_out[i] (hamball proportion) per motif; *n_uniq_first = unique k-mers of the first round. */
int kb_find_motif(uint8_t *seq, int64_t n, const int64_t *borders, int64_t n_seq, int k, int max_ham_dist, double p_unif,
                  double ratio_cutoff, int top_k, int n_trial, int revcom, int rep_mode, uint32_t *cons_out, double *prop_out,
                  int64_t *n_uniq_first, int threads) {
    const size_t n_bins = (size_t)1 << (2 * k);
    uint32_t *hist = (uint32_t *)malloc(n_bins * 4), *uniq = (uint32_t *)malloc(n_bins * 4);
    int64_t *cnt = (int64_t *)malloc(n_bins * 8);
    kb_count(seq, n, borders, n_seq, k, !rep_mode, hist, threads);
    int64_t m = kb_merge_compact(hist, k, revcom, uniq, cnt);
    if (n_uniq_first) *n_uniq_first = m;
    int64_t n_total = 0;
    for (int64_t i = 0; i < m; ++i) n_total += cnt[i];
    if (k < 16) n_total = (int32_t)n_total;                   /* the reference sums int32 scalars (wraps) */
    int found = 0;
    for (int t = 0; t < n_trial; ++t) {
        if (top_k > m) break;
        uint32_t cand[64];
        int64_t cc[64];
        int nc = 0;
        for (int64_t i = 0; i < m; ++i) {                     /* top_k by (count desc, hash asc) */
            int pos = nc;
            while (pos > 0 && cnt[i] > cc[pos - 1]) --pos;
            if (pos >= top_k) continue;
            const int last = nc < top_k ? nc : top_k - 1;
            for (int q = last; q > pos; --q) { cand[q] = cand[q - 1]; cc[q] = cc[q - 1]; }
            cand[pos] = uniq[i];
            cc[pos] = cnt[i];
            if (nc < top_k) ++nc;
        }
        double mass[64];
        kb_hamball_mass(uniq, cnt, m, k, cand, nc, max_ham_dist, revcom, mass, threads);
        int best = 0;
        for (int c = 1; c < nc; ++c)
            if (mass[c] > mass[best]) best = c;
        const double prop = mass[best] / (double)n_total, ratio = prop / p_unif;
        if (!(ratio > ratio_cutoff)) break;
        cons_out[found] = cand[best];
        prop_out[found] = prop;
        ++found;
        uint32_t cons[2] = {cand[best], kb_rc(cand[best], k)};
        int32_t rr[2] = {max_ham_dist, max_ham_dist};
        kb_mask(seq, n, k, cons, rr, revcom ? 2 : 1, threads);
        kb_count(seq, n, borders, n_seq, k, 0, hist, threads);   /* later rounds: no per-read dedupe (:695-699) */
        m = kb_merge_compact(hist, k, revcom, uniq, cnt);
    }
    free(hist);
    free(uniq);
    free(cnt);
    return found;
}

/* ---- embedding: hd_prob from unsmoothed Hamming distances (a stand-in for the smoothed matrix with the same value
 * structure: P[i][j] = lut[n_nb^2 * ham(kh_i, kh_j)], diagonal = lut[0]; the iteration's cost does not depend on the values) */
void kb_fill_prob(const uint32_t *kh, int64_t n, const float *lut, int per_mismatch, float *P, int threads) {
    kb_threads(threads);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) P[i * n + j] = lut[per_mismatch * kb_ham(kh[i], kh[j])];
}

/* ---- one umap iteration body fused into a single pass over rows [r0, r1) (visualization.py:296-317, taichi_core.py:252-326):
 * q = clip(1 / (1 + d^2), 1e-3, 1 - 1e-3); loss += CE(p, q) for j > i (eps = 1e-10 branches on p); T = q / (1 - q) * (p - q);
 * g[c][i] = sum_{j != i} T * (y[c][i] - y[c][j]) in f32, j ascending (the reference's order; no FMA: -ffp-contract=off).
 * g: float[2][n] (rows outside [r0, r1) untouched); returns the CE partial of these rows (float64; the reference: f32 np.sum). */
double kb_embed_forces_src(const float *P, const uint32_t *kh, const float *lut, int per_mismatch, const float *y, int64_t n,
                           int64_t r0, int64_t r1, float *g, int threads);
double kb_embed_forces(const float *P, const float *y, int64_t n, int64_t r0, int64_t r1, float *g, int threads) {
    return kb_embed_forces_src(P, NULL, NULL, 0, y, n, r0, r1, g, threads);
}
/* the same pass with p looked up on the fly, p_ij = lut[per_mismatch * ham(kh_i, kh_j)] (0 on the diagonal), when P == NULL:
 * no N x N matrix in host memory (10 GB at N = 50 000) -- a cheaper data flow than the reference's, same arithmetic per pair */
double kb_embed_forces_src(const float *P, const uint32_t *kh, const float *lut, int per_mismatch, const float *y, int64_t n,
                           int64_t r0, int64_t r1, float *g, int threads) {
    kb_threads(threads);
    const float lo = 1e-3f, hi = 0.999f, eps = 1e-10f, one = 1.0f;
    double loss = 0.0;
#pragma omp parallel for reduction(+ : loss) schedule(dynamic, 16)
    for (int64_t i = r0; i < r1; ++i) {
        const float xi = y[i], yi = y[n + i];
        const float *p = P ? P + i * n : NULL;
        const uint32_t ki = kh ? kh[i] : 0;
        float gx = 0.0f, gy = 0.0f;
        double l = 0.0;
        for (int64_t j = 0; j < n; ++j) {
            if (j == i) continue;
            const float dx = xi - y[j], dy = yi - y[n + j];
            float q = one / (one + (dx * dx + dy * dy));
            q = q < lo ? lo : (q > hi ? hi : q);
            const float pv = p ? p[j] : lut[per_mismatch * kb_ham(ki, kh[j])];
            if (j > i) {
                float ce;
                if (pv < eps) ce = -logf(one - q);
                else if (pv > one - eps) ce = -logf(q);
                else ce = -pv * logf(q) - (one - pv) * logf(one - q);
                l += (double)ce;
            }
            const float t = q / (one - q) * (pv - q);
            gx = gx + t * dx;
            gy = gy + t * dy;
        }
        g[i] = gx;
        g[n + i] = gy;
        loss += l;
    }
    return loss;
}

/* The same row sums for a SAMPLE of rows whose probabilities arrive as a slab: Pslab[r * n + j] = p(rows[r], j), g = float[2][n_rows]
 * (g[r], g[n_rows + r] of row rows[r]).  The checker of the device's SEQ kernel at sizes where no N x N matrix fits the host
 * (tests/test_gpu_fullsize.py); the loop body is the one above (taichi_core.py:305-326, visualization.py:131-145). */
void kb_embed_forces_rows(const float *Pslab, const int64_t *rows, int64_t n_rows, const float *y, int64_t n, float *g, int threads) {
    kb_threads(threads);
    const float lo = 1e-3f, hi = 0.999f, one = 1.0f;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t i = rows[r];
        const float xi = y[i], yi = y[n + i];
        const float *p = Pslab + r * n;
        float gx = 0.0f, gy = 0.0f;
        for (int64_t j = 0; j < n; ++j) {
            if (j == i) continue;
            const float dx = xi - y[j], dy = yi - y[n + j];
            float q = one / (one + (dx * dx + dy * dy));
            q = q < lo ? lo : (q > hi ? hi : q);
            const float t = q / (one - q) * (p[j] - q);
            gx = gx + t * dx;
            gy = gy + t * dy;
        }
        g[r] = gx;
        g[n_rows + r] = gy;
    }
}

/* y += -(4 g) lr (visualization.py:145,316) */
void kb_embed_update(float *y, const float *g, int64_t n, float lr) {
    for (int64_t i = 0; i < 2 * n; ++i) y[i] = y[i] + (-(4.0f * g[i]) * lr);
}

int kb_max_threads(void) { return omp_get_max_threads(); }
/* OpenMP's thread count for this host thread's later parallel regions, in ANY library of the process (libgomp is shared):
 * bench.py pins the oracle's OpenMP loops to the cgroup's CPU share with it */
void kb_set_threads(int threads) { kb_threads(threads); }
