// counts_internal.h -- the counts handle shared by counts.hip (histogram path) and counts_sort.hip (k >= 17)
#pragma once
#include "common.h"

struct kmap_counts {
    int k = 0;
    int narrow = 1;          // hash dtype uint32 (k < 16)
    int64_t n_uniq = 0;
    void *uniq = nullptr;    // H[n_uniq]
    uint32_t *cnt = nullptr; // uint32[n_uniq]
    size_t cap = 0;          // entries allocated
    uint32_t *bins = nullptr;
    size_t bins_cap = 0;     // bins allocated
    // the 4^k-bin table is ONE per device (scratch arena), shared by all handles: (bins_dev, bins_gen) say which fill of it this
    // handle made; kmap_counts_bins / kmap_counts_finish refuse to touch a table another handle has written since
    int bins_dev = -1;
    uint64_t bins_gen = 0;
};
// does the shared table still hold THIS handle's last histogram? (KMAP_E_STATE + message otherwise)
int kmap_counts_bins_check(const kmap_counts *c, const char *who);


// k >= 17: sort + run-length encode + revcom merge (counts_sort.hip)
int kmap_counts_sort_path(kmap_counts *c, const uint64_t *hash_dev, int64_t n, int k, int merge, int64_t *n_uniq,
                          hipStream_t st);
int kmap_counts_prepare_bins(kmap_counts *c, int k, hipStream_t st);
int kmap_counts_finish_hist(kmap_counts *c, int k, int merge, int64_t *n_uniq, hipStream_t st);
int kmap_counts_hist_hashes(kmap_counts *c, const void *hash_dev, int64_t n, int k, hipStream_t st);
int kmap_counts_reserve_bins(kmap_counts *c, int k);   // allocation only (the partitioned histogram writes every bin)
// 11 <= k <= 16, hashes as uint32: bucket-partitioned histogram without global atomics (counts_part.hip); k = 16: the valid
// windows of the all-T 16-mer (hash 0xFFFFFFFF = the uint32 invalid marker) are counted aside and added with part_add_bin
bool kmap_counts_part_applies(int k, int64_t n);
int kmap_counts_part_hist_u32(kmap_counts *c, const uint32_t *hash_dev, int64_t n, int k, hipStream_t st);
// the same with the keys hashed on the fly from the 2-bit packed reads (+ per-read dedupe skip bits); adds the all-T 16-mer itself
int kmap_counts_part_hist_packed(kmap_counts *c, const uint32_t *codes_dev, const uint16_t *inval_dev, const uint32_t *skip_dev,
                                 int64_t n, int k, hipStream_t st);
// 8 <= k <= 14: the same with 16-bit keys, <= 65 536 bins per bucket and XCD-private bucket streams (counts_fine.hip); hash_dev or
// (codes_dev, inval_dev, skip_dev) as above
bool kmap_counts_fine_applies(int k);
int kmap_counts_fine_hist(kmap_counts *c, const uint32_t *hash_dev, const uint32_t *codes_dev, const uint16_t *inval_dev,
                          const uint32_t *skip_dev, int64_t n, int k, hipStream_t st);
int kmap_counts_part_add_bin(kmap_counts *c, size_t bin, const unsigned long long *extra_dev, hipStream_t st);

// ---- key-space-sharded counting (multi-GPU, 11 <= k <= 16): every rank holds ALL reads and counts only the windows that decide the
// entries of ITS key range [lo, lo + len) of the table -- no table collective.  The entry at position y of the merged table
// (merge_revcom, kmer_count.py:643-685) depends on c(y) and c(rc y) only, so a window with key x is kept when x lies in the range
// (virtual key x - lo: table T1) or, failing that, when rc(x) does (virtual key half + rc(x) - lo: table T2 = the partner counts of
// the range's positions, already transposed: no reverse-complement transpose of the table is needed afterwards).  The virtual keys
// live in a 4^vk-bin table (half = 4^vk / 2 >= len), so the partitioned histogram of a smaller k serves them: at G = 8, k = 14 a
// rank's pass is a k = 13-shaped histogram over a quarter of the windows.
struct kmap_key_range {
    uint32_t lo, len;        // the rank's positions [lo, lo + len)
    uint32_t half;           // first virtual key of the partner table T2 (0: no reverse-complement merge, own keys only)
    int sh;                  // 32 - 2 k
};
// counts_range.hip: ONE pass over the packed reads -> the dense list of the virtual keys of the windows the range keeps (in the stream's
// KMAP_SLOT_HASH scratch, padded with 0xFFFFFFFF to whole 32 768-key chunks); the partitioned histogram of a 4^vk-bin table then runs
// on that list (kmap_counts_part_hist_u32 with k = vk)
int kmap_counts_range_stage(const uint32_t *codes_dev, const uint16_t *inval_dev, const uint32_t *skip_dev, int64_t n, int k, kmap_key_range kr,
                            uint32_t **keys_out, int64_t *n_keys, hipStream_t st);
// compaction of a range-mode table (T1 = bins [0, len), T2 = bins [half, half + len)) into the handle's uniq / cnt arrays
int kmap_counts_finish_key_range(kmap_counts *c, int k, kmap_key_range r, int64_t *n_uniq, hipStream_t st);
// the WHOLE table is in c's bins (small inputs, reads too long for the LDS dedupe): merged in place, positions [first, first + n_bins) compacted
int kmap_counts_finish_hist_slice(kmap_counts *c, int k, int merge, uint64_t first, uint64_t n_bins, int64_t *n_uniq, hipStream_t st);
