/*
 * kmap_hip.h -- C ABI of libkmap_hip.so: the MI355X (gfx950) implementation of kmap's
 * data-parallel hot path.  Plain pointers and sizes only; no torch / numpy types.
 *
 * The reference (chengl7-lab/kmap) has no FFI layer: its de-facto operator API is the set of
 * numpy-in/numpy-out Python functions whose bodies launch Taichi kernels.
 * Every entry point below names the reference operator + kernel it replaces
 * (file:line relative to the reference's src/kmap/).  kmap_amd/_ffi.py is the ctypes binding;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (KMAP_E_*); kmap_last_error() returns a
 *     thread-local message.
 *   - `*_dev` pointers are device (HBM) pointers; plain names are host pointers.  Host-pointer
 *     entry points are blocking.  Device-pointer entry points are asynchronous on `stream`
 *     (a hipStream_t passed as void*, NULL = default stream) unless stated otherwise.
 *   - the caller allocates every output (like the reference: np.empty then kernel); the library
 *     returns heap memory only behind opaque handles.
 *   - hashes are uint32 for k < 16 and uint64 for 16 <= k < 32, invalid hash = all ones,
 *     counts int32 / int64 (kmer_count.py:351-370).  k >= 32 or k <= 0 -> KMAP_E_INVAL.
 */
#ifndef KMAP_HIP_H
#define KMAP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMAP_OK 0
#define KMAP_E_INVAL (-1)   /* bad argument (k range, null pointer, size) */
#define KMAP_E_HIP (-2)     /* HIP runtime error (message has hipGetErrorString) */
#define KMAP_E_NOMEM (-3)   /* device allocation failed */
#define KMAP_E_UNSUP (-4)   /* valid request this build cannot serve */
#define KMAP_E_STATE (-5)   /* handle used out of order */
#define KMAP_E_IO (-6)      /* a file operation of the host-side writers failed (message has strerror) */

/* ---- library / device -------------------------------------------------------------------- */
int kmap_version(void);                       /* 1000*major + minor */
const char *kmap_last_error(void);
int kmap_device_count(int *n);
int kmap_set_device(int dev);
int kmap_get_device(int *dev);
/* arch name of the current device, e.g. "gfx950:sramecc+:xnack-" */
int kmap_device_arch(char *buf, int buflen);

/* raw device memory, so a ctypes caller can keep data resident between calls */
int kmap_malloc(void **dev_ptr, size_t bytes);
int kmap_free(void *dev_ptr);
int kmap_memset(void *dev_ptr, int value, size_t bytes, void *stream);
int kmap_memcpy_h2d(void *dev_dst, const void *host_src, size_t bytes, void *stream);
int kmap_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes, void *stream);
int kmap_memcpy_d2d(void *dev_dst, const void *dev_src, size_t bytes, void *stream);
/* pitched device->host copy: `rows` rows of `width` bytes, device pitch `dpitch`, dense host */
int kmap_memcpy2d_d2h(void *host_dst, size_t hpitch, const void *dev_src, size_t dpitch, size_t width, size_t rows,
                      void *stream);
/* The library caches its transient device buffers (hash arrays, the 4^k-bin histogram table -- 16 GiB at k = 16 --, partition
 * keys) per device between calls.  kmap_scratch_release frees those of at least `min_bytes` on the current device (synchronises
 * it first); counts handles that were between kmap_counts_hist_packed_dev and kmap_counts_finish then fail with KMAP_E_STATE. */
int kmap_scratch_release(size_t min_bytes);
int kmap_stream_sync(void *stream);
int kmap_stream_create(void **stream);
int kmap_stream_destroy(void *stream);
/* HIP-event timing on the stream the kernels run on (bench.py's roofline leg) */
int kmap_event_create(void **ev);
int kmap_event_destroy(void *ev);
int kmap_event_record(void *ev, void *stream);
int kmap_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms); /* synchronises ev_stop */

/* ---- k-mer hashing: comp_kmer_hash_taichi kmer_count.py:449-473, kernels taichi_core.py:3-61
 * hash at every array index; invalid if the window touches a 255 byte or passes the end. */
int kmap_hash_kmers_u32_dev(const uint8_t *seq_dev, int64_t n, int k, uint32_t *out_dev, void *stream);
int kmap_hash_kmers_u64_dev(const uint8_t *seq_dev, int64_t n, int k, uint64_t *out_dev, void *stream);
int kmap_hash_kmers_u32(const uint8_t *seq, int64_t n, int k, uint32_t *out);
int kmap_hash_kmers_u64(const uint8_t *seq, int64_t n, int k, uint64_t *out);

/* ---- per-read de-duplication: remove_duplicate_hash_per_seq kmer_count.py:743-760
 * borders: (n_seq,2) int64 [start, end); keeps the first occurrence of each value per read. */
int kmap_dedupe_per_read_u32_dev(uint32_t *hash_dev, int64_t n, const int64_t *borders_dev, int64_t n_seq,
                                 void *stream);
int kmap_dedupe_per_read_u64_dev(uint64_t *hash_dev, int64_t n, const int64_t *borders_dev, int64_t n_seq,
                                 void *stream);
int kmap_dedupe_per_read_u32(uint32_t *hash, int64_t n, const int64_t *borders, int64_t n_seq);
int kmap_dedupe_per_read_u64(uint64_t *hash, int64_t n, const int64_t *borders, int64_t n_seq);

/* ---- reverse complement: get_revcom_hash_arr kmer_count.py:613-623, taichi_core.py:181-224 */
int kmap_revcom_u32_dev(const uint32_t *in_dev, int64_t n, int k, uint32_t *out_dev, void *stream);
int kmap_revcom_u64_dev(const uint64_t *in_dev, int64_t n, int k, uint64_t *out_dev, void *stream);
int kmap_revcom_u32(const uint32_t *in, int64_t n, int k, uint32_t *out);
int kmap_revcom_u64(const uint64_t *in, int64_t n, int k, uint64_t *out);

/* ---- Hamming 1-vs-N: cal_hamming_dist / _head / _tail kmer_count.py:494-577,
 * kernels taichi_core.py:63-177.  dist = #nonzero 2-bit groups of ((h >> shift_bits) ^ cons) over
 * the low `clen` groups.  full: shift 0, clen k; head: shift 2(k-clen); tail: shift 0. */
int kmap_hamdist_1vN_u32_dev(const uint32_t *h_dev, int64_t n, uint32_t cons, int shift_bits, int clen,
                             uint8_t *out_dev, void *stream);
int kmap_hamdist_1vN_u64_dev(const uint64_t *h_dev, int64_t n, uint64_t cons, int shift_bits, int clen,
                             uint8_t *out_dev, void *stream);
int kmap_hamdist_1vN_u32(const uint32_t *h, int64_t n, uint32_t cons, int shift_bits, int clen, uint8_t *out);
int kmap_hamdist_1vN_u64(const uint64_t *h, int64_t n, uint64_t cons, int shift_bits, int clen, uint8_t *out);

/* ---- masking: mask_input kmer_count.py:580-610 (one kmer_len, several consensuses).
 * In place on the uint8 sequence array; hashes are taken from the array as it is on entry. */
int kmap_mask_hamball_dev(uint8_t *seq_dev, int64_t n, int k, const uint64_t *cons, const int32_t *radius,
                          int n_cons, void *stream);                       /* cons/radius: host arrays */
int kmap_mask_hamball(uint8_t *seq, int64_t n, int k, const uint64_t *cons, const int32_t *radius, int n_cons);

/* ---- counting: count_uniq_hash kmer_count.py:476-491 (+ remove_duplicate..., merge_revcom
 * kmer_count.py:643-685) fused on device.  Two-call pattern: *_run returns the number of unique
 * k-mers, *_fetch copies them out (uniq ascending like np.unique; after merge_revcom in the
 * reference's order).  The counts object owns device memory; free it with kmap_counts_destroy. */
typedef struct kmap_counts kmap_counts;
int kmap_counts_create(kmap_counts **c);
int kmap_counts_destroy(kmap_counts *c);
/* from the sequence array (hash + optional per-read dedupe + count + optional revcom merge) */
int kmap_counts_run_seq_dev(kmap_counts *c, const uint8_t *seq_dev, int64_t n, const int64_t *borders_dev,
                            int64_t n_seq, int k, int dedupe_per_read, int merge_revcom, int64_t *n_uniq,
                            void *stream);
/* from a materialised hash array (drop-in for count_uniq_hash; invalid hashes dropped) */
int kmap_counts_run_hashes_dev(kmap_counts *c, const void *hash_dev, int64_t n, int k, int merge_revcom,
                               int64_t *n_uniq, void *stream);
/* load counts computed earlier (e.g. from kmer_count/k{k}.pkl, find_motif motif_discovery.py:621-624):
 * uniq uint32/uint64 by k, cnt int32/int64 by k, host arrays */
int kmap_counts_load(kmap_counts *c, const void *uniq, const void *cnt, int64_t n_uniq, int k);
/* uniq_out: uint32[n_uniq] (k<16) or uint64[n_uniq]; cnt_out: int32 (k<16) or int64 */
int kmap_counts_fetch(kmap_counts *c, void *uniq_out, void *cnt_out);
/* the same copy on a caller-chosen stream through pinned staging buffers (conversion on host threads): lets a background host
 * thread drain a finished table (multi-GB at k >= 14) while the default stream keeps counting into another handle */
int kmap_counts_fetch_stream(kmap_counts *c, void *uniq_out, void *cnt_out, void *stream);
/* elements [first, first + count) of one array of the table in the reference's dtype: which = 0 unique hashes, 1 counts */
int kmap_counts_fetch_range(kmap_counts *c, int which, int64_t first, int64_t count, void *out, void *stream);
/* the same range written into an open file instead (pwrite at file_offset; the descriptor's own position is untouched): pinned
 * staging, the next chunk crossing PCIe while the current one is written, counts widened on the device -- the k{k}.pkl writers of
 * multi-GB tables (reference motif_discovery.py:642-645 pickles [k, uniq, cnt]) */
int kmap_counts_write_range(kmap_counts *c, int which, int64_t first, int64_t count, int fd, int64_t file_offset, void *stream);
/* device addresses of the resident table (uniq uint32 for k < 16 else uint64; cnt uint32 whatever k; counts > 2^32 wrap like the
 * histogram bins); valid until the next count / load / destroy on the handle.  Lets sample_disp_kmer (motif_discovery.py:812-921)
 * label a multi-GB table where it lies instead of re-reading k{k}.pkl */
int kmap_counts_table_dev(kmap_counts *c, void **uniq_dev, void **cnt_dev, int64_t *n_uniq);
int kmap_counts_total(kmap_counts *c, int64_t *total);          /* sum of counts */
/* the top_k (<= 16) most frequent k-mers: largest count first, ties by lowest index (np.argpartition's tie order,
 * motif_discovery.py:661, is numpy-specific; this rule is used when the arrays are too large to fetch per trial).
 * idx_out int64[top_k], kh_out uint64[top_k], cnt_out int64[top_k]; returns the number found in *n_found. */
int kmap_counts_topk(kmap_counts *c, int top_k, int64_t *idx_out, uint64_t *kh_out, int64_t *cnt_out, int *n_found);
/* Hamming-ball mass of candidates over the counted k-mers: find_motif motif_discovery.py:666-673 */
int kmap_counts_hamball_mass(kmap_counts *c, const uint64_t *cands, int n_cand, int radius, int revcom,
                             double *mass_out);

/* ---- 2-bit-packed reads (device resident): the uint8 array contract (kmer_count.py:244-347) packed 16 positions per
 * group: codes uint32[groups] (first base most significant) + invalid bitmask uint16[groups] (255 / past the end),
 * groups = kmap_packed_groups(n): the data groups, at least two all-invalid halo groups, an even total.  Masking ORs bits into the invalid mask; the
 * codes never change.  The packed entry points do what their uint8 counterparts above do, on the packed stream. */
int64_t kmap_packed_groups(int64_t n);
int kmap_pack_reads_dev(const uint8_t *seq_dev, int64_t n, uint32_t *codes_dev, uint16_t *inval_dev, void *stream);
int kmap_unpack_reads_dev(const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n, uint8_t *seq_out_dev,
                          void *stream);
/* out_dev: uint32[n] (k < 16) or uint64[n] */
int kmap_hash_kmers_packed_dev(const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n, int k, void *out_dev,
                               void *stream);
int kmap_counts_run_packed_dev(kmap_counts *c, const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n,
                               const int64_t *borders_dev, int64_t n_seq, int k, int dedupe_per_read, int merge_revcom,
                               int64_t *n_uniq, void *stream);
/* multi-GPU counting by KEY SPACE (11 <= k <= 16; reference kmer_count.py:476-491,643-685): every rank holds all packed reads and
 * computes the positions [first_bin, first_bin + n_bins) -- in key order, first_bin a multiple of 8 -- of the table
 * kmap_counts_run_packed_dev would produce from the same input, from the windows that decide those entries alone (a window with k-mer
 * x is kept when x, or else rc(x), lies in the range): no table collective, the histogram passes shrink with the number of ranks.
 * The handle then holds the shard (kmap_counts_table_dev / _fetch); shards concatenated in rank order are the single-GPU table. */
int kmap_counts_run_packed_range_dev(kmap_counts *c, const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n,
                                     const int64_t *borders_dev, int64_t n_seq, int k, int dedupe_per_read, int merge_revcom,
                                     uint64_t first_bin, uint64_t n_bins, int64_t *n_uniq, void *stream);
/* multi-GPU counting: every rank histograms ITS reads (`kmap_counts_hist_packed_dev`, k <= 16, bins zeroed
 * first; the per-read dedupe is local to a read), the caller all-reduces the 4^k uint32 bins in place
 * (`kmap_counts_bins` returns the device pointer), then every rank compacts identically (`kmap_counts_finish`). */
int kmap_counts_hist_packed_dev(kmap_counts *c, const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n,
                                const int64_t *borders_dev, int64_t n_seq, int k, int dedupe_per_read, void *stream);
/* All handles of a device share ONE table: kmap_counts_bins / kmap_counts_finish return KMAP_E_STATE if another handle has
 * counted (or kmap_scratch_release ran) since this handle's kmap_counts_hist_packed_dev. */
int kmap_counts_bins(kmap_counts *c, void **bins_dev, int64_t *n_bins);
int kmap_counts_finish(kmap_counts *c, int k, int merge_revcom, int64_t *n_uniq, void *stream);
/* Key-range-sharded variant (11 <= k <= 16) of the same job: instead of all-reducing the whole table, every rank ends up with
 * the summed counts of ITS slice of the bins and compacts that slice; the (uniq, cnt) shards are then all-gathered in rank order.
 *   kmap_counts_hist_packed_dev                         local table
 *   [merge_revcom] kmap_counts_presence_dev             one nibble per bin (0 / 1): bin x -> byte x / 2, low nibble for even x;
 *                  caller: SUM all-reduce of the 4^k / 2 bytes (<= 15 ranks)
 *                  kmap_counts_merge_presence_dev       table <- local part of the merged counts (which pair member survives is
 *                                                        decided by the global presence), so that merged(sum) == sum(merged)
 *   caller: SUM-reduce slice r of the table to rank r (in place)
 *   kmap_counts_finish_range(first_bin, n_bins)         compaction of the own slice into the handle (first_bin % 8 == 0)
 *   caller: all-gather of the shards (kmap_counts_table_dev), kmap_counts_adopt_dev(concatenation)
 * Replaces the all-reduce of /root/reference's single-process count_uniq_hash + merge_revcom (kmer_count.py:643-760) across ranks. */
int kmap_counts_presence_dev(kmap_counts *c, int k, void *nib_dev, void *stream);
int kmap_counts_merge_presence_dev(kmap_counts *c, int k, const void *nib_dev, void *stream);
int kmap_counts_finish_range(kmap_counts *c, int k, int merged, uint64_t first_bin, uint64_t n_bins, int64_t *n_uniq, void *stream);
int kmap_counts_adopt_dev(kmap_counts *c, const void *uniq_dev, const void *cnt_dev, int64_t n_uniq, int k);
/* planes_dev (optional, from kmap_pack_planes_dev): with it and k <= 16 the Hamming-ball test of all windows runs bit-sliced
 * over positions (csrc/bitslice.hip) instead of window by window; NULL keeps the per-window kernels.  Results are identical. */
int kmap_mask_hamball_packed_dev(const uint32_t *codes_dev, uint16_t *inval_dev, int64_t n, int k, const uint64_t *cons,
                                 const int32_t *radius, int n_cons, const uint32_t *planes_dev, void *stream);
/* positions [0, m) of the packed reads become invalid (bit 15 - i of group word g = position 16 g + i): what a consensus within its
 * radius of the all-T k-mer does to the k - 1 positions behind a separator (kmer_count.py:580-610) when that separator belongs to
 * the previous rank's shard of the reads (kmap_amd/distributed.py DistDeviceSeq.mask); stream-ordered like every other call. */
int kmap_inval_set_prefix_dev(uint16_t *inval_dev, int64_t m, void *stream);  /* cons/radius: host */
/* bit planes of the packed reads, built once per upload: planes[2w] / planes[2w + 1] = the high / low bits of the 32 base
 * codes of groups 2w, 2w + 1, first position most significant (an opaque input of the two calls above -- the layout may change
 * between library versions); planes_dev: uint32[kmap_packed_groups(n)] (always an even number of groups); both arrays 8-byte aligned */
int kmap_pack_planes_dev(const uint32_t *codes_dev, int64_t n, uint32_t *planes_dev, void *stream);

/* ---- motif occurrence scan: get_motif_occurence motif_discovery.py:1422-1477.
 * For every read and one consensus: positions p in the reference's slice [0 : len-k+1] whose
 * min(fwd, revcom) distance is <= radius AND equals the read's minimum.  Two-call: run returns the
 * total number of hits; fetch writes per-read hit counts (int32[n_seq]), min distance (int8[n_seq],
 * -1 = none) and the concatenated positions (int32[total], read order, ascending per read). */
typedef struct kmap_scan kmap_scan;
int kmap_scan_create(kmap_scan **s);
int kmap_scan_destroy(kmap_scan *s);
int kmap_scan_run_dev(kmap_scan *s, const uint8_t *seq_dev, int64_t n, const int64_t *borders_dev, int64_t n_seq,
                      int k, uint64_t cons, int radius, int revcom, int64_t *total_hits, void *stream);
/* Optional: the caller declares that read s of `borders_dev` is [s * stride, s * stride + read_len) (fixed-length reads, as the reference's
 * FASTA encoder lays them out: motif_discovery.py:1422-1477 walks such reads one by one).  Verified on the device; *accepted = 1 when
 * true.  Runs of this handle on the same border array then derive the borders from s instead of loading them (16 B per read).  The
 * declaration ends when the handle runs on other borders; renew it if the CONTENT of the array at that address changes. */
int kmap_scan_declare_uniform(kmap_scan *s, const int64_t *borders_dev, int64_t n_seq, int64_t read_len, int64_t stride, int *accepted, void *stream);
int kmap_scan_run_packed_dev(kmap_scan *s, const uint32_t *codes_dev, const uint16_t *inval_dev, int64_t n,
                             const int64_t *borders_dev, int64_t n_seq, int k, uint64_t cons, int radius, int revcom,
                             int64_t *total_hits, const uint32_t *planes_dev /* optional, see above */, void *stream);
int kmap_scan_fetch(kmap_scan *s, int32_t *hits_per_read, int8_t *min_dist, int32_t *positions);
/* the last run's results without (or before) a fetch: the two numbers scan_motif's candidate table needs of a hit list --
 * reads with a hit (get_motif_seq_num, motif_discovery.py:1345-1393) and the largest per-read hit count (> 20 triggers the
 * reference's subsampling, :1466-1469) -- and a fetch on the caller's stream, for worker threads that write the occurrence
 * CSV while the launching thread goes on (the handle keeps the lists until its next run) */
int kmap_scan_summary(kmap_scan *s, int64_t *reads_with_hits, int32_t *max_hits, void *stream);
int kmap_scan_fetch_stream(kmap_scan *s, int32_t *hits_per_read, int32_t *positions, void *stream);
/* device addresses of the last run's lists (hits int32[n_seq], positions int32[total]; valid until the handle's next run), for
 * callers that gather the shards of a read-sharded scan with a device collective ("scan hits: gather in read order") */
int kmap_scan_result_dev(kmap_scan *s, void **hits_dev, void **pos_dev, int64_t *n_seq, int64_t *total);
/* hit counts as bytes (narrowed on the device; for lists whose summary max_hits <= 255, larger counts saturate) */
int kmap_scan_fetch_stream_u8(kmap_scan *s, uint8_t *hits_u8, int32_t *positions, void *stream);

/* host-side writer of the occurrence table (gen_motif_occurence_file motif_discovery.py:1396-1419):
 * rows "seq_ind;loc,loc;...;seq_len" for every read with at least one hit; per consensus c the arrays
 * hits[c] (int32[n_seq]) and pos[c] (int32, concatenated in read order, already subsampled/sorted).
 * header is written verbatim as the first line.  Native because the reference's per-read Python
 * formatting dominates at 10^7 reads. */
int kmap_write_occurrence_csv(const char *path, const char *header, int64_t n_seq, int n_cons,
                              const int32_t *const *hits, const int32_t *const *pos, const int64_t *read_len,
                              int64_t *rows_written);
/* the same file from byte-sized hit counts (kmap_scan_fetch_stream_u8) */
int kmap_write_occurrence_csv_u8(const char *path, const char *header, int64_t n_seq, int n_cons,
                              const uint8_t *const *hits, const int32_t *const *pos, const int64_t *read_len,
                              int64_t *rows_written);

/* one line of the co-occurrence distance file (write_co_occurence_dist_arr motif_discovery.py:1143-1162): the n values formatted like
 * Python's f"{x:.2f}", tab-separated, '\n' at the end, written to the open descriptor `fd` at its position (n = 0: an empty line).
 * Native for the same reason as the occurrence table: 3 M values per motif pair at 10^7 reads. */
int kmap_write_f2_tsv_line(int fd, const double *v, int64_t n);
/* np.median of every cell of one motif's hit list (get_motif_co_occurence_mat motif_discovery.py:1189-1253 takes it per row of the
 * CSV): hits[r] ascending locations of read r, back to back in pos[n_pos]; med[r] = mean of the two middle ones, NaN when empty */
int kmap_cell_medians_i32(const int32_t *hits, const int32_t *pos, int64_t n_seq, int64_t n_pos, double *med);

/* ---- consumers of the counts / the hit list -------------------------------------------
 * Hamming ball of a consensus over counted k-mers (ex_hamball_kh_arr motif_discovery.py:924-975) fused with the
 * position count matrix (cal_cnt_mat :978-986).  uniq/cnt are HOST arrays as stored in k{k}.pkl (u32+i32 for k<16,
 * u64+i64 otherwise); members whose reverse complement is strictly closer are re-oriented (:968-973); output order =
 * input order.  out_kh/out_cnt: caller-allocated, capacity n; *n_out = members; cnt_mat (optional) int64[4][k]. */
int kmap_hamball_extract(const void *uniq_host, const void *cnt_host, int64_t n, int k, uint64_t conseq_kh, int max_ham_dist,
                         int revcom_mode, void *out_kh, void *out_cnt, int64_t *n_out, int64_t *cnt_mat);
/* the same ball over the table a counts handle still holds in HBM (scan_motif keeps the table of the longest final consensus' k for
 * the labelled sampling: the Hamming-ball matrices need not read the multi-GB k{k}.pkl back).  Two-call pattern: *n_out = members;
 * they are written (hashes u32 / u64, counts i32 / i64 by k as above, table order) only when cap >= *n_out. */
int kmap_counts_hamball_extract(kmap_counts *c, uint64_t conseq_kh, int max_ham_dist, int revcom_mode, int64_t cap, void *out_kh,
                                void *out_cnt, int64_t *n_out, int64_t *cnt_mat);
/* motif position density (get_motif_pos_density motif_discovery.py:1255-1327): for every read with m > 0 hits,
 * density[x] += (sum_i normpdf(x; loc_i / (seq_len - kmer_len + 1), x_step)) / m, f64.  HOST arrays: hits int32[n_seq],
 * offs int64[n_seq+1] (exclusive prefix sums of hits), pos int32[offs[n_seq]], seq_len int64[n_seq]; density f64[nx].
 * Per-term arithmetic follows scipy's norm.pdf operation order; reads are summed in chunks of 256 in read order. */
int kmap_pos_density(const int32_t *hits, const int64_t *offs, const int32_t *pos, const int64_t *seq_len, int64_t n_seq,
                     int kmer_len, const double *x_arr, int nx, double x_step, double *density);

/* labelled sampling of the counted k-mers on the device (sample_disp_kmer motif_discovery.py:812-921), for k-mer tables too
 * large for the reference's n_conseq x n_uniq host matrices.  label: nearest consensus by head (forward) / tail (reverse
 * complement) partial Hamming distance, distances above the consensus' own radius count as k, first minimum wins, noise
 * label n_cons when the minimum exceeds radius_k; members whose reverse complement is closer are re-oriented IN uniq_dev. */
int kmap_label_kmers_dev(void *uniq_dev, int64_t n, int k, int n_cons, const uint64_t *cons_kh, const int32_t *cons_len,
                         const int32_t *cons_radius, int radius_k, int revcom_mode, uint8_t *label_dev, void *stream);
/* per label l < n_labels (<= 64): sum of the counts (np.bincount weights) and number of members; cnt int32 or int64 */
int kmap_label_sums_dev(const uint8_t *label_dev, const void *cnt_dev, int cnt64, int64_t n, int n_labels, int64_t *weight_sums,
                        int64_t *member_counts);
/* excl_dev[0..n] (uint64) = exclusive prefix sums of label c's weights (counts; 1 per member if cnt_dev is NULL), [n] = total */
int kmap_label_prefix_dev(const uint8_t *label_dev, const void *cnt_dev, int cnt64, int64_t n, int c, uint32_t *scratch_dev,
                          uint64_t *excl_dev, void *stream);
/* idx_out[j] = first i whose inclusive prefix exceeds targets[j] (np.searchsorted(cdf, x, "right") on integer weights) */
int kmap_prefix_search_dev(const uint64_t *excl_dev, int64_t n, const int64_t *targets, int64_t m, int64_t *idx_out);
/* ascending indices of label c's members (excl_dev from kmap_label_prefix_dev with cnt_dev = NULL; m = excl[n]) */
int kmap_label_members_dev(const uint8_t *label_dev, const uint64_t *excl_dev, int64_t n, int c, int64_t m, int64_t *idx_out);
/* out[j] = src_dev[idx[j]], elements of 1, 4 or 8 bytes; idx / out on the host */
int kmap_gather_dev(const void *src_dev, int elem_bytes, const int64_t *idx, int64_t m, void *out);

/* ---- FASTA -> uint8 array contract (host side; proc_input / dna2arr / convert_fasta_to_binary, kmer_count.py:182-347).
 * Header lines start with '>', sequence lines are concatenated with white space removed, A/C/G/T (either case) -> 0..3,
 * anything else -> 255, one 255 separator after every record; borders [start, end) with end = index of the separator; text
 * before the first header is ignored.  Two-call pattern: open reports the sizes, read fills the caller's arrays.  A plain file
 * is mapped and cut behind newlines into ranges that host threads walk (KMAP_IO_THREADS, default <= 16): open counts them, read
 * encodes each range straight into seq_out / borders_out; a gzip stream is encoded by one thread while it is inflated. */
typedef struct kmap_fasta kmap_fasta;
int kmap_fasta_open(const char *path, kmap_fasta **f, int64_t *n_bytes, int64_t *n_seq);
int kmap_fasta_read(kmap_fasta *f, uint8_t *seq_out, int64_t *borders_out);
int kmap_fasta_close(kmap_fasta *f);

/* ---- synthetic workload generator (benchmarks / tests; no reference operator corresponds to it): seeded reads in the array
 * contract above, generated in HBM -- BASELINE config C5 is 15 GB, minutes of numpy on the host.  Fixed-length reads, uniform
 * bases; the first fractions[0] * n_reads reads carry motif 0, the next fractions[1] * n_reads motif 1, ... at a uniform
 * position with per-base substitution rate mutation_rate (the style of the reference's tests/kmap_tests.py:75-114).
 * seq_dev: uint8[n_reads * (read_len + 1)], 16-byte aligned; borders_dev (optional): int64[n_reads][2]; motif_codes: the
 * motifs' base codes 0..3 concatenated (host), motif_len / fractions: host arrays of n_motifs <= 4 entries. */
int kmap_synth_reads_dev(uint8_t *seq_dev, int64_t *borders_dev, int64_t n_reads, int read_len, uint64_t seed,
                         const uint8_t *motif_codes, const int32_t *motif_len, const double *fractions, int n_motifs,
                         double mutation_rate, void *stream);

/* ---- all-pairs Hamming matrix: cal_samp_kmer_hamdist_mat motif_discovery.py:759-808
 * (one launch instead of n_uniq launches + Python block expansion).  kh: N hashes (already
 * expanded by counts), label: N int32; pairs sharing label l with clen[l] < k are compared on the
 * first clen[l] bases only.  Writes rows [row0, row0+nrows) as uint8 with leading dimension ld.
 * Fastest path (k <= 16, N >= 4096): ld = kmap_hamdist_pitch(n) -- a multiple of 4 KiB with an odd number of 4-KiB chunks
 * per row -- and a 4-KiB aligned out_dev; any other ld >= n works through the general kernel (ld % 16 == 0 and a 16-byte
 * aligned out_dev keep its vector stores). */
int64_t kmap_hamdist_pitch(int64_t n);
int kmap_hamdist_matrix_u32_dev(const uint32_t *kh_dev, const int32_t *label_dev, int64_t n, int k,
                                const int32_t *clen, int n_lab, int64_t row0, int64_t nrows, uint8_t *out_dev,
                                int64_t ld, void *stream);                 /* clen: host array */
int kmap_hamdist_matrix_u64_dev(const uint64_t *kh_dev, const int32_t *label_dev, int64_t n, int k,
                                const int32_t *clen, int n_lab, int64_t row0, int64_t nrows, uint8_t *out_dev,
                                int64_t ld, void *stream);
/* host convenience: dense n x n uint8 out */
int kmap_hamdist_matrix_u8(const uint64_t *kh, const int32_t *label, int64_t n, int k, const int32_t *clen,
                           int n_lab, uint8_t *out);

/* ---- kNN smoothing: knn_smooth visualization.py:90-109, kernel taichi_core.py:227-249
 * integer form: sums[i,j] = sum_{a in nb[i], b in nb[j]} D[a,b]  (exact; S = f32(sums)/n_nb/n_nb),
 * diagonal forced to 0.  D: uint8 rows [0,n) with leading dimension ldd; nb: int32 [n, n_nb];
 * writes uint16 rows [row0,row0+nrows) with leading dimension lds. */
int kmap_knn_sums_u8_dev(const uint8_t *D_dev, int64_t ldd, const int32_t *nb_dev, int64_t n, int n_nb,
                         int64_t row0, int64_t nrows, uint16_t *sums_dev, int64_t lds, void *stream);
/* the same sums straight from the k-mers, without reading D: the double sum over neighbour pairs of mismatching bases is
 * n_nb^2 k minus the dot product of two per-position base-count profiles (plus a tail correction per short consensus label,
 * the matrix rule of cal_samp_kmer_hamdist_mat motif_discovery.py:789-800); identical to kmap_knn_sums_u8_dev on the matrix
 * kmap_hamdist_matrix_* writes for (kh, label, clen).  k <= 16, at most 4 labels with clen < k, n_nb^2 k <= 65535; returns
 * KMAP_E_UNSUP otherwise (use the matrix-based entry point). */
/* n_nb | KMAP_KNN_NATURAL_DIAG: S[i][i] keeps the value the formula gives it (= S[i][i'] for a repeated k-mer i') instead of the reference's
 * 0 (visualization.py:103,107).  Nothing on the embedding path reads the diagonal (taichi_core.py:320-326 skips j == i, the loss takes
 * i < j), and with it the rows of a repeated k-mer are equal byte for byte, so that they can be stored once (kmap_embed_set_row_map). */
#define KMAP_KNN_NATURAL_DIAG (1 << 30)
int kmap_knn_sums_kmers_u32_dev(const uint32_t *kh_dev, const int32_t *label_dev, int64_t n, int k, const int32_t *clen, int n_lab,
                                const int32_t *nb_dev, int n_nb, int64_t row0, int64_t nrows, uint16_t *sums_dev, int64_t lds,
                                void *stream);
int kmap_knn_sums_kmers_u64_dev(const uint64_t *kh_dev, const int32_t *label_dev, int64_t n, int k, const int32_t *clen, int n_lab,
                                const int32_t *nb_dev, int n_nb, int64_t row0, int64_t nrows, uint16_t *sums_dev, int64_t lds,
                                void *stream);
/* deterministic device k-NN selection for rows [row0,row0+nrows): the n_nb smallest entries of each row of D,
 * ties broken by the lowest column index (self included, like the reference's argpartition over the full row).
 * The reference uses np.argpartition (visualization.py:100) whose choice among ties is numpy/ISA specific; this
 * rule is used where the int64 matrix is never materialised on the host (N > 16384).  nb_out: int32 [nrows, n_nb]. */
int kmap_knn_select_u8_dev(const uint8_t *D_dev, int64_t ldd, int64_t n, int n_nb, int64_t row0, int64_t nrows,
                           int32_t *nb_out_dev, void *stream);
/* Helpers of the host-side neighbour choice (the reference's np.argpartition call, visualization.py:100, run on rows streamed back
 * from the device matrix): the matrix expands every sampled k-mer to its count (motif_discovery.py:759-772, "each uniq kmer is
 * expanded uniq_kmer_cnt times"), the row of a repeated k-mer equals the row above it, and equal rows partition alike -- fresh_dev[lr] (uint8) = row row0 + lr differs from the row above
 * (lr = 0: 1); out row o = row idx_dev[o] of D (n bytes, pitch ldo): the rows that need a partition, compacted for the copy. */
int kmap_rows_fresh_u8_dev(const uint8_t *D_dev, int64_t ldd, int64_t n, int64_t row0, int64_t nrows, uint8_t *fresh_dev, void *stream);
int kmap_gather_rows_u8_dev(const uint8_t *D_dev, int64_t ldd, int64_t n, const int32_t *idx_dev, int64_t n_idx, uint8_t *out_dev,
                            int64_t ldo, void *stream);
/* generic float form in the reference's summation order (any distance matrix) */
int kmap_knn_smooth_f32(const float *D, const int32_t *nb, int64_t n, int n_nb, float *S_out);

/* ---- embedding operators (drop-in L3): visualization.py:131-176,235-256, taichi_core.py:252-326 */
int kmap_ld_prob_mat_f32(const float *ld_2xn, int64_t n, float *q_out_nxn);          /* incl. clip */
int kmap_cross_entropy_f32(const float *p_nxn, const float *q_nxn, int64_t n, float *loss_out);
int kmap_gradient_loss_f32(const float *p_nxn, const float *q_nxn, const float *ld_2xn, int64_t n,
                           float *grad_out_2xn);                                       /* incl. x4 */

/* ---- device-resident embedding loop: umap visualization.py:270-326 ------------------------
 * The session owns the high-dimensional probabilities for rows [row0,row0+nrows) of the N x N
 * problem (either a float matrix, or uint16 neighbour sums + a LUT of (n_nb^2 * k + 1) floats
 * computed on the host with the reference's numpy expression chain), the 2 x N coordinates,
 * n_best snapshots and the scalar loop state.  One `step` = one reference iteration:
 * q -> loss -> best-list insert -> early-stop test -> gradient -> update -> jitter. */
typedef struct kmap_embed kmap_embed;
#define KMAP_EMBED_FAST 0      /* wavefront-parallel row sums (order differs from the reference) */
#define KMAP_EMBED_SEQ 1       /* one lane per row, j ascending, no FMA: the reference's f32 order */
int kmap_embed_create(kmap_embed **e, int64_t n, int64_t row0, int64_t nrows, int n_best, float learning_rate,
                      int mode);
/* FAST mode sharded over `world` ranks with each unordered pair evaluated once: rank r owns the 256-row blocks
 * I = r, r + world, ... (cyclic: the upper-triangle work of a block shrinks with I).  Its probability rows are passed block
 * after block (local block b = rows [256 (r + world b), +256) of the matrix; kmap_embed_cyclic_blocks gives their number).
 * kmap_embed_forces then writes partial gradients for ALL points (rows and columns the rank touched) into grad_dev -- the
 * ranks' buffers are summed by the all-reduce (a true sum here, unlike the row-sharded sessions' concatenation). */
int kmap_embed_create_cyclic(kmap_embed **e, int64_t n, int world, int rank, int n_best, float learning_rate);
int64_t kmap_embed_cyclic_blocks(int64_t n, int world, int rank);
int kmap_embed_destroy(kmap_embed *e);
int kmap_embed_set_prob_f32(kmap_embed *e, const float *p_rows_dev, int64_t ld);       /* device rows */
int kmap_embed_set_prob_lut(kmap_embed *e, const uint16_t *sums_rows_dev, int64_t ld, const float *lut,
                            int lut_len);                                               /* lut: host */
/* SEQ sessions, after kmap_embed_set_prob_lut: the sums of session row r are row rowmap_dev[r] (int32, device, kept by the caller
 * like the sums) of a matrix of src_rows rows -- for samples that repeat their k-mers in runs (motif_discovery.py:759-772), whose
 * repeated rows are stored once.  The map starts at 0, ends at src_rows - 1 and steps by 0 or 1 (checked).  NULL: one stored row per
 * session row again.  Results are those of the expanded matrix, bit for bit (same values through another address). */
int kmap_embed_set_row_map(kmap_embed *e, const int32_t *rowmap_dev, int64_t src_rows);
/* init: 2 x N coordinates, n_best placeholder snapshots (n_best x 2 x N), host arrays */
int kmap_embed_set_coords(kmap_embed *e, const float *coords_2xn, const float *placeholders);
/* jitter normals N(0, 0.01) pre-drawn (float64, as numpy draws them) from the host RNG stream,
 * consumed in order; replaces any earlier buffer (the consumed count keeps running) */
int kmap_embed_set_jitter(kmap_embed *e, const double *normals, int n_normals);
/* external (e.g. torch-allocated) buffers for multi-GPU: grads of the local rows are written to
 * grad_dev (2 x N, local rows filled, rest 0) and loss partial to loss_dev (double[1]) by
 * `kmap_embed_forces` (entries of other rows are NOT touched: zero the buffer before every call,
 * an in-place all-reduce leaves the other ranks' rows behind); after the all-reduce the caller runs
 * `kmap_embed_apply`.  NULL pointers select the session's own buffers. */
int kmap_embed_forces(kmap_embed *e, float *grad_dev_2xn, double *loss_dev, void *stream);
int kmap_embed_apply(kmap_embed *e, const float *grad_dev_2xn, const double *loss_dev, void *stream);
/* one-message form of the pair above (the multi-GPU loop: ONE collective per iteration, no memset, no float64 collective):
 * msg_dev = float[kmap_embed_msg_floats(n)] = [gradient 2 x N | the rank's loss partial as an exact integer: f64 -> 48.48 fixed
 * point -> six 16-bit limbs stored as floats | flag | pad].  `kmap_embed_forces_msg` writes the rank's entries and its limbs,
 * the caller SUM-all-reduces the whole buffer as float32 (limb sums of up to 256 ranks are exact in any order, so every rank
 * decodes the same loss whatever algorithm the collective uses), `kmap_embed_apply_msg` decodes and applies it -- and, for
 * row-sharded sessions, zeroes every entry it has read, so the next message again starts from x + 0 + ... + 0.  The buffer must
 * be zero before the first call. */
int64_t kmap_embed_msg_floats(int64_t n);
/* Peer-direct exchange of the iteration message ("one-hop direct all-gather over the xGMI links" instead of a
 * collective-library call per iteration).  Every rank owns a fine-grained receive area of 2 (iteration parity) x world slots
 * of kmap_embed_msg_floats(n) floats + one flag word per slot, exported through an IPC handle; after kmap_peer_connect it holds
 * device pointers to all peers' areas.  kmap_embed_step_peer runs n_iter iterations without any host involvement between them:
 * forces_msg into a local buffer -> push kernel (the message is copied into slot [parity][rank] of EVERY rank's area, then a
 * system-scope release store of the iteration number into the slot's flag) -> apply kernel (waits -- bounded -- for the world
 * flags of the iteration, adds the world slots in rank order, decodes the loss limbs, applies).  Sums in rank order are the
 * same on every rank, so all ranks take identical decisions.  A wait that exceeds its bound (kmap_peer_set_timeout_ms, default
 * 10 s) sets a sticky flag (kmap_peer_status); the blocks that gave up do not apply the iteration, the host raises after the
 * segment and the coordinates are to be discarded: no kernel spins for ever.
 * Before the first iteration a binding validates the mapping: kmap_peer_bus_id of every rank -> kmap_peer_can_access (same device,
 * or hipDeviceCanAccessPeer), then kmap_peer_hello_push (a tagged word into every rank's area) -> barrier -> kmap_peer_hello_check
 * (this rank's area holds every rank's word): a wrong handle or a link that does not carry stores is found here, not in iteration 1. */
typedef struct kmap_peer kmap_peer;
#define KMAP_PEER_HANDLE_BYTES 64
#define KMAP_PEER_BUS_ID_BYTES 32
int kmap_peer_create(kmap_peer **p, int world, int rank, int64_t msg_floats);
int kmap_peer_handle(kmap_peer *p, void *handle_out /* KMAP_PEER_HANDLE_BYTES */);
int kmap_peer_bus_id(char *bus_id_out /* KMAP_PEER_BUS_ID_BYTES, NUL-terminated PCI bus id of the current device */);
int kmap_peer_can_access(const char *bus_id /* a rank's kmap_peer_bus_id */, int *can_access);
int kmap_peer_connect(kmap_peer *p, const void *handles /* world x KMAP_PEER_HANDLE_BYTES, rank order */);
int kmap_peer_hello_push(kmap_peer *p, uint64_t token);                  /* token + rank + 1 -> word [rank] of every rank's area */
int kmap_peer_hello_check(kmap_peer *p, uint64_t token, int *n_missing); /* words of the own area that are not token + q + 1 */
int kmap_peer_set_timeout_ms(kmap_peer *p, int64_t ms);                  /* bound of the apply kernel's wait (> 0) */
int kmap_peer_status(kmap_peer *p, int *timed_out, int64_t *iterations);
int kmap_peer_destroy(kmap_peer *p);
int kmap_embed_step_peer(kmap_embed *e, kmap_peer *p, int n_iter, void *stream);
int kmap_embed_forces_msg(kmap_embed *e, float *msg_dev, void *stream);
int kmap_embed_apply_msg(kmap_embed *e, float *msg_dev, void *stream);
/* single-GPU convenience: n_iter iterations of forces+apply on `stream` */
int kmap_embed_step(kmap_embed *e, int n_iter, void *stream);
/* loop state: iterations done, stopped flag, last loss, best loss, jitter normals consumed */
int kmap_embed_state(kmap_embed *e, int64_t *iters, int *stopped, float *last_loss, float *best_loss,
                     int *jitter_used, void *stream);
int kmap_embed_get_coords(kmap_embed *e, float *coords_2xn, void *stream);      /* current iterate */
int kmap_embed_get_best(kmap_embed *e, float *coords_2xn, void *stream);        /* lowest-loss snapshot */
int kmap_embed_get_losses(kmap_embed *e, float *losses, int64_t max_n, int64_t *n_out, void *stream);
void *kmap_embed_coords_dev(kmap_embed *e);                                      /* device ptr (2 x N f32) */

/* ---- device self-tests (test support): exhaustive bit comparison of the SEQ force kernel's two shortened f32 divisions
 * (csrc/seq_div.h) with IEEE division over EVERY float bit pattern in [lo_bits, hi_bits].  which = 0: clip(1 / s1, 1e-3, 0.999)
 * with rcp_steps Newton steps; which = 1: q / (1 - q) with rcp_steps steps on the reciprocal and quo_steps residual corrections.
 * *n_bad = number of operands whose result differs, *first_bad_bits = the smallest such bit pattern. */
int kmap_selftest_seq_div(int which, int rcp_steps, int quo_steps, uint32_t lo_bits, uint32_t hi_bits, uint64_t *n_bad,
                          uint32_t *first_bad_bits);

#ifdef __cplusplus
}
#endif
#endif /* KMAP_HIP_H */
