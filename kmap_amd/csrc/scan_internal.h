// scan_internal.h -- the occurrence-scan handle shared by scan.hip (uint8 input) and packed.hip (2-bit input)
#pragma once
#include "common.h"

struct kmap_scan {
    int64_t n_seq = 0, total = 0, cap_seq = 0, cap_pos = 0;
    int32_t *hits = nullptr;
    int8_t *mind = nullptr;
    uint64_t *offs = nullptr;           // exclusive scan of hits (run); afterwards scratch: summary words, byte-narrowed hits
    int32_t *pos = nullptr;
    // the caller's declaration (kmap_scan_declare_uniform, verified on the device) that read s of these borders is
    // [s * stride, s * stride + len): the per-read kernels then take the borders from s instead of loading 16 bytes per read (bitslice.hip)
    const void *geo_borders = nullptr;
    int64_t geo_n_seq = -1, geo_len = 0, geo_stride = 0;   // stride 0: not uniform
};
int kmap_scan_reserve(kmap_scan *s, int64_t n_seq);
int kmap_scan_reserve_pos(kmap_scan *s, uint64_t total);

// bitslice.hip: hit bits of every window (k <= 16) from the bit planes of the reads (words: one uint32 per 32 windows for the
// scan; else uint16 per group for the mask's coverage pass), and the scan's per-read passes on them
int kmap_bitslice_hits(const uint32_t *planes, const uint16_t *inval, int64_t n, int k, const uint64_t *cons, const int32_t *radius,
                       int n_cons, int revcom_pairs, uint16_t *hit16, bool words, hipStream_t st);
int kmap_bitslice_scan_reads(bool write, const uint32_t *hit32, const uint32_t *codes, const uint16_t *inval, int64_t n,
                             const int64_t *borders, int64_t n_seq, int k, uint64_t cons, int revcom, int radius, kmap_scan *s,
                             hipStream_t st);
// the whole per-read part in one call: one pass + a copy of the blocks' segments into read order (two passes when the hits do not
// fit the temporary buffer)
int kmap_bitslice_scan_reads_all(const uint32_t *hit32, const uint32_t *codes, const uint16_t *inval, int64_t n, const int64_t *borders,
                                 int64_t n_seq, int k, uint64_t cons, int revcom, int radius, kmap_scan *s, uint64_t *total_out,
                                 hipStream_t st);
int kmap_bitslice_declare_uniform(kmap_scan *s, const int64_t *borders, int64_t n_seq, int64_t len, int64_t stride, int *accepted, hipStream_t st);
