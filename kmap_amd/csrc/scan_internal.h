// scan_internal.h -- the occurrence-scan handle shared by scan.hip (uint8 input) and packed.hip (2-bit input)
#pragma once
#include "common.h"

struct kmap_scan {
    int64_t n_seq = 0, total = 0, cap_seq = 0, cap_pos = 0;
    int32_t *hits = nullptr;
    int8_t *mind = nullptr;
    uint64_t *offs = nullptr;           // exclusive scan of hits (run); afterwards scratch: summary words, byte-narrowed hits
    int32_t *pos = nullptr;
};
int kmap_scan_reserve(kmap_scan *s, int64_t n_seq);
int kmap_scan_reserve_pos(kmap_scan *s, uint64_t total);
