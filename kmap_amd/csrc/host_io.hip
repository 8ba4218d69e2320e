// host_io.hip -- native host-side writers for the file contracts on the hot path (no device code).
#include <errno.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace {
// append the decimal form of v to buf, return the new end
inline char *put_int(char *p, long long v) {
    if (v < 0) { *p++ = '-'; v = -v; }
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
}  // namespace

extern "C" int kmap_write_occurrence_csv(const char *path, const char *header, int64_t n_seq, int n_cons,
                                         const int32_t *const *hits, const int32_t *const *pos, const int64_t *read_len,
                                         int64_t *rows_written) {
    KMAP_REQUIRE(path && header && n_seq >= 0 && n_cons >= 0, "write_occurrence_csv: bad arguments");
    KMAP_REQUIRE(n_seq == 0 || n_cons == 0 || (hits && pos && read_len), "write_occurrence_csv: null arrays");
    FILE *fh = fopen(path, "w");
    if (!fh) {
        kmap_set_error("write_occurrence_csv: cannot open %s: %s", path, strerror(errno));
        return KMAP_E_INVAL;
    }
    fputs(header, fh);
    fputc('\n', fh);
    std::vector<int64_t> cursor((size_t)n_cons, 0);
    std::vector<char> buf(1 << 20);
    size_t used = 0;
    int64_t rows = 0;
    for (int64_t i = 0; i < n_seq; ++i) {
        size_t need = 64;
        bool any = false;
        for (int c = 0; c < n_cons; ++c) {
            need += 12 * (size_t)hits[c][i] + 2;
            any |= hits[c][i] > 0;
        }
        if (any) {
            if (used + need > buf.size()) {
                fwrite(buf.data(), 1, used, fh);
                used = 0;
                if (need > buf.size()) buf.resize(need * 2);
            }
            char *p = buf.data() + used;
            p = put_int(p, i);
            for (int c = 0; c < n_cons; ++c) {
                *p++ = ';';
                const int32_t *q = pos[c] + cursor[c];
                for (int32_t h = 0; h < hits[c][i]; ++h) {
                    if (h) *p++ = ',';
                    p = put_int(p, q[h]);
                }
            }
            *p++ = ';';
            p = put_int(p, read_len[i]);
            *p++ = '\n';
            used = (size_t)(p - buf.data());
            ++rows;
        }
        for (int c = 0; c < n_cons; ++c) cursor[c] += hits[c][i];
    }
    fwrite(buf.data(), 1, used, fh);
    const int rc = fclose(fh);
    if (rows_written) *rows_written = rows;
    if (rc != 0) {
        kmap_set_error("write_occurrence_csv: write to %s failed", path);
        return KMAP_E_INVAL;
    }
    return KMAP_OK;
}
