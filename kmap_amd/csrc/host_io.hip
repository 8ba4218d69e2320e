// host_io.hip -- native host-side writers for the file contracts on the hot path (no device code).
#include <errno.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
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
    // format in parallel: reads are cut into chunks, every chunk gets its own cursor (prefix of hits) and buffer,
    // buffers are written in chunk order
    const int n_threads = (int)std::min<int64_t>(std::max<unsigned>(1u, std::thread::hardware_concurrency()), 32);
    const int64_t n_chunks = std::max<int64_t>(1, std::min<int64_t>((n_seq + 65535) / 65536, 4096));
    const int64_t per = (n_seq + n_chunks - 1) / n_chunks;
    std::vector<std::vector<int64_t>> start((size_t)n_chunks, std::vector<int64_t>((size_t)n_cons, 0));
    {
        std::vector<int64_t> cur((size_t)n_cons, 0);
        for (int64_t ch = 0; ch < n_chunks; ++ch) {
            start[(size_t)ch] = cur;
            const int64_t lo = ch * per, hi = std::min(n_seq, lo + per);
            for (int c = 0; c < n_cons; ++c) {
                int64_t s2 = 0;
                for (int64_t i = lo; i < hi; ++i) s2 += hits[c][i];
                cur[(size_t)c] += s2;
            }
        }
    }
    std::vector<std::vector<char>> bufs((size_t)n_chunks);
    std::vector<int64_t> rows_of((size_t)n_chunks, 0);
    std::atomic<int64_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const int64_t ch = next.fetch_add(1);
            if (ch >= n_chunks) return;
            const int64_t lo = ch * per, hi = std::min(n_seq, lo + per);
            std::vector<int64_t> cursor = start[(size_t)ch];
            std::vector<char> &buf = bufs[(size_t)ch];
            size_t need = 0;
            for (int64_t i = lo; i < hi; ++i) {
                size_t r = 48;
                for (int c = 0; c < n_cons; ++c) r += 12 * (size_t)hits[c][i] + 2;
                need += r;
            }
            buf.resize(need + 64);
            char *p = buf.data();
            int64_t rows = 0;
            for (int64_t i = lo; i < hi; ++i) {
                bool any = false;
                for (int c = 0; c < n_cons; ++c) any |= hits[c][i] > 0;
                if (any) {
                    p = put_int(p, i);
                    for (int c = 0; c < n_cons; ++c) {
                        *p++ = ';';
                        const int32_t *q = pos[c] + cursor[(size_t)c];
                        for (int32_t h = 0; h < hits[c][i]; ++h) {
                            if (h) *p++ = ',';
                            p = put_int(p, q[h]);
                        }
                    }
                    *p++ = ';';
                    p = put_int(p, read_len[i]);
                    *p++ = '\n';
                    ++rows;
                }
                for (int c = 0; c < n_cons; ++c) cursor[(size_t)c] += hits[c][i];
            }
            buf.resize((size_t)(p - buf.data()));
            rows_of[(size_t)ch] = rows;
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker);
        for (auto &t : pool) t.join();
    }
    int64_t rows = 0;
    for (int64_t ch = 0; ch < n_chunks; ++ch) {
        if (!bufs[(size_t)ch].empty()) fwrite(bufs[(size_t)ch].data(), 1, bufs[(size_t)ch].size(), fh);
        rows += rows_of[(size_t)ch];
    }
    const int rc = fclose(fh);
    if (rows_written) *rows_written = rows;
    if (rc != 0) {
        kmap_set_error("write_occurrence_csv: write to %s failed", path);
        return KMAP_E_INVAL;
    }
    return KMAP_OK;
}
