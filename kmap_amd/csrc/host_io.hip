// host_io.hip -- native host-side writers for the file contracts on the hot path (no device code).
#include <errno.h>
#include <stdlib.h>
#include <zlib.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <thread>
#include <vector>

#include "common.h"

namespace {
// decimal text of 0..9999: four characters (left-aligned, and zero-padded) + the length of the unpadded form.  The CSV rows
// are nothing but small integers -- read index, positions < read length, read length -- so one lookup and one 4-byte store
// per number replaces the digit loop (8 M rows per file at C3: 50 -> ~15 ns per row and core)
struct Dec4 {
    char plain[10000][4];
    char padded[10000][4];
    unsigned char len[10000];
    Dec4() {
        for (int v = 0; v < 10000; ++v) {
            const char d[4] = {(char)('0' + v / 1000), (char)('0' + v / 100 % 10), (char)('0' + v / 10 % 10), (char)('0' + v % 10)};
            const int skip = v >= 1000 ? 0 : v >= 100 ? 1 : v >= 10 ? 2 : 3;
            for (int i = 0; i < 4; ++i) {
                padded[v][i] = d[i];
                plain[v][i] = i + skip < 4 ? d[i + skip] : '0';
            }
            len[v] = (unsigned char)(4 - skip);
        }
    }
};
const Dec4 g_dec4;

// append the decimal form of v to buf (which has >= 4 bytes of slack behind the number), return the new end
inline char *put_int(char *p, long long v) {
    if (v >= 0 && v < 10000) {
        memcpy(p, g_dec4.plain[v], 4);
        return p + g_dec4.len[v];
    }
    if (v >= 0 && v < 100000000) {
        const int hi = (int)(v / 10000), lo = (int)(v % 10000);
        memcpy(p, g_dec4.plain[hi], 4);
        p += g_dec4.len[hi];
        memcpy(p, g_dec4.padded[lo], 4);
        return p + 4;
    }
    if (v < 0) { *p++ = '-'; v = -v; }
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
}  // namespace

namespace {
template <typename HT>
int write_occurrence_csv_impl(const char *path, const char *header, int64_t n_seq, int n_cons, const HT *const *hits,
                              const int32_t *const *pos, const int64_t *read_len, int64_t *rows_written) {
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
    static const int thread_cap = getenv("KMAP_IO_THREADS") ? std::max(1, atoi(getenv("KMAP_IO_THREADS"))) : 16;   // several writers run at once (scan_motif)
    const int n_threads = (int)std::min<int64_t>(std::max<unsigned>(1u, std::thread::hardware_concurrency()), thread_cap);
    const int64_t n_chunks = std::max<int64_t>(1, std::min<int64_t>((n_seq + 65535) / 65536, 4096));
    const int64_t per = (n_seq + n_chunks - 1) / n_chunks;
    std::vector<std::vector<int64_t>> start((size_t)n_chunks, std::vector<int64_t>((size_t)n_cons, 0));
    {   // per-chunk hit sums in parallel, then a serial prefix over the (few thousand) chunks
        std::atomic<int64_t> nx{0};
        auto sum_worker = [&]() {
            for (;;) {
                const int64_t ch = nx.fetch_add(1);
                if (ch >= n_chunks) return;
                const int64_t lo = ch * per, hi = std::min(n_seq, lo + per);
                for (int c = 0; c < n_cons; ++c) {
                    int64_t s2 = 0;
                    for (int64_t i = lo; i < hi; ++i) s2 += hits[c][i];
                    start[(size_t)ch][(size_t)c] = s2;
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(sum_worker);
        for (auto &t : pool) t.join();
        std::vector<int64_t> cur((size_t)n_cons, 0);
        for (int64_t ch = 0; ch < n_chunks; ++ch)
            for (int c = 0; c < n_cons; ++c) {
                const int64_t s2 = start[(size_t)ch][(size_t)c];
                start[(size_t)ch][(size_t)c] = cur[(size_t)c];
                cur[(size_t)c] += s2;
            }
    }
    // uninitialised buffers: a vector<char>::resize would zero-fill (and page-fault) every byte before it is formatted over
    struct Chunk {
        std::unique_ptr<char[]> mem;
        size_t len = 0;
        char *data() const { return mem.get(); }
        size_t size() const { return len; }
        bool empty() const { return len == 0; }
    };
    std::vector<Chunk> bufs((size_t)n_chunks);
    std::vector<int64_t> rows_of((size_t)n_chunks, 0);
    std::atomic<int64_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const int64_t ch = next.fetch_add(1);
            if (ch >= n_chunks) return;
            const int64_t lo = ch * per, hi = std::min(n_seq, lo + per);
            std::vector<int64_t> cursor = start[(size_t)ch];
            Chunk &buf = bufs[(size_t)ch];
            size_t need = 0;
            for (int64_t i = lo; i < hi; ++i) {
                size_t r = 48;
                for (int c = 0; c < n_cons; ++c) r += 12 * (size_t)hits[c][i] + 2;
                need += r;
            }
            buf.mem.reset(new char[need + 64]);
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
                        for (int32_t h = 0; h < (int32_t)hits[c][i]; ++h) {
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
            buf.len = (size_t)(p - buf.data());
            rows_of[(size_t)ch] = rows;
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker);
        for (auto &t : pool) t.join();
    }
    // the chunks go to their final offsets with parallel pwrite()s (one memcpy into the page cache per thread instead of
    // one serial stream: 185 MB per file at C3)
    int64_t rows = 0;
    fflush(fh);
    const int fd = fileno(fh);
    std::vector<int64_t> off((size_t)n_chunks + 1, 0);
    off[0] = (int64_t)ftello(fh);
    for (int64_t ch = 0; ch < n_chunks; ++ch) {
        off[(size_t)ch + 1] = off[(size_t)ch] + (int64_t)bufs[(size_t)ch].size();
        rows += rows_of[(size_t)ch];
    }
    std::atomic<bool> write_ok{true};
    {
        std::atomic<int64_t> nx{0};
        auto write_worker = [&]() {
            for (;;) {
                const int64_t ch = nx.fetch_add(1);
                if (ch >= n_chunks) return;
                const char *src = bufs[(size_t)ch].data();
                size_t left = bufs[(size_t)ch].size();
                int64_t at = off[(size_t)ch];
                while (left) {
                    const ssize_t w = pwrite(fd, src, left, (off_t)at);
                    if (w <= 0) {
                        write_ok = false;
                        return;
                    }
                    src += w;
                    at += w;
                    left -= (size_t)w;
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < std::min(n_threads, 2); ++t) pool.emplace_back(write_worker);   // one inode lock: more threads only wait (measured 4..64: 21-27 ms)
        for (auto &t : pool) t.join();
    }
    const int rc = fclose(fh);
    if (rows_written) *rows_written = rows;
    if (rc != 0 || !write_ok) {
        kmap_set_error("write_occurrence_csv: write to %s failed", path);
        return KMAP_E_INVAL;
    }
    return KMAP_OK;
}
}  // namespace

extern "C" int kmap_write_occurrence_csv(const char *path, const char *header, int64_t n_seq, int n_cons,
                                         const int32_t *const *hits, const int32_t *const *pos, const int64_t *read_len,
                                         int64_t *rows_written) {
    return write_occurrence_csv_impl<int32_t>(path, header, n_seq, n_cons, hits, pos, read_len, rows_written);
}
// same file from byte-sized hit counts (kmap_scan_fetch_stream_u8: a quarter of the bytes to fetch and to walk)
extern "C" int kmap_write_occurrence_csv_u8(const char *path, const char *header, int64_t n_seq, int n_cons,
                                            const uint8_t *const *hits, const int32_t *const *pos, const int64_t *read_len,
                                            int64_t *rows_written) {
    return write_occurrence_csv_impl<uint8_t>(path, header, n_seq, n_cons, hits, pos, read_len, rows_written);
}

// ---- FASTA encoder --------------------------------------------------------------------------------------------
struct kmap_fasta {
    std::vector<uint8_t> seq;
    std::vector<int64_t> borders;   // start, end pairs
};

extern "C" int kmap_fasta_open(const char *path, kmap_fasta **out, int64_t *n_bytes, int64_t *n_seq) {
    KMAP_REQUIRE(path && out && n_bytes && n_seq, "fasta_open: null argument");
    gzFile gz = gzopen(path, "rb");   // transparently reads plain files too
    if (!gz) {
        kmap_set_error("fasta_open: cannot open %s: %s", path, strerror(errno));
        return KMAP_E_INVAL;
    }
    gzbuffer(gz, 1 << 20);
    uint8_t lut[256];
    memset(lut, 255, sizeof lut);
    lut[(int)'A'] = lut[(int)'a'] = 0;
    lut[(int)'C'] = lut[(int)'c'] = 1;
    lut[(int)'G'] = lut[(int)'g'] = 2;
    lut[(int)'T'] = lut[(int)'t'] = 3;
    kmap_fasta *f = new kmap_fasta();
    std::vector<uint8_t> buf(1 << 22);
    bool in_header = false, in_record = false, at_line_start = true;
    int64_t start = 0;
    for (;;) {
        const int got = gzread(gz, buf.data(), (unsigned)buf.size());
        if (got < 0) {
            int errnum = 0;
            kmap_set_error("fasta_open: read error in %s: %s", path, gzerror(gz, &errnum));
            gzclose(gz);
            delete f;
            return KMAP_E_INVAL;
        }
        if (got == 0) break;
        for (int i = 0; i < got; ++i) {
            const uint8_t ch = buf[(size_t)i];
            if (in_header) {
                if (ch == '\n') { in_header = false; at_line_start = true; }
                continue;
            }
            if (ch == '\n') { at_line_start = true; continue; }
            if (at_line_start && ch == '>') {
                if (in_record) {   // close the previous record
                    f->borders.push_back(start);
                    f->borders.push_back((int64_t)f->seq.size());
                    f->seq.push_back(255);
                }
                in_record = true;
                in_header = true;
                start = (int64_t)f->seq.size();
                continue;
            }
            at_line_start = false;
            if (!in_record) continue;                                   // text before the first header
            if (ch == ' ' || ch == '\t' || ch == '\r' || ch == '\v' || ch == '\f') continue;
            f->seq.push_back(lut[ch]);
        }
    }
    gzclose(gz);
    if (in_record) {
        f->borders.push_back(start);
        f->borders.push_back((int64_t)f->seq.size());
        f->seq.push_back(255);
    }
    *out = f;
    *n_bytes = (int64_t)f->seq.size();
    *n_seq = (int64_t)(f->borders.size() / 2);
    return KMAP_OK;
}

extern "C" int kmap_fasta_read(kmap_fasta *f, uint8_t *seq_out, int64_t *borders_out) {
    KMAP_REQUIRE(f, "fasta_read: null handle");
    KMAP_REQUIRE((f->seq.empty() || seq_out) && (f->borders.empty() || borders_out), "fasta_read: null output");
    if (!f->seq.empty()) memcpy(seq_out, f->seq.data(), f->seq.size());
    if (!f->borders.empty()) memcpy(borders_out, f->borders.data(), f->borders.size() * sizeof(int64_t));
    return KMAP_OK;
}

extern "C" int kmap_fasta_close(kmap_fasta *f) {
    delete f;
    return KMAP_OK;
}
