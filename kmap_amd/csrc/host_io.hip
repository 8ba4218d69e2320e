// host_io.hip -- native host-side writers for the file contracts on the hot path (no device code).
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <zlib.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <string>
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

// ---- "%.2f" rows (the co-occurrence distance file, reference motif_discovery.py:1143-1162) ---------------------------------
// One line of n doubles formatted like Python's f"{x:.2f}", tab-separated, '\n' at the end, written to `fd` at its position.
// The reference formats every value in a Python generator (1.2 s for the 3 M median differences of a C3 run); the values are
// differences of medians of integer positions, i.e. multiples of 0.5, whose "%.2f" form is their integer part + ".00" / ".50" --
// exact, no rounding involved; anything else goes through snprintf (correctly rounded, like Python's own formatting).
namespace {
inline char *put_f2(char *p, double v) {
    const double t = v * 2.0;
    if (v == v && t > -2e15 && t < 2e15 && t == (double)(long long)t) {     // a multiple of 0.5 of moderate size (NaN fails v == v)
        long long h = (long long)t;
        if (h < 0 || (h == 0 && std::signbit(v))) {
            *p++ = '-';
            h = -h;
        }
        p = put_int(p, h >> 1);
        *p++ = '.';
        *p++ = (h & 1) ? '5' : '0';
        *p++ = '0';
        return p;
    }
    if (v != v) {                                                            // Python prints 'nan' whatever the sign bit
        memcpy(p, "nan", 3);
        return p + 3;
    }
    return p + snprintf(p, 336, "%.2f", v);                                  // <= 1 + 309 + 3 characters + the terminator
}
}  // namespace

static int write_f2_tsv_line_impl(int fd, const double *v, int64_t n) {
    KMAP_REQUIRE(fd >= 0 && n >= 0 && (n == 0 || v), "write_f2_tsv_line: bad descriptor / null values");
    const int64_t per = 1 << 16;
    const int64_t n_chunks = (n + per - 1) / per;
    std::vector<std::string> parts((size_t)std::max<int64_t>(n_chunks, 1));
    static const int thread_cap = getenv("KMAP_IO_THREADS") ? std::max(1, atoi(getenv("KMAP_IO_THREADS"))) : 16;
    const int n_threads = (int)std::min<int64_t>(std::min<int64_t>(std::max<unsigned>(1u, std::thread::hardware_concurrency()), thread_cap),
                                                 std::max<int64_t>(n_chunks, 1));
    std::atomic<int64_t> next{0};
    std::atomic<int> oom{0};
    auto worker = [&]() {
        char tmp[352];
        for (;;) {
            const int64_t c = next.fetch_add(1);
            if (c >= n_chunks) return;
            const int64_t lo = c * per, hi = std::min(n, lo + per);
            try {
                std::string &out = parts[(size_t)c];
                out.reserve((size_t)(hi - lo) * 8);
                for (int64_t i = lo; i < hi; ++i) {
                    char *e = put_f2(tmp, v[i]);
                    *e++ = (i + 1 == n) ? '\n' : '\t';
                    out.append(tmp, (size_t)(e - tmp));
                }
            } catch (...) {
                oom.store(1);
                return;
            }
        }
    };
    {
        std::vector<std::thread> pool;
        try {
            for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
        } catch (...) {
        }
        worker();
        for (auto &th : pool) th.join();
    }
    if (oom.load()) {
        kmap_set_error("write_f2_tsv_line: out of memory");
        return KMAP_E_NOMEM;
    }
    if (n == 0) parts[0] = "\n";
    for (const std::string &part : parts) {
        const char *p = part.data();
        size_t left = part.size();
        while (left) {
            const ssize_t w = write(fd, p, left);
            if (w < 0) {
                if (errno == EINTR) continue;
                kmap_set_error("write_f2_tsv_line: write failed: %s", strerror(errno));
                return KMAP_E_IO;
            }
            p += w;
            left -= (size_t)w;
        }
    }
    return KMAP_OK;
}

extern "C" int kmap_write_f2_tsv_line(int fd, const double *v, int64_t n) {
    try {                                              // no C++ exception crosses the C ABI
        return write_f2_tsv_line_impl(fd, v, n);
    } catch (const std::bad_alloc &) {
        kmap_set_error("write_f2_tsv_line: out of memory");
        return KMAP_E_NOMEM;
    }
}

// np.median of every cell of one motif's hit list (reports.Occurrence.medians): hits[r] locations of read r, ascending, stored back to
// back in pos; med[r] = the mean of the two middle ones, NaN for an empty cell.  One pass instead of numpy's six 10^7-element
// temporaries (cumulated offsets, masks, two gathers).
extern "C" int kmap_cell_medians_i32(const int32_t *hits, const int32_t *pos, int64_t n_seq, int64_t n_pos, double *med) {
    KMAP_REQUIRE(n_seq >= 0 && n_pos >= 0 && (n_seq == 0 || (hits && med)) && (n_pos == 0 || pos), "cell_medians: bad sizes / null pointer");
    int64_t o = 0;
    for (int64_t r = 0; r < n_seq; ++r) {
        const int64_t h = hits[r];
        if (h < 0 || o + h > n_pos) {
            kmap_set_error("cell_medians: the hit counts do not add up to the %lld locations given (read %lld)", (long long)n_pos, (long long)r);
            return KMAP_E_INVAL;
        }
        med[r] = h ? ((double)pos[o + (h - 1) / 2] + (double)pos[o + h / 2]) / 2.0 : std::nan("");
        o += h;
    }
    if (o != n_pos) {
        kmap_set_error("cell_medians: the hit counts add up to %lld of the %lld locations given", (long long)o, (long long)n_pos);
        return KMAP_E_INVAL;
    }
    return KMAP_OK;
}

// ---- FASTA encoder --------------------------------------------------------------------------------------------
// The contract (kmap_hip.h; reference kmer_count.py:244-347 through Bio.SeqIO): a record starts at a line whose FIRST byte is '>',
// its sequence is the following lines with white space removed, A/C/G/T in either case -> 0..3, anything else -> 255, one 255
// after every record, text before the first header is ignored.
//
// Line-oriented and parallel (round 5; the byte-at-a-time state machine it replaces took 2.8 s for C3's 1.6-GB FASTA on the GPU
// box's host, 12 s in the authoring container; this one 0.06 s on 16 threads, 0.65 s on one).  The ENCODING of a range of the file that starts at a line start is: every
// sequence byte translated through a 256-entry table whose white-space entries do not advance the output, and ONE 255 where a
// header line begins.  Encodings of consecutive ranges concatenate, and the separators of the file are exactly those 255s -- all
// but the first of the file, before which everything is dropped -- plus one at the very end: a range needs to know nothing about
// its neighbours but where its output starts and how many headers came before it.  A plain file is mapped and cut behind
// newlines into a few ranges per thread: kmap_fasta_open counts every range (output bytes, headers), kmap_fasta_read encodes
// every range straight into the caller's arrays -- no intermediate copy of the 1.5 GB.  A gzip stream cannot be cut: it is
// encoded once, buffer by buffer (the state carried across the cuts: inside a header line / at a line start), into a buffer
// that kmap_fasta_read copies out.
namespace {
struct FaState {
    bool in_header = false, at_line_start = true;
};
constexpr uint8_t FA_WS = 254;         // table entry of the white space removed from sequence lines (never part of the output)
struct FaLut {
    uint8_t v[256];
    FaLut() {
        memset(v, 255, sizeof v);
        v[(int)'A'] = v[(int)'a'] = 0;
        v[(int)'C'] = v[(int)'c'] = 1;
        v[(int)'G'] = v[(int)'g'] = 2;
        v[(int)'T'] = v[(int)'t'] = 3;
        v[(int)' '] = v[(int)'\t'] = v[(int)'\r'] = v[(int)'\v'] = v[(int)'\f'] = FA_WS;
    }
};
const FaLut g_fa_lut;

// Walk [p0, p0 + n): returns the output position behind the range (it starts at `k`).  WRITE: out[...] receives the encoding
// (at most one byte per input byte, and exactly the bytes the counting walk of the same range counted); otherwise only
// positions are counted.  on_header(input offset of the '>', output
// position of its 255) is called for every header line that begins in the range.
// cap (WRITE): positions the walk may fill; a walk that would go past it stops and returns FA_OVERRUN (the mapped file is not
// what kmap_fasta_open counted any more: somebody wrote to it in between).
constexpr size_t FA_OVERRUN = ~(size_t)0;
template <bool WRITE, typename OnHeader>
size_t fa_walk(const uint8_t *p0, size_t n, FaState &s, uint8_t *out, size_t k, size_t cap, OnHeader on_header) {
    const uint8_t *p = p0;
    const uint8_t *const e = p0 + n;
    const uint8_t *const lut = g_fa_lut.v;
    while (p < e) {
        if (s.in_header) {                                              // the rest of a header line
            const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
            if (!nl) break;                                             // continues in the next buffer
            p = nl + 1;
            s.in_header = false;
            s.at_line_start = true;
            continue;
        }
        if (s.at_line_start) {
            if (*p == '>') {
                if (WRITE && k >= cap) return FA_OVERRUN;
                on_header((size_t)(p - p0), k);
                if (WRITE) out[k] = 255;
                ++k;
                s.in_header = true;
                ++p;
                continue;
            }
            if (*p == '\n') {                                           // empty line: the next byte starts a line again
                ++p;
                continue;
            }
            s.at_line_start = false;                                    // any other byte, white space too: a '>' later in the line is sequence
        }
        const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
        const uint8_t *const seg = nl ? nl : e;
        const uint8_t *body = seg;
        if (body > p && body[-1] == '\r') --body;                       // CRLF files: the line's own '\r' is white space like any other
        uint32_t ws = 0;                                                // ' ', \t, \v, \f, \r (a segment holds no \n): byte compares, vectorised
        for (const uint8_t *q = p; q < body; ++q) ws += (uint32_t)((*q == 32) | ((uint8_t)(*q - 9) <= 4));
        if (WRITE && k + ((size_t)(body - p) - ws) > cap) return FA_OVERRUN;
        if (!WRITE) {
            k += (size_t)(body - p) - ws;
        } else if (ws == 0) {
            // the table as arithmetic on the byte, so that the loop vectorises: upper-cased A / C / G / T have (c >> 1) & 3 = 0, 1, 3, 2
            uint8_t *o = out + k;
            const size_t m = (size_t)(body - p);
            for (size_t i = 0; i < m; ++i) {
                const uint8_t c = p[i], u = (uint8_t)(c & 0xDF), t = (uint8_t)((c >> 1) & 3);
                const bool acgt = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
                o[i] = acgt ? (uint8_t)(t ^ (t >> 1)) : (uint8_t)255;
            }
            k += m;
        } else {
            uint8_t *o = out + k;
            for (const uint8_t *q = p; q < body; ++q) {                 // nothing is stored for a white-space byte: the position behind a
                const uint8_t v = lut[*q];                              // range's output belongs to the thread of the next range
                if (v != FA_WS) *o++ = v;
            }
            k = (size_t)(o - out);
        }
        p = seg;
        if (nl) {
            ++p;
            s.at_line_start = true;
        }
    }
    return k;
}

struct FaRange {                        // a range of the mapped file and what kmap_fasta_open counted in it
    size_t lo = 0, hi = 0;             // input bytes [lo, hi)
    size_t out = 0, headers = 0;       // output bytes (every header's 255 included), header lines
    size_t first_hdr_in = 0, first_hdr_out = 0;   // input offset (from lo) / output position of the first header (headers > 0)
    size_t base = 0, rank0 = 0;        // filled by fa_finish: output offset of the range's first byte, headers before the range
};
}  // namespace

struct kmap_fasta {
    // mapped plain file
    const uint8_t *map = nullptr;
    size_t map_len = 0;
    std::vector<FaRange> ranges;
    size_t first_range = 0;             // the range that holds the first header of the file; earlier ranges are text to ignore
    // gzip stream: the whole encoding (header 255s included) + their positions
    uint8_t *buf = nullptr;
    size_t len = 0, cap = 0;
    std::vector<size_t> seps;
    int64_t n_bytes = 0, n_seq = 0;
    unsigned threads = 1;
    ~kmap_fasta() {
        if (map) munmap((void *)map, map_len);
        free(buf);
    }
};

namespace {
unsigned fa_threads() {
    const char *v = getenv("KMAP_IO_THREADS");
    const int cap = v ? std::max(1, atoi(v)) : 16;
    return (unsigned)std::min<int>((int)std::max(1u, std::thread::hardware_concurrency()), cap);
}
// fn(t) for t in [0, n) on up to `nt` threads (the calling thread takes its share; threads that cannot be started are not missed)
template <typename F>
void fa_parallel(size_t n, unsigned nt, F fn) {
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t t = next.fetch_add(1);
            if (t >= n) return;
            fn(t);
        }
    };
    std::vector<std::thread> pool;
    try {
        for (unsigned i = 1; i < nt && i < n; ++i) pool.emplace_back(worker);
    } catch (...) {
    }
    worker();
    for (auto &th : pool) th.join();
}
bool fa_reserve(kmap_fasta *f, size_t want) {   // room for `want` more bytes in the stream buffer
    if (f->len + want <= f->cap) return true;
    size_t ncap = f->cap + f->cap / 2;
    if (ncap < f->len + want) ncap = f->len + want;
    uint8_t *nb = (uint8_t *)realloc(f->buf, ncap ? ncap : 1);
    if (!nb) return false;
    f->buf = nb;
    f->cap = ncap;
    return true;
}
// what a gzread that returned `got` <= 0 means: "" for a clean end of the stream.  A file that ends in the middle of a gzip stream is
// NOT a clean end: gzread hands out what it could inflate and then returns 0 like at a real end -- only gzerror says Z_BUF_ERROR
// (until round 5 a truncated .fa.gz was encoded as far as it went, without a word).
std::string fa_gz_end(gzFile gz, int got) {
    int errnum = Z_OK;
    const char *msg = gzerror(gz, &errnum);
    if (got < 0 || (errnum != Z_OK && errnum != Z_STREAM_END)) {
        if (errnum == Z_BUF_ERROR) return "the file ends in the middle of a gzip stream (truncated?)";
        return (msg && *msg) ? msg : "read error";
    }
    return "";
}
// gzip (or anything that cannot be mapped): the stream is inflated by a helper thread into a ring of three 4-MiB buffers while
// this thread encodes the buffer before -- inflating is the slower half (zlib: a few hundred MB/s of text), the walk hides
// behind it.  size_hint: the uncompressed size the gzip trailer names (modulo 2^32; 0 = unknown): one allocation instead of a
// growing one when it holds.
int fa_open_stream(const char *path, kmap_fasta *f, size_t size_hint) {
    gzFile gz = gzopen(path, "rb");   // transparently reads plain files too
    if (!gz) {
        kmap_set_error("fasta_open: cannot open %s: %s", path, strerror(errno));
        return KMAP_E_INVAL;
    }
    gzbuffer(gz, 1 << 20);
    if (size_hint) (void)fa_reserve(f, size_hint + 1);           // a failed hint is not an error: the buffer grows as before
    constexpr int RING = 3;
    constexpr size_t BUF = (size_t)1 << 22;
    std::vector<uint8_t> ring[RING];
    int got_of[RING] = {0, 0, 0};
    for (auto &b : ring) b.resize(BUF);
    std::mutex mu;
    std::condition_variable cv;
    int filled = 0;                    // buffers inflated and not yet encoded
    bool done = false, stop = false;   // producer reached the end (or failed) / consumer gave up
    std::string read_error;
    auto inflate = [&]() {
        for (int head = 0;; head = (head + 1) % RING) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return filled < RING || stop; });
                if (stop) return;
            }
            const int got = gzread(gz, ring[head].data(), (unsigned)BUF);
            std::lock_guard<std::mutex> lk(mu);
            if (got <= 0) {
                read_error = fa_gz_end(gz, got);
                done = true;
                cv.notify_all();
                return;
            }
            got_of[head] = got;
            ++filled;
            cv.notify_all();
        }
    };
    std::thread producer;
    bool threaded = true;
    try {
        producer = std::thread(inflate);
    } catch (...) {
        threaded = false;              // no helper thread to be had: inflate and encode in turn
    }
    int rc = KMAP_OK;
    FaState st;
    auto encode = [&](const uint8_t *p, size_t n) -> bool {
        if (!fa_reserve(f, n + 1)) return false;
        f->len = fa_walk<true>(p, n, st, f->buf, f->len, f->cap, [&](size_t, size_t at) { f->seps.push_back(at); });
        return true;
    };
    if (threaded) {
        for (int tail = 0;; tail = (tail + 1) % RING) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return filled > 0 || done; });
                if (filled == 0) break;                          // done and drained
            }
            const bool ok = encode(ring[tail].data(), (size_t)got_of[tail]);
            std::lock_guard<std::mutex> lk(mu);
            --filled;
            if (!ok) {
                rc = KMAP_E_NOMEM;
                stop = true;
            }
            cv.notify_all();
            if (!ok) break;
        }
        producer.join();
    } else {
        for (;;) {
            const int got = gzread(gz, ring[0].data(), (unsigned)BUF);
            if (got <= 0) {
                read_error = fa_gz_end(gz, got);
                break;
            }
            if (!encode(ring[0].data(), (size_t)got)) {
                rc = KMAP_E_NOMEM;
                break;
            }
        }
    }
    if (rc == KMAP_OK && !read_error.empty()) {
        kmap_set_error("fasta_open: read error in %s: %s", path, read_error.c_str());
        rc = KMAP_E_INVAL;
    } else if (rc == KMAP_E_NOMEM) {
        kmap_set_error("fasta_open: out of memory");
    }
    gzclose(gz);
    if (rc != KMAP_OK) return rc;
    f->n_seq = (int64_t)f->seps.size();
    f->n_bytes = f->seps.empty() ? 0 : (int64_t)(f->len - (f->seps[0] + 1) + 1);   // without the text before the first header and its 255; + the last separator
    return KMAP_OK;
}
}  // namespace

static int fasta_open_impl(const char *path, kmap_fasta **out, int64_t *n_bytes, int64_t *n_seq) {
    KMAP_REQUIRE(path && out && n_bytes && n_seq, "fasta_open: null argument");
    std::unique_ptr<kmap_fasta> f(new kmap_fasta());
    f->threads = fa_threads();
    size_t gz_hint = 0;
    {
        const int fd = open(path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) {
            kmap_set_error("fasta_open: cannot open %s: %s", path, strerror(errno));
            return KMAP_E_INVAL;
        }
        struct stat sb;
        unsigned char magic[2] = {0, 0};
        if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && pread(fd, magic, 2, 0) >= 1) {
            if (!(magic[0] == 0x1f && magic[1] == 0x8b)) {
                void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
                if (m != MAP_FAILED) {
                    f->map = (const uint8_t *)m;
                    f->map_len = (size_t)sb.st_size;
                }
            } else if (sb.st_size >= 18) {                      // gzip trailer: ISIZE, the uncompressed length modulo 2^32
                unsigned char t[4];
                // trusted only as far as it is plausible for one member (deflate does not shrink text below ~1 / 1000 of its size)
                if (pread(fd, t, 4, sb.st_size - 4) == 4) {
                    const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
                    if (isize / 1024 <= (size_t)sb.st_size) gz_hint = isize;
                }
            }
        }
        close(fd);
    }
    if (!f->map) {
        KMAP_TRY(fa_open_stream(path, f.get(), gz_hint));
    } else {
        // ranges of >= KMAP_FASTA_MIN_CHUNK bytes (default 4 MiB; the tests cut small files finer), each cut behind a newline
        const char *mc = getenv("KMAP_FASTA_MIN_CHUNK");
        const size_t min_chunk = mc ? (size_t)std::max(1ll, atoll(mc)) : ((size_t)4 << 20);
        const size_t n = f->map_len;
        const size_t want = std::max<size_t>(1, std::min<size_t>((size_t)f->threads * 4, n / min_chunk));
        std::vector<size_t> cut;
        cut.push_back(0);
        for (size_t t = 1; t < want; ++t) {
            const size_t from = std::max(cut.back(), (n / want) * t);
            if (from >= n) break;
            const uint8_t *nl = (const uint8_t *)memchr(f->map + from, '\n', n - from);
            if (!nl || (size_t)(nl - f->map) + 1 >= n) break;
            const size_t at = (size_t)(nl - f->map) + 1;
            if (at > cut.back()) cut.push_back(at);
        }
        cut.push_back(n);
        f->ranges.resize(cut.size() - 1);
        fa_parallel(f->ranges.size(), f->threads, [&](size_t i) {
            FaRange &r = f->ranges[i];
            r.lo = cut[i];
            r.hi = cut[i + 1];
            FaState st;
            r.out = fa_walk<false>(f->map + r.lo, r.hi - r.lo, st, nullptr, 0, FA_OVERRUN, [&](size_t in_at, size_t out_at) {
                if (r.headers++ == 0) {
                    r.first_hdr_in = in_at;
                    r.first_hdr_out = out_at;
                }
            });
        });
        size_t n_hdr = 0, at = 0;
        bool found = false;
        f->first_range = f->ranges.size();
        for (size_t i = 0; i < f->ranges.size(); ++i) {
            FaRange &r = f->ranges[i];
            r.rank0 = n_hdr;
            n_hdr += r.headers;
            if (!found) {
                if (!r.headers) continue;                   // text before the first header of the file
                found = true;
                f->first_range = i;
                r.base = 0;                                 // its output starts BEHIND its first header's 255 (fa_read walks from there)
                at = r.out - (r.first_hdr_out + 1);
            } else {
                r.base = at;
                at += r.out;
            }
        }
        f->n_seq = (int64_t)n_hdr;
        f->n_bytes = found ? (int64_t)(at + 1) : 0;         // + the separator of the last record
    }
    *n_bytes = f->n_bytes;
    *n_seq = f->n_seq;
    *out = f.release();
    return KMAP_OK;
}

extern "C" int kmap_fasta_open(const char *path, kmap_fasta **out, int64_t *n_bytes, int64_t *n_seq) {
    try {                                              // the handle's vectors / the header positions of a gzip stream may not fit
        return fasta_open_impl(path, out, n_bytes, n_seq);
    } catch (const std::bad_alloc &) {
        kmap_set_error("fasta_open: out of memory");
        return KMAP_E_NOMEM;
    }
}

static int fasta_read_impl(kmap_fasta *f, uint8_t *seq_out, int64_t *borders_out) {
    KMAP_REQUIRE(f, "fasta_read: null handle");
    KMAP_REQUIRE((f->n_bytes == 0 || seq_out) && (f->n_seq == 0 || borders_out), "fasta_read: null output");
    if (f->n_seq == 0) return KMAP_OK;
    const size_t total = (size_t)f->n_bytes;
    // the header of rank r >= 1 (file order) has its 255 at output position g: record r - 1 ends there, record r starts behind it
    auto border = [&](size_t r, size_t g) {
        borders_out[2 * (r - 1) + 1] = (int64_t)g;
        borders_out[2 * r] = (int64_t)(g + 1);
    };
    borders_out[0] = 0;
    std::atomic<int> changed{0};
    if (f->map) {
        fa_parallel(f->ranges.size() - f->first_range, f->threads, [&](size_t j) {
            const size_t i = f->first_range + j;
            const FaRange &r = f->ranges[i];
            const uint8_t *p = f->map + r.lo;
            size_t n = r.hi - r.lo, rank = r.rank0;
            FaState st;
            if (i == f->first_range) {                      // from behind the '>' of the file's first header: rank 0 writes nothing
                p += r.first_hdr_in + 1;
                n -= r.first_hdr_in + 1;
                st.in_header = true;
                rank = 1;
            }
            // exactly the bytes / headers counted at open: a file that changed in between must not write past its share
            const size_t share = i == f->first_range ? r.out - (r.first_hdr_out + 1) : r.out;
            const size_t rank_end = r.rank0 + r.headers;
            const size_t end = fa_walk<true>(p, n, st, seq_out + r.base, 0, share, [&](size_t, size_t at) {
                if (rank < rank_end) border(rank, r.base + at);
                ++rank;
            });
            if (end != share || rank != rank_end) changed.store(1);
        });
        if (changed.load()) {
            kmap_set_error("fasta_read: the file is not what fasta_open counted (modified while it was read?)");
            return KMAP_E_STATE;
        }
    } else {
        const size_t drop = f->seps[0] + 1;
        memcpy(seq_out, f->buf + drop, f->len - drop);
        for (size_t r = 1; r < f->seps.size(); ++r) border(r, f->seps[r] - drop);
    }
    seq_out[total - 1] = 255;
    borders_out[2 * ((size_t)f->n_seq - 1) + 1] = (int64_t)(total - 1);
    return KMAP_OK;
}

extern "C" int kmap_fasta_read(kmap_fasta *f, uint8_t *seq_out, int64_t *borders_out) {
    try {
        return fasta_read_impl(f, seq_out, borders_out);
    } catch (const std::bad_alloc &) {
        kmap_set_error("fasta_read: out of memory");
        return KMAP_E_NOMEM;
    }
}

extern "C" int kmap_fasta_close(kmap_fasta *f) {
    delete f;
    return KMAP_OK;
}
