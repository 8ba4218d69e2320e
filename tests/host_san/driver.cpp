// tests/host_san/driver.cpp -- CPU sanitizer harness for the threaded HOST code of libkmap_hip (test infrastructure, never shipped):
// the FASTA(.gz) reader and the occurrence-CSV formatter + pwrite pool of kmap_amd/csrc/host_io.hip, and the conversion pool of
// kmap_amd/csrc/host_pool.h (used by counts.hip's table fetch).  Built twice by the Makefile next to it -- -fsanitize=address,undefined
// and -fsanitize=thread -- from the product's own sources compiled host-only; tests/test_host_sanitizers.py drives it.
//   driver fasta <in.fa[.gz]> <seq.bin> <borders.bin>      arrays as kmap_fasta_open / _read return them
//   driver csv <out_i32.csv> <out_u8.csv> <n_seq> <seed>    the same synthetic hit lists through both CSV entry points
//   driver pool <n>                                         u32 -> i64 (unaligned destination), u32 -> u64, u32 -> u32 conversions
//   driver f2 <values.f64> <out.tsv> <medians.f64>          the "%.2f" line writer of the co-occurrence file and the per-read medians
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <vector>

#include "../../include/kmap_hip.h"
#include "../../kmap_amd/csrc/host_pool.h"

static char g_err[512];
void kmap_set_error(const char *fmt, ...) {      // api_core.hip's thread-local message buffer is device-side code's; a plain one here
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

static int fail(const char *what, int rc) {
    fprintf(stderr, "driver: %s failed (rc %d): %s\n", what, rc, g_err);
    return 2;
}
static bool dump(const char *path, const void *p, size_t bytes) {
    FILE *fh = fopen(path, "wb");
    if (!fh) return false;
    const bool ok = bytes == 0 || fwrite(p, 1, bytes, fh) == bytes;
    return fclose(fh) == 0 && ok;
}

int main(int argc, char **argv) {
    if (argc >= 5 && !strcmp(argv[1], "fasta")) {
        kmap_fasta *f = nullptr;
        int64_t nb = 0, ns = 0;
        int rc = kmap_fasta_open(argv[2], &f, &nb, &ns);
        if (rc != KMAP_OK) return fail("kmap_fasta_open", rc);
        std::vector<uint8_t> seq((size_t)nb);
        std::vector<int64_t> borders((size_t)ns * 2);
        rc = kmap_fasta_read(f, seq.data(), borders.data());
        if (rc != KMAP_OK) return fail("kmap_fasta_read", rc);
        kmap_fasta_close(f);
        if (!dump(argv[3], seq.data(), seq.size()) || !dump(argv[4], borders.data(), borders.size() * 8)) return fail("dump", -1);
        printf("%lld %lld\n", (long long)nb, (long long)ns);
        return 0;
    }
    if (argc >= 6 && !strcmp(argv[1], "csv")) {
        const int64_t n_seq = atoll(argv[4]);
        uint64_t x = strtoull(argv[5], nullptr, 10) * 2862933555777941757ull + 3037000493ull;
        auto rnd = [&]() { x = x * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(x >> 33); };
        const int n_cons = 3;
        std::vector<int32_t> hits[3], pos[3];
        std::vector<uint8_t> hits8[3];
        std::vector<int64_t> read_len((size_t)n_seq);
        for (int64_t s = 0; s < n_seq; ++s) read_len[(size_t)s] = 30 + rnd() % 20000;     // numbers on both sides of the 4- and 8-digit fast paths
        for (int c = 0; c < n_cons; ++c) {
            hits[c].resize((size_t)n_seq);
            hits8[c].resize((size_t)n_seq);
            for (int64_t s = 0; s < n_seq; ++s) {
                const uint32_t r = rnd() % 16;
                const int h = r < 10 ? 0 : r < 14 ? 1 : (int)(rnd() % 21);
                hits[c][(size_t)s] = h;
                hits8[c][(size_t)s] = (uint8_t)h;
                int32_t p = 0;
                for (int i = 0; i < h; ++i) {
                    p += (int32_t)(rnd() % 997);
                    pos[c].push_back(p);
                }
            }
            if (pos[c].empty()) pos[c].push_back(0);
        }
        const int32_t *hp[3] = {hits[0].data(), hits[1].data(), hits[2].data()};
        const uint8_t *hp8[3] = {hits8[0].data(), hits8[1].data(), hits8[2].data()};
        const int32_t *pp[3] = {pos[0].data(), pos[1].data(), pos[2].data()};
        int64_t rows = 0, rows8 = 0;
        int rc = kmap_write_occurrence_csv(argv[2], "seq_ind;motif_0_A;motif_1_C;motif_2_G;seq_len", n_seq, n_cons, hp, pp, read_len.data(), &rows);
        if (rc != KMAP_OK) return fail("kmap_write_occurrence_csv", rc);
        rc = kmap_write_occurrence_csv_u8(argv[3], "seq_ind;motif_0_A;motif_1_C;motif_2_G;seq_len", n_seq, n_cons, hp8, pp, read_len.data(), &rows8);
        if (rc != KMAP_OK) return fail("kmap_write_occurrence_csv_u8", rc);
        rc = kmap_write_occurrence_csv("/nonexistent-dir/x.csv", "h", n_seq, n_cons, hp, pp, read_len.data(), &rows8);
        if (rc == KMAP_OK) return fail("write to a missing directory must fail", rc);
        printf("%lld\n", (long long)rows);
        return rows == rows8 ? 0 : 3;
    }
    if (argc >= 3 && !strcmp(argv[1], "pool")) {
        const size_t n = (size_t)atoll(argv[2]);
        std::vector<uint32_t> src(n);
        for (size_t i = 0; i < n; ++i) src[i] = (uint32_t)(i * 2654435761u);
        std::vector<char> raw(n * 8 + 16);
        int64_t *un = (int64_t *)(raw.data() + 3);                           // deliberately unaligned (the pickle-file view)
        kmap_convert_pool<uint32_t, int64_t>(un, src.data(), n, 7);
        for (size_t i = 0; i < n; i += (n / 1000) + 1) {
            int64_t v;
            memcpy(&v, raw.data() + 3 + i * 8, 8);
            if (v != (int64_t)src[i]) return fail("u32 -> i64", (int)i);
        }
        std::vector<uint64_t> w(n);
        kmap_convert_pool<uint32_t, uint64_t>(w.data(), src.data(), n, 16);
        std::vector<uint32_t> same(n);
        kmap_convert_pool<uint32_t, uint32_t>(same.data(), src.data(), n, 3);
        for (size_t i = 0; i < n; ++i)
            if (w[i] != src[i] || same[i] != src[i]) return fail("u32 -> u64 / u32", (int)i);
        kmap_convert_pool<uint32_t, uint64_t>(w.data(), src.data(), 0, 4);  // empty
        printf("%zu\n", n);
        return 0;
    }
    if (argc >= 5 && !strcmp(argv[1], "f2")) {      // driver f2 <values.f64> <out.tsv> <medians.f64>: the "%.2f" line writer + the cell medians
        FILE *fh = fopen(argv[2], "rb");
        if (!fh) return fail("open values", -1);
        std::vector<double> v;
        double x;
        while (fread(&x, 8, 1, fh) == 1) v.push_back(x);
        fclose(fh);
        FILE *out = fopen(argv[3], "wb");
        if (!out) return fail("open output", -1);
        fputs("pair\n", out);
        fflush(out);
        int rc = kmap_write_f2_tsv_line(fileno(out), v.data(), (int64_t)v.size());
        if (rc == KMAP_OK) rc = kmap_write_f2_tsv_line(fileno(out), nullptr, 0);          // an empty line
        fclose(out);
        if (rc != KMAP_OK) return fail("kmap_write_f2_tsv_line", rc);
        if (kmap_write_f2_tsv_line(-1, v.data(), 1) == KMAP_OK) return fail("a bad descriptor must fail", 0);
        // medians: cells of 0..4 ascending locations cut from the values' integer parts
        std::vector<int32_t> hits, pos;
        size_t at = 0;
        for (size_t r = 0; at < v.size() && r < 200000; ++r) {
            const int h = (int)(r % 5);
            int32_t p = 0;
            int got = 0;
            for (; got < h && at < v.size(); ++got, ++at) {
                const double a = std::isfinite(v[at]) ? std::fabs(v[at]) : 0.0;
                p += (int32_t)std::fmod(a, 1000.0);
                pos.push_back(p);
            }
            hits.push_back(got);
        }
        std::vector<double> med(hits.size());
        rc = kmap_cell_medians_i32(hits.data(), pos.data(), (int64_t)hits.size(), (int64_t)pos.size(), med.data());
        if (rc != KMAP_OK) return fail("kmap_cell_medians_i32", rc);
        if (!hits.empty() && kmap_cell_medians_i32(hits.data(), pos.data(), (int64_t)hits.size(), (int64_t)pos.size() + 1, med.data()) == KMAP_OK)
            return fail("hit counts that do not add up must fail", 0);
        std::vector<double> packed;                                                         // hits, then positions, then medians, as f64
        for (int32_t h : hits) packed.push_back(h);
        for (int32_t p : pos) packed.push_back(p);
        for (double m : med) packed.push_back(m);
        if (!dump(argv[4], packed.data(), packed.size() * 8)) return fail("dump", -1);
        printf("%zu %zu %zu\n", v.size(), hits.size(), pos.size());
        return 0;
    }
    fprintf(stderr, "usage: driver fasta|csv|pool|f2 ...\n");
    return 64;
}
