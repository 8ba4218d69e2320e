// tests/host_san/driver.cpp -- CPU sanitizer harness for the threaded HOST code of libkmap_hip (test infrastructure, never shipped):
// the FASTA(.gz) reader and the occurrence-CSV formatter + pwrite pool of kmap_amd/csrc/host_io.hip, and the conversion pool of
// kmap_amd/csrc/host_pool.h (used by counts.hip's table fetch).  Built twice by the Makefile next to it -- -fsanitize=address,undefined
// and -fsanitize=thread -- from the product's own sources compiled host-only; tests/test_host_sanitizers.py drives it.
//   driver fasta <in.fa[.gz]> <seq.bin> <borders.bin>      arrays as kmap_fasta_open / _read return them
//   driver csv <out_i32.csv> <out_u8.csv> <n_seq> <seed>    the same synthetic hit lists through both CSV entry points
//   driver pool <n>                                         u32 -> i64 (unaligned destination), u32 -> u64, u32 -> u32 conversions
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
    fprintf(stderr, "usage: driver fasta|csv|pool ...\n");
    return 64;
}
