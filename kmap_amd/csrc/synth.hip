// synth.hip -- seeded synthetic reads generated in HBM, in the array contract of the reference's preprocessing
// (uint8 codes 0..3, one 255 separator after every read, (n_seq, 2) int64 borders: kmer_count.py:244-347).
//
// Workload generator for the benchmark configurations that are too large to build on the host inside a benchmark run
// (BASELINE config C5: 50 M x 300 bp = 15 GB; numpy needs minutes for it): the style of the reference's test generator
// (tests/kmap_tests.py:75-114) as SURVEY.md 8(d) fixes it -- fixed-length reads, uniform bases, the first fractions[0] of
// the reads carry motif 0, the next fractions[1] motif 1, ... at a uniform position with per-base substitution rate
// `mutation_rate`, the rest pure random; no N.  Counter-based: every byte is a function of (seed, read, offset), so the
// array does not depend on the launch shape.  Not part of the product path (no reference operator corresponds to it).
#include "common.h"

namespace {

struct SynthMotifs {
    int n;
    int len[4];
    int64_t bound[4];              // reads [bound[m-1], bound[m]) carry motif m
    uint8_t codes[4][32];
};

__device__ __forceinline__ uint64_t mix64(uint64_t x) {      // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// thread = 16 consecutive bytes of the array (one 16-byte store)
__global__ __launch_bounds__(256) void synth_reads_kernel(uint8_t *__restrict__ seq, int64_t n_bytes, int read_len, uint64_t seed,
                                                          SynthMotifs mo, uint32_t mut_thresh) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t p0 = t * 16;
    if (p0 >= n_bytes) return;
    const int row = read_len + 1;
    int64_t r = p0 / row;
    int off = (int)(p0 - r * row);
    uint32_t w[4] = {0, 0, 0, 0};
    int m = -1, mpos = 0;
    auto setup = [&](int64_t read) {
        m = -1;
        for (int i = mo.n - 1; i >= 0; --i)
            if (read < mo.bound[i]) m = i;
        if (m >= 0) mpos = (int)(mix64(seed ^ 0xA5A5A5A5ull ^ ((uint64_t)read * 0x100000001B3ull)) % (uint64_t)(read_len - mo.len[m] + 1));
    };
    setup(r);
    for (int b = 0; b < 16; ++b) {
        const int64_t p = p0 + b;
        uint32_t v = 0;
        if (p < n_bytes) {
            if (off == read_len) {
                v = 255;
            } else {
                const uint64_t h = mix64(seed + (uint64_t)p * 0x9E3779B97F4A7C15ull);
                v = (uint32_t)(h & 3u);
                if (m >= 0 && off >= mpos && off < mpos + mo.len[m] && (uint32_t)(h >> 32) >= mut_thresh) v = mo.codes[m][off - mpos];
            }
        }
        w[b >> 2] |= v << (8 * (b & 3));
        if (++off == row) {
            off = 0;
            ++r;
            setup(r);
        }
    }
    if (p0 + 16 <= n_bytes) {
        *reinterpret_cast<uint4 *>(seq + p0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (int b = 0; b < 16 && p0 + b < n_bytes; ++b) seq[p0 + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
    }
}

__global__ __launch_bounds__(256) void synth_borders_kernel(int64_t *__restrict__ borders, int64_t n_reads, int read_len) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    const int64_t st = r * (int64_t)(read_len + 1);
    borders[2 * r] = st;
    borders[2 * r + 1] = st + read_len;
}

}  // namespace

extern "C" {

int kmap_synth_reads_dev(uint8_t *seq_dev, int64_t *borders_dev, int64_t n_reads, int read_len, uint64_t seed,
                         const uint8_t *motif_codes, const int32_t *motif_len, const double *fractions, int n_motifs,
                         double mutation_rate, void *stream) {
    KMAP_REQUIRE(n_reads >= 0 && read_len > 0, "synth_reads: bad sizes");
    KMAP_REQUIRE(n_motifs >= 0 && n_motifs <= 4, "synth_reads: at most 4 motifs");
    KMAP_REQUIRE(mutation_rate >= 0.0 && mutation_rate <= 1.0, "synth_reads: mutation_rate out of [0, 1]");
    if (n_reads == 0) return KMAP_OK;
    KMAP_REQUIRE(seq_dev && ((uintptr_t)seq_dev % 16) == 0, "synth_reads: seq_dev must be 16-byte aligned");
    SynthMotifs mo;
    memset(&mo, 0, sizeof mo);
    mo.n = n_motifs;
    int64_t acc = 0;
    const uint8_t *src = motif_codes;
    for (int i = 0; i < n_motifs; ++i) {
        KMAP_REQUIRE(motif_codes && motif_len && fractions, "synth_reads: null motif arrays");
        KMAP_REQUIRE(motif_len[i] > 0 && motif_len[i] <= 32 && motif_len[i] <= read_len, "synth_reads: motif %d length out of range", i);
        mo.len[i] = motif_len[i];
        for (int b = 0; b < motif_len[i]; ++b) {
            KMAP_REQUIRE(src[b] < 4, "synth_reads: motif codes must be 0..3");
            mo.codes[i][b] = src[b];
        }
        src += motif_len[i];
        acc += (int64_t)((double)n_reads * fractions[i]);
        mo.bound[i] = acc < n_reads ? acc : n_reads;
    }
    const int64_t n_bytes = n_reads * (int64_t)(read_len + 1);
    const uint32_t mut = (uint32_t)(mutation_rate * 4294967295.0);       // substitution iff the byte's high hash word < mut
    hipStream_t st = as_stream(stream);
    synth_reads_kernel<<<(unsigned)(((n_bytes + 15) / 16 + 255) / 256), 256, 0, st>>>(seq_dev, n_bytes, read_len, seed, mo, mut);
    if (borders_dev) synth_borders_kernel<<<(unsigned)((n_reads + 255) / 256), 256, 0, st>>>(borders_dev, n_reads, read_len);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

}  // extern "C"
