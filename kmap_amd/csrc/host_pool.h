// host_pool.h -- pure host code shared by counts.hip and the CPU sanitizer harness (tests/host_san): element-wise conversion of a
// fetched table chunk on several host threads.  No HIP types: the harness compiles this with -fsanitize=address,undefined and
// -fsanitize=thread.
#pragma once
#include <stddef.h>
#include <string.h>

#include <thread>
#include <type_traits>
#include <vector>

// dst[i] = (DST)src[i] for i in [0, n) on up to `nt` threads.  `dst` may be unaligned (a view into a memory-mapped pickle file:
// byte-wise typed stores).  Never throws across the caller (a C ABI function): if threads cannot be started, the rest is converted
// on the calling thread.
template <typename SRC, typename DST>
inline void kmap_convert_pool(DST *dst, const SRC *src, size_t n, unsigned nt) {
    typedef DST __attribute__((aligned(1))) DSTu;
    auto part = [=](size_t lo, size_t hi) {
        if (std::is_same<SRC, DST>::value) {
            memcpy((void *)((char *)dst + lo * sizeof(DST)), (const void *)(src + lo), (hi - lo) * sizeof(SRC));
        } else {
            DSTu *du = (DSTu *)dst;
            for (size_t i = lo; i < hi; ++i) du[i] = (DST)src[i];
        }
    };
    if (nt < 1) nt = 1;
    if (n < ((size_t)1 << 16)) nt = 1;                                // not worth a thread
    std::vector<std::thread> pool;
    size_t done_to = 0;                                               // entries [0, done_to) are covered by started threads
    try {
        for (unsigned t = 0; t + 1 < nt; ++t) {
            const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
            pool.emplace_back(part, lo, hi);
            done_to = hi;
        }
    } catch (...) {
    }
    part(done_to, n);                                                 // the last share (or everything left) on this thread
    for (auto &th : pool) th.join();
}
