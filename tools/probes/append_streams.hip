// append_streams.hip -- can a counting sort with MANY buckets (4096) append short runs (16..128 B) to its bucket streams at a
// useful rate on a part with eight L2s?  Every block iteration ("tile") reserves RUN bytes in each of NB streams with one
// returning atomic per stream and stores the run.  Variables: NB, RUN, cursor per stream (classes = 1) or per (stream, XCD class
// = blockIdx % 8; cursors laid out [class][stream] so that the classes do not share lines) (classes = 8: a stream's partial lines live in ONE L2), atomic scope.
//   hipcc --offload-arch=gfx950 -O3 -o append_streams append_streams.hip && ./append_streams
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int RUN, int SCOPE>
__global__ __launch_bounds__(1024) void append_kernel(unsigned long long *__restrict__ cursor, unsigned char *__restrict__ out, int nb,
                                                      int classes, int tiles_per_block) {
    const int cls = classes == 1 ? 0 : (int)(blockIdx.x % classes);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    for (int t = 0; t < tiles_per_block; ++t)
        for (int s = threadIdx.x; s < nb; s += 1024) {
            const unsigned long long at = __hip_atomic_fetch_add(&cursor[(size_t)cls * nb + s], (unsigned long long)RUN, __ATOMIC_RELAXED, SCOPE);
            u32x4 v = {(unsigned)s, (unsigned)t, blockIdx.x, 7u};
#pragma unroll
            for (int o = 0; o < RUN; o += 16) *reinterpret_cast<u32x4 *>(out + at + o) = v;
        }
}
// only the atomics
template <int SCOPE>
__global__ __launch_bounds__(1024) void atomics_kernel(unsigned long long *__restrict__ cursor, int nb, int classes, int tiles_per_block,
                                                       unsigned long long *__restrict__ sink) {
    const int cls = classes == 1 ? 0 : (int)(blockIdx.x % classes);
    unsigned long long acc = 0;
    for (int t = 0; t < tiles_per_block; ++t)
        for (int s = threadIdx.x; s < nb; s += 1024)
            acc += __hip_atomic_fetch_add(&cursor[(size_t)cls * nb + s], 16ull, __ATOMIC_RELAXED, SCOPE);
    if (acc == 1) *sink = acc;
}

template <int RUN, int SCOPE>
static void run(int nb, int classes, const char *scope_name) {
    const int blocks = 1024;
    const size_t total = (size_t)3 << 30;                                  // bytes appended
    const int tiles = (int)(total / ((size_t)blocks * nb * RUN));
    const size_t per_stream = (size_t)tiles * RUN * blocks / classes;     // bytes per (stream, class)
    unsigned long long *cursor;
    unsigned char *out;
    hipMalloc(&cursor, (size_t)nb * classes * 8);
    hipMalloc(&out, (size_t)nb * classes * per_stream + 4096);
    unsigned long long *h = (unsigned long long *)malloc((size_t)nb * classes * 8);
    for (size_t i = 0; i < (size_t)nb * classes; ++i) h[i] = ((i % nb) * classes + i / nb) * per_stream;   // a stream's classes side by side
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemcpy(cursor, h, (size_t)nb * classes * 8, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        append_kernel<RUN, SCOPE><<<blocks, 1024>>>(cursor, out, nb, classes, tiles);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)tiles * blocks * nb * RUN;
    printf("append  NB %5d  run %4d B  classes %d  scope %-9s  %7.3f ms  %6.2f TB/s  %6.1f G atomics/s\n", nb, RUN, classes, scope_name, best,
           bytes / best * 1e-9, bytes / RUN / best * 1e-6);
    fflush(stdout);
    hipFree(cursor);
    hipFree(out);
    free(h);
}
template <int SCOPE>
static void run_atomics(int nb, int classes, const char *scope_name) {
    const int blocks = 1024, tiles = 64;
    unsigned long long *cursor, *sink;
    hipMalloc(&cursor, (size_t)nb * classes * 8 + 8);
    hipMemset(cursor, 0, (size_t)nb * classes * 8 + 8);
    sink = cursor + (size_t)nb * classes;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        atomics_kernel<SCOPE><<<blocks, 1024>>>(cursor, nb, classes, tiles, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("atomics NB %5d  classes %d  scope %-9s  %7.3f ms  %6.1f G atomics/s\n", nb, classes, scope_name, best,
           (double)tiles * blocks * nb / best * 1e-6);
    fflush(stdout);
    hipFree(cursor);
}

int main() {
    for (int nb : {1024, 4096}) {
        for (int classes : {1, 8}) {
            run_atomics<__HIP_MEMORY_SCOPE_AGENT>(nb, classes, "agent");
            run_atomics<__HIP_MEMORY_SCOPE_WORKGROUP>(nb, classes, "workgroup");
        }
    }
    for (int nb : {1024, 4096})
        for (int classes : {1, 8}) {
            run<16, __HIP_MEMORY_SCOPE_AGENT>(nb, classes, "agent");
            run<32, __HIP_MEMORY_SCOPE_AGENT>(nb, classes, "agent");
            run<64, __HIP_MEMORY_SCOPE_AGENT>(nb, classes, "agent");
            run<128, __HIP_MEMORY_SCOPE_AGENT>(nb, classes, "agent");
            run<16, __HIP_MEMORY_SCOPE_WORKGROUP>(nb, classes, "workgroup");
            run<64, __HIP_MEMORY_SCOPE_WORKGROUP>(nb, classes, "workgroup");
        }
    return 0;
}
