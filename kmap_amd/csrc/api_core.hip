// api_core.hip -- library/device/memory/event entry points of the C ABI (include/kmap_hip.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void kmap_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

#include <map>
#include <mutex>
#include <tuple>

namespace {
struct Scratch {
    void *p = nullptr;
    size_t bytes = 0;
};
std::mutex g_scratch_mu;
std::map<std::tuple<int, void *, int>, Scratch> g_scratch;
}  // namespace

int kmap_scratch(void **ptr, size_t bytes, hipStream_t stream, int slot) {
    int dev = 0;
    KMAP_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    Scratch &sc = g_scratch[std::make_tuple(dev, (void *)stream, slot)];
    if (sc.bytes < bytes || !sc.p) {
        if (sc.p) {
            KMAP_CHECK_HIP(hipStreamSynchronize(stream));   // earlier users of the old buffer
            KMAP_CHECK_HIP(hipFree(sc.p));
            sc.p = nullptr;
            sc.bytes = 0;
        }
        size_t want = bytes < 4096 ? 4096 : bytes + bytes / 8;   // head-room against regrowth
        hipError_t e = hipMalloc(&sc.p, want);
        if (e != hipSuccess) {
            sc.p = nullptr;
            kmap_set_error("scratch allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
            return KMAP_E_NOMEM;
        }
        sc.bytes = want;
    }
    *ptr = sc.p;
    return KMAP_OK;
}

// frees the cached buffers of at least `min_bytes` on the current device (all streams, all slots); the device is synchronised
// first, so no work using them is in flight
static int scratch_release(size_t min_bytes) {
    int dev = 0;
    KMAP_CHECK_HIP(hipGetDevice(&dev));
    KMAP_CHECK_HIP(hipDeviceSynchronize());
    bool bins = false;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        for (auto it = g_scratch.begin(); it != g_scratch.end();) {
            if (std::get<0>(it->first) == dev && it->second.bytes >= min_bytes) {
                if (it->second.p) (void)hipFree(it->second.p);
                bins = bins || std::get<2>(it->first) == KMAP_SLOT_BINS;
                it = g_scratch.erase(it);
            } else {
                ++it;
            }
        }
    }
    if (bins) kmap_counts_bins_invalidate(dev);
    return KMAP_OK;
}

int kmap_allow_lds(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> done;     // (kernel, device) -> bytes granted
    int dev = 0;
    KMAP_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    int &have = done[std::make_pair(kernel, dev)];
    if (have < bytes) {
        // the library is built for gfx950 (160 KiB of LDS per workgroup); say so instead of failing inside the launch elsewhere
        int limit = 0;
        KMAP_CHECK_HIP(hipDeviceGetAttribute(&limit, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
        KMAP_REQUIRE(bytes <= limit, "a kernel needs %d bytes of LDS per workgroup, device %d offers %d (this library targets MI355X / gfx950)", bytes, dev, limit);
        KMAP_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        have = bytes;
    }
    return KMAP_OK;
}

extern "C" {

int kmap_version(void) { return 1000 * 0 + 1; }
int kmap_scratch_release(size_t min_bytes) { return scratch_release(min_bytes); }
const char *kmap_last_error(void) { return g_err; }

int kmap_device_count(int *n) {
    KMAP_REQUIRE(n, "kmap_device_count: null");
    KMAP_CHECK_HIP(hipGetDeviceCount(n));
    return KMAP_OK;
}
int kmap_set_device(int dev) {
    KMAP_CHECK_HIP(hipSetDevice(dev));
    return KMAP_OK;
}
int kmap_get_device(int *dev) {
    KMAP_REQUIRE(dev, "kmap_get_device: null");
    KMAP_CHECK_HIP(hipGetDevice(dev));
    return KMAP_OK;
}
int kmap_device_arch(char *buf, int buflen) {
    KMAP_REQUIRE(buf && buflen > 0, "kmap_device_arch: bad buffer");
    int dev = 0;
    KMAP_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    KMAP_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    snprintf(buf, (size_t)buflen, "%s", p.gcnArchName);
    return KMAP_OK;
}

int kmap_malloc(void **dev_ptr, size_t bytes) {
    KMAP_REQUIRE(dev_ptr, "kmap_malloc: null");
    KMAP_CHECK_HIP(hipMalloc(dev_ptr, bytes ? bytes : 16));
    return KMAP_OK;
}
int kmap_free(void *dev_ptr) {
    if (dev_ptr) KMAP_CHECK_HIP(hipFree(dev_ptr));
    return KMAP_OK;
}
int kmap_memset(void *dev_ptr, int value, size_t bytes, void *stream) {
    if (bytes) KMAP_CHECK_HIP(hipMemsetAsync(dev_ptr, value, bytes, as_stream(stream)));
    return KMAP_OK;
}
int kmap_memcpy_h2d(void *d, const void *h, size_t bytes, void *stream) {
    if (!bytes) return KMAP_OK;
    KMAP_CHECK_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    KMAP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));   // pageable host memory: keep it simple
    return KMAP_OK;
}
int kmap_memcpy_d2h(void *h, const void *d, size_t bytes, void *stream) {
    if (!bytes) return KMAP_OK;
    KMAP_CHECK_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    KMAP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    return KMAP_OK;
}
int kmap_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes) KMAP_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return KMAP_OK;
}
int kmap_memcpy2d_d2h(void *host_dst, size_t hpitch, const void *dev_src, size_t dpitch, size_t width, size_t rows,
                      void *stream) {
    if (!width || !rows) return KMAP_OK;
    KMAP_CHECK_HIP(hipMemcpy2DAsync(host_dst, hpitch, dev_src, dpitch, width, rows, hipMemcpyDeviceToHost,
                                    as_stream(stream)));
    KMAP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    return KMAP_OK;
}
int kmap_stream_sync(void *stream) {
    KMAP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    return KMAP_OK;
}
int kmap_stream_create(void **stream) {
    KMAP_REQUIRE(stream, "kmap_stream_create: null");
    hipStream_t s;
    KMAP_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return KMAP_OK;
}
int kmap_stream_destroy(void *stream) {
    if (stream) KMAP_CHECK_HIP(hipStreamDestroy(as_stream(stream)));
    return KMAP_OK;
}
int kmap_event_create(void **ev) {
    KMAP_REQUIRE(ev, "kmap_event_create: null");
    hipEvent_t e;
    KMAP_CHECK_HIP(hipEventCreate(&e));
    *ev = (void *)e;
    return KMAP_OK;
}
int kmap_event_destroy(void *ev) {
    if (ev) KMAP_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
    return KMAP_OK;
}
int kmap_event_record(void *ev, void *stream) {
    KMAP_CHECK_HIP(hipEventRecord((hipEvent_t)ev, as_stream(stream)));
    return KMAP_OK;
}
int kmap_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms) {
    KMAP_REQUIRE(ms, "kmap_event_elapsed_ms: null");
    KMAP_CHECK_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    KMAP_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return KMAP_OK;
}

}  // extern "C"
