// common.h -- shared host/device helpers for libkmap_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/kmap_hip.h"

#define KMAP_WAVE 64

void kmap_set_error(const char *fmt, ...);

#define KMAP_CHECK_HIP(expr)                                                                      \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            kmap_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (_e == hipErrorOutOfMemory) ? KMAP_E_NOMEM : KMAP_E_HIP;                       \
        }                                                                                         \
    } while (0)

#define KMAP_REQUIRE(cond, ...)                                                                   \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            kmap_set_error(__VA_ARGS__);                                                          \
            return KMAP_E_INVAL;                                                                  \
        }                                                                                         \
    } while (0)

#define KMAP_TRY(expr)                                                                            \
    do {                                                                                          \
        int _r = (expr);                                                                          \
        if (_r != KMAP_OK) return _r;                                                             \
    } while (0)

static inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }

// Scratch arena: cached hipMalloc buffers keyed by (device, stream, slot), grown on demand and
// kept until process exit.  NEVER use hipMallocAsync/hipFreeAsync here: on this ROCm 7.2 / gfx950
// stack, sub-sector (4-byte) stores from workgroups on different XCDs into stream-ordered pool
// memory were observed to be lost (only one XCD's bytes of each 32-byte sector survived), while the
// same kernels on hipMalloc memory are correct.  Work that uses a slot is ordered by its stream.
enum { KMAP_SLOT_A = 0, KMAP_SLOT_B = 1, KMAP_SLOT_C = 2, KMAP_SLOT_D = 3, KMAP_SLOT_HASH = 4, KMAP_SLOT_PART = 5, KMAP_SLOT_BINS = 6 };
int kmap_scratch(void **ptr, size_t bytes, hipStream_t stream, int slot);
// counts.hip: the shared histogram table of `dev` changed hands or was freed -- every handle's cached pointer is stale
void kmap_counts_bins_invalidate(int dev);
// raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) once per (kernel, device); thread-safe
int kmap_allow_lds(const void *kernel, int bytes);

// RAII device scratch buffer for the blocking host-pointer entry points
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int alloc(size_t b) {
        bytes = b;
        if (b == 0) b = 16;
        hipError_t e = hipMalloc(&p, b);
        if (e != hipSuccess) {
            p = nullptr;
            kmap_set_error("hipMalloc(%zu) failed: %s", b, hipGetErrorString(e));
            return KMAP_E_NOMEM;
        }
        return KMAP_OK;
    }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    template <typename T>
    T *as() { return (T *)p; }
};

// ---- device helpers ---------------------------------------------------------------------------
// number of non-zero 2-bit groups of x (x already restricted to the groups of interest)
__device__ __forceinline__ int popc2(uint32_t x) {
    return __builtin_popcount((x | (x >> 1)) & 0x55555555u);
}
__device__ __forceinline__ int popc2(uint64_t x) {
    return __builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull);
}
template <typename H>
__host__ __device__ __forceinline__ H low_mask(int k) {   // mask of the low 2k bits
    return (2 * k >= (int)(8 * sizeof(H))) ? (H)~(H)0 : (H)((((H)1) << (2 * k)) - 1);
}
// reverse complement of a k-mer hash: complement = mask - h, then reverse the 2-bit groups
// (taichi_core.py:181-206); done with a bit-reversal instead of the reference's k-step loop.
__device__ __forceinline__ uint32_t revcom_hash(uint32_t h, int k) {
    uint32_t com = low_mask<uint32_t>(k) - h;               // u32 wrap-around like the reference
    uint32_t r = __builtin_bitreverse32(com);               // reverses bits; fix the order inside pairs
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    return (r >> (32 - 2 * k)) & low_mask<uint32_t>(k);
}
__device__ __forceinline__ uint64_t revcom_hash(uint64_t h, int k) {
    uint64_t com = low_mask<uint64_t>(k) - h;
    uint64_t r = __builtin_bitreverse64(com);
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    return (r >> (64 - 2 * k)) & low_mask<uint64_t>(k);
}
