// peer_exchange.hip -- the receive areas of the peer-direct message exchange of the multi-GPU embedding loop (kmap_hip.h:
// kmap_peer_*): allocation as fine-grained device memory (remote stores and system-scope flag updates are visible to a running
// kernel of the owner), export / import through HIP IPC handles -- across the GPUs of a node, or between processes that share one
// GPU (the one-GPU rehearsal).  The kernels that use them are in embed.hip (kmap_embed_step_peer).
#include <string.h>

#include "embed_internal.h"

static_assert(sizeof(hipIpcMemHandle_t) == KMAP_PEER_HANDLE_BYTES, "IPC handle size");

extern "C" {

int kmap_peer_create(kmap_peer **out, int world, int rank, int64_t msg_floats) {
    KMAP_REQUIRE(out && world >= 1 && world <= KMAP_PEER_MAX && rank >= 0 && rank < world, "peer_create: world must be 1..%d and rank inside it", KMAP_PEER_MAX);
    KMAP_REQUIRE(msg_floats > 0, "peer_create: empty message");
    kmap_peer *p = new kmap_peer();
    p->world = world;
    p->rank = rank;
    p->msg_floats = msg_floats;
    p->slot_floats = (msg_floats + 3) & ~(int64_t)3;
    p->area_bytes = kmap_peer_hello_offset(p) + (size_t)world * 8;
    hipError_t e = hipExtMallocWithFlags(&p->area, p->area_bytes, hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipMemset(p->area, 0, p->area_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&p->msg_local, (size_t)p->slot_floats * 4);
    if (e == hipSuccess) e = hipMemset(p->msg_local, 0, (size_t)p->slot_floats * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&p->done, 16);
    if (e == hipSuccess) e = hipMemset(p->done, 0, 16);
    if (e != hipSuccess) {
        kmap_set_error("peer_create: %s", hipGetErrorString(e));
        kmap_peer_destroy(p);
        return e == hipErrorOutOfMemory ? KMAP_E_NOMEM : KMAP_E_HIP;
    }
    p->peer_area[rank] = p->area;
    *out = p;
    return KMAP_OK;
}

int kmap_peer_handle(kmap_peer *p, void *handle_out) {
    KMAP_REQUIRE(p && p->area && handle_out, "peer_handle: null");
    hipIpcMemHandle_t h;
    KMAP_CHECK_HIP(hipIpcGetMemHandle(&h, p->area));
    memcpy(handle_out, &h, sizeof h);
    return KMAP_OK;
}

int kmap_peer_bus_id(char *bus_id_out) {
    KMAP_REQUIRE(bus_id_out, "peer_bus_id: null");
    int dev = 0;
    KMAP_CHECK_HIP(hipGetDevice(&dev));
    memset(bus_id_out, 0, KMAP_PEER_BUS_ID_BYTES);
    KMAP_CHECK_HIP(hipDeviceGetPCIBusId(bus_id_out, KMAP_PEER_BUS_ID_BYTES - 1, dev));
    return KMAP_OK;
}

// can the current device store into memory of the device with this PCI bus id: the same device (ranks sharing a GPU), or a device
// of this process's view that hipDeviceCanAccessPeer allows; a device this process does not see cannot be judged -> 0
int kmap_peer_can_access(const char *bus_id, int *can_access) {
    KMAP_REQUIRE(bus_id && can_access, "peer_can_access: null");
    *can_access = 0;
    int dev = 0, other = -1;
    KMAP_CHECK_HIP(hipGetDevice(&dev));
    char mine[KMAP_PEER_BUS_ID_BYTES] = {0};
    KMAP_CHECK_HIP(hipDeviceGetPCIBusId(mine, KMAP_PEER_BUS_ID_BYTES - 1, dev));
    if (strncmp(mine, bus_id, KMAP_PEER_BUS_ID_BYTES) == 0) {
        *can_access = 1;
        return KMAP_OK;
    }
    if (hipDeviceGetByPCIBusId(&other, bus_id) != hipSuccess || other < 0) {
        (void)hipGetLastError();
        return KMAP_OK;                                      // not visible here (HIP_VISIBLE_DEVICES): unknown = no
    }
    int can = 0;
    KMAP_CHECK_HIP(hipDeviceCanAccessPeer(&can, dev, other));
    *can_access = can ? 1 : 0;
    return KMAP_OK;
}

int kmap_peer_connect(kmap_peer *p, const void *handles) {
    KMAP_REQUIRE(p && handles, "peer_connect: null");
    for (int q = 0; q < p->world; ++q) {
        if (q == p->rank || p->opened[q]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + (size_t)q * KMAP_PEER_HANDLE_BYTES, sizeof h);
        void *ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess || !ptr) {
            (void)hipGetLastError();
            kmap_set_error("peer_connect: rank %d's receive area cannot be mapped (hipIpcOpenMemHandle: %s)", q, hipGetErrorString(e));
            return KMAP_E_HIP;
        }
        p->peer_area[q] = ptr;
        p->opened[q] = true;
    }
    return KMAP_OK;
}

namespace {
__global__ void peer_hello_kernel(PeerTab tab, size_t hello_off_words, int world, int rank, unsigned long long word) {
    const int q = (int)threadIdx.x;
    if (q < world) {
        unsigned long long *hello = reinterpret_cast<unsigned long long *>(tab.slots[q]) + hello_off_words;
        __hip_atomic_store(hello + rank, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
}  // namespace

int kmap_peer_hello_push(kmap_peer *p, uint64_t token) {
    KMAP_REQUIRE(p, "peer_hello_push: null");
    PeerTab tab;
    memset(&tab, 0, sizeof tab);
    for (int q = 0; q < p->world; ++q) {
        KMAP_REQUIRE(p->peer_area[q], "peer_hello_push: rank %d's receive area is not connected", q);
        tab.slots[q] = (float *)p->peer_area[q];
    }
    peer_hello_kernel<<<1, 64>>>(tab, kmap_peer_hello_offset(p) / 8, p->world, p->rank, (unsigned long long)token + (unsigned long long)p->rank + 1ull);
    KMAP_CHECK_HIP(hipGetLastError());
    KMAP_CHECK_HIP(hipDeviceSynchronize());                 // a store that faults surfaces here, as an error of this call
    return KMAP_OK;
}

int kmap_peer_hello_check(kmap_peer *p, uint64_t token, int *n_missing) {
    KMAP_REQUIRE(p && n_missing, "peer_hello_check: null");
    unsigned long long w[KMAP_PEER_MAX] = {0};
    KMAP_CHECK_HIP(hipMemcpy(w, (const char *)p->area + kmap_peer_hello_offset(p), (size_t)p->world * 8, hipMemcpyDeviceToHost));
    int miss = 0;
    for (int q = 0; q < p->world; ++q) miss += w[q] != (unsigned long long)token + (unsigned long long)q + 1ull;
    *n_missing = miss;
    return KMAP_OK;
}

int kmap_peer_set_timeout_ms(kmap_peer *p, int64_t ms) {
    KMAP_REQUIRE(p && ms > 0, "peer_set_timeout_ms: the bound must be positive");
    p->timeout_ticks = (unsigned long long)ms * 100000ull;   // wall_clock64: 100 MHz
    return KMAP_OK;
}

int kmap_peer_status(kmap_peer *p, int *timed_out, int64_t *iterations) {
    KMAP_REQUIRE(p, "peer_status: null");
    unsigned long long w[2] = {0, 0};
    KMAP_CHECK_HIP(hipMemcpy(w, p->done, 16, hipMemcpyDeviceToHost));   // blocking: everything issued so far has run
    if (timed_out) *timed_out = w[1] != 0;
    if (iterations) *iterations = (int64_t)p->iter;
    return KMAP_OK;
}

int kmap_peer_destroy(kmap_peer *p) {
    if (!p) return KMAP_OK;
    (void)hipDeviceSynchronize();
    for (int q = 0; q < p->world; ++q)
        if (p->opened[q] && p->peer_area[q]) (void)hipIpcCloseMemHandle(p->peer_area[q]);
    if (p->area) (void)hipFree(p->area);
    if (p->msg_local) (void)hipFree(p->msg_local);
    if (p->done) (void)hipFree(p->done);
    delete p;
    return KMAP_OK;
}

}  // extern "C"
