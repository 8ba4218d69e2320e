// peer_exchange.hip -- the receive areas of the peer-direct message exchange of the multi-GPU embedding loop (kmap_hip.h:
// kmap_peer_*): allocation as fine-grained device memory (remote stores and system-scope flag updates are visible to a running
// kernel of the owner), export / import through HIP IPC handles -- across the GPUs of a node, or between processes that share one
// GPU (the one-GPU rehearsal).  The kernels that use them are in embed.hip (kmap_embed_step_peer).
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
    p->area_bytes = kmap_peer_slots_bytes(p) + (size_t)2 * world * 8;
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

int kmap_peer_connect(kmap_peer *p, const void *handles) {
    KMAP_REQUIRE(p && handles, "peer_connect: null");
    for (int q = 0; q < p->world; ++q) {
        if (q == p->rank || p->opened[q]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + (size_t)q * KMAP_PEER_HANDLE_BYTES, sizeof h);
        void *ptr = nullptr;
        KMAP_CHECK_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        p->peer_area[q] = ptr;
        p->opened[q] = true;
    }
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
