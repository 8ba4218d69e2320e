// embed.hip -- the 2-D embedding loop (reference visualization.py:259-326, kernels taichi_core.py:252-326), device resident:
// loss reductions, the per-iteration apply kernels, the session API and the drop-in float operators.  The force kernels live in
// embed_fast.hip / embed_seq.hip (embed_internal.h), the smoothing step in knn_smooth.hip / knn_profile.hip.
//
//  * forces     : one pass over the rows a GPU owns: q_ij, clip, cross-entropy partial (j > i),
//                 T_ij = q/(1-q)*(p-q), g_i = sum_j T_ij (y_i - y_j).  p comes either from an f32 matrix
//                 or from LUT[sums[i,j]] (the LUT holds the reference's numpy-evaluated
//                 exp(-sigmoid(s/n_nb/n_nb)/0.5) for every possible integer sum).
//  * apply      : loss -> best-list insert (bisect.insort_right) -> early-stop test -> y += -(4 g) lr
//                 -> add_jitter (as written in the reference: only points 0 and 1 are ever touched),
//                 all on device; the host only pre-draws the jitter normals from numpy's RNG stream.
//
// Everything float here is compiled with -ffp-contract=off and no fast-math: the per-pair values of SEQ mode are
// bit-identical to numpy's f32 scalar arithmetic; FAST mode differs from the reference in the last bits of the per-pair
// values and in the order of the row sums.
#include <math.h>
#include <stdlib.h>

#include <vector>

#include <type_traits>

#include "embed_internal.h"

namespace {

constexpr int BLK = EMB_BLK;

// =================================================================================================
// per-pair arithmetic shared by all force kernels (IEEE f32, numpy scalar order)
// =================================================================================================
struct Pair {
    float q, t, ce;
};
__device__ __forceinline__ float q_of(float dx, float dy) {
    const float d2 = dx * dx + dy * dy;              // (dx*dx) + (dy*dy), no FMA (taichi_core.py:254)
    float q = 1.0f / (1.0f + d2);                    // :255
    q = fminf(q, 0.999f);                            // np.minimum(prob, 1 - 1e-3)   visualization.py:254
    q = fmaxf(q, 0.001f);                            // np.maximum(prob, 1e-3)       visualization.py:255
    return q;
}
__device__ __forceinline__ float t_of(float p, float q) {
    return (q / (1.0f - q)) * (p - q);               // visualization.py:132-134
}
template <bool EXACT_LOG>
__device__ __forceinline__ float ce_of(float p, float q) {
    // taichi_core.py:279-303: eps = 1e-10 branches (q is already clipped to [1e-3, 1-1e-3]); branch-free.
    // !EXACT_LOG: v_log_f32 (log2, 1 ulp) * ln2 -- arguments lie in [1e-3, 0.999], no denormal handling needed.
    const float eps = 1e-10f;
    const float lq = EXACT_LOG ? logf(q) : __builtin_amdgcn_logf(q) * 0.69314718f;
    const float l1q = EXACT_LOG ? logf(1.0f - q) : __builtin_amdgcn_logf(1.0f - q) * 0.69314718f;
    const float full = -p * lq - (1.0f - p) * l1q;
    const float hi = (p > 1.0f - eps) ? -lq : full;
    return (p < eps) ? -l1q : hi;
}

// deterministic reduction of the per-block loss partials inside the fused apply kernel (every block computes the same total):
// thread t adds part[t], part[t + 256], ... in order, then a fixed tree over the 256 thread sums
__device__ __forceinline__ double loss_total_256(const double *__restrict__ part, int n_part, double *sh /* [256] LDS */) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n_part; i += BLK) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = BLK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    const double total = sh[0];
    __syncthreads();
    return total;
}
// standalone reduction (symmetric kernels: ~2 10^4 partials; multi-GPU: a rank's partials before the all-reduce): 1024 threads,
// thread t adds part[t], part[t + 1024], ... in order, then a fixed tree (a 256-thread block took 30 us on 19 600 partials)
constexpr int RL_TPB = 1024;
__global__ __launch_bounds__(RL_TPB) void reduce_loss_kernel(const double *__restrict__ part, int n_part,
                                                             double *__restrict__ loss_out) {
    __shared__ double sh[RL_TPB];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_part; i += RL_TPB) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = RL_TPB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = sh[0];
}

// ---- multi-GPU one-message protocol ---------------------------------------------------------------
// A rank's contribution to an iteration is ONE float buffer: [0, 2N) its gradient entries (own rows, or partial sums for all
// points in the cyclic layout) and MSG_EXTRA floats behind them that carry its loss partial as an exact integer: the f64 value
// in 48.48 fixed point, cut into six 16-bit limbs, each stored as a float.  A float32 SUM all-reduce adds limbs of up to 256
// ranks without rounding (6 x < 2^24), in any order, so every rank decodes the same total whatever algorithm the collective
// library picks -- one collective per iteration instead of a float32 and a float64 one, and the stop / snapshot decisions
// (which come from the loss alone) cannot diverge between ranks.  Limb 6 flags a non-finite or out-of-range partial (-> NaN).
__device__ __forceinline__ void loss_to_limbs(double v, float *__restrict__ tail) {
    const bool bad = !(v >= 0.0) || !(v < 140737488355328.0);            // NaN, negative or >= 2^47
    unsigned long long hi = 0, lo = 0;
    if (!bad) {
        hi = (unsigned long long)v;                                       // floor (v >= 0)
        lo = (unsigned long long)((v - (double)hi) * 281474976710656.0);  // exact difference, truncated at 2^-48
    }
    for (int i = 0; i < 3; ++i) tail[i] = (float)((lo >> (16 * i)) & 0xFFFFull);
    for (int i = 0; i < 3; ++i) tail[3 + i] = (float)((hi >> (16 * i)) & 0xFFFFull);
    tail[6] = bad ? 1.0f : 0.0f;
    tail[7] = 0.0f;
}
__device__ __forceinline__ double loss_from_limbs(const float *__restrict__ tail) {
    if (tail[6] != 0.0f) return (double)NAN;
    unsigned long long lo = 0, hi = 0;                                    // summed limbs carry past 16 bits: integer Horner
    for (int i = 2; i >= 0; --i) lo = (lo << 16) + (unsigned long long)tail[i];
    for (int i = 2; i >= 0; --i) hi = (hi << 16) + (unsigned long long)tail[3 + i];
    hi += lo >> 48;
    lo &= 0xFFFFFFFFFFFFull;
    return (double)hi + (double)lo * (1.0 / 281474976710656.0);
}
// reduce_loss_kernel with the total written as limbs behind the gradient (n_part = 0: a rank without rows sends zero)
__global__ __launch_bounds__(RL_TPB) void reduce_loss_limbs_kernel(const double *__restrict__ part, int n_part,
                                                                   float *__restrict__ tail) {
    __shared__ double sh[RL_TPB];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_part; i += RL_TPB) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = RL_TPB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_to_limbs(sh[0], tail);
}

// =================================================================================================
// loop state + apply
// =================================================================================================

// what one reference iteration decides from the loss (visualization.py:303-311), identical in every thread
struct StepDecision {
    float loss;
    int slot;               // snapshot buffer that takes the iterate if `insert`
    bool halted, insert, stop;
};
__device__ __forceinline__ StepDecision step_decide(const LoopState *__restrict__ st, double loss_total) {
    StepDecision d;
    d.halted = st->stopped != 0;
    d.loss = (float)(2.0 * loss_total);                                  // np.sum(ce) * 2 (visualization.py:176)
    d.insert = d.loss < st->worst_loss;                                  // :303 (false for NaN)
    d.stop = fabsf(st->prev_loss - d.loss) < 1e-7f * fabsf(d.loss);      // :310
    d.slot = st->worst_slot;
    return d;
}
// one coordinate (not point 0 / 1, which the leader owns): snapshot, then the gradient step
__device__ __forceinline__ void step_element(const StepDecision &d, int64_t idx, float g_raw, float *__restrict__ Y,
                                             float *__restrict__ snaps, int64_t n, float lr) {
    const float y = Y[idx];
    if (d.insert) snaps[(int64_t)d.slot * 2 * n + idx] = y;             // snapshot of the iterate that produced `loss`
    if (!d.stop) {
        const float g = 4.0f * g_raw;                                    // gradient_loss_taichi returns 4.0 * ret (:145)
        Y[idx] = y + (-g * lr);                                          // ld_data += (-grad_loss * learning_rate) (:316)
    }
}
// the leader thread: loop record (best list, stop flag, loss log) and points 0 / 1 incl. add_jitter.  gsp = raw gradient of
// x0, x1, y0, y1
__device__ void step_leader(LoopState *__restrict__ states, int cur, const StepDecision &d, float *__restrict__ Y, const float (&gsp)[4],
                            float *__restrict__ snaps, int64_t n, float lr, const double *__restrict__ normals,
                            const int *__restrict__ n_normals_dev, float *__restrict__ loss_log, int64_t loss_log_cap) {
    const LoopState st = states[cur];
    if (d.halted) {
        states[cur ^ 1] = st;
        return;
    }
    const int nb = st.n_best, slot = d.slot;
    const float loss = d.loss;
    LoopState ns = st;
    ns.iters = st.iters + 1;
    ns.last_loss = loss;
    if (st.iters < loss_log_cap) loss_log[st.iters] = loss;
    if (d.insert) {   // best_res_list[:-1] then bisect.insort_right by loss (:304-308)
        int pos = 0;
        while (pos < nb - 1 && !(st.best_loss[pos] > loss)) ++pos;
        for (int t = nb - 1; t > pos; --t) {
            ns.best_loss[t] = st.best_loss[t - 1];
            ns.best_slot[t] = st.best_slot[t - 1];
        }
        ns.best_loss[pos] = loss;
        ns.best_slot[pos] = slot;
        ns.worst_loss = ns.best_loss[nb - 1];
        ns.worst_slot = ns.best_slot[nb - 1];
        for (int p = 0; p < 2 && p < n; ++p)
            for (int c = 0; c < 2; ++c) snaps[(int64_t)slot * 2 * n + (int64_t)c * n + p] = Y[(int64_t)c * n + p];
    }
    if (d.stop) {
        ns.stopped = 1;
    } else {
        ns.prev_loss = loss;
        const int n_normals = *n_normals_dev;
        // update + add_jitter for points 0 and 1 (visualization.py:179-196 indexes the 2 x N array as N x 2,
        // so `ld_data[:, p]` is the (x_p, y_p) pair of point p)
        for (int p = 0; p < 2 && p < n; ++p) {
            float v[2];
            for (int c = 0; c < 2; ++c) {
                const float g = 4.0f * gsp[2 * c + p];
                v[c] = Y[(int64_t)c * n + p] + (-g * lr);
            }
            const int lo_i = (v[1] < v[0]) ? 1 : 0;                     // argsort of two values (stable)
            const float diff = v[1 - lo_i] - v[lo_i];                   // np.diff of the sorted pair
            if (diff < 0.1f) {
                const double nrm = (ns.jitter_used < n_normals) ? normals[ns.jitter_used] : 0.0;
                ns.jitter_used += 1;
                v[lo_i] = (float)((double)v[lo_i] + nrm);               // f32 array element += f64 draw
            }
            for (int c = 0; c < 2; ++c) Y[(int64_t)c * n + p] = v[c];
        }
    }
    states[cur ^ 1] = ns;
}

// apply with the gradient in memory.  FUSED_LOSS: the block first reduces the force kernel's loss partials itself (single-GPU
// step: forces -> this kernel); otherwise the total comes from loss_in[0] (multi-GPU: after the all-reduce)
template <bool FUSED_LOSS>
__global__ __launch_bounds__(BLK) void apply_kernel(LoopState *__restrict__ states, int cur, float *__restrict__ Y,
                                                    const float *__restrict__ G, const double *__restrict__ loss_in, int n_part,
                                                    float *__restrict__ snaps, int64_t n, float lr,
                                                    const double *__restrict__ normals, const int *__restrict__ n_normals_dev,
                                                    float *__restrict__ loss_log, int64_t loss_log_cap) {
    __shared__ double sh[FUSED_LOSS ? BLK : 1];
    const double total = FUSED_LOSS ? loss_total_256(loss_in, n_part, sh) : loss_in[0];
    const StepDecision d = step_decide(&states[cur], total);   // every block reads the same, already complete record
    const int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const bool special = (idx == 0 || idx == 1 || idx == n || idx == n + 1);   // owned by the leader (jitter)
    if (!d.halted && idx < 2 * n && !special) step_element(d, idx, G[idx], Y, snaps, n, lr);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float gsp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int c = 0; c < 2; ++c)
            for (int p = 0; p < 2 && p < n; ++p) gsp[2 * c + p] = G[(int64_t)c * n + p];
        step_leader(states, cur, d, Y, gsp, snaps, n, lr, normals, n_normals_dev, loss_log, loss_log_cap);
    }
}

// apply for the one-message protocol: the summed message M = [gradient 2N | loss limbs].  CLEAR (row-sharded sessions, whose
// force kernel fills only the rank's own rows): every entry is zeroed once it has been read, so the next iteration's message
// starts from x + 0 + ... + 0 without a memset launch (the cyclic layout overwrites all 2N entries itself).
template <bool CLEAR>
__global__ __launch_bounds__(BLK) void apply_msg_kernel(LoopState *__restrict__ states, int cur, float *__restrict__ Y,
                                                        float *__restrict__ M, float *__restrict__ snaps, int64_t n, float lr,
                                                        const double *__restrict__ normals, const int *__restrict__ n_normals_dev,
                                                        float *__restrict__ loss_log, int64_t loss_log_cap) {
    const double total = loss_from_limbs(M + 2 * n);
    const StepDecision d = step_decide(&states[cur], total);
    const int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const bool special = (idx == 0 || idx == 1 || idx == n || idx == n + 1);   // owned by the leader (jitter)
    if (idx < 2 * n && !special) {
        const float g = M[idx];
        if (CLEAR) M[idx] = 0.0f;
        if (!d.halted) step_element(d, idx, g, Y, snaps, n, lr);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float gsp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int c = 0; c < 2; ++c)
            for (int p = 0; p < 2 && p < n; ++p) {
                gsp[2 * c + p] = M[(int64_t)c * n + p];
                if (CLEAR) M[(int64_t)c * n + p] = 0.0f;
            }
        step_leader(states, cur, d, Y, gsp, snaps, n, lr, normals, n_normals_dev, loss_log, loss_log_cap);
    }
}

// ---- peer-direct exchange (kmap_hip.h: kmap_embed_step_peer) -------------------------------------------------------------------
// push: this rank's message -> slot [parity][rank] of EVERY rank's receive area (own included), 16 bytes per thread and peer;
// every block fences its stores system-wide and counts itself done; the block that finds itself last publishes the iteration
// number in the slot's flag on every rank (release, system scope).
__global__ __launch_bounds__(BLK) void peer_push_kernel(PeerTab tab, const float *__restrict__ msg, int64_t msg_floats, int world, int rank,
                                                        int parity, unsigned long long iter_tag, unsigned long long *__restrict__ done) {
    const int64_t i4 = ((int64_t)blockIdx.x * BLK + threadIdx.x) * 4;
    const size_t slot = ((size_t)parity * world + rank) * (size_t)msg_floats;
    if (i4 < msg_floats) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(msg + i4);
        for (int q = 0; q < world; ++q) *reinterpret_cast<f32x4 *>(tab.slots[q] + slot + i4) = v;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long before = atomicAdd(done, 1ull);
        if (before == (unsigned long long)gridDim.x - 1) {                 // every block's stores are fenced: publish
            *done = 0;
            __threadfence_system();
            for (int q = 0; q < world; ++q)
                __hip_atomic_store(tab.flags[q] + (size_t)parity * world + rank, iter_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// apply for the peer protocol: wait for the iteration's world flags in the own area (bounded: `timeout_ticks` of the 100 MHz wall
// clock; then the sticky time-out word is set and this block leaves the iteration unapplied -- the host raises after the segment
// and discards the coordinates), add the world slots in rank order, decode the loss limbs of the sum, apply.
// slots / flags are written by OTHER GPUs while this kernel runs: no const / __restrict__ on them (that would license scalar or
// non-coherent loads of "memory that does not change during the dispatch"), and every thread fences (system-scope acquire) behind
// the barrier before it reads a slot.
__global__ __launch_bounds__(BLK) void apply_peer_kernel(LoopState *__restrict__ states, int cur, float *__restrict__ Y,
                                                         float *slots, unsigned long long *flags,
                                                         int world, int parity, unsigned long long iter_tag, int64_t msg_floats,
                                                         unsigned long long *timed_out, unsigned long long timeout_ticks,
                                                         float *__restrict__ snaps, int64_t n,
                                                         float lr, const double *__restrict__ normals, const int *__restrict__ n_normals_dev,
                                                         float *__restrict__ loss_log, int64_t loss_log_cap) {
    __shared__ int gave_up_s;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        bool gave_up = __hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // sticky: after one time-out the
        for (int q = 0; q < world && !gave_up; ++q) {                        //  rest of the segment does not wait per iteration again
            unsigned long long *f = flags + (size_t)parity * world + q;
            while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != iter_tag) {
                if (wall_clock64() - t0 > timeout_ticks) {                   // a peer is gone (or a host stalled beyond the bound)
                    atomicMax(timed_out, 1ull);
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        gave_up_s = gave_up;
    }
    __syncthreads();
    if (gave_up_s) return;                                                   // block-uniform: nothing of a partial sum is applied
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                            // system scope, every thread: the peers' slot stores
    float *base = slots + (size_t)parity * world * (size_t)msg_floats;
    auto summed = [&](int64_t i) {
        float s = __builtin_nontemporal_load(base + i);
        for (int q = 1; q < world; ++q) s += __builtin_nontemporal_load(base + (size_t)q * msg_floats + i);   // rank order: the same float sum on every rank
        return s;
    };
    float tail[MSG_EXTRA];
#pragma unroll
    for (int t = 0; t < MSG_EXTRA; ++t) tail[t] = summed(2 * n + t);
    const double total = loss_from_limbs(tail);
    const StepDecision d = step_decide(&states[cur], total);
    const int64_t idx = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const bool special = (idx == 0 || idx == 1 || idx == n || idx == n + 1);   // owned by the leader (jitter)
    if (idx < 2 * n && !special && !d.halted) step_element(d, idx, summed(idx), Y, snaps, n, lr);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float gsp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int c = 0; c < 2; ++c)
            for (int p = 0; p < 2 && p < n; ++p) gsp[2 * c + p] = summed((int64_t)c * n + p);
        step_leader(states, cur, d, Y, gsp, snaps, n, lr, normals, n_normals_dev, loss_log, loss_log_cap);
    }
}

// symmetric FAST kernel, single GPU: the sum of the row / column partials (sym_reduce_kernel's order: 8 lanes per element, lane l
// adds partials l, l + 8, ..., fixed butterfly) fused with apply -- the gradient never goes to memory.  The first four 8-lane
// groups of block 0 take x0, x1, y0, y1 and hand them to the leader through LDS; the other groups take the remaining 2N - 4
// coordinates in order.
__global__ __launch_bounds__(BLK) void sym_apply_kernel(LoopState *__restrict__ states, int cur, float *__restrict__ Y,
                                                        const float *__restrict__ rowpart, const float *__restrict__ colpart,
                                                        int64_t n_lblocks, int64_t nJ, const double *__restrict__ loss_sum,
                                                        float *__restrict__ snaps, int64_t n, float lr,
                                                        const double *__restrict__ normals, const int *__restrict__ n_normals_dev,
                                                        float *__restrict__ loss_log, int64_t loss_log_cap) {
    constexpr int SPLIT = 8;
    __shared__ float gsp_s[4];
    const StepDecision d = step_decide(&states[cur], loss_sum[0]);
    const int64_t grp = ((int64_t)blockIdx.x * BLK + threadIdx.x) / SPLIT;
    const int l = threadIdx.x & (SPLIT - 1);
    // group -> coordinate index: 0..3 -> x0, x1, y0, y1; 4.. -> the others in order
    int64_t idx;
    if (grp < 4) idx = (grp >> 1) * n + (grp & 1);
    else idx = (grp - 4 < n - 2) ? grp - 4 + 2 : grp - 4 + 4;
    const bool live = idx < 2 * n && (grp >= 4 || (grp & 1) < n);
    float g = 0.0f;
    if (live && !d.halted) {
        const int c = (int)(idx / n);
        const int64_t i = idx % n;
        const int64_t Ii = i / SY_R, Ji = i / SY_C;
        for (int64_t J = l; J < nJ; J += SPLIT)
            if (sy_tile_live(Ii, J)) g += rowpart[(J * 2 + c) * n + i];
        for (int64_t b = l; b < n_lblocks; b += SPLIT)
            if (sy_tile_live(b, Ji)) g += colpart[(b * 2 + c) * n + i];
    }
    g += __shfl_xor(g, 1);
    g += __shfl_xor(g, 2);
    g += __shfl_xor(g, 4);
    if (blockIdx.x == 0) {   // block-uniform
        if (grp < 4 && l == 0) gsp_s[grp] = g;
        __syncthreads();
    }
    if (live && !d.halted && grp >= 4 && l == 0) step_element(d, idx, g, Y, snaps, n, lr);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float gsp[4] = {gsp_s[0], gsp_s[1], gsp_s[2], gsp_s[3]};
        step_leader(states, cur, d, Y, gsp, snaps, n, lr, normals, n_normals_dev, loss_log, loss_log_cap);
    }
}

}  // namespace

namespace {
void drop_graph(kmap_embed *e) {   // kernel arguments changed: the captured iterations are stale
    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    e->gexec = nullptr;
}
}  // namespace

extern "C" {

// ---- session ---------------------------------------------------------------------------------
static int embed_create_impl(kmap_embed **out, int64_t n, int64_t row0, int64_t nrows, int n_best, float learning_rate,
                             int mode, int world, int rank);

int kmap_embed_create(kmap_embed **out, int64_t n, int64_t row0, int64_t nrows, int n_best, float learning_rate,
                      int mode) {
    return embed_create_impl(out, n, row0, nrows, n_best, learning_rate, mode, 1, 0);
}

int64_t kmap_embed_cyclic_blocks(int64_t n, int world, int rank) {
    if (n <= 0 || world <= 0 || rank < 0 || rank >= world) return 0;
    const int64_t nI = (n + SY_R - 1) / SY_R;
    return nI > rank ? (nI - rank + world - 1) / world : 0;
}

int kmap_embed_create_cyclic(kmap_embed **out, int64_t n, int world, int rank, int n_best, float learning_rate) {
    KMAP_REQUIRE(world >= 1 && rank >= 0 && rank < world, "embed_create_cyclic: bad world / rank");
    return embed_create_impl(out, n, 0, n, n_best, learning_rate, KMAP_EMBED_FAST, world, rank);
}

static int embed_create_impl(kmap_embed **out, int64_t n, int64_t row0, int64_t nrows, int n_best, float learning_rate,
                             int mode, int world, int rank) {
    KMAP_REQUIRE(out, "embed_create: null");
    KMAP_REQUIRE(n > 0 && row0 >= 0 && nrows >= 0 && row0 + nrows <= n, "embed_create: bad row range");
    KMAP_REQUIRE(n < ((int64_t)1 << 31) - 64, "embed_create: n too large");
    KMAP_REQUIRE(n_best > 0 && n_best <= MAX_BEST, "embed_create: n_best must be in [1,%d]", MAX_BEST);
    KMAP_REQUIRE(mode == KMAP_EMBED_FAST || mode == KMAP_EMBED_SEQ, "embed_create: unknown mode %d", mode);
    kmap_embed *e = new kmap_embed();
    e->n = n; e->row0 = row0; e->nrows = nrows; e->n_best = n_best; e->lr = learning_rate; e->mode = mode;
    {   // symmetric FAST kernel: single-GPU all-rows sessions (KMAP_EMBED_SYM=0 runs the row-wise kernel instead: the kernel
        // row-sharded multi-GPU sessions use, so that it can be compared on one GPU)
        const char *env = getenv("KMAP_EMBED_SYM");
        e->sym = (mode == KMAP_EMBED_FAST) && row0 == 0 && nrows == n && n >= 16384 && !(env && env[0] == '0');   // below ~16k the tall tiles leave CUs idle
        if (world > 1) e->sym = true;                       // the cyclic creator always runs the symmetric kernel
        e->symI = (n + SY_R - 1) / SY_R;
        e->symJ = (n + SY_C - 1) / SY_C;
        e->world = world;
        e->rank = rank;
        e->n_lblocks = kmap_embed_cyclic_blocks(n, world, rank);
    }
    if (mode == KMAP_EMBED_SEQ) kmap_embed_seq_split(e);
    e->n_part = kmap_embed_force_blocks(e) > 0 ? kmap_embed_force_blocks(e) : 1;
    if (mode == KMAP_EMBED_SEQ && kmap_embed_seq_blocks_max(e) > e->n_part) e->n_part = kmap_embed_seq_blocks_max(e);
    hipError_t err = hipSuccess;
    auto A = [&](void **p, size_t b) { if (err == hipSuccess) err = hipMalloc(p, b ? b : 16); };
    A((void **)&e->Y, ((size_t)2 * n + 64) * 4);            // + 64 floats: embed_seq.hip's coordinate loads of the batch that reaches past column n - 1 run up to 31 floats past Yy
    A((void **)&e->G, (size_t)2 * n * 4);
    A((void **)&e->snaps, (size_t)n_best * 2 * n * 4);
    A((void **)&e->loss_log, (size_t)e->loss_log_cap * 4);
    A((void **)&e->loss_part, (size_t)e->n_part * 8);
    A((void **)&e->loss_sum, 8);
    A((void **)&e->states, 2 * sizeof(LoopState));
    A((void **)&e->lut_dev, F_LUT_LDS * 4);
    A((void **)&e->n_normals_dev, 16);
    if (e->sym) {
        A((void **)&e->rowpart, (size_t)e->symJ * 2 * n * 4);
        A((void **)&e->colpart, (size_t)(e->n_lblocks ? e->n_lblocks : 1) * 2 * n * 4);
    }
    if (err != hipSuccess) {
        kmap_set_error("embed_create: %s", hipGetErrorString(err));
        kmap_embed_destroy(e);
        return KMAP_E_NOMEM;
    }
    LoopState s0;
    memset(&s0, 0, sizeof s0);
    s0.n_best = n_best;
    s0.prev_loss = INFINITY;
    s0.last_loss = INFINITY;
    for (int b = 0; b < MAX_BEST; ++b) {
        s0.best_loss[b] = INFINITY;
        s0.best_slot[b] = b;
    }
    s0.worst_loss = INFINITY;
    s0.worst_slot = n_best - 1;
    KMAP_CHECK_HIP(hipMemcpy(&e->states[0], &s0, sizeof s0, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(&e->states[1], &s0, sizeof s0, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemset(e->G, 0, (size_t)2 * n * 4));
    KMAP_CHECK_HIP(hipMemset(e->Y + 2 * n, 0, 64 * 4));
    KMAP_CHECK_HIP(hipMemset(e->loss_part, 0, (size_t)e->n_part * 8));
    KMAP_CHECK_HIP(hipMemset(e->n_normals_dev, 0, 16));
    *out = e;
    return KMAP_OK;
}

int kmap_embed_destroy(kmap_embed *e) {
    if (!e) return KMAP_OK;
    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    if (e->gstream) (void)hipStreamDestroy(e->gstream);
    void *ptrs[] = {e->Y, e->G, e->snaps, e->loss_log, e->loss_part, e->loss_sum, e->states, e->lut_dev, e->normals,
                    e->rowpart, e->colpart, e->n_normals_dev};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete e;
    return KMAP_OK;
}

int kmap_embed_set_prob_f32(kmap_embed *e, const float *p_rows_dev, int64_t ld) {
    KMAP_REQUIRE(e && p_rows_dev && ld >= e->n, "embed_set_prob_f32: bad arguments");
    if (e->sym) {   // the symmetric tile kernel reads u16 sums through the LUT; an f32 matrix goes through the row-wise kernel
        KMAP_REQUIRE(e->world == 1, "embed_set_prob_f32: the cyclic multi-GPU layout needs the neighbour-sum + LUT source");
        e->sym = false;
        const int n_part = kmap_embed_force_blocks(e) > 0 ? kmap_embed_force_blocks(e) : 1;
        if (n_part > e->n_part) {
            double *lp = nullptr;
            KMAP_CHECK_HIP(hipMalloc((void **)&lp, (size_t)n_part * 8));
            KMAP_CHECK_HIP(hipMemset(lp, 0, (size_t)n_part * 8));
            (void)hipFree(e->loss_part);
            e->loss_part = lp;
        }
        e->n_part = n_part;
    }
    e->src = ProbSrc{p_rows_dev, nullptr, nullptr, ld, 0, nullptr, e->nrows};
    e->have_prob = true;
    drop_graph(e);
    return KMAP_OK;
}

int kmap_embed_set_prob_lut(kmap_embed *e, const uint16_t *sums_rows_dev, int64_t ld, const float *lut, int lut_len) {
    KMAP_REQUIRE(e && sums_rows_dev && lut && ld >= e->n, "embed_set_prob_lut: bad arguments");
    KMAP_REQUIRE(lut_len > 0 && lut_len <= F_LUT_LDS, "embed_set_prob_lut: lut_len=%d exceeds %d", lut_len, F_LUT_LDS);
    KMAP_CHECK_HIP(hipMemcpy(e->lut_dev, lut, (size_t)lut_len * 4, hipMemcpyHostToDevice));
    e->src = ProbSrc{nullptr, sums_rows_dev, e->lut_dev, ld, lut_len, nullptr, e->nrows};
    e->have_prob = true;
    drop_graph(e);
    return KMAP_OK;
}

int kmap_embed_set_row_map(kmap_embed *e, const int32_t *rowmap_dev, int64_t src_rows) {
    KMAP_REQUIRE(e && e->have_prob && e->src.ps, "embed_set_row_map: set the neighbour-sum + LUT source first");
    if (!rowmap_dev) {
        e->src.rowmap = nullptr;
        e->src.src_rows = e->nrows;
        drop_graph(e);
        return KMAP_OK;
    }
    KMAP_REQUIRE(e->mode == KMAP_EMBED_SEQ && !e->sym, "embed_set_row_map: SEQ sessions only (the FAST kernels read one stored row per session row)");
    KMAP_REQUIRE(src_rows >= 1 && src_rows <= e->nrows, "embed_set_row_map: src_rows must be in [1, rows of the session]");
    std::vector<int32_t> m((size_t)e->nrows);
    KMAP_CHECK_HIP(hipMemcpy(m.data(), rowmap_dev, (size_t)e->nrows * 4, hipMemcpyDeviceToHost));
    bool ok = e->nrows > 0 && m[0] == 0 && (int64_t)m[(size_t)e->nrows - 1] == src_rows - 1;
    for (int64_t r = 1; ok && r < e->nrows; ++r) ok = m[(size_t)r] == m[(size_t)r - 1] || m[(size_t)r] == m[(size_t)r - 1] + 1;
    KMAP_REQUIRE(ok, "embed_set_row_map: the map must start at 0, end at src_rows - 1 and step by 0 or 1");
    e->src.rowmap = rowmap_dev;
    e->src.src_rows = src_rows;
    drop_graph(e);
    return KMAP_OK;
}

int kmap_embed_set_coords(kmap_embed *e, const float *coords_2xn, const float *placeholders) {
    KMAP_REQUIRE(e && coords_2xn, "embed_set_coords: null");
    KMAP_CHECK_HIP(hipMemcpy(e->Y, coords_2xn, (size_t)2 * e->n * 4, hipMemcpyHostToDevice));
    if (placeholders)
        KMAP_CHECK_HIP(hipMemcpy(e->snaps, placeholders, (size_t)e->n_best * 2 * e->n * 4, hipMemcpyHostToDevice));
    else
        KMAP_CHECK_HIP(hipMemset(e->snaps, 0, (size_t)e->n_best * 2 * e->n * 4));
    e->have_coords = true;
    return KMAP_OK;
}

int kmap_embed_set_jitter(kmap_embed *e, const double *normals, int n_normals) {
    KMAP_REQUIRE(e && n_normals >= 0 && (n_normals == 0 || normals), "embed_set_jitter: bad arguments");
    KMAP_CHECK_HIP(hipDeviceSynchronize());
    if (n_normals > e->normals_cap) {           // grow geometrically: the pointer (a kernel argument) rarely changes
        int cap = e->normals_cap ? e->normals_cap : 8192;
        while (cap < n_normals) cap *= 2;
        if (e->normals) KMAP_CHECK_HIP(hipFree(e->normals));
        e->normals = nullptr;
        e->normals_cap = 0;
        KMAP_CHECK_HIP(hipMalloc((void **)&e->normals, (size_t)cap * 8));
        e->normals_cap = cap;
        drop_graph(e);
    }
    if (n_normals) KMAP_CHECK_HIP(hipMemcpy(e->normals, normals, (size_t)n_normals * 8, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(e->n_normals_dev, &n_normals, 4, hipMemcpyHostToDevice));
    e->n_normals = n_normals;
    return KMAP_OK;
}

namespace {
// the force kernel of the session (+ the row / column partial sum into G for the symmetric kernel when `reduce_sym`);
// loss partials go to e->loss_part
int launch_forces(kmap_embed *e, float *G, bool reduce_sym, hipStream_t st) {
    if (e->sym) return kmap_embed_launch_sym(e, G, reduce_sym, st);
    if (e->mode == KMAP_EMBED_SEQ) return kmap_embed_launch_seq(e, G, st);
    return kmap_embed_launch_fast_rows(e, G, st);
}

// one single-GPU iteration with the fused tails: forces -> apply<loss reduction fused> (2 launches), or for the symmetric
// kernels forces -> loss reduction -> partial sums + apply (3 launches instead of 4; the gradient never goes to memory)
int launch_iteration(kmap_embed *e, hipStream_t st) {
    KMAP_TRY(launch_forces(e, e->G, false, st));
    const int nblk = kmap_embed_force_blocks(e);
    if (e->sym) {
        reduce_loss_kernel<<<1, RL_TPB, 0, st>>>(e->loss_part, nblk, e->loss_sum);
        sym_apply_kernel<<<(unsigned)(((2 * e->n + 4) * 8 + BLK - 1) / BLK), BLK, 0, st>>>(
            e->states, e->cur, e->Y, e->rowpart, e->colpart, e->n_lblocks, e->symJ, e->loss_sum, e->snaps, e->n, e->lr, e->normals,
            e->n_normals_dev, e->loss_log, e->loss_log_cap);
    } else {
        apply_kernel<true><<<(unsigned)((2 * e->n + BLK - 1) / BLK), BLK, 0, st>>>(e->states, e->cur, e->Y, e->G, e->loss_part, nblk, e->snaps,
                                                                                  e->n, e->lr, e->normals, e->n_normals_dev,
                                                                                  e->loss_log, e->loss_log_cap);
    }
    KMAP_CHECK_HIP(hipGetLastError());
    e->cur ^= 1;
    return KMAP_OK;
}

// capture two iterations (parities cur, cur ^ 1) into a graph; any failure switches graph replay off for the session
bool ensure_graph(kmap_embed *e) {
    if (e->graph_failed) return false;
    if (e->gexec) return true;
    // opt-in (KMAP_EMBED_GRAPH=1): measured on ROCm 7.2 / MI355X, replaying the captured pair of iterations is SLOWER than
    // launching the same 2-3 kernels directly (N = 50 k FAST 0.958 vs 0.882 ms / iteration, N = 5 k SEQ 0.175 vs 0.163, FAST 0.047 vs
    // 0.041): the launches are already queued ahead of the GPU, and the graph adds per-node dispatch cost
    static const bool on = [] { const char *v = getenv("KMAP_EMBED_GRAPH"); return v && v[0] == '1'; }();
    if (!on) { e->graph_failed = true; return false; }
    if (!e->gstream && hipStreamCreateWithFlags(&e->gstream, hipStreamNonBlocking) != hipSuccess) { e->graph_failed = true; return false; }
    hipGraph_t graph = nullptr;
    const int cur0 = e->cur;
    bool ok = hipStreamBeginCapture(e->gstream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
        ok = launch_iteration(e, e->gstream) == KMAP_OK && launch_iteration(e, e->gstream) == KMAP_OK;
        ok = (hipStreamEndCapture(e->gstream, &graph) == hipSuccess) && ok && graph;
    }
    e->cur = cur0;                                   // captured, not executed
    if (ok) ok = hipGraphInstantiate(&e->gexec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
    if (!ok) {
        (void)hipGetLastError();
        e->gexec = nullptr;
        e->graph_failed = true;
        return false;
    }
    e->graph_cur = cur0;
    return true;
}
}  // namespace

int kmap_embed_forces(kmap_embed *e, float *grad_dev_2xn, double *loss_dev, void *stream) {
    KMAP_REQUIRE(e && e->have_prob && e->have_coords, "embed_forces: probabilities/coordinates not set");
    hipStream_t st = as_stream(stream);
    float *G = grad_dev_2xn ? grad_dev_2xn : e->G;
    double *L = loss_dev ? loss_dev : e->loss_sum;
    const int nblk = kmap_embed_force_blocks(e);
    if (nblk == 0) {
        KMAP_CHECK_HIP(hipMemsetAsync(L, 0, 8, st));
        return KMAP_OK;
    }
    KMAP_TRY(launch_forces(e, G, true, st));
    reduce_loss_kernel<<<1, RL_TPB, 0, st>>>(e->loss_part, nblk, L);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

int kmap_embed_apply(kmap_embed *e, const float *grad_dev_2xn, const double *loss_dev, void *stream) {
    KMAP_REQUIRE(e && e->have_coords, "embed_apply: coordinates not set");
    hipStream_t st = as_stream(stream);
    const float *G = grad_dev_2xn ? grad_dev_2xn : e->G;
    const double *L = loss_dev ? loss_dev : e->loss_sum;
    const unsigned grid = (unsigned)((2 * e->n + BLK - 1) / BLK);
    apply_kernel<false><<<grid, BLK, 0, st>>>(e->states, e->cur, e->Y, G, L, 0, e->snaps, e->n, e->lr, e->normals, e->n_normals_dev,
                                              e->loss_log, e->loss_log_cap);
    KMAP_CHECK_HIP(hipGetLastError());
    e->cur ^= 1;
    return KMAP_OK;
}

int64_t kmap_embed_msg_floats(int64_t n) { return n > 0 ? 2 * n + MSG_EXTRA : 0; }

int kmap_embed_forces_msg(kmap_embed *e, float *msg_dev, void *stream) {
    KMAP_REQUIRE(e && e->have_prob && e->have_coords, "embed_forces_msg: probabilities/coordinates not set");
    KMAP_REQUIRE(msg_dev, "embed_forces_msg: null message buffer");
    hipStream_t st = as_stream(stream);
    const int nblk = kmap_embed_force_blocks(e);
    if (nblk > 0) KMAP_TRY(launch_forces(e, msg_dev, true, st));
    reduce_loss_limbs_kernel<<<1, RL_TPB, 0, st>>>(e->loss_part, nblk, msg_dev + 2 * e->n);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

int kmap_embed_apply_msg(kmap_embed *e, float *msg_dev, void *stream) {
    KMAP_REQUIRE(e && e->have_coords, "embed_apply_msg: coordinates not set");
    KMAP_REQUIRE(msg_dev, "embed_apply_msg: null message buffer");
    hipStream_t st = as_stream(stream);
    const unsigned grid = (unsigned)((2 * e->n + BLK - 1) / BLK);
    const bool clear = !e->sym && !(e->row0 == 0 && e->nrows == e->n);      // the force kernel leaves other ranks' rows alone
    if (clear)
        apply_msg_kernel<true><<<grid, BLK, 0, st>>>(e->states, e->cur, e->Y, msg_dev, e->snaps, e->n, e->lr, e->normals, e->n_normals_dev,
                                                     e->loss_log, e->loss_log_cap);
    else
        apply_msg_kernel<false><<<grid, BLK, 0, st>>>(e->states, e->cur, e->Y, msg_dev, e->snaps, e->n, e->lr, e->normals, e->n_normals_dev,
                                                      e->loss_log, e->loss_log_cap);
    KMAP_CHECK_HIP(hipGetLastError());
    e->cur ^= 1;
    return KMAP_OK;
}

int kmap_embed_step_peer(kmap_embed *e, kmap_peer *p, int n_iter, void *stream) {
    KMAP_REQUIRE(e && p && e->have_prob && e->have_coords, "embed_step_peer: session / exchange not ready");
    KMAP_REQUIRE(p->msg_floats == kmap_embed_msg_floats(e->n), "embed_step_peer: the exchange was created for another message length");
    for (int q = 0; q < p->world; ++q) KMAP_REQUIRE(p->peer_area[q], "embed_step_peer: rank %d's receive area is not connected", q);
    hipStream_t st = as_stream(stream);
    PeerTab tab;
    memset(&tab, 0, sizeof tab);
    for (int q = 0; q < p->world; ++q) {
        tab.slots[q] = (float *)p->peer_area[q];
        tab.flags[q] = (unsigned long long *)((char *)p->peer_area[q] + kmap_peer_slots_bytes(p));
    }
    const unsigned pgrid = (unsigned)((p->slot_floats / 4 + BLK - 1) / BLK);
    const unsigned agrid = (unsigned)((2 * e->n + BLK - 1) / BLK);
    for (int it = 0; it < n_iter; ++it) {
        const int parity = (int)(p->iter & 1);
        const unsigned long long tag = p->iter + 1;
        KMAP_TRY(kmap_embed_forces_msg(e, p->msg_local, stream));
        peer_push_kernel<<<pgrid, BLK, 0, st>>>(tab, p->msg_local, p->slot_floats, p->world, p->rank, parity, tag, p->done);
        apply_peer_kernel<<<agrid, BLK, 0, st>>>(e->states, e->cur, e->Y, (float *)p->area,
                                                 (unsigned long long *)((char *)p->area + kmap_peer_slots_bytes(p)), p->world, parity, tag,
                                                 p->slot_floats, p->done + 1, p->timeout_ticks, e->snaps, e->n, e->lr, e->normals, e->n_normals_dev,
                                                 e->loss_log, e->loss_log_cap);
        KMAP_CHECK_HIP(hipGetLastError());
        e->cur ^= 1;
        p->iter += 1;
    }
    return KMAP_OK;
}

int kmap_embed_step(kmap_embed *e, int n_iter, void *stream) {
    KMAP_REQUIRE(e && e->row0 == 0 && e->nrows == e->n && e->world == 1, "embed_step: single-GPU convenience needs all rows local");
    KMAP_REQUIRE(e->have_prob && e->have_coords, "embed_step: probabilities/coordinates not set");
    hipStream_t st = as_stream(stream);
    int it = 0;
    if (n_iter >= 4 && ensure_graph(e)) {
        if (e->cur != e->graph_cur && it < n_iter) {   // realign the parity the graph was captured at
            KMAP_TRY(launch_iteration(e, st));
            ++it;
        }
        for (; it + 2 <= n_iter; it += 2) {
            if (hipGraphLaunch(e->gexec, st) != hipSuccess) {   // e.g. a stream the runtime cannot launch graphs into
                (void)hipGetLastError();
                drop_graph(e);
                e->graph_failed = true;
                break;
            }
        }
    }
    for (; it < n_iter; ++it) KMAP_TRY(launch_iteration(e, st));
    return KMAP_OK;
}

int kmap_embed_state(kmap_embed *e, int64_t *iters, int *stopped, float *last_loss, float *best_loss, int *jitter_used,
                     void *stream) {
    KMAP_REQUIRE(e, "embed_state: null");
    KMAP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    LoopState s;
    KMAP_CHECK_HIP(hipMemcpy(&s, &e->states[e->cur], sizeof s, hipMemcpyDeviceToHost));
    if (iters) *iters = s.iters;
    if (stopped) *stopped = s.stopped;
    if (last_loss) *last_loss = s.last_loss;
    if (best_loss) *best_loss = s.best_loss[0];
    if (jitter_used) *jitter_used = s.jitter_used;
    return KMAP_OK;
}

int kmap_embed_get_coords(kmap_embed *e, float *coords_2xn, void *stream) {
    KMAP_REQUIRE(e && coords_2xn, "embed_get_coords: null");
    KMAP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    KMAP_CHECK_HIP(hipMemcpy(coords_2xn, e->Y, (size_t)2 * e->n * 4, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

int kmap_embed_get_best(kmap_embed *e, float *coords_2xn, void *stream) {
    KMAP_REQUIRE(e && coords_2xn, "embed_get_best: null");
    KMAP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    LoopState s;
    KMAP_CHECK_HIP(hipMemcpy(&s, &e->states[e->cur], sizeof s, hipMemcpyDeviceToHost));
    KMAP_CHECK_HIP(hipMemcpy(coords_2xn, e->snaps + (size_t)s.best_slot[0] * 2 * e->n, (size_t)2 * e->n * 4,
                             hipMemcpyDeviceToHost));   // best_res_list[0][1] (visualization.py:325)
    return KMAP_OK;
}

int kmap_embed_get_losses(kmap_embed *e, float *losses, int64_t max_n, int64_t *n_out, void *stream) {
    KMAP_REQUIRE(e && n_out, "embed_get_losses: null");
    KMAP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
    LoopState s;
    KMAP_CHECK_HIP(hipMemcpy(&s, &e->states[e->cur], sizeof s, hipMemcpyDeviceToHost));
    int64_t m = s.iters < e->loss_log_cap ? s.iters : e->loss_log_cap;
    if (m > max_n) m = max_n;
    if (m > 0 && losses) KMAP_CHECK_HIP(hipMemcpy(losses, e->loss_log, (size_t)m * 4, hipMemcpyDeviceToHost));
    *n_out = m;
    return KMAP_OK;
}

void *kmap_embed_coords_dev(kmap_embed *e) { return e ? (void *)e->Y : nullptr; }

// ---- drop-in L3 float operators (host pointers, blocking) -----------------------------------------
}  // extern "C"

namespace {
__global__ __launch_bounds__(BLK) void ld_prob_kernel(const float *__restrict__ Y, int64_t n, float *__restrict__ Q) {
    const int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (t >= n * n) return;
    const int64_t i = t / n, j = t % n;
    // np.ones then the kernel fills i != j; clip applies to the diagonal's 1.0 too (visualization.py:251-255)
    Q[t] = (i == j) ? 0.999f : q_of(Y[i] - Y[j], Y[n + i] - Y[n + j]);
}
__global__ __launch_bounds__(BLK) void ce_rows_kernel(const float *__restrict__ P, const float *__restrict__ Q, int64_t n,
                                                      double *__restrict__ part) {
    __shared__ double sh[BLK];
    double s = 0.0;
    const int64_t i = blockIdx.x;
    for (int64_t j = i + 1 + threadIdx.x; j < n; j += BLK) {
        float q = Q[i * n + j];
        const float eps = 1e-10f;
        q = (q < eps) ? eps : ((q > 1.0f - eps) ? 1.0f - eps : q);   // taichi_core.py:283-288
        s += (double)ce_of<true>(P[i * n + j], q);
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = BLK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[i] = sh[0];
}
__global__ __launch_bounds__(KMAP_WAVE) void grad_rows_kernel(const float *__restrict__ P, const float *__restrict__ Q,
                                                              const float *__restrict__ Y, int64_t n,
                                                              float *__restrict__ G) {
    const int64_t i = (int64_t)blockIdx.x * KMAP_WAVE + threadIdx.x;
    if (i >= n) return;
    const float xi = Y[i], yi = Y[n + i];
    float gx = 0.0f, gy = 0.0f;
    for (int64_t j = 0; j < n; ++j) {
        if (j == i) continue;
        const float t = t_of(P[i * n + j], Q[i * n + j]);
        gx = gx + t * (xi - Y[j]);
        gy = gy + t * (yi - Y[n + j]);
    }
    G[i] = 4.0f * gx;
    G[n + i] = 4.0f * gy;
}
}  // namespace

extern "C" {

int kmap_ld_prob_mat_f32(const float *ld_2xn, int64_t n, float *q_out_nxn) {
    KMAP_REQUIRE(n >= 0, "ld_prob_mat: n<0");
    if (n == 0) return KMAP_OK;
    KMAP_REQUIRE(ld_2xn && q_out_nxn, "ld_prob_mat: null pointer");
    DevBuf dy, dq;
    KMAP_TRY(dy.alloc((size_t)2 * n * 4));
    KMAP_TRY(dq.alloc((size_t)n * n * 4));
    KMAP_CHECK_HIP(hipMemcpy(dy.p, ld_2xn, (size_t)2 * n * 4, hipMemcpyHostToDevice));
    ld_prob_kernel<<<(unsigned)((n * n + BLK - 1) / BLK), BLK>>>(dy.as<float>(), n, dq.as<float>());
    KMAP_CHECK_HIP(hipGetLastError());
    KMAP_CHECK_HIP(hipMemcpy(q_out_nxn, dq.p, (size_t)n * n * 4, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

int kmap_cross_entropy_f32(const float *p_nxn, const float *q_nxn, int64_t n, float *loss_out) {
    KMAP_REQUIRE(n >= 0 && loss_out, "cross_entropy: bad arguments");
    *loss_out = 0.0f;
    if (n == 0) return KMAP_OK;
    KMAP_REQUIRE(p_nxn && q_nxn, "cross_entropy: null pointer");
    DevBuf dp, dq, dpart, dsum;
    KMAP_TRY(dp.alloc((size_t)n * n * 4));
    KMAP_TRY(dq.alloc((size_t)n * n * 4));
    KMAP_TRY(dpart.alloc((size_t)n * 8));
    KMAP_TRY(dsum.alloc(8));
    KMAP_CHECK_HIP(hipMemcpy(dp.p, p_nxn, (size_t)n * n * 4, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(dq.p, q_nxn, (size_t)n * n * 4, hipMemcpyHostToDevice));
    ce_rows_kernel<<<(unsigned)n, BLK>>>(dp.as<float>(), dq.as<float>(), n, dpart.as<double>());
    reduce_loss_kernel<<<1, RL_TPB>>>(dpart.as<double>(), (int)n, dsum.as<double>());
    KMAP_CHECK_HIP(hipGetLastError());
    double s = 0.0;
    KMAP_CHECK_HIP(hipMemcpy(&s, dsum.p, 8, hipMemcpyDeviceToHost));
    *loss_out = (float)(2.0 * s);
    return KMAP_OK;
}

int kmap_gradient_loss_f32(const float *p_nxn, const float *q_nxn, const float *ld_2xn, int64_t n, float *grad_out_2xn) {
    KMAP_REQUIRE(n >= 0, "gradient_loss: n<0");
    if (n == 0) return KMAP_OK;
    KMAP_REQUIRE(p_nxn && q_nxn && ld_2xn && grad_out_2xn, "gradient_loss: null pointer");
    DevBuf dp, dq, dy, dg;
    KMAP_TRY(dp.alloc((size_t)n * n * 4));
    KMAP_TRY(dq.alloc((size_t)n * n * 4));
    KMAP_TRY(dy.alloc((size_t)2 * n * 4));
    KMAP_TRY(dg.alloc((size_t)2 * n * 4));
    KMAP_CHECK_HIP(hipMemcpy(dp.p, p_nxn, (size_t)n * n * 4, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(dq.p, q_nxn, (size_t)n * n * 4, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy(dy.p, ld_2xn, (size_t)2 * n * 4, hipMemcpyHostToDevice));
    grad_rows_kernel<<<(unsigned)((n + KMAP_WAVE - 1) / KMAP_WAVE), KMAP_WAVE>>>(dp.as<float>(), dq.as<float>(),
                                                                                 dy.as<float>(), n, dg.as<float>());
    KMAP_CHECK_HIP(hipGetLastError());
    KMAP_CHECK_HIP(hipMemcpy(grad_out_2xn, dg.p, (size_t)2 * n * 4, hipMemcpyDeviceToHost));
    return KMAP_OK;
}

}  // extern "C"
