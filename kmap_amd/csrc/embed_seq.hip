// embed_seq.hip -- SEQ forces of the embedding iteration: the reference's arithmetic and summation order, bit for bit
// (taichi_core.py:305-326 with T from visualization.py:131-145: IEEE f32, j ascending, j != i, no FMA).
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "embed_internal.h"
#include "seq_div.h"

namespace {
// =================================================================================================
// SEQ forces: the reference's summation order (taichi_core.py:305-326: ret_val += diff[i,j] * (y[k,i] - y[k,j]),
// j ascending, j != i), IEEE f32, no FMA.  A row is owned by the 4 lanes of a quad: for a group of 4 columns
// each sub-lane evaluates one term (q, t, t*dx, t*dy -- independent work), then ALL lanes of the row add the
// 4 terms in column order (DPP quad broadcasts), so they carry identical accumulators and the sum order is exactly
// j = 0, 1, 2, ...  The loss needs no order (f64 accumulation of f32 terms), each sub-lane keeps its own.
// =================================================================================================
// acc + (value of `v` in lane S of the caller's quad) as ONE v_add_f32 with a DPP quad_perm source.
// hipcc does not fold __builtin_amdgcn_update_dpp into the add, so the instruction is written out; the DPP operand
// `v` is always produced more than 2 VALU instructions earlier (the DPP read-after-VALU-write hazard, cdna_hip 5.7).
template <int S>
__device__ __forceinline__ float add_quad_bcast(float acc, float v) {
    float r;
    if constexpr (S == 0) asm("v_add_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(acc));
    if constexpr (S == 1) asm("v_add_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(acc));
    if constexpr (S == 2) asm("v_add_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(acc));
    if constexpr (S == 3) asm("v_add_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(acc));
    return r;
}
// the eight columns of quad lane S, x and y chains alternating, as ONE asm statement: between single-instruction asm statements the
// compiler puts an s_nop behind every other pair of adds (24 per 64-add batch; it guards the accumulator, written two instructions
// earlier, as if it were the DPP source -- only src0 goes through the DPP network, and the terms are written long before)
template <int S>
__device__ __forceinline__ void add_quad_block(float &gx, float &gy, const float (&tx)[8], const float (&ty)[8]) {
#define KMAP_QP8(SS, P)                                                                                                            \
    if constexpr (S == SS)                                                                                                        \
        asm("v_add_f32_dpp %0, %2, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %10, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %3, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %11, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %4, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %12, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %5, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %13, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %6, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %14, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %7, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %15, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %8, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %16, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %9, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %17, %1 quad_perm:" P " row_mask:0xf bank_mask:0xf"        \
            : "+v"(gx), "+v"(gy)                                                                                                  \
            : "v"(tx[0]), "v"(tx[1]), "v"(tx[2]), "v"(tx[3]), "v"(tx[4]), "v"(tx[5]), "v"(tx[6]), "v"(tx[7]), "v"(ty[0]), "v"(ty[1]), "v"(ty[2]),  \
              "v"(ty[3]), "v"(ty[4]), "v"(ty[5]), "v"(ty[6]), "v"(ty[7]));
    KMAP_QP8(0, "[0,0,0,0]") KMAP_QP8(1, "[1,1,1,1]") KMAP_QP8(2, "[2,2,2,2]") KMAP_QP8(3, "[3,3,3,3]")
    KMAP_QP8(4, "[0,0,2,2]") KMAP_QP8(5, "[1,1,3,3]")        // two sub-lanes per row (pair form): lane 0 / 1 of each lane pair
#undef KMAP_QP8
}
// Ordered adds with the x chain and the y chain on DIFFERENT lanes of a row, in ONE accumulator register.  Merged operands:
// U[c] (even lanes: own t dx, odd lanes: the left neighbour's t dy) and W[c] (odd lanes: own t dx, even lanes: the right neighbour's
// t dy).  Every row sum gets the same operands in the same order as with one register per chain.
__device__ __forceinline__ void seq_merge_xy(float (&u)[8], float (&w)[8], const float (&tx)[8], const float (&ty)[8]) {
    const uint64_t even = 0x5555555555555555ull, odd = 0xaaaaaaaaaaaaaaaaull;
    // D = vcc ? src1 (own t dx) : dpp(src0 = the neighbour's t dy)
#define KMAP_MERGE8(OUT, MASK, P)                                                                                                         \
    asm volatile("s_mov_b64 vcc, %24\n\t"                                                                                                 \
                 "v_cndmask_b32_dpp %0, %16, %8, vcc quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"                                       \
                 "v_cndmask_b32_dpp %1, %17, %9, vcc quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"                                       \
                 "v_cndmask_b32_dpp %2, %18, %10, vcc quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"                                      \
                 "v_cndmask_b32_dpp %3, %19, %11, vcc quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"                                      \
                 "v_cndmask_b32_dpp %4, %20, %12, vcc quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"                                      \
                 "v_cndmask_b32_dpp %5, %21, %13, vcc quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"                                      \
                 "v_cndmask_b32_dpp %6, %22, %14, vcc quad_perm:" P " row_mask:0xf bank_mask:0xf\n\t"                                      \
                 "v_cndmask_b32_dpp %7, %23, %15, vcc quad_perm:" P " row_mask:0xf bank_mask:0xf"                                          \
                 : "=&v"(OUT[0]), "=&v"(OUT[1]), "=&v"(OUT[2]), "=&v"(OUT[3]), "=&v"(OUT[4]), "=&v"(OUT[5]), "=&v"(OUT[6]), "=&v"(OUT[7])   \
                 : "v"(tx[0]), "v"(tx[1]), "v"(tx[2]), "v"(tx[3]), "v"(tx[4]), "v"(tx[5]), "v"(tx[6]), "v"(tx[7]), "v"(ty[0]), "v"(ty[1]),    \
                   "v"(ty[2]), "v"(ty[3]), "v"(ty[4]), "v"(ty[5]), "v"(ty[6]), "v"(ty[7]), "s"(MASK)                                       \
                 : "vcc")
    KMAP_MERGE8(u, even, "[0,0,2,2]");
    KMAP_MERGE8(w, odd, "[1,1,3,3]");
#undef KMAP_MERGE8
}
// quad form (four lanes per row): the x chain on lanes 0 / 1 of the quad, the y chain on lanes 2 / 3 -- U[c] serves the owners 0 and 2,
// W[c] the owners 1 and 3: 16 merges + 32 adds instead of 64 adds
__device__ __forceinline__ void add_quad_split(float &g, const float (&tx)[8], const float (&ty)[8]) {
    float u[8], w[8];
    seq_merge_xy(u, w, tx, ty);
    // one statement per add, not volatile: the compiler may place the next batch's instructions between them (the chain waits 8.7
    // cycles per add; r05: no-far 1.95 -> 1.89 ms with the pair form's adds written this way)
#define KMAP_ADDQ(SRC, P) asm("v_add_f32_dpp %0, %1, %0 quad_perm:" P " row_mask:0xf bank_mask:0xf" : "+v"(g) : "v"(SRC))
#pragma unroll
    for (int c = 0; c < 8; ++c) KMAP_ADDQ(u[c], "[0,0,1,1]");
#pragma unroll
    for (int c = 0; c < 8; ++c) KMAP_ADDQ(w[c], "[1,1,0,0]");
#pragma unroll
    for (int c = 0; c < 8; ++c) KMAP_ADDQ(u[c], "[2,2,3,3]");
#pragma unroll
    for (int c = 0; c < 8; ++c) KMAP_ADDQ(w[c], "[3,3,2,2]");
#undef KMAP_ADDQ
}
// a wave-uniform 64-bit value, moved to scalar registers (the compiler cannot see that `threadIdx.x >> 6` is uniform)
__device__ __forceinline__ int64_t seq_uniform(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
constexpr int SQ_CPL = 8;                       // consecutive columns per lane per batch (one 16-byte load of u16 sums)

struct SeqBatch {                                // raw operands of one batch of one lane, as column pairs (c, c + 1)
    uint32_t w[4];                               // 8 u16 sums (LUT source) ...
    f32x2 pf[SQ_CPL / 2];                        // ... or 8 f32 probabilities
    f32x2 x[SQ_CPL / 2], y[SQ_CPL / 2];
};
// One batch of one lane.  The SAME number of load instructions on every path through a step -- no branch around or inside this
// function: the waitcnt pass counts outstanding loads, loads return in order, and at a control-flow join it must assume the
// smaller count was issued after the loads it waits for.  With a conditional prefetch (or a vector / scalar choice per batch) the
// wait in front of the CURRENT batch's first use became vmcnt(0): it also waited for the prefetch just issued, one exposed L2
// round trip per batch -- 2.5 of the kernel's 6.4 SIMD cycles per instruction (r04 what-if run without the loads: 1.0 instead of
// 2.5 ms).  So the prefetch is unconditional (behind the last batch it re-reads the current one) and VEC is a compile-time choice.
// VEC (n % 4 == 0, sums pitch % 8 == 0): one 16-byte load of sums, four of coordinates.  A lane's 8 columns may reach past column
// n - 1 (the terms are masked there): the sums row is read inside its pitch, the coordinates up to 7 floats behind X / Yy -- X is
// followed by Yy, Yy by the session's 64 floats of padding.  A lane entirely behind column n - 1 reads the last lane that is not.
template <bool LUTSRC, bool VEC>
__device__ __forceinline__ void seq_load(SeqBatch &b, const ProbSrc &src, const float *__restrict__ X,
                                         const float *__restrict__ Yy, int64_t lrc, int64_t jl, int64_t n) {
    if constexpr (VEC) {
        const int64_t jv = jl < n ? jl : ((n - 1) & ~(int64_t)7);
        if (LUTSRC) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(src.ps + lrc * src.ld + jv);
            b.w[0] = v.x; b.w[1] = v.y; b.w[2] = v.z; b.w[3] = v.w;
        } else {
#pragma unroll
            for (int d = 0; d < SQ_CPL / 2; ++d) {
                const int64_t ja = (jv + 2 * d < n) ? jv + 2 * d : n - 1, jb = (jv + 2 * d + 1 < n) ? jv + 2 * d + 1 : n - 1;
                b.pf[d] = f32x2{src.pf[lrc * src.ld + ja], src.pf[lrc * src.ld + jb]};
            }
        }
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(X + jv), a1 = *reinterpret_cast<const f32x4 *>(X + jv + 4);
        const f32x4 c0 = *reinterpret_cast<const f32x4 *>(Yy + jv), c1 = *reinterpret_cast<const f32x4 *>(Yy + jv + 4);
        b.x[0] = f32x2{a0.x, a0.y}; b.x[1] = f32x2{a0.z, a0.w}; b.x[2] = f32x2{a1.x, a1.y}; b.x[3] = f32x2{a1.z, a1.w};
        b.y[0] = f32x2{c0.x, c0.y}; b.y[1] = f32x2{c0.z, c0.w}; b.y[2] = f32x2{c1.x, c1.y}; b.y[3] = f32x2{c1.z, c1.w};
    } else {
#pragma unroll
        for (int d = 0; d < SQ_CPL / 2; ++d) {
            const int64_t ja = (jl + 2 * d < n) ? jl + 2 * d : n - 1, jb = (jl + 2 * d + 1 < n) ? jl + 2 * d + 1 : n - 1;
            if (LUTSRC) b.w[d] = (uint32_t)src.ps[lrc * src.ld + ja] | ((uint32_t)src.ps[lrc * src.ld + jb] << 16);
            else b.pf[d] = f32x2{src.pf[lrc * src.ld + ja], src.pf[lrc * src.ld + jb]};
            b.x[d] = f32x2{X[ja], X[jb]};
            b.y[d] = f32x2{Yy[ja], Yy[jb]};
        }
    }
}

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// The LUT is the first object of every SEQ kernel's LDS and the kernels have no static LDS: it starts at LDS address 0 (checked on
// entry, seq_lut_at_zero), so the byte offset of an entry IS its address and one SDWA shift per entry (pick the 16-bit half, scale by
// 4) replaces and / shift + shift-add (embed_fast.hip uses the same gather).
__device__ __forceinline__ f32x2 seq_lut_pair(uint32_t w) {
    uint32_t a, b;
    const uint32_t two = 2u;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(a) : "s"(two), "v"(w));
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(b) : "s"(two), "v"(w));
    typedef const __attribute__((address_space(3))) float *lds_f32;
    return f32x2{*reinterpret_cast<lds_f32>(a), *reinterpret_cast<lds_f32>(b)};
}
__device__ __forceinline__ void seq_lut_at_zero(const float *lut_s) {
    typedef __attribute__((address_space(3))) const float *lds_f32;
    if ((uint32_t)(uintptr_t)(lds_f32)lut_s != 0u) __builtin_trap();
}

// The 8 terms of one lane's batch: t * dx, t * dy of columns jl32 .. jl32 + 7 against point (xi, yi) = row i32, and the batch's
// cross-entropy contribution in log2 units.  SLOW: generic IEEE divisions (some squared distance beyond 1e30); LOSS: the batch has
// columns right of the wave's rows; MASK: per-term predicates (the batch reaches past column n - 1 or contains the diagonal of
// one of the wave's rows).  The two divisions are the exhaustively verified short sequences of seq_div.h.
// Everything is written on column PAIRS as v_pk_{add,mul,fma}_f32 -- the same IEEE operations, two columns per issue slot: per
// pair of terms 2 SDWA shifts (LUT byte offsets), dx, dy, dx^2, dy^2, +, 1 +, 2 rcp, 2 pk_fma, 2 med3, 1 -, 2 rcp, pk_mul,
// 2 pk_fma, p - q, t, t dx, t dy = 26 instructions (13 per term; left to its own vectoriser the compiler reached 17.6).
// dx, dy come from the dispatcher, which needs them for its range test anyway.
template <bool LUTSRC, bool SLOW, bool LOSS, bool MASK>
__device__ __forceinline__ void seq_terms(const SeqBatch &cur, const float *__restrict__ lut_s, const f32x2 (&dx)[SQ_CPL / 2],
                                          const f32x2 (&dy)[SQ_CPL / 2], const f32x2 (&d2)[SQ_CPL / 2], int i32, int n32, int jl32,
                                          float (&tx)[SQ_CPL], float (&ty)[SQ_CPL], float &ce2) {
    const f32x2 one2 = {1.0f, 1.0f};
    f32x2 es2 = {0.0f, 0.0f};
    float prod = 1.0f;
#pragma unroll
    for (int d = 0; d < SQ_CPL / 2; ++d) {                               // 4 independent pairs of terms
        const f32x2 p = LUTSRC ? seq_lut_pair(cur.w[d]) : cur.pf[d];
        const f32x2 s1 = one2 + d2[d];                                   // d2 = (dx*dx) + (dy*dy), no FMA (taichi_core.py:254)
        f32x2 q;                                                         // 1 / (1 + d2)   :255
        if (SLOW) {
            q = f32x2{1.0f / s1.x, 1.0f / s1.y};
        } else {                                                         // seq_rcp<1>: v_rcp_f32 + one Newton step
            static_assert(KMAP_SEQ_RCP_STEPS == 1 && KMAP_SEQ_QUO_RSTEPS == 0 && KMAP_SEQ_QUO_STEPS == 1, "the packed sequences below");
            const f32x2 r = {__builtin_amdgcn_rcpf(s1.x), __builtin_amdgcn_rcpf(s1.y)};
            q = pk_fma(pk_fma(-s1, r, one2), r, r);
        }
        q = f32x2{__builtin_amdgcn_fmed3f(q.x, 0.001f, 0.999f), __builtin_amdgcn_fmed3f(q.y, 0.001f, 0.999f)};   // np.minimum(.., 1 - 1e-3), np.maximum(.., 1e-3)
        const f32x2 omq = one2 - q;
        f32x2 u;                                                         // q / (1 - q)   visualization.py:132-134
        if (SLOW) {
            u = f32x2{q.x / omq.x, q.y / omq.y};
        } else {                                                         // seq_quo<0, 1>: q * rcp(1 - q) + one residual correction
            const f32x2 r2 = {__builtin_amdgcn_rcpf(omq.x), __builtin_amdgcn_rcpf(omq.y)};
            const f32x2 u0 = q * r2;
            u = pk_fma(pk_fma(-omq, u0, q), r2, u0);
        }
        const f32x2 t = u * (p - q);
        f32x2 tx2 = t * dx[d], ty2 = t * dy[d];                          // products rounded on their own (-ffp-contract=off)
        const int ja = jl32 + 2 * d, jb = ja + 1;
        if (MASK) {
            const bool ua = (ja < n32) && (ja != i32), ub = (jb < n32) && (jb != i32);
            tx2 = f32x2{ua ? tx2.x : 0.0f, ub ? tx2.y : 0.0f};
            ty2 = f32x2{ua ? ty2.x : 0.0f, ub ? ty2.y : 0.0f};
        }
        tx[2 * d] = tx2.x; tx[2 * d + 1] = tx2.y;
        ty[2 * d] = ty2.x; ty[2 * d + 1] = ty2.y;
        if (LOSS) {
            // -(p ln q + (1-p) ln(1-q)) = -ln2 (log2(1-q) + p log2(q/(1-q))): one log per pair + one log of the product
            // of the eight (1-q); the reference's eps branches change a term by < 1e-9 relative (p < 1e-10) or not at
            // all (p = 1), and the loss is not part of the bit-pinned path
            const f32x2 lg = {__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)};
            f32x2 pl = p, om = omq;
            if (MASK) {                                                  // a plain batch with loss lies right of all the wave's rows
                const bool la = (ja < n32) && (ja > i32), lb = (jb < n32) && (jb > i32);
                pl = f32x2{la ? p.x : 0.0f, lb ? p.y : 0.0f};
                om = f32x2{la ? omq.x : 1.0f, lb ? omq.y : 1.0f};
            }
            es2 = pk_fma(pl, lg, es2);
            // ONE chain in column order, not two packed half-products: near the loss floor every (1 - q) is the same float
            // (0.999), the rounding of its powers is then systematic, and the chain's happens to sit 2.4e-6 from the reference's
            // per-term logs where the pairwise product sits 9.6e-6 (tests/test_gpu_embed.py::test_early_stop_golden)
            prod = (prod * om.x) * om.y;
        }
    }
    if (LOSS) ce2 = __builtin_amdgcn_logf(prod) + (es2.x + es2.y);
}
// FAR batches: every pair of the wave's batch has d2 >= 1000, so every q = 1 / (1 + d2) <= 1 / 1001 is clipped to 0.001f
// (visualization.py:254-255) and q, 1 - q, q / (1 - q), their logs and the product of the eight (1 - q) are the SAME floats for all
// pairs: the two divisions, the clip and the logs are replaced by constants -- constants computed once per thread by the very
// instruction sequences of the general path, on the hardware (seq_far_consts), so a batch evaluated here has bit for bit the terms
// and the loss partial the general path gives it.  12 instead of 26 instructions per pair of terms.  Once an embedding has spread
// out most batches are far (r04, N = 5000, two motif clusters + noise: 33 % of all pairs after 100 iterations, 74 % after 2500).
struct SeqFar {
    float q, u, lg, logprod8;
};
__device__ __forceinline__ SeqFar seq_far_consts() {
    SeqFar f;
    float q = 0.001f;
    asm volatile("" : "+v"(q));                   // evaluated by the instructions below, not folded by the compiler
    f.q = __builtin_amdgcn_fmed3f(q, 0.001f, 0.999f);
    const float omq = 1.0f - f.q;
    f.u = seq_quo<KMAP_SEQ_QUO_RSTEPS, KMAP_SEQ_QUO_STEPS>(f.q, omq);
    f.lg = __builtin_amdgcn_logf(f.u);
    float prod = 1.0f;
#pragma unroll
    for (int c = 0; c < SQ_CPL; ++c) prod = prod * omq;   // the general path's chain, column by column
    f.logprod8 = __builtin_amdgcn_logf(prod);
    return f;
}
template <bool LUTSRC, bool LOSS>
__device__ __forceinline__ void seq_terms_far(const SeqBatch &cur, const float *__restrict__ lut_s, const f32x2 (&dx)[SQ_CPL / 2],
                                              const f32x2 (&dy)[SQ_CPL / 2], const SeqFar &far, float (&tx)[SQ_CPL], float (&ty)[SQ_CPL],
                                              float &ce2) {
    const f32x2 q2 = {far.q, far.q}, u2 = {far.u, far.u}, lg2 = {far.lg, far.lg};
    f32x2 es2 = {0.0f, 0.0f};
#pragma unroll
    for (int d = 0; d < SQ_CPL / 2; ++d) {
        const f32x2 p = LUTSRC ? seq_lut_pair(cur.w[d]) : cur.pf[d];
        const f32x2 t = u2 * (p - q2);
        const f32x2 tx2 = t * dx[d], ty2 = t * dy[d];
        tx[2 * d] = tx2.x; tx[2 * d + 1] = tx2.y;
        ty[2 * d] = ty2.x; ty[2 * d + 1] = ty2.y;
        if (LOSS) es2 = pk_fma(p, lg2, es2);
    }
    if (LOSS) ce2 = far.logprod8 + (es2.x + es2.y);
}

// min / max of three as ONE instruction (the d2 are results of arithmetic: nothing to canonicalise first, which the compiler's
// own min / max lowering does with a v_max x, x per operand)
__device__ __forceinline__ float seq_min3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float seq_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// wave-uniform dispatch over the variants of one batch.  `plain`: the batch neither reaches past column n - 1 nor contains the diagonal
// of any of the wave's rows (no per-term masks); `want_loss`: some column of the batch lies right of some row of the wave (each
// unordered pair is charged to its j > i side).  Both are the caller's, who knows them as scalars.  The far test comes first and needs
// only the smallest d2 of the batch; the largest (some squared distance too large for the short divisions -> generic division) is
// looked at by the batches that do divide: a far batch has no division, its q is the clip constant whatever the distance.
template <bool LUTSRC>
__device__ __forceinline__ void seq_terms_select(const SeqBatch &cur, const float *__restrict__ lut_s, float xi, float yi, int i32, int n32,
                                                 int jl32, bool plain, bool want_loss, const SeqFar &far, float (&tx)[SQ_CPL],
                                                 float (&ty)[SQ_CPL], float &ce2) {
    const f32x2 xi2 = {xi, xi}, yi2 = {yi, yi};
    f32x2 dx[SQ_CPL / 2], dy[SQ_CPL / 2], d2[SQ_CPL / 2];
#pragma unroll
    for (int d = 0; d < SQ_CPL / 2; ++d) {
        dx[d] = xi2 - cur.x[d];
        dy[d] = yi2 - cur.y[d];
        d2[d] = dx[d] * dx[d] + dy[d] * dy[d];
    }
    ce2 = 0.0f;
    if (plain) {
        const float l = seq_min3(seq_min3(seq_min3(d2[0].x, d2[0].y, d2[1].x), d2[1].y, d2[2].x), d2[2].y, fminf(d2[3].x, d2[3].y));
        if (!__any(!(l >= 1000.0f))) {                                   // a NaN distance gives NaN terms on either path
            if (want_loss) seq_terms_far<LUTSRC, true>(cur, lut_s, dx, dy, far, tx, ty, ce2);
            else seq_terms_far<LUTSRC, false>(cur, lut_s, dx, dy, far, tx, ty, ce2);
            return;
        }
    }
    const float m = seq_max3(seq_max3(seq_max3(d2[0].x, d2[0].y, d2[1].x), d2[1].y, d2[2].x), d2[2].y, fmaxf(d2[3].x, d2[3].y));
    if (__any(!(m < 1e30f))) {   // rare (coordinates beyond 1e15): one generic instantiation
        seq_terms<LUTSRC, true, true, true>(cur, lut_s, dx, dy, d2, i32, n32, jl32, tx, ty, ce2);
    } else if (plain) {
        if (want_loss) seq_terms<LUTSRC, false, true, false>(cur, lut_s, dx, dy, d2, i32, n32, jl32, tx, ty, ce2);
        else seq_terms<LUTSRC, false, false, false>(cur, lut_s, dx, dy, d2, i32, n32, jl32, tx, ty, ce2);
    } else {
        seq_terms<LUTSRC, false, true, true>(cur, lut_s, dx, dy, d2, i32, n32, jl32, tx, ty, ce2);
    }
}
// rows_in_wave consecutive rows from wave_row_min; batch = columns [j0, j0 + batch_cols)
template <bool LUTSRC>
__device__ __forceinline__ void seq_terms_dispatch(const SeqBatch &cur, const float *__restrict__ lut_s, float xi, float yi, int i32,
                                                   int64_t n, int64_t j0, int batch_cols, int64_t wave_row_min, int rows_in_wave,
                                                   int jl32, const SeqFar &far, float (&tx)[SQ_CPL], float (&ty)[SQ_CPL], float &ce2) {
    const bool want_loss = (j0 + batch_cols - 1) > wave_row_min;
    const bool plain = (j0 + batch_cols <= n) && (j0 + batch_cols - 1 < wave_row_min || j0 > wave_row_min + rows_in_wave - 1);
    seq_terms_select<LUTSRC>(cur, lut_s, xi, yi, i32, (int)n, jl32, plain, want_loss, far, tx, ty, ce2);
}

// quad form, block `bid` of the rows [0, nrows) (lut_s: the block's LUT copy in LDS, already filled; wl: SQ_WAVES doubles of LDS)
// SUB = 4: the quad form.  SUB = 2 (pair form): two sub-lanes per row, 32 rows per wave -- every ordered add then serves 32 rows instead
// of 16 (4 instead of 8 add instructions per pair), at twice the columns per lane; one round of pair waves replaces two rounds of quad
// waves (12.6 N against 2 x 7.3 N instructions per SIMD).  Rows [lrow0, lrow0 + ...) of the local range, bounded by `nrows`.
template <bool LUTSRC, int SUB>
__device__ __forceinline__ void seq_quad_body(const ProbSrc &src, const float *__restrict__ Y, int64_t n, int64_t row0, int64_t lrow0,
                                              int64_t nrows, float *__restrict__ G, double *__restrict__ loss_part, int64_t bid,
                                              const float *__restrict__ lut_s, double *wl) {
    constexpr int SQ_SUB = SUB, SQ_ROWS = KMAP_WAVE / SUB, SQ_BATCH = SUB * SQ_CPL;   // shadow the quad form's constants
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (SQ_SUB - 1);
    const int64_t lr = lrow0 + (bid * SQ_WAVES + wave) * SQ_ROWS + (lane / SQ_SUB);
    const bool valid = lr < nrows;
    const int64_t lrc = valid ? lr : nrows - 1;
    const int64_t srow = prob_src_row(src, lrc);             // the stored row of its sums (ProbSrc::rowmap)
    const int64_t i = row0 + lrc;
    const float *X = Y, *Yy = Y + n;
    const float xi = X[i], yi = Yy[i];
    const bool vec = ((n & 3) == 0) && (src.ld % 8 == 0);
    const int i32 = (int)i;                      // n < 2^31 (checked by the host)
    const int64_t wave_row_min = row0 + lrow0 + (bid * SQ_WAVES + wave) * SQ_ROWS;   // smallest global row of the wave
    float gx = 0.0f, gy = 0.0f, ce_acc = 0.0f;
    double loss = 0.0;
    const SeqFar far = seq_far_consts();
    auto run = [&](auto vec_tag) {
    constexpr bool VEC = decltype(vec_tag)::value;
    // two batch buffers in alternating roles (no register copies between batches): while `cur` is evaluated, `nxt` is in flight
    SeqBatch bufA, bufB;
    seq_load<LUTSRC, VEC>(bufA, src, X, Yy, srow, (int64_t)sub * SQ_CPL, n);
    auto step = [&](const SeqBatch &cur, SeqBatch &nxt, int64_t j0) {
        const int64_t jl = j0 + (int64_t)sub * SQ_CPL;                       // this lane's first column of the batch
        seq_load<LUTSRC, VEC>(nxt, src, X, Yy, srow, (j0 + SQ_BATCH < n) ? jl + SQ_BATCH : jl, n);   // prefetch, always (see seq_load)
        float tx[SQ_CPL], ty[SQ_CPL];
        float ce2;                                                           // loss terms in log2 units (order-free)
        seq_terms_dispatch<LUTSRC>(cur, lut_s, xi, yi, i32, n, j0, SQ_BATCH, wave_row_min, SQ_ROWS, (int)jl, far, tx, ty, ce2);
        ce_acc += ce2;
        // ordered accumulation over the batch's 32 columns: column j0 + 8*s2 + c lives in sub-lane s2, slot c
        asm volatile("s_nop 1");
        if constexpr (SUB == 4) {
            add_quad_block<0>(gx, gy, tx, ty);
            add_quad_block<1>(gx, gy, tx, ty);
            add_quad_block<2>(gx, gy, tx, ty);
            add_quad_block<3>(gx, gy, tx, ty);
        } else {
            add_quad_block<4>(gx, gy, tx, ty);
            add_quad_block<5>(gx, gy, tx, ty);
        }
        if (((j0 / SQ_BATCH) & 7) == 7) {
            loss += (double)ce_acc;
            ce_acc = 0.0f;
        }
    };
    for (int64_t j0 = 0; j0 < n; j0 += 2 * SQ_BATCH) {
        step(bufA, bufB, j0);
        if (j0 + SQ_BATCH < n) step(bufB, bufA, j0 + SQ_BATCH);
    }
    };
    // The product's shape (u16 sums + LUT, n % 4 == 0, pitch % 8 == 0): everything that says WHICH batch lives in scalar registers.
    // The wave's rows and the batch's columns are wave-uniform, so the load addresses are a scalar base (advanced by scalar adds) + a
    // per-lane byte offset computed once, and "left of the rows / on their diagonal / right of them / past column n - 1" are scalar
    // compares: no vector instruction of a step goes into bookkeeping (r05: 113 of the 265 instructions of a far step did).
    auto run_scalar = [&]() {
        const int64_t wave_lr0 = seq_uniform(lrow0 + (bid * SQ_WAVES + wave) * SQ_ROWS);
        const int64_t base_lr = wave_lr0 < nrows ? wave_lr0 : nrows - 1;           // a wave entirely behind the rows reads the last one
        // 32-bit scalars (n < 2^24 here): the scalar unit compares 32-bit integers, 64-bit compares would be vector instructions
        const int n32 = (int)n, wrow_min = (int)(row0 + wave_lr0), wrow_max = wrow_min + SQ_ROWS - 1;
        const int j0_last = ((n32 - 1) / SQ_BATCH) * SQ_BATCH;                     // the batch that may reach past column n - 1
        // Buffer loads: address = descriptor base + per-lane offset (computed once) + scalar offset (the batch: one scalar add per
        // stream and step instead of a 64-bit pointer per load).  The batch that reaches past column n - 1 reads what lies there -- the
        // rest of the sums row's pitch or the next row, Yy behind X, the session's 64 floats of padding behind Yy -- or, past the end of
        // the descriptor, zeros: those columns are masked out of the terms by selects, never multiplied away.
        const int64_t base_s = seq_uniform(prob_src_row(src, base_lr));            // stored rows never step back: srow >= base_s
        const uint32_t row_off = (uint32_t)((srow - base_s) * src.ld * 2);         // < 64 rows x pitch x 2 bytes
        const uint32_t so = row_off + (uint32_t)sub * SQ_CPL * 2u, co = (uint32_t)sub * SQ_CPL * 4u;
        const uint64_t sums_bytes = (uint64_t)(src.src_rows - base_s) * (uint64_t)src.ld * 2u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(src.ps + base_s * src.ld), 0,
                                                                          sums_bytes < 0xffffffffull ? (int)(uint32_t)sums_bytes : -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X), 0, (int)((2 * n32 + 64) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Yy), 0, (int)((n32 + 64) * 4), 0x00020000);
        // j0: scalar.  The sums come from HBM, two steps ahead; the coordinates from L2, one step ahead and issued BEFORE the sums
        // load of the same step: loads return in order, so whatever is issued ahead of a load that is awaited one step later has one
        // step to arrive as well
        auto load_w = [&](SeqBatch &b, int j0) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)so, j0 * 2, 0);
            b.w[0] = v.x; b.w[1] = v.y; b.w[2] = v.z; b.w[3] = v.w;
        };
        auto load_xy = [&](SeqBatch &b, int j0) {
            const u32x4 a0 = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)co, j0 * 4, 0), a1 = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)co + 16, j0 * 4, 0);
            const u32x4 c0 = __builtin_amdgcn_raw_buffer_load_b128(ry, (int)co, j0 * 4, 0), c1 = __builtin_amdgcn_raw_buffer_load_b128(ry, (int)co + 16, j0 * 4, 0);
            auto f2 = [](uint32_t lo, uint32_t hi) { return f32x2{__uint_as_float(lo), __uint_as_float(hi)}; };
            b.x[0] = f2(a0.x, a0.y); b.x[1] = f2(a0.z, a0.w); b.x[2] = f2(a1.x, a1.y); b.x[3] = f2(a1.z, a1.w);
            b.y[0] = f2(c0.x, c0.y); b.y[1] = f2(c0.z, c0.w); b.y[2] = f2(c1.x, c1.y); b.y[3] = f2(c1.z, c1.w);
        };
        // Two stages, one batch apart, in one loop pass: stage A of batch s + 1 (differences, squared distances, the far test, the LUT
        // gathers issued) and stage B of batch s (its terms from what stage A left, the ordered adds).  A wave is a chain of latencies
        // -- the min tree feeding a scalar branch, the gathers' LDS round trip, the add chain -- and a SIMD has two or three of these
        // waves: in one pass the latencies of a batch now lie beside the other batch's instructions instead of in front of its own
        // (r05 what-ifs at N = 50 000, all batches far: without the adds 1.50 instead of 1.53 ms, without the gathers 1.49 -- nothing
        // was busy, everything waited).
        struct Staged {
            f32x2 dx[SQ_CPL / 2], dy[SQ_CPL / 2], d2[SQ_CPL / 2];
            SeqBatch g;                                                            // g.pf: the gathered probabilities
        };
        auto stage_a = [&](const SeqBatch &raw, Staged &st) -> bool {              // -> every pair of the batch is far (scalar)
            const f32x2 xi2 = {xi, xi}, yi2 = {yi, yi};
#pragma unroll
            for (int d = 0; d < SQ_CPL / 2; ++d) {
                st.g.pf[d] = seq_lut_pair(raw.w[d]);
                st.dx[d] = xi2 - raw.x[d];
                st.dy[d] = yi2 - raw.y[d];
                st.d2[d] = st.dx[d] * st.dx[d] + st.dy[d] * st.dy[d];
            }
            const float l = seq_min3(seq_min3(seq_min3(st.d2[0].x, st.d2[0].y, st.d2[1].x), st.d2[1].y, st.d2[2].x), st.d2[2].y,
                                     fminf(st.d2[3].x, st.d2[3].y));
            return !__any(!(l >= 1000.0f));
        };
        const int jsub = sub * SQ_CPL;
        // REGION 0: batches left of the wave's rows (no masks, no loss); 1: right of them and inside column n - 1 (no masks, loss);
        // 2: anything (the batches on the diagonal and the one that reaches past column n - 1)
        auto stage_b = [&](auto region_tag, const Staged &st, bool all_far, int j0) {
            constexpr int REGION = decltype(region_tag)::value;
            float tx[SQ_CPL], ty[SQ_CPL];
            float ce2 = 0.0f;
            bool want_loss = REGION == 1, plain = REGION != 2;
            if constexpr (REGION == 2) {
                want_loss = (j0 + SQ_BATCH - 1) > wrow_min;
                plain = (j0 + SQ_BATCH <= n32) && (j0 + SQ_BATCH - 1 < wrow_min || j0 > wrow_max);
            }
            const int jl32 = j0 + jsub;
            if (plain && all_far) {
                if (want_loss) seq_terms_far<false, true>(st.g, lut_s, st.dx, st.dy, far, tx, ty, ce2);
                else seq_terms_far<false, false>(st.g, lut_s, st.dx, st.dy, far, tx, ty, ce2);
            } else {
                const float m = seq_max3(seq_max3(seq_max3(st.d2[0].x, st.d2[0].y, st.d2[1].x), st.d2[1].y, st.d2[2].x), st.d2[2].y,
                                         fmaxf(st.d2[3].x, st.d2[3].y));
                if (__any(!(m < 1e30f))) {
                    seq_terms<false, true, true, true>(st.g, lut_s, st.dx, st.dy, st.d2, i32, n32, jl32, tx, ty, ce2);
                } else if (plain) {
                    if (want_loss) seq_terms<false, false, true, false>(st.g, lut_s, st.dx, st.dy, st.d2, i32, n32, jl32, tx, ty, ce2);
                    else seq_terms<false, false, false, false>(st.g, lut_s, st.dx, st.dy, st.d2, i32, n32, jl32, tx, ty, ce2);
                } else {
                    seq_terms<false, false, true, true>(st.g, lut_s, st.dx, st.dy, st.d2, i32, n32, jl32, tx, ty, ce2);
                }
            }
            if constexpr (REGION != 0) {                                         // left of the rows ce2 is 0 and ce_acc still is
                ce_acc += ce2;
                if (((j0 / SQ_BATCH) & 7) == 7) {
                    loss += (double)ce_acc;
                    ce_acc = 0.0f;
                }
            }
            asm volatile("s_nop 1");
            if constexpr (SUB == 4) {
                add_quad_split(gx, tx, ty);                                      // gx: the x sum in lanes 0 / 1 of the quad, the y sum in lanes 2 / 3
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    asm("v_add_f32_dpp %0, %2, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %3, %1 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf"
                        : "+v"(gx), "+v"(gy) : "v"(tx[c]), "v"(ty[c]));
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    asm("v_add_f32_dpp %0, %2, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %3, %1 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf"
                        : "+v"(gx), "+v"(gy) : "v"(tx[c]), "v"(ty[c]));
            }
        };
        // raw batches (past the last one: the last one again).  At the top of a pass for batch j0: stA = stage A of batch j0; rawB =
        // batch j0 + 1, both parts issued; rawA.w = the sums of batch j0 + 2, issued; rawA's coordinates free
        SeqBatch rawA, rawB;
        Staged stA, stB;
        auto at = [&](int j0) { return j0 < j0_last ? j0 : j0_last; };
        load_xy(rawA, 0);
        load_w(rawA, 0);
        load_xy(rawB, at(SQ_BATCH));
        load_w(rawB, at(SQ_BATCH));
        bool farA = stage_a(rawA, stA), farB = false;
        load_w(rawA, at(2 * SQ_BATCH));
        auto span = [&](auto region_tag, int ja, int jb) {                       // ja: a multiple of two batches
            for (int j0 = ja; j0 < jb; j0 += 2 * SQ_BATCH) {
                load_xy(rawA, at(j0 + 2 * SQ_BATCH));
                farB = stage_a(rawB, stB);
                load_w(rawB, at(j0 + 3 * SQ_BATCH));
                stage_b(region_tag, stA, farA, j0);
                load_xy(rawB, at(j0 + 3 * SQ_BATCH));
                farA = stage_a(rawA, stA);
                load_w(rawA, at(j0 + 4 * SQ_BATCH));
                if (j0 + SQ_BATCH < jb) stage_b(region_tag, stB, farB, j0 + SQ_BATCH);
            }
        };
        // the three regions as three loops, their borders rounded to pairs of batches towards the diagonal region (which can do any
        // batch): what a batch needs is then known where the code is written, not compared in every step
        constexpr int B2 = 2 * SQ_BATCH;
        const int jn = (n32 / B2) * B2;                                          // whole pairs of batches inside column n - 1
        int ja = (wrow_min / B2) * B2, jb = ((wrow_max + B2) / B2) * B2;
        if (ja > jn) ja = jn;
        if (jb > jn) jb = n32;                                                   // no room right of the rows: the diagonal region runs to the end
        span(std::integral_constant<int, 0>{}, 0, ja);
        span(std::integral_constant<int, 2>{}, ja, jb);
        if (jb < n32) {
            span(std::integral_constant<int, 1>{}, jb, jn);
            span(std::integral_constant<int, 2>{}, jn, n32);
        }
        if constexpr (SUB == 4) gy = __shfl(gx, (lane & ~3) | 2);                // the y chain lives in lanes 2 / 3 (add_quad_split)
    };
    if (LUTSRC && vec && src.ld < ((int64_t)1 << 24)) run_scalar();
    else if (vec) run(std::true_type{});
    else run(std::false_type{});
    loss += (double)ce_acc;
    loss *= -0.6931471805599453;   // log2 units -> -ln
    if (valid && sub == 0) {
        G[i] = gx;
        G[n + i] = gy;
    }
    if (!valid) loss = 0.0;
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_down(loss, o);
    if (lane == 0) wl[wave] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < SQ_WAVES; ++w) t += wl[w];
        loss_part[bid] = t;
    }
}

// SEQ forces, WIDE form: a row is owned by GW lanes (GW = 8 .. 64; 64 / GW rows per wave).  Same terms, same order of the row
// sum; what differs is how the terms reach the accumulator: every lane writes its 8 (t dx, t dy) pairs to a per-wave LDS strip,
// and the row's first lane then adds the strip's 8 GW pairs in column order (one packed f32 add per pair).
// Per column and wave that costs 288 / (8 GW) + 1 VALU instructions instead of the quad form's 11 -- but GW / 4 times the lanes
// per row, i.e. more total work: it is for the rows that do NOT fill the machine.  With 16 rows per quad wave and 3 waves per SIMD
// (145 VGPRs), 49 152 rows fill an MI355X exactly; the 848 remaining rows of N = 50 000 were a fourth round of 14 blocks that ran
// alone for a full millisecond (N = 49 152: 3.00 ms, N = 49 216: 3.98 ms).  As 848 one-row waves they take ~0.16 ms.
// wide form, block `bid` of the local rows [lrow0, nrows); xch_all: SQ_WAVES x 64 x 8 (t dx, t dy) pairs of LDS
// acc + (value of `v` in lane K of the caller's 16-lane DPP row), valid in lane 0 of the row: row_ror:(16 - K) makes lane L read lane
// (L + K) mod 16.
template <int K>
__device__ __forceinline__ float add_row16_lane(float acc, float v) {
    float r;
    if constexpr (K == 0) { r = acc + v; return r; }
#define KMAP_ROR(KK, N) if constexpr (K == KK) asm("v_add_f32_dpp %0, %1, %2 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "v"(acc));
    KMAP_ROR(1, 15) KMAP_ROR(2, 14) KMAP_ROR(3, 13) KMAP_ROR(4, 12) KMAP_ROR(5, 11) KMAP_ROR(6, 10) KMAP_ROR(7, 9) KMAP_ROR(8, 8)
    KMAP_ROR(9, 7) KMAP_ROR(10, 6) KMAP_ROR(11, 5) KMAP_ROR(12, 4) KMAP_ROR(13, 3) KMAP_ROR(14, 2) KMAP_ROR(15, 1)
#undef KMAP_ROR
    return r;
}
// the eight columns of source lane K (K >= 1) of a batch, x and y chains alternating, as ONE asm statement: left to itself the
// compiler puts an s_nop behind every pair of these adds (it treats the accumulator, written two instructions earlier, as if it
// were the DPP source: 112 s_nop for the 240 adds of a batch).  Only src0 goes through the DPP network, and tx / ty are written
// long before.
template <int K>
__device__ __forceinline__ void add_row16_block(float &gx, float &gy, const float (&tx)[SQ_CPL], const float (&ty)[SQ_CPL]) {
#define KMAP_ROR8(KK, N)                                                                                                          \
    if constexpr (K == KK)                                                                                                        \
        asm("v_add_f32_dpp %0, %2, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %10, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %3, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %11, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %4, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %12, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %5, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %13, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %6, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %14, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %7, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %15, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %8, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %16, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
            "v_add_f32_dpp %0, %9, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %17, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf"        \
            : "+v"(gx), "+v"(gy)                                                                                                  \
            : "v"(tx[0]), "v"(tx[1]), "v"(tx[2]), "v"(tx[3]), "v"(tx[4]), "v"(tx[5]), "v"(tx[6]), "v"(tx[7]), "v"(ty[0]), "v"(ty[1]), "v"(ty[2]),  \
              "v"(ty[3]), "v"(ty[4]), "v"(ty[5]), "v"(ty[6]), "v"(ty[7]));
    KMAP_ROR8(1, 15) KMAP_ROR8(2, 14) KMAP_ROR8(3, 13) KMAP_ROR8(4, 12) KMAP_ROR8(5, 11) KMAP_ROR8(6, 10) KMAP_ROR8(7, 9) KMAP_ROR8(8, 8)
    KMAP_ROR8(9, 7) KMAP_ROR8(10, 6) KMAP_ROR8(11, 5) KMAP_ROR8(12, 4) KMAP_ROR8(13, 3) KMAP_ROR8(14, 2) KMAP_ROR8(15, 1)
#undef KMAP_ROR8
}
// Sixteen lanes per row (one DPP row), four rows per wave, no LDS: for sessions with fewer rows than one round of quad waves, where
// a wave is a chain of dependent adds and not a share of issue slots.  Lane s of a row computes the terms of columns
// j0 + 8 s .. + 7 of a 128-column batch; lane 0 of the row adds them in column order through row_ror sources (the other lanes
// execute the same adds on rotated operands and are ignored).  The x and y chains alternate, so consecutive adds of one chain are
// two instructions apart -- their latency -- and nothing waits for an LDS round trip as in the strip-exchange form.
constexpr int SR_SUB = 16, SR_ROWS = KMAP_WAVE / SR_SUB, SR_BATCH = SR_SUB * SQ_CPL;
template <bool LUTSRC>
__device__ __forceinline__ void seq_row16_body(const ProbSrc &src, const float *__restrict__ Y, int64_t n, int64_t row0, int64_t lrow0,
                                               int64_t nrows, float *__restrict__ G, double *__restrict__ loss_part, int64_t bid,
                                               const float *__restrict__ lut_s, double *wl) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (SR_SUB - 1);
    const int64_t wave_lr = lrow0 + (bid * SQ_WAVES + wave) * SR_ROWS;    // first local row of the wave
    const int64_t lr = wave_lr + (lane / SR_SUB);
    const bool valid = lr < nrows;
    const int64_t lrc = valid ? lr : nrows - 1;
    const int64_t srow = prob_src_row(src, lrc);             // the stored row of its sums (ProbSrc::rowmap)
    const int64_t i = row0 + lrc;
    const float *X = Y, *Yy = Y + n;
    const float xi = X[i], yi = Yy[i];
    const bool vec = ((n & 3) == 0) && (src.ld % 8 == 0);
    const int i32 = (int)i;
    const int64_t wave_row_min = row0 + wave_lr;
    float gx = 0.0f, gy = 0.0f, ce_acc = 0.0f;
    double loss = 0.0;
    const SeqFar far = seq_far_consts();
    auto run = [&](auto vec_tag) {
    constexpr bool VEC = decltype(vec_tag)::value;
    // (computing the terms of batch b + 1 in the same loop body as the adds of batch b -- a software pipeline for the scheduler to
    // interleave -- measured slower: 0.066 vs 0.058 ms at N = 4000)
    SeqBatch bufA, bufB;
    seq_load<LUTSRC, VEC>(bufA, src, X, Yy, srow, (int64_t)sub * SQ_CPL, n);
    auto step = [&](const SeqBatch &cur, SeqBatch &nxt, int64_t j0) {
        const int64_t jl = j0 + (int64_t)sub * SQ_CPL;
        seq_load<LUTSRC, VEC>(nxt, src, X, Yy, srow, (j0 + SR_BATCH < n) ? jl + SR_BATCH : jl, n);   // prefetch, always (see seq_load)
        float tx[SQ_CPL], ty[SQ_CPL];
        float ce2;
        seq_terms_dispatch<LUTSRC>(cur, lut_s, xi, yi, i32, n, j0, SR_BATCH, wave_row_min, SR_ROWS, (int)jl, far, tx, ty, ce2);
        ce_acc += ce2;
        asm volatile("s_nop 1");
#define SEQ_ADD16(K)                                                                                  \
        _Pragma("unroll") for (int c = 0; c < SQ_CPL; ++c) {                                          \
            gx = add_row16_lane<K>(gx, tx[c]);                                                        \
            gy = add_row16_lane<K>(gy, ty[c]);                                                        \
        }
        SEQ_ADD16(0)
#undef SEQ_ADD16
        add_row16_block<1>(gx, gy, tx, ty); add_row16_block<2>(gx, gy, tx, ty); add_row16_block<3>(gx, gy, tx, ty);
        add_row16_block<4>(gx, gy, tx, ty); add_row16_block<5>(gx, gy, tx, ty); add_row16_block<6>(gx, gy, tx, ty);
        add_row16_block<7>(gx, gy, tx, ty); add_row16_block<8>(gx, gy, tx, ty); add_row16_block<9>(gx, gy, tx, ty);
        add_row16_block<10>(gx, gy, tx, ty); add_row16_block<11>(gx, gy, tx, ty); add_row16_block<12>(gx, gy, tx, ty);
        add_row16_block<13>(gx, gy, tx, ty); add_row16_block<14>(gx, gy, tx, ty); add_row16_block<15>(gx, gy, tx, ty);
        if (((j0 / SR_BATCH) & 7) == 7) {
            loss += (double)ce_acc;
            ce_acc = 0.0f;
        }
    };
    for (int64_t j0 = 0; j0 < n; j0 += 2 * SR_BATCH) {
        step(bufA, bufB, j0);
        if (j0 + SR_BATCH < n) step(bufB, bufA, j0 + SR_BATCH);
    }
    };
    if (vec) run(std::true_type{});
    else run(std::false_type{});
    loss += (double)ce_acc;
    loss *= -0.6931471805599453;   // log2 units -> -ln
    if (valid && sub == 0) {
        G[i] = gx;
        G[n + i] = gy;
    }
    if (!valid) loss = 0.0;
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_down(loss, o);
    if (lane == 0) wl[wave] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < SQ_WAVES; ++w) t += wl[w];
        loss_part[bid] = t;
    }
}

template <bool LUTSRC, int GW>
__device__ __forceinline__ void seq_wide_body(const ProbSrc &src, const float *__restrict__ Y, int64_t n, int64_t row0, int64_t lrow0,
                                              int64_t nrows, float *__restrict__ G, double *__restrict__ loss_part, int64_t bid,
                                              const float *__restrict__ lut_s, double *wl, f32x2 *xch_all) {
    constexpr int RW = KMAP_WAVE / GW;                       // rows per wave
    constexpr int BC = GW * SQ_CPL;                          // columns per batch
    f32x2 *xch = xch_all + (size_t)(threadIdx.x >> 6) * (KMAP_WAVE * SQ_CPL);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (GW - 1), grp = lane / GW;
    const int64_t wave_lr = lrow0 + (bid * SQ_WAVES + wave) * RW;    // first local row of the wave
    const int64_t lr = wave_lr + grp;
    const bool valid = lr < nrows;
    const int64_t lrc = valid ? lr : nrows - 1;
    const int64_t srow = prob_src_row(src, lrc);             // the stored row of its sums (ProbSrc::rowmap)
    const int64_t i = row0 + lrc;
    const float *X = Y, *Yy = Y + n;
    const float xi = X[i], yi = Yy[i];
    const bool vec = ((n & 3) == 0) && (src.ld % 8 == 0);
    const int i32 = (int)i;
    const int64_t wave_row_min = row0 + wave_lr;
    f32x2 acc = {0.0f, 0.0f};
    float ce_acc = 0.0f;
    double loss = 0.0;
    const SeqFar far = seq_far_consts();
    auto run = [&](auto vec_tag) {
    constexpr bool VEC = decltype(vec_tag)::value;
    SeqBatch bufA, bufB;
    seq_load<LUTSRC, VEC>(bufA, src, X, Yy, srow, (int64_t)sub * SQ_CPL, n);
    f32x2 *mine = xch + (size_t)lane * SQ_CPL;                         // = strip of row grp, columns sub * 8 .. + 7
    const f32x4 *strip = reinterpret_cast<const f32x4 *>(xch + (size_t)grp * BC);
    int batch = 0;
    auto step = [&](const SeqBatch &cur, SeqBatch &nxt, int64_t j0) {
        const int64_t jl = j0 + (int64_t)sub * SQ_CPL;
        seq_load<LUTSRC, VEC>(nxt, src, X, Yy, srow, (j0 + BC < n) ? jl + BC : jl, n);      // prefetch, always (see seq_load)
        float tx[SQ_CPL], ty[SQ_CPL];
        float ce2;
        seq_terms_dispatch<LUTSRC>(cur, lut_s, xi, yi, i32, n, j0, BC, wave_row_min, RW, (int)jl, far, tx, ty, ce2);
        ce_acc += ce2;
#pragma unroll
        for (int c = 0; c < SQ_CPL; c += 2) {
            const f32x4 v = {tx[c], ty[c], tx[c + 1], ty[c + 1]};
            *reinterpret_cast<f32x4 *>(mine + c) = v;
        }
        // same wave writes and reads: LDS operations of a wave execute in order; the compiler must not move the reads up
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (sub == 0) {   // ONE lane per row carries the sum: 64 lanes reading the same 16 bytes made the strip reads LDS-bandwidth bound
#pragma unroll 16
            for (int m = 0; m < BC / 2; ++m) {                               // ordered: columns j0 + 2m, j0 + 2m + 1
                const f32x4 v = strip[m];
                acc += f32x2{v.x, v.y};
                acc += f32x2{v.z, v.w};
            }
        }
        asm volatile("" ::: "memory");                                       // the next batch's writes stay behind these reads
        if ((batch & 7) == 7) {
            loss += (double)ce_acc;
            ce_acc = 0.0f;
        }
        ++batch;
    };
    for (int64_t j0 = 0; j0 < n; j0 += 2 * BC) {
        step(bufA, bufB, j0);
        if (j0 + BC < n) step(bufB, bufA, j0 + BC);
    }
    };
    if (vec) run(std::true_type{});
    else run(std::false_type{});
    loss += (double)ce_acc;
    loss *= -0.6931471805599453;
    if (valid && sub == 0) {
        G[i] = acc.x;
        G[n + i] = acc.y;
    }
    if (!valid) loss = 0.0;
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_down(loss, o);
    if (lane == 0) wl[wave] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < SQ_WAVES; ++w) t += wl[w];
        loss_part[bid] = t;
    }
}

// =================================================================================================
// SEQ forces, PRODUCER / ADDER form.  Same terms, same order of every row sum as the forms above; what differs is who adds.
// In the quad / pair / row16 forms every wave both evaluates terms and walks the ordered add chain, and an ordered add serves only
// the 16 / 32 / 4 rows of its wave: 8 / 4 / 30 add instructions per column and wave.  A session with few rows (a rank's share of a
// row-sharded run: 6 250 x 50 000 at G = 8) cannot fill the SIMDs with such waves and runs at the latency of their chains and of
// their one load in flight (r05 shard proxy: 0.68 ms per shard against 2.15 ms for all 50 000 rows -- 3.2x on 8 GPUs).  Here a
// block owns R <= RP rows and splits the roles:
//   * PRODUCER waves evaluate the terms of a chunk of CH columns for all R rows -- lanes along the columns, a step = RPS rows x CH
//     columns = 512 terms -- into an LDS buffer.  Their only memory stream is the sums (one 16-byte load per lane and step, PF steps
//     ahead: the steps of a producer are static, p, p + NP, p + 2 NP, ...); the chunk's coordinates and the block's row coordinates
//     wait in LDS (loads return in order: a short-latency load issued behind a prefetch would wait for it);
//   * ONE ADDER wave, a row per lane, reads the terms back -- (column pair, row) cells of 16 bytes, one ds_read_b128 per two
//     columns -- and adds them in column order with one v_pk_add_f32 per column (x and y chains in one instruction): 1.5
//     instructions per column for up to 64 rows, on a chain nothing else delays (s_setprio).  It also stages the coordinates of the
//     chunk after next (its memory pipe is idle otherwise).
// Two term buffers: while the adder consumes chunk u the producers fill chunk u + 1; one barrier per chunk.
// LDS cell of (column pair cp = 4 g + d, row r) in units of 16 B: g * (4 RP + 1) + d * RP + r -- one cell of padding per column group g
// (= producer lane): the producers' ds_write_b128 (8 lanes x 16 B per 128-B bank window: 8 consecutive column groups, odd pitch -> 8
// different cells of the window) and the adder's ds_read_b128 (16 consecutive rows) are conflict-free, and the adder's address is
// its row's 16 bytes + a compile-time offset: no address arithmetic inside its chain.
// Waves w and w + 4 of a block share a SIMD (round-robin placement): wave 3 is the adder, waves 7 / 11 / 15 are producers only as far
// as the launch says (SA_EA of them) and leave at once otherwise, so that the adder's SIMD carries the adder + SA_EA producers and the
// other three SIMDs M producers each -- equal issue load (the adder issues about as much as three producers' steps).
// =================================================================================================
constexpr int SA_MAX_WAVES = 16;                 // 1 adder + up to 15 producers (the launch picks the number: a divisor-like match of the steps per chunk)
constexpr int SA_ADDER = 3;                      // the adder's wave index
constexpr int SA_PF = 4;                         // sums loads in flight per producer lane
template <int RP, int CH>
struct SaGeom {
    static constexpr int LPR = CH / SQ_CPL;              // lanes per row in a producer step
    static constexpr int RPS = KMAP_WAVE / LPR;          // rows per step
    static constexpr int QP = 4 * RP + 1;                // cells (16 B) per column group: its four column pairs x RP rows + one of padding
    static constexpr size_t BUF_BYTES = (size_t)(CH / 8) * QP * 16;
    static constexpr size_t XY_FLOATS = 2 * (size_t)CH;  // one chunk's x | y
    static constexpr size_t FIXED_BYTES = 2 * BUF_BYTES + 2 * XY_FLOATS * 4 + (size_t)RP * 8 + 16 * 8 + (size_t)RP * 4;   // term buffers, coordinate buffers, row coordinates, loss partials, sums-row offsets
    static_assert(LPR * RPS == KMAP_WAVE && (RP == 32 || RP == 64) && CH % 8 == 0 && CH <= 256, "geometry");
};
template <int RP, int CH>
__global__ __launch_bounds__(KMAP_WAVE *SA_MAX_WAVES) void forces_seqa_kernel(ProbSrc src, const float *__restrict__ Y, int64_t n, int64_t row0,
                                                                 int64_t nrows, int R, float *__restrict__ G,
                                                                 double *__restrict__ loss_part) {
    using GEO = SaGeom<RP, CH>;
    constexpr int LPR = GEO::LPR, RPS = GEO::RPS;
    extern __shared__ __attribute__((aligned(16))) float smem[];             // LUT (at LDS address 0) | term buffers | coordinates | loss partials
    float *lut_s = smem;
    seq_lut_at_zero(lut_s);
    f32x4 *buf0 = reinterpret_cast<f32x4 *>(smem + (((size_t)src.lut_len + 3) & ~(size_t)3));
    f32x4 *buf1 = buf0 + GEO::BUF_BYTES / 16;
    float *xy0 = reinterpret_cast<float *>(buf1 + GEO::BUF_BYTES / 16);      // x[CH] | y[CH] of the even chunks
    float *xy1 = xy0 + GEO::XY_FLOATS;
    f32x2 *rowxy = reinterpret_cast<f32x2 *>(xy1 + GEO::XY_FLOATS);          // (x, y) of the block's rows
    double *wl = reinterpret_cast<double *>(rowxy + RP);
    uint32_t *rowoff = reinterpret_cast<uint32_t *>(wl + 16);                // byte offset of block row r's sums row from the block's first (ProbSrc::rowmap)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // roles (see above): M producers on each of the three other SIMD classes, EA on the adder's
    const int M = R >> 8 & 15, EA = R >> 12 & 3, SA_NP = 3 * M + EA;
    R &= 255;
    const int cls = wave & 3, idx = wave >> 2;
    const float *X = Y, *Yy = Y + n;
    const int64_t lr0 = (int64_t)blockIdx.x * R;                       // first local row of the block
    const int Rb = (int)((nrows - lr0) < (int64_t)R ? (nrows - lr0) : (int64_t)R);   // rows of this block (>= 1)
    const int S = (Rb + RPS - 1) / RPS;                                // producer steps per chunk
    const int64_t nch = (n + CH - 1) / CH;                             // chunks
    if (threadIdx.x < SA_MAX_WAVES) wl[threadIdx.x] = 0.0;
    for (int t = threadIdx.x; t < src.lut_len && t < F_LUT_LDS; t += blockDim.x) lut_s[t] = src.lut[t];
    for (int t = threadIdx.x; t < 2 * CH; t += blockDim.x) {           // coordinates of chunks 0 and 1 (columns past n - 1: the last point's)
        const int64_t j = t < n ? t : n - 1;
        (t < CH ? xy0 : xy1 - CH)[t] = X[j];
        (t < CH ? xy0 : xy1 - CH)[CH + t] = Yy[j];
    }
    const int64_t srow0 = prob_src_row(src, lr0);                           // block-uniform
    if (threadIdx.x < RP) {
        const int64_t lr = lr0 + (threadIdx.x < Rb ? threadIdx.x : Rb - 1);
        rowxy[threadIdx.x] = f32x2{X[row0 + lr], Yy[row0 + lr]};
        rowoff[threadIdx.x] = (uint32_t)((prob_src_row(src, lr) - srow0) * src.ld * 2);
    }
    __syncthreads();
    if (cls == 3 ? (idx > EA) : (idx >= M)) return;                    // a placeholder wave: it helped with the staging and leaves (an ended wave no longer counts at barriers)
    if (wave == SA_ADDER) {
        // ---------------- the adder: lane r = row lr0 + r; its u-th barrier = "chunk u is complete"
        __builtin_amdgcn_s_setprio(3);
        const int r = lane & (RP - 1);
        f32x2 acc = {0.0f, 0.0f};
        for (int64_t u = 0; u < nch; ++u) {
            __syncthreads();
            // coordinates of chunk u + 2 -> registers now, -> the buffer chunk u's producers have just finished with at the end
            float sx[4], sy[4];
            const int64_t jc = (u + 2) * CH + 4 * lane;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t j = jc + c < n ? jc + c : n - 1;
                sx[c] = X[j];
                sy[c] = Yy[j];
            }
            const f32x4 *buf = (u & 1) ? buf1 : buf0;
            const int ncols = (int)((n - u * CH) < (int64_t)CH ? (n - u * CH) : (int64_t)CH);
            if (ncols == CH) {
                // sixteen columns (two producer lanes' column groups = eight cells) per round, the next round's cells in flight
                auto cells = [&](f32x4 (&v)[8], int g2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 *cell = buf + (size_t)(2 * g2 + h) * GEO::QP + r;
                        v[4 * h] = cell[0]; v[4 * h + 1] = cell[RP]; v[4 * h + 2] = cell[2 * RP]; v[4 * h + 3] = cell[3 * RP];
                    }
                };
                auto adds = [&](const f32x4 (&v)[8]) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        acc += f32x2{v[c].x, v[c].y};
                        acc += f32x2{v[c].z, v[c].w};
                    }
                };
                f32x4 va[8], vb[8];
                cells(va, 0);
                static_assert(CH % 32 == 0, "two rounds per loop pass");
#pragma unroll
                for (int g2 = 0; g2 < CH / 16; g2 += 2) {                  // fully unrolled: every read is base + immediate offset
                    cells(vb, g2 + 1);
                    adds(va);
                    cells(va, g2 + 2 < CH / 16 ? g2 + 2 : g2);              // behind the last round: a harmless re-read
                    adds(vb);
                }
            } else {                                                   // the last chunk: only the columns that exist
                for (int c = 0; c < ncols; ++c) {
                    const int cp = c >> 1;
                    const f32x4 v = buf[(size_t)(cp >> 2) * GEO::QP + (size_t)(cp & 3) * RP + r];
                    acc += (c & 1) ? f32x2{v.z, v.w} : f32x2{v.x, v.y};
                }
            }
            if (4 * lane < CH) {
                float *xy = (u & 1) ? xy1 : xy0;
                *reinterpret_cast<f32x4 *>(xy + 4 * lane) = f32x4{sx[0], sx[1], sx[2], sx[3]};
                *reinterpret_cast<f32x4 *>(xy + CH + 4 * lane) = f32x4{sy[0], sy[1], sy[2], sy[3]};
            }
        }
        if (lane < RP && lane < Rb) {
            const int64_t i = row0 + lr0 + lane;
            G[i] = acc.x;
            G[n + i] = acc.y;
        }
        wl[wave] = 0.0;
    } else {
        // ---------------- a producer: steps p, p + NP, ... of the sequence (chunk 0: S steps), (chunk 1: S steps), ...
        // Everything that says WHICH step is wave-uniform and lives in scalar registers (readfirstlane tells the compiler); a lane adds
        // constants it computed once: its row inside the step, its eight columns, and their clamped forms for the block's last step
        // (rows past the block -> its last row) and the last chunk (columns past n - 1 -> the last aligned group).
        const int p = __builtin_amdgcn_readfirstlane(cls == 3 ? 3 * M + idx - 1 : 3 * idx + cls);
        const int NP = __builtin_amdgcn_readfirstlane(SA_NP), nch32 = (int)nch;
        const int sub = lane & (LPR - 1), rin = lane / LPR;
        const int rows_last = Rb - (S - 1) * RPS;                          // rows of the block's last step (1 .. RPS)
        const int rin_last = rin < rows_last ? rin : rows_last - 1;
        const int64_t jl_last = (int64_t)(nch32 - 1) * CH + sub * SQ_CPL;
        const uint32_t off_col = (uint32_t)sub * SQ_CPL * 2u,
                       off_col_last = (uint32_t)((jl_last < n ? jl_last : ((n - 1) & ~(int64_t)7)) - (int64_t)(nch32 - 1) * CH) * 2u;
        const SeqFar far = seq_far_consts();
        double loss = 0.0;
        int pu = 0, pw = p;                                                // (chunk, step in chunk) of the next step to prefetch
        auto norm = [&](int &u, int &w) {
            while (w >= S) {
                w -= S;
                ++u;
            }
        };
        norm(pu, pw);
        const char *sums_base = reinterpret_cast<const char *>(src.ps + srow0 * src.ld);
        auto issue = [&](u32x4 &q) {                                   // always a load: behind the last step the last step's again
            const int u = pu < nch32 ? pu : nch32 - 1;
            const char *col = sums_base + (int64_t)u * CH * 2;                                        // scalar
            const uint32_t off = rowoff[pw * RPS + (pw == S - 1 ? rin_last : rin)] + (u == nch32 - 1 ? off_col_last : off_col);
            q = *reinterpret_cast<const u32x4 *>(col + off);
            pw += NP;
            norm(pu, pw);
        };
        u32x4 q[SA_PF];
#pragma unroll
        for (int i = 0; i < SA_PF; ++i) issue(q[i]);
        int cu = 0, cw = p, passed = 0;
        norm(cu, cw);
        const float *xy_lane = xy0 + SQ_CPL * sub;
        f32x4 *cell_lane = buf0 + (size_t)sub * GEO::QP + rin;
        const int i_lane = (int)(row0 + lr0) + rin, i_lane_last = (int)(row0 + lr0) + rin_last;
        auto step = [&](const u32x4 &sums) {
            SeqBatch cur;
            cur.w[0] = sums.x; cur.w[1] = sums.y; cur.w[2] = sums.z; cur.w[3] = sums.w;
            while (passed < cu) {                                      // chunk cu's buffers are free once barrier cu - 1 has been passed
                __syncthreads();
                ++passed;
            }
            const bool last = cw == S - 1;                             // scalar
            const float *xy = xy_lane + (cu & 1) * (int)GEO::XY_FLOATS;
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(xy), a1 = *reinterpret_cast<const f32x4 *>(xy + 4);
            const f32x4 c0 = *reinterpret_cast<const f32x4 *>(xy + CH), c1 = *reinterpret_cast<const f32x4 *>(xy + CH + 4);
            cur.x[0] = f32x2{a0.x, a0.y}; cur.x[1] = f32x2{a0.z, a0.w}; cur.x[2] = f32x2{a1.x, a1.y}; cur.x[3] = f32x2{a1.z, a1.w};
            cur.y[0] = f32x2{c0.x, c0.y}; cur.y[1] = f32x2{c0.z, c0.w}; cur.y[2] = f32x2{c1.x, c1.y}; cur.y[3] = f32x2{c1.z, c1.w};
            const int r_step = cw * RPS;                               // the step's first row inside the block (scalar)
            const f32x2 rxy = rowxy[r_step + (last ? rin_last : rin)];
            const int j0 = cu * CH;
            float tx[SQ_CPL], ty[SQ_CPL];
            float ce2;
            seq_terms_dispatch<true>(cur, lut_s, rxy.x, rxy.y, r_step + (last ? i_lane_last : i_lane), n, (int64_t)j0, CH,
                                     row0 + lr0 + r_step, RPS, j0 + sub * SQ_CPL, far, tx, ty, ce2);
            if (!last || rin < rows_last) loss += (double)ce2;         // a clamped duplicate of the block's last row does not count
            f32x4 *cell = cell_lane + (size_t)r_step + (cu & 1) * (GEO::BUF_BYTES / 16);   // row r_step + rin < S RPS <= RP
#pragma unroll
            for (int d = 0; d < SQ_CPL / 2; ++d) cell[(size_t)d * RP] = f32x4{tx[2 * d], ty[2 * d], tx[2 * d + 1], ty[2 * d + 1]};
            cw += NP;
            norm(cu, cw);
        };
        while (cu < nch32) {
#pragma unroll
            for (int i = 0; i < SA_PF; ++i) {
                const u32x4 sums = q[i];
                issue(q[i]);                                           // refill at once, on every path (see seq_load): SA_PF loads in flight
                if (cu < nch32) step(sums);
            }
        }
        while (passed < nch32) {                                       // the barriers the adder still waits at
            __syncthreads();
            ++passed;
        }
        loss *= -0.6931471805599453;   // log2 units -> -ln
        for (int o = 32; o > 0; o >>= 1) loss += __shfl_down(loss, o);
        if (lane == 0) wl[wave] = loss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tsum = 0.0;
        for (int w = 0; w < SA_MAX_WAVES; ++w) tsum += wl[w];
        loss_part[blockIdx.x] = tsum;
    }
}

// One launch for both forms: blocks [0, nb_tail) take the left-over rows in the wide form (GW lanes per row; GW = 0: none), the
// blocks behind them the whole rounds in the quad form.  Launched together the wide waves share their SIMDs with three quad
// waves each, which hides the latency of their N-step dependent add chain (alone on the machine -- as a second launch -- the
// 848 one-row waves of N = 50 000 took 0.5 - 0.6 ms; as a fourth wave per SIMD they cost their ~8 % of issue slots).  Wide
// blocks come first in the grid so that they are placed before the CUs fill up.
template <bool LUTSRC, int GW>
__global__ __launch_bounds__(KMAP_WAVE *SQ_WAVES) void forces_seq_kernel(ProbSrc src, const float *__restrict__ Y, int64_t n,
                                                               int64_t row0, int64_t pair_rows, int64_t main_rows, int64_t nrows,
                                                               int nb_tail, int nb_pair, int nb_main, float *__restrict__ G,
                                                               double *__restrict__ loss_part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // LUT (at LDS address 0) | exchange strips of the wide form | SQ_WAVES loss partials
    constexpr size_t XCH_FLOATS = GW ? (size_t)SQ_WAVES * KMAP_WAVE * SQ_CPL * 2 : 0;
    float *lut_s = smem;
    const size_t lut_floats = LUTSRC ? (((size_t)src.lut_len + 3) & ~(size_t)3) : 4;
    float *xch_s = smem + lut_floats;
    double *wl = reinterpret_cast<double *>(xch_s + XCH_FLOATS);
    if (LUTSRC) {
        seq_lut_at_zero(lut_s);
        for (int t = threadIdx.x; t < src.lut_len && t < F_LUT_LDS; t += blockDim.x) lut_s[t] = src.lut[t];
        __syncthreads();
    }
    if constexpr (GW != 0) {
        if ((int)blockIdx.x < nb_tail) {   // block-uniform
            if constexpr (GW == 16)   // one DPP row per matrix row: no exchange through LDS
                seq_row16_body<LUTSRC>(src, Y, n, row0, main_rows, nrows, G, loss_part + nb_main, (int64_t)blockIdx.x, lut_s, wl);
            else
                seq_wide_body<LUTSRC, GW>(src, Y, n, row0, main_rows, nrows, G, loss_part + nb_main, (int64_t)blockIdx.x, lut_s, wl,
                                          reinterpret_cast<f32x2 *>(xch_s));
            return;
        }
    }
    // grid order: wide blocks, pair blocks (rows [0, pair_rows)), quad blocks (rows [pair_rows, main_rows)); loss partials: quad |
    // pair | wide (nb_main = quad + pair blocks)
    const int b = (int)blockIdx.x - nb_tail;
    if (b < nb_pair) seq_quad_body<LUTSRC, 2>(src, Y, n, row0, 0, pair_rows, G, loss_part + (nb_main - nb_pair), (int64_t)b, lut_s, wl);
    else seq_quad_body<LUTSRC, 4>(src, Y, n, row0, pair_rows, main_rows, G, loss_part, (int64_t)(b - nb_pair), lut_s, wl);
}
}  // namespace

int kmap_embed_seq_pair_blocks(const kmap_embed *e) { return (int)(e->seq_pair_rows / (2 * SQ_ROWS * SQ_WAVES)); }
// quad + pair blocks
static int seq_main_blocks(const kmap_embed *e) {
    return kmap_embed_seq_pair_blocks(e) + (int)((e->seq_main_rows - e->seq_pair_rows + SQ_ROWS * SQ_WAVES - 1) / (SQ_ROWS * SQ_WAVES));
}
static int seq_tail_blocks(const kmap_embed *e) {
    if (!e->seq_tail_g) return 0;
    const int64_t rows_per_block = (int64_t)SQ_WAVES * (KMAP_WAVE / e->seq_tail_g);
    return (int)((e->nrows - e->seq_main_rows + rows_per_block - 1) / rows_per_block);
}
// the producer / adder form needs its two 64-KiB term buffers + the LUT in the CU's 160 KiB of LDS
constexpr size_t SA_LDS_MAX = 160 * 1024 - 512;
static bool seq_adder_form(const kmap_embed *e) {
    if (!e->seq_R) return false;
    if (e->src.pf) return false;                          // f32 probability rows (the drop-in float operators): classic forms
    const size_t lut = ((size_t)e->src.lut_len * 4 + 15) & ~(size_t)15;
    // 16-byte loads of the sums: rows 16-byte aligned (the product's pitch is a multiple of 128 entries)
    return SaGeom<32, 256>::FIXED_BYTES + lut <= SA_LDS_MAX && (e->src.ps == nullptr || e->src.ld % 8 == 0);
}
static int seq_adder_blocks(const kmap_embed *e) { return e->seq_R ? (int)((e->nrows + e->seq_R - 1) / e->seq_R) : 0; }
int kmap_embed_seq_blocks(const kmap_embed *e) { return seq_adder_form(e) ? seq_adder_blocks(e) : seq_main_blocks(e) + seq_tail_blocks(e); }
int kmap_embed_seq_blocks_max(const kmap_embed *e) {
    const int a = seq_adder_blocks(e), c = seq_main_blocks(e) + seq_tail_blocks(e);
    return a > c ? a : c;
}

// How the SEQ rows are split between the quad kernel and the wide kernel.  The kernels are VALU-issue bound and every wave of a
// SIMD shares its issue slots, so the cost of a set of waves is (waves on the fullest SIMD) x (instructions per wave); per
// column a quad wave issues ~9.4 instructions (8 terms x 27 + 64 adds + ~20 per 32 columns), a wide wave with g lanes per row
// ~(30 / g + 1.2).  Whole rounds of quad waves (one wave on every SIMD) are the cheapest way to do rows; what is left over is
// given to whichever form finishes it soonest.
void kmap_embed_seq_split(kmap_embed *e) {
    e->seq_main_rows = e->nrows;
    e->seq_tail_g = 0;
    e->seq_pair_rows = 0;
    e->seq_R = e->seq_RP = 0;
    {   // producer / adder form: KMAP_SEQ_FORM=adder|classic forces it on / off (A/B switch + the bit-identity tests of the forms)
        const char *v = getenv("KMAP_SEQ_FORM");
        const bool force_on = v && v[0] == 'a', force_off = v && v[0] == 'c';
        int dev0 = 0, cus0 = 256;
        if (hipGetDevice(&dev0) != hipSuccess || hipDeviceGetAttribute(&cus0, hipDeviceAttributeMultiprocessorCount, dev0) != hipSuccess || cus0 <= 0) cus0 = 256;
        // r05 measurements (tools/seqa_check.py, ms per force evaluation, classic / this form): 300 x 300: .019 / .013; 1000 x 1000: .023 / .016;
        // 5000 x 5000: .082 / .045; 12 000 x 12 000: .26 / .17; 6250 x 50 000 (an eighth of C3): .67 / .35; 16 461 x 17 413: .79 / .38;
        // 25 000 x 200 000 (an eighth of C4): 9.3 / 5.3; 30 000 x 30 000: .93 / .99; 40 000 x 40 000: 1.53 / 1.89; 50 000 x 50 000: 2.1 / 2.6
        // -- the quad / pair forms win once their whole rounds (16 384 rows each) carry most of the rows.  Below n = 3072 the difference is
        // microseconds and the small golden traces keep the form they were recorded against.
        const bool want = force_on || (!force_off && e->nrows > 0 && e->n >= 3072 && 4 * e->nrows < 7 * (int64_t)(4 * cus0) * SQ_ROWS);
        if (want && e->nrows > 0) {
            // one block per CU and round; the fewest rounds the row slots allow, then the smallest R that still fits them (a block's
            // producer time grows with R, its adder time does not)
            const int rp = e->nrows > (int64_t)32 * cus0 ? 64 : 32;
            const int64_t rounds = (e->nrows + (int64_t)rp * cus0 - 1) / ((int64_t)rp * cus0);
            int R = (int)((e->nrows + rounds * cus0 - 1) / (rounds * cus0));
            if (R < 1) R = 1;
            if (R > rp) R = rp;
            e->seq_R = R;
            e->seq_RP = rp;
        }
    }
    static const int off = [] { const char *v = getenv("KMAP_SEQ_TAIL"); return v && v[0] == '0'; }();   // A/B switch
    if (off || e->nrows <= 0) return;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int64_t simds = 4 * (int64_t)cus;
    const int64_t round_rows = simds * SQ_ROWS;                       // rows of one wave on every SIMD
    const int64_t main_rows = (e->nrows / round_rows) * round_rows;
    const int64_t rem = e->nrows - main_rows;
    // two rounds of quad waves -> one round of pair waves (every ordered add serves 32 rows: 12.6 N against 14.6 N instructions per
    // SIMD); an odd round stays in the quad form and shares the SIMDs with the pair round
    static const int pair_on = [] { const char *v = getenv("KMAP_SEQ_PAIR"); return !(v && v[0] == '0'); }();   // A/B switch
    // ... but only next to quad waves: a pair round alone on the SIMDs (one wave each) exposes its add chain (N = 33 000: 1.52 against
    // 1.29 ms for two quad rounds), so at least one quad round stays -- R rounds of quad rows become (R - 1) / 2 pair rounds + the rest
    const int64_t rounds_q = main_rows / round_rows;
    const int64_t pair_rounds = rounds_q >= 3 ? (rounds_q - 1) / 2 : 0;
    e->seq_pair_rows = pair_on ? pair_rounds * 2 * round_rows : 0;
    if (rem == 0) return;
    auto rounds = [&](int64_t waves) { return (double)((waves + simds - 1) / simds); };
    double best = rounds((rem + SQ_ROWS - 1) / SQ_ROWS) * 9.4;       // the remainder as quad waves
    int best_g = 0;
    for (int g : {8, 16, 32, 64}) {
        const int64_t waves = (rem + (KMAP_WAVE / g) - 1) / (KMAP_WAVE / g);
        const double cost = rounds(waves) * (30.0 / g + 1.2);
        if (cost < best) { best = cost; best_g = g; }
    }
    if (main_rows == 0) {
        // fewer rows than one round of quad waves (N < 16 384 on this part): the waves are dependent-add chains, not issue slots, and
        // whole-round counting misjudges 1.2 waves per SIMD as two rounds.  Measured (tools/bench_embed.py --modes seq, ms per
        // iteration, g = quad / 8 / 16 / 32 / 64): N = 1000: .036 .035 .026 .022 .021; 3000: .085 .073 .049 .051 .060;
        // 5000: .138 .115 .105 .105 .133; 8000: .214 .174 .155 .199 .307; 10 000: .265 .293 .249 .297 .462; 14 000: .372 .412 .429 ...
        best_g = rem <= 1500 ? 64 : rem <= 2500 ? 32 : rem <= 12000 ? 16 : 0;
    }
    if (best_g) {
        e->seq_main_rows = main_rows;
        e->seq_tail_g = best_g;
    }
}

int kmap_embed_launch_seq(kmap_embed *e, float *G, hipStream_t st) {
    const bool lut = e->src.ps != nullptr;
    const size_t lds = lut ? (((size_t)e->src.lut_len * 4 + 15) & ~(size_t)15) : 16;
    if (seq_adder_form(e)) {
        const int nb = seq_adder_blocks(e);
#define KMAP_SEQA(RP, CH)                                                                                                          \
    do {                                                                                                                           \
        const size_t lds_a = SaGeom<RP, CH>::FIXED_BYTES + lds;                                                                    \
        KMAP_TRY(kmap_allow_lds((const void *)forces_seqa_kernel<RP, CH>, (int)lds_a));                                            \
        const int steps = (e->seq_R + SaGeom<RP, CH>::RPS - 1) / SaGeom<RP, CH>::RPS;     /* producer steps per chunk */                   \
        /* producers: one step each per chunk where 3 M + EA can equal the steps (7 .. 13 of them), else as many as fit */          \
        int M = 4, EA = 1;                                                                                                         \
        if (steps <= 13) { M = steps >= 12 ? 4 : steps >= 9 ? 3 : 2; EA = steps - 3 * M; if (EA < 0) { EA = 0; } if (EA > 3) { EA = 3; } }          \
        const int waves = 4 * (M > EA + 1 ? M : EA + 1);                                                                           \
        const int geo = e->seq_R | (M << 8) | (EA << 12);                                                                          \
        forces_seqa_kernel<RP, CH><<<nb, KMAP_WAVE * waves, lds_a, st>>>(e->src, e->Y, e->n, e->row0, e->nrows, geo, G, e->loss_part); \
    } while (0)
        if (e->seq_RP == 64) KMAP_SEQA(64, 128); else KMAP_SEQA(32, 256);
#undef KMAP_SEQA
        KMAP_CHECK_HIP(hipGetLastError());
        return KMAP_OK;
    }
    const int nb_main = seq_main_blocks(e), nb_tail = seq_tail_blocks(e);
    const size_t lds_w = (nb_tail ? (size_t)SQ_WAVES * KMAP_WAVE * SQ_CPL * 8 : 0) + lds + SQ_WAVES * 8;   // LUT | strips | loss partials
#define KMAP_SEQ(LUT, GW)                                                                                                          \
    do {                                                                                                                           \
        KMAP_TRY(kmap_allow_lds((const void *)forces_seq_kernel<LUT, GW>, (int)lds_w));                                            \
        forces_seq_kernel<LUT, GW><<<nb_tail + nb_main, KMAP_WAVE * SQ_WAVES, lds_w, st>>>(e->src, e->Y, e->n, e->row0, e->seq_pair_rows, \
                                                                                          e->seq_main_rows, e->nrows, nb_tail,      \
                                                                                          kmap_embed_seq_pair_blocks(e), nb_main, G, e->loss_part); \
    } while (0)
#define KMAP_SEQ_G(LUT)                                                                                       \
    do {                                                                                                      \
        const int g = nb_tail ? e->seq_tail_g : 0;                                                            \
        if (g == 0) KMAP_SEQ(LUT, 0); else if (g == 8) KMAP_SEQ(LUT, 8); else if (g == 16) KMAP_SEQ(LUT, 16); \
        else if (g == 32) KMAP_SEQ(LUT, 32); else KMAP_SEQ(LUT, 64);                                            \
    } while (0)
    if (lut) KMAP_SEQ_G(true); else KMAP_SEQ_G(false);
#undef KMAP_SEQ_G
#undef KMAP_SEQ
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}
