// embed_fast.hip -- FAST forces of the embedding iteration (reference visualization.py:296-317, kernels taichi_core.py:252-326):
// per-pair values as the reference computes them up to the last bits (v_rcp / v_log, FMAs), row sums reduced wavefront-parallel.
// Not bit-pinned; embed_seq.hip is.  Row-wise kernel (row-sharded multi-GPU sessions, small N) and the symmetric tile kernel
// (every unordered pair once; single GPU at N >= 16 384 and the cyclic multi-GPU layout).
#include <math.h>

#include "embed_internal.h"

namespace {
constexpr int BLK = EMB_BLK;

// =================================================================================================
// FAST forces: a wave owns F_RPW rows at a time and sweeps the columns, 8 columns per lane per step.
// Not bit-pinned (row sums are reduced wavefront-parallel), so the per-pair math uses v_rcp_f32 /
// v_log_f32 and explicit FMAs: ~25 VALU + 4 transcendental issues per pair instead of ~96.
//   q   = clamp(1/(1+d2)),  t = q/(1-q)*(p-q),  g += t*(y_i-y_j)
//   ce  = -(p*ln q + (1-p)*ln(1-q)) = -ln2 * (log2(1-q) + p*(log2 q - log2(1-q)))   (eps branches of the
//         reference change ce by < 1e-9 relative and are dropped here; SEQ mode keeps them)
// The diagonal needs no predicate for the gradient (dx = dy = 0 -> t*0 = 0); the loss takes j > i only.
// =================================================================================================

// FAST per-pair core.  q = clip(1/(1+d2), 1e-3, 1-1e-3) is obtained by clamping d2 to [1/999, 999] (the same interval),
// which turns q, 1-q = d2/(1+d2) and q/(1-q) = 1/d2 into products of ONE reciprocal: r = 1/(d2 (1+d2)), q = r d2,
// q/(1-q) = r (1+d2).  The cross-entropy term -(p ln q + (1-p) ln(1-q)) = -ln2 (log2(1-q) - p log2 d2) needs one log per
// pair plus one log of the product of the (1-q) of a lane's 8 columns (each in [1e-3, 0.999]: the product stays normal in
// f32, and for far pairs -- q = 1e-3, the bulk of the sum -- the product form has a smaller systematic error than 8 logs).
constexpr float FAST_D2_MIN = 1.0f / 999.0f, FAST_D2_MAX = 999.0f;
__device__ __forceinline__ void fast_core(float dx, float dy, float p, float &t, float &omq, float &d2c) {
    d2c = __builtin_amdgcn_fmed3f(__builtin_fmaf(dx, dx, dy * dy), FAST_D2_MIN, FAST_D2_MAX);
    const float s1 = 1.0f + d2c;
    const float r = __builtin_amdgcn_rcpf(d2c * s1);
    const float q = r * d2c;
    omq = 1.0f - q;
    t = (r * s1) * (p - q);
}
enum { PL_NONE = 0, PL_ALL = 1, PL_MASK = 2 };   // loss terms: none of the 8 pairs / all of them / only j > i (and j < n)
template <int PL, bool GUARD>
__device__ __forceinline__ void fast_pairs(const float (&p)[F_CPL], const float (&xj)[F_CPL], const float (&yj)[F_CPL],
                                           float xi, float yi, int64_t gi, int64_t j0, int64_t n, float &gx, float &gy,
                                           float &ce2) {
    float esum = 0.0f, prod = 1.0f;
#pragma unroll
    for (int c = 0; c < F_CPL; ++c) {
        const float dx = xi - xj[c], dy = yi - yj[c];
        float t, omq, d2c;
        fast_core(dx, dy, p[c], t, omq, d2c);
        const int64_t j = j0 + c;
        if (GUARD) t = (j < n) ? t : 0.0f;
        gx = __builtin_fmaf(t, dx, gx);
        gy = __builtin_fmaf(t, dy, gy);
        if (PL == PL_ALL) {
            esum = __builtin_fmaf(p[c], __builtin_amdgcn_logf(d2c), esum);
            prod *= omq;
        } else if (PL == PL_MASK) {
            const bool live = (j > gi) && (j < n);
            esum += live ? p[c] * __builtin_amdgcn_logf(d2c) : 0.0f;
            prod *= live ? omq : 1.0f;
        }
    }
    if (PL != PL_NONE) ce2 += __builtin_amdgcn_logf(prod) - esum;
}

template <bool LUTSRC>
__global__ __launch_bounds__(KMAP_WAVE *F_WAVES) void forces_fast_kernel(ProbSrc src, const float *__restrict__ Y,
                                                                          int64_t n, int64_t row0, int64_t nrows,
                                                                          float *__restrict__ G,
                                                                          double *__restrict__ loss_part) {
    extern __shared__ __attribute__((aligned(16))) float lut_s[];   // lut_len floats (dynamic: sized by the launch)
    __shared__ double wloss[F_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (LUTSRC) {
        for (int t = threadIdx.x; t < src.lut_len && t < F_LUT_LDS; t += blockDim.x) lut_s[t] = src.lut[t];
        __syncthreads();
    }
    const float *X = Y, *Yy = Y + n;
    double wave_loss = 0.0;
    const int64_t rbase = ((int64_t)blockIdx.x * F_WAVES + wave) * F_RPW;
    if (rbase < nrows) {
        float xi[F_RPW], yi[F_RPW], gx[F_RPW], gy[F_RPW];
        int64_t gi[F_RPW], lrow[F_RPW];
#pragma unroll
        for (int r = 0; r < F_RPW; ++r) {
            lrow[r] = (rbase + r < nrows) ? rbase + r : nrows - 1;   // clamped duplicate rows are discarded below
            gi[r] = row0 + lrow[r];
            xi[r] = X[gi[r]];
            yi[r] = Yy[gi[r]];
            gx[r] = gy[r] = 0.0f;
        }
        const bool vec_ok = (src.ld % 8 == 0) && (n % 4 == 0 || true);
        float ce2 = 0.0f;   // log2 units, f32 partial flushed into f64 every 16 steps
        int step = 0;
        for (int64_t j0 = (int64_t)lane * F_CPL; j0 < n; j0 += (int64_t)KMAP_WAVE * F_CPL, ++step) {
            const bool full = (j0 + F_CPL <= n);
            float xj[F_CPL], yj[F_CPL];
            if (full && (n % 4 == 0)) {   // 16-byte aligned coordinate rows
                const f32x4 a0 = *reinterpret_cast<const f32x4 *>(X + j0), a1 = *reinterpret_cast<const f32x4 *>(X + j0 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(Yy + j0), b1 = *reinterpret_cast<const f32x4 *>(Yy + j0 + 4);
                xj[0] = a0.x; xj[1] = a0.y; xj[2] = a0.z; xj[3] = a0.w; xj[4] = a1.x; xj[5] = a1.y; xj[6] = a1.z; xj[7] = a1.w;
                yj[0] = b0.x; yj[1] = b0.y; yj[2] = b0.z; yj[3] = b0.w; yj[4] = b1.x; yj[5] = b1.y; yj[6] = b1.z; yj[7] = b1.w;
            } else {
#pragma unroll
                for (int c = 0; c < F_CPL; ++c) {
                    const int64_t j = (j0 + c < n) ? j0 + c : n - 1;
                    xj[c] = X[j];
                    yj[c] = Yy[j];
                }
            }
#pragma unroll
            for (int r = 0; r < F_RPW; ++r) {
                float p[F_CPL];
                if (LUTSRC) {
                    const uint16_t *row = src.ps + lrow[r] * src.ld + j0;
                    if (full && vec_ok) {
                        const u32x4 w = *reinterpret_cast<const u32x4 *>(row);
                        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                        for (int c = 0; c < F_CPL; ++c) p[c] = lut_s[(ws[c >> 1] >> (16 * (c & 1))) & 0xFFFFu];
                    } else {
#pragma unroll
                        for (int c = 0; c < F_CPL; ++c) p[c] = (j0 + c < n) ? lut_s[row[c]] : 0.0f;
                    }
                } else {
                    const float *row = src.pf + lrow[r] * src.ld + j0;
#pragma unroll
                    for (int c = 0; c < F_CPL; ++c) p[c] = (j0 + c < n) ? row[c] : 0.0f;
                }
                float e = 0.0f;
                // wave-uniform choice: the step's 512 columns lie right of the diagonal (all loss terms), left of it (none:
                // each unordered pair is charged once, to its j > i side) or straddle it / the end of the row (masked)
                const int64_t sj0 = (int64_t)step * (KMAP_WAVE * F_CPL);
                const bool wfull = sj0 + KMAP_WAVE * F_CPL <= n;
                if (wfull && sj0 > gi[r]) fast_pairs<PL_ALL, false>(p, xj, yj, xi[r], yi[r], gi[r], j0, n, gx[r], gy[r], e);
                else if (wfull && sj0 + KMAP_WAVE * F_CPL - 1 <= gi[r]) fast_pairs<PL_NONE, false>(p, xj, yj, xi[r], yi[r], gi[r], j0, n, gx[r], gy[r], e);
                else fast_pairs<PL_MASK, true>(p, xj, yj, xi[r], yi[r], gi[r], j0, n, gx[r], gy[r], e);
                ce2 += (rbase + r < nrows) ? e : 0.0f;
            }
            if ((step & 15) == 15) {
                wave_loss += (double)ce2;
                ce2 = 0.0f;
            }
        }
        wave_loss += (double)ce2;
        wave_loss *= -0.6931471805599453;   // log2 -> -ln
#pragma unroll
        for (int r = 0; r < F_RPW; ++r) {
            for (int o = 32; o > 0; o >>= 1) {
                gx[r] += __shfl_down(gx[r], o);
                gy[r] += __shfl_down(gy[r], o);
            }
            if (lane == 0 && rbase + r < nrows) {
                G[gi[r]] = gx[r];
                G[n + gi[r]] = gy[r];
            }
        }
        for (int o = 32; o > 0; o >>= 1) wave_loss += __shfl_down(wave_loss, o);
    }
    if (lane == 0) wloss[wave] = wave_loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < F_WAVES; ++w) s += wloss[w];
        loss_part[blockIdx.x] = s;
    }
}
// =================================================================================================
// FAST forces, symmetric form (single-GPU, all rows local): every unordered pair {i<j} is evaluated once.
// A wave owns a tile of SY_R rows x 512 columns (lane = 8 consecutive columns); t*(y_i-y_j) goes to the row's sum
// (wave reduction per row) and, negated, to the lane's column accumulators (registers, over the tile's rows).
// Tiles write disjoint slices of two partial buffers -- rowpart[J][2][N] and colpart[I][2][N] -- and a second kernel
// adds the partials in a fixed order: deterministic, no atomics.  Tiles entirely below the diagonal are skipped.
// =================================================================================================
// wave-wide sum by DPP (no LDS): inclusive row scan (row_shr 1,2,4,8), then row_bcast15 / row_bcast31; lane 63 = total
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0x111, 0xF>(v);   // row_shr:1
    v = dpp_add<0x112, 0xF>(v);   // row_shr:2
    v = dpp_add<0x114, 0xF>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF>(v);   // row_shr:8   -> lane 15 of every row holds the row's sum
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
    return v;
}
// =================================================================================================
// FAST forces, symmetric form, second generation (u16 sums + LUT source).  Same partial buffers and loss layout as
// forces_sym_kernel, other work split and instruction stream.  PMC passes of the first kernel (profiles/r02_pmc*.json) showed
// 75 % VALU issue utilisation at ~23 VALU instructions per pair, 55 % of the wave-cycles parked on loads, and only ~2.4 waves
// per wave slot over the whole launch (9.8 k waves of 256 rows x 512 columns on 4096 slots: a long tail).  Here:
//   * a block = one 256-row x 512-column tile, its four waves take one 64-row block each (4x finer work units, 39 k waves);
//     the waves' column-side sums meet in LDS, so the column partials stay one slice per 256-row block;
//   * rows go in groups of S2_G; the 16-byte sums loads of group g+1 are in flight while group g is evaluated (the old kernel
//     loaded, waited, gathered, waited, computed -- per row);
//   * all per-pair arithmetic on column PAIRS as v_pk_{add,mul,fma}_f32 (two pairs per issue slot): with m = d2c (1 + d2c),
//     r = 1/m:  q = r d2c,  q/(1-q) = r + q  (~16 issue slots per pair incl. the two 8-cycle transcendentals, down from ~25);
//   * LUT byte offsets by one SDWA shift per pair (v_lshlrev_b32_sdwa picks the 16-bit half and scales it by 4), the LUT at LDS
//     address 0 so that the shift result is the ds_read address;
//   * the row-side partial sums of a group (gx, gy of S2_G rows) are reduced together: transposed through a per-wave LDS
//     scratch (ds_write per value, one ds_read_b128 per lane, a few adds, DPP row shifts) instead of 2 x 6 DPP steps per row.
// Row blocks that touch the diagonal or the right / bottom edge take the masked generic path (the first kernel's arithmetic).
// =================================================================================================
constexpr int S2_G = 2;                        // rows per group
constexpr int S2_TSTRIDE = 68;                 // dwords between value types in the transpose scratch (64 lanes + 4: conflict-free)
constexpr int S2_SCRATCH = 2 * S2_G * S2_TSTRIDE;   // dwords per wave (2 S2_G value types)
constexpr int S2_CS = 16 * 64;                 // dwords per wave of the column-sum exchange: 16 components x 64 lanes

__device__ __forceinline__ uint32_t lut_off_lo(uint32_t w, uint32_t two) {   // (w & 0xFFFF) << 2
    uint32_t a;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(a) : "s"(two), "v"(w));
    return a;
}
__device__ __forceinline__ uint32_t lut_off_hi(uint32_t w, uint32_t two) {   // (w >> 16) << 2
    uint32_t a;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(a) : "s"(two), "v"(w));
    return a;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// float at an absolute LDS byte address.  The LUT is the kernel's first LDS object and starts at address 0 (checked on entry), so
// the SDWA result IS the ds_read address: no per-pair add of the (relocated, zero) base of the extern __shared__ symbol.
__device__ __forceinline__ float lds_f32_at(uint32_t byte_addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) float *>(byte_addr);
}

__global__ __launch_bounds__(KMAP_WAVE *SY_WAVES, 4) void forces_sym2_kernel(ProbSrc src, const float *__restrict__ Y, int64_t n,
                                                                             float *__restrict__ rowpart, float *__restrict__ colpart,
                                                                             double *__restrict__ loss_part, int64_t nJ, int64_t part_ld,
                                                                             int world, int rank, int lut_pad) {
    static_assert(SY_WAVES == SY_NRB, "one wave per 64-row block of the tile");
    extern __shared__ __attribute__((aligned(16))) float lut_s[];   // [lut_pad] LUT | SY_WAVES transpose scratches | column-sum exchange
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blockIdx.x = column tile J, blockIdx.y = local 256-row block; global row block I (cyclic over the ranks)
    const int64_t Il = blockIdx.y, I = (int64_t)rank + (int64_t)world * Il, J = blockIdx.x;
    if (!sy_tile_live(I, J)) {                                      // tile entirely below the diagonal (block-uniform)
        if (threadIdx.x == 0) loss_part[Il * part_ld + J] = 0.0;
        return;
    }
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)lut_s != 0u) __builtin_trap();   // see lds_f32_at
    for (int t = threadIdx.x; t < src.lut_len && t < F_LUT_LDS; t += blockDim.x) lut_s[t] = src.lut[t];
    __syncthreads();
    float *scratch = lut_s + lut_pad + wave * S2_SCRATCH;
    float *colx = lut_s + lut_pad + SY_WAVES * S2_SCRATCH;          // [wave][component 0..15][lane]
    const float *X = Y, *Yy = Y + n;
    const int64_t j0 = J * SY_C + (int64_t)lane * F_CPL;
    f32x2 cgx[4], cgy[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) cgx[d] = cgy[d] = f32x2{0.0f, 0.0f};
    double wave_loss = 0.0;
    const int64_t r0 = I * SY_R + (int64_t)wave * SY_RB;            // this wave's 64 rows
    if (r0 < n && (J + 1) * SY_C - 1 > r0) {
        f32x2 xj[4], yj[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int64_t ja = (j0 + 2 * d < n) ? j0 + 2 * d : n - 1, jb = (j0 + 2 * d + 1 < n) ? j0 + 2 * d + 1 : n - 1;
            xj[d] = f32x2{X[ja], X[jb]};
            yj[d] = f32x2{Yy[ja], Yy[jb]};
        }
        const int nr = (int)((n - r0 < SY_RB) ? n - r0 : SY_RB);
        const int64_t myrow = (r0 + lane < n) ? r0 + lane : n - 1;
        float xrv = X[myrow], yrv = Yy[myrow];                      // lane r holds row r's coordinates
        const uint16_t *rows = src.ps + (Il * SY_R + (int64_t)wave * SY_RB) * src.ld + j0;   // row r of the block: rows + r * ld
        float ce2 = 0.0f;
        // interior: 64 rows, all left of the tile's first column, all 512 columns exist -> no j > i / j < n tests
        const bool interior = ((J + 1) * SY_C <= n) && (src.ld % 8 == 0) && nr == SY_RB && (J * SY_C > r0 + nr - 1);
        if (interior) {
            const uint32_t two = 2;
            u32x4 wn[S2_G];
#pragma unroll
            for (int a = 0; a < S2_G; ++a) wn[a] = *reinterpret_cast<const u32x4 *>(rows + (int64_t)a * src.ld);
#pragma unroll 1
            for (int g = 0; g < SY_RB / S2_G; ++g) {
                u32x4 wc[S2_G];
#pragma unroll
                for (int a = 0; a < S2_G; ++a) wc[a] = wn[a];
                if (g + 1 < SY_RB / S2_G) {
#pragma unroll
                    for (int a = 0; a < S2_G; ++a)
                        wn[a] = *reinterpret_cast<const u32x4 *>(rows + (int64_t)((g + 1) * S2_G + a) * src.ld);
                }
                float part[2 * S2_G];
                f32x2 es2 = f32x2{0.0f, 0.0f};
#pragma unroll
                for (int a = 0; a < S2_G; ++a) {
                    const int r = g * S2_G + a;
                    float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xrv), r));
                    float yi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yrv), r));
                    const uint32_t ws[4] = {wc[a].x, wc[a].y, wc[a].z, wc[a].w};
                    f32x2 p2[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d)   // LUT gather: one SDWA shift (16-bit half -> byte offset) + one ds_read_b32 per pair
                        p2[d] = f32x2{lds_f32_at(lut_off_lo(ws[d], two)), lds_f32_at(lut_off_hi(ws[d], two))};
                    f32x2 gx2 = f32x2{0.0f, 0.0f}, gy2 = f32x2{0.0f, 0.0f}, pr2 = f32x2{1.0f, 1.0f};
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const f32x2 xi2 = f32x2{xi, xi}, yi2 = f32x2{yi, yi};
                        const f32x2 dx = xi2 - xj[d], dy = yi2 - yj[d];
                        const f32x2 d2 = pk_fma(dx, dx, dy * dy);
                        const f32x2 d2c = f32x2{__builtin_amdgcn_fmed3f(d2.x, FAST_D2_MIN, FAST_D2_MAX),
                                                __builtin_amdgcn_fmed3f(d2.y, FAST_D2_MIN, FAST_D2_MAX)};
                        const f32x2 m = pk_fma(d2c, d2c, d2c);                       // d2c (1 + d2c)
                        const f32x2 rr = f32x2{__builtin_amdgcn_rcpf(m.x), __builtin_amdgcn_rcpf(m.y)};
                        const f32x2 q = rr * d2c;                                    // 1 / (1 + d2c), clipped through d2c
                        const f32x2 u = rr + q;                                      // q / (1 - q) = 1 / d2c
                        const f32x2 t = u * (p2[d] - q);
                        const f32x2 omq = f32x2{1.0f, 1.0f} - q;
                        gx2 = pk_fma(t, dx, gx2);                                    // row side: + t (y_i - y_j)
                        gy2 = pk_fma(t, dy, gy2);
                        cgx[d] = pk_fma(-t, dx, cgx[d]);                             // column side (negation = operand modifier)
                        cgy[d] = pk_fma(-t, dy, cgy[d]);
                        const f32x2 lg = f32x2{__builtin_amdgcn_logf(d2c.x), __builtin_amdgcn_logf(d2c.y)};
                        es2 = pk_fma(p2[d], lg, es2);
                        pr2 = pr2 * omq;
                        // two column pairs at a time: the empty asm ties their results to the inputs of the next two (row
                        // coordinates), so only two of the group's independent chains are interleaved -- enough to fill most
                        // trans / packed-op wait states; left alone the scheduler overlaps all of them and needs > 200 VGPRs.
                        // The LUT gathers above stay free to issue early.
                        if (d & 1)
                            asm volatile("" : "+s"(xi), "+s"(yi), "+v"(xrv), "+v"(yrv), "+v"(gx2), "+v"(gy2), "+v"(es2), "+v"(pr2),
                                         "+v"(cgx[d]), "+v"(cgy[d]), "+v"(cgx[d - 1]), "+v"(cgy[d - 1]));
                    }
                    part[2 * a] = gx2.x + gx2.y;
                    part[2 * a + 1] = gy2.x + gy2.y;
                    ce2 += __builtin_amdgcn_logf(pr2.x * pr2.y);
                }
                ce2 -= es2.x + es2.y;
                // the 2 S2_G partial sums of the group, summed over the wave's 64 lanes: value type t goes to scratch[t][lane],
                // lane L then adds the S2_E entries [L / S2_L][S2_E (L % S2_L) ..] and DPP row shifts finish the groups of S2_L lanes
                constexpr int S2_T = 2 * S2_G, S2_L = 64 / S2_T, S2_E = 64 / S2_L;   // value types; lanes per type; entries per lane
#pragma unroll
                for (int t = 0; t < S2_T; ++t) scratch[t * S2_TSTRIDE + lane] = part[t];
                __builtin_amdgcn_wave_barrier();
                const float *mine = scratch + (lane / S2_L) * S2_TSTRIDE + (lane % S2_L) * S2_E;
                float sred;
                if constexpr (S2_E == 8) {
                    const f32x4 va = *reinterpret_cast<const f32x4 *>(mine), vb = *reinterpret_cast<const f32x4 *>(mine + 4);
                    sred = ((va.x + va.y) + (va.z + va.w)) + ((vb.x + vb.y) + (vb.z + vb.w));
                } else {
                    static_assert(S2_E == 4 || S2_E == 8, "rows per group: 2 or 4");
                    const f32x4 va = *reinterpret_cast<const f32x4 *>(mine);
                    sred = (va.x + va.y) + (va.z + va.w);
                }
                __builtin_amdgcn_wave_barrier();
                sred = dpp_add<0x111, 0xF>(sred);   // row_shr:1
                sred = dpp_add<0x112, 0xF>(sred);   // row_shr:2
                sred = dpp_add<0x114, 0xF>(sred);   // row_shr:4
                if constexpr (S2_L == 16) sred = dpp_add<0x118, 0xF>(sred);   // row_shr:8
                if ((lane % S2_L) == S2_L - 1) {   // the last lane of each group holds the total of its value type
                    const int t = lane / S2_L;                               // row t >> 1 of the group, x / y
                    rowpart[(J * 2 + (t & 1)) * n + r0 + g * S2_G + (t >> 1)] = sred;
                }
                if (((g + 1) * S2_G) % 16 == 0) {
                    wave_loss += (double)ce2;
                    ce2 = 0.0f;
                }
            }
        } else {
            for (int r = 0; r < nr; ++r) {
                const int64_t gi = r0 + r;
                const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xrv), r));
                const float yi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yrv), r));
                float gx = 0.0f, gy = 0.0f;
                if (j0 + F_CPL - 1 > gi && j0 < n) {   // this lane has at least one column right of the diagonal
                    float esum = 0.0f, prod = 1.0f;
#pragma unroll
                    for (int c = 0; c < F_CPL; ++c) {
                        const int64_t j = j0 + c;
                        const float p = (j < n) ? lut_s[rows[(int64_t)r * src.ld + c]] : 0.0f;
                        const float xc = (c & 1) ? xj[c >> 1].y : xj[c >> 1].x, yc = (c & 1) ? yj[c >> 1].y : yj[c >> 1].x;
                        const float dx = xi - xc, dy = yi - yc;
                        float t, omq, d2c;
                        fast_core(dx, dy, p, t, omq, d2c);
                        const float lterm = p * __builtin_amdgcn_logf(d2c);
                        const bool live = (j > gi) && (j < n);
                        t = live ? t : 0.0f;
                        esum += live ? lterm : 0.0f;
                        prod *= live ? omq : 1.0f;
                        gx = __builtin_fmaf(t, dx, gx);
                        gy = __builtin_fmaf(t, dy, gy);
                        if (c & 1) {
                            cgx[c >> 1].y = __builtin_fmaf(-t, dx, cgx[c >> 1].y);
                            cgy[c >> 1].y = __builtin_fmaf(-t, dy, cgy[c >> 1].y);
                        } else {
                            cgx[c >> 1].x = __builtin_fmaf(-t, dx, cgx[c >> 1].x);
                            cgy[c >> 1].x = __builtin_fmaf(-t, dy, cgy[c >> 1].x);
                        }
                    }
                    ce2 += __builtin_amdgcn_logf(prod) - esum;
                }
                gx = wave_sum_to_lane63(gx);
                gy = wave_sum_to_lane63(gy);
                if (lane == 63) {
                    rowpart[(J * 2 + 0) * n + gi] = gx;
                    rowpart[(J * 2 + 1) * n + gi] = gy;
                }
                if ((r & 15) == 15) {
                    wave_loss += (double)ce2;
                    ce2 = 0.0f;
                }
            }
        }
        wave_loss += (double)ce2;
        wave_loss *= -0.6931471805599453;
        for (int o = 32; o > 0; o >>= 1) wave_loss += __shfl_down(wave_loss, o);
    }
    // column side: the four waves' sums over their 64 rows meet in LDS; one slice per 256-row block leaves the CU
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        colx[(wave * 16 + 2 * d) * 64 + lane] = cgx[d].x;
        colx[(wave * 16 + 2 * d + 1) * 64 + lane] = cgx[d].y;
        colx[(wave * 16 + 8 + 2 * d) * 64 + lane] = cgy[d].x;
        colx[(wave * 16 + 8 + 2 * d + 1) * 64 + lane] = cgy[d].y;
    }
    double *wl = reinterpret_cast<double *>(colx + SY_WAVES * S2_CS);
    if (lane == 0) wl[wave] = wave_loss;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = q * 256 + threadIdx.x;              // output element: x / y component, column of the tile
        const int cxy = e >> 9, col = e & 511, comp = cxy * 8 + (col & 7), l = col >> 3;
        float s = colx[(0 * 16 + comp) * 64 + l];
        s += colx[(1 * 16 + comp) * 64 + l];
        s += colx[(2 * 16 + comp) * 64 + l];
        s += colx[(3 * 16 + comp) * 64 + l];
        const int64_t j = J * SY_C + col;
        if (j < n) colpart[(Il * 2 + cxy) * n + j] = s;
    }
    if (threadIdx.x == 0) loss_part[Il * part_ld + J] = ((wl[0] + wl[1]) + (wl[2] + wl[3]));
}

// G[c][i] = sum_J rowpart[J][c][i] (tiles right of i's row block) + sum_I colpart[I][c][i] (row blocks above / at i)
__global__ __launch_bounds__(BLK) void sym_reduce_kernel(const float *__restrict__ rowpart, const float *__restrict__ colpart,
                                                         int64_t n, int64_t n_lblocks, int64_t nJ, int world, int rank,
                                                         float *__restrict__ G) {
    // 8 lanes per output element (a one-lane walk over ~300 partials is latency-bound: 113 us at N = 50 k): lane l adds the
    // partials l, l+8, ... in order, then the 8 lane sums are combined in a fixed butterfly -- deterministic, no atomics.
    // Sharded: row partials exist only for the rows of this rank's blocks, column partials for its local blocks; the ranks'
    // G buffers are then summed by the all-reduce.
    constexpr int SPLIT = 8;
    const int64_t t = ((int64_t)blockIdx.x * BLK + threadIdx.x) / SPLIT;
    const int l = threadIdx.x & (SPLIT - 1);
    float g = 0.0f;
    if (t < 2 * n) {
        const int c = (int)(t / n);
        const int64_t i = t % n;
        const int64_t Ii = i / SY_R, Ji = i / SY_C;
        if (Ii % world == rank)
            for (int64_t J = l; J < nJ; J += SPLIT)
                if (sy_tile_live(Ii, J)) g += rowpart[(J * 2 + c) * n + i];
        for (int64_t b = l; b < n_lblocks; b += SPLIT)
            if (sy_tile_live((int64_t)rank + (int64_t)world * b, Ji)) g += colpart[(b * 2 + c) * n + i];
    }
    g += __shfl_xor(g, 1);
    g += __shfl_xor(g, 2);
    g += __shfl_xor(g, 4);
    if (t < 2 * n && l == 0) G[t] = g;
}
}  // namespace

int kmap_embed_launch_fast_rows(kmap_embed *e, float *G, hipStream_t st) {
    const int nblk = kmap_embed_force_blocks(e);
    const bool lut = e->src.ps != nullptr;
    const size_t lds = lut ? (((size_t)e->src.lut_len * 4 + 15) & ~(size_t)15) : 16;
    if (lut) forces_fast_kernel<true><<<nblk, KMAP_WAVE * F_WAVES, lds, st>>>(e->src, e->Y, e->n, e->row0, e->nrows, G, e->loss_part);
    else forces_fast_kernel<false><<<nblk, KMAP_WAVE * F_WAVES, lds, st>>>(e->src, e->Y, e->n, e->row0, e->nrows, G, e->loss_part);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}

// symmetric tile kernel (u16 sums + LUT source only); reduce_into_G: also add the row / column partials into G (the fused
// single-GPU step does that inside its apply kernel instead)
int kmap_embed_launch_sym(kmap_embed *e, float *G, bool reduce_into_G, hipStream_t st) {
    KMAP_REQUIRE(e->src.ps != nullptr, "embed: the symmetric kernel needs the neighbour-sum + LUT source");
    const size_t lds = ((size_t)e->src.lut_len * 4 + 15) & ~(size_t)15;
    const int lut_pad = (int)(lds / 4);
    const size_t lds2 = lds + ((size_t)SY_WAVES * S2_SCRATCH + (size_t)SY_WAVES * S2_CS) * 4 + SY_WAVES * 8;
    const int64_t part_ld = ((e->symJ + SY_WAVES - 1) / SY_WAVES) * SY_WAVES;   // loss partials: one row of part_ld entries per local row block
    forces_sym2_kernel<<<dim3((unsigned)e->symJ, (unsigned)e->n_lblocks), KMAP_WAVE * SY_WAVES, lds2, st>>>(
        e->src, e->Y, e->n, e->rowpart, e->colpart, e->loss_part, e->symJ, part_ld, e->world, e->rank, lut_pad);
    if (reduce_into_G)
        sym_reduce_kernel<<<(unsigned)((2 * e->n * 8 + BLK - 1) / BLK), BLK, 0, st>>>(e->rowpart, e->colpart, e->n, e->n_lblocks, e->symJ, e->world, e->rank, G);
    KMAP_CHECK_HIP(hipGetLastError());
    return KMAP_OK;
}
