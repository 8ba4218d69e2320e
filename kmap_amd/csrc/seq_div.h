// seq_div.h -- the two f32 divisions of the reference's gradient, as short instruction sequences that are EXACT (= IEEE
// round-to-nearest division, bit for bit) on the operand ranges the SEQ force kernel meets them on; shared by embed.hip (the
// kernel) and selftest.hip (the exhaustive bit test behind tests/test_gpu_embed.py::test_seq_divisions_exhaustive).
//
//   q = 1 / (1 + d2)            taichi_core.py:255      s1 = 1 + d2 in [1, 2^100): only the value of clip(q, 1e-3, 0.999) matters
//   u = q / (1 - q)             visualization.py:132    q in [1e-3, 0.999] (clipped), so u is a function of ONE float
//
// hipcc's generic f32 division is v_div_scale x2, v_rcp, 2 FMAs on the reciprocal, mul, 3 FMAs on the quotient, v_div_fmas,
// v_div_fixup = 11 instructions; without the scale / fixup wrappers (exact when nothing needs rescaling) 8.  Every variant below
// can be checked over ALL operand bit patterns of its range on the device (kmap_selftest_seq_div).
#pragma once

// Result of the exhaustive scan (tools/seq_div_scan.py, MI355X, profiles/r03_seq_div_scan.txt):
//   clip(1 / s1): v_rcp_f32 alone differs on 8 952 101 of the 8.3e8 operands; with ONE Newton step (3 instructions) on none.
//   q / (1 - q):  v_rcp_f32 * q differs on 29 541 946 of the 8.2e7 operands; with ONE residual correction (4 instructions) on none
//                 (refining the reciprocal first does not help without the correction: 28 198 039 differ).
// So 3 + 4 = 7 instructions replace 16.
#define KMAP_SEQ_RCP_STEPS 1      // Newton steps on v_rcp_f32 for 1 / s1 (each: 2 FMAs)
#define KMAP_SEQ_QUO_RSTEPS 0     // Newton steps on the reciprocal used by the quotient
#define KMAP_SEQ_QUO_STEPS 1      // residual corrections on q * r for q / (1 - q) (each: 2 FMAs)

template <int STEPS>
__device__ __forceinline__ float seq_rcp(float b) {                  // 1 / b
    float r = __builtin_amdgcn_rcpf(b);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const float e = __builtin_fmaf(-b, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
    }
    return r;
}
template <int RSTEPS, int QSTEPS>
__device__ __forceinline__ float seq_quo(float a, float b) {         // a / b
    const float r = seq_rcp<RSTEPS>(b);
    float q = a * r;
#pragma unroll
    for (int s = 0; s < QSTEPS; ++s) {
        const float m = __builtin_fmaf(-b, q, a);
        q = __builtin_fmaf(m, r, q);
    }
    return q;
}
