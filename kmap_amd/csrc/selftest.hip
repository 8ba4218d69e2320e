// selftest.hip -- exhaustive device-side bit tests of arithmetic shortcuts the kernels rely on (test support; no reference
// operator corresponds to it).  kmap_selftest_seq_div: for EVERY float bit pattern in [lo_bits, hi_bits], compare a candidate
// division sequence of seq_div.h with the compiler's IEEE division.
#include "common.h"
#include "seq_div.h"

namespace {

template <int RS>
__device__ __forceinline__ bool rcp_ok(float s1) {
    // what the kernel uses of q: the clipped value
    const float want = __builtin_amdgcn_fmed3f(1.0f / s1, 0.001f, 0.999f);
    const float got = __builtin_amdgcn_fmed3f(seq_rcp<RS>(s1), 0.001f, 0.999f);
    return __float_as_uint(want) == __float_as_uint(got);
}
template <int RS, int QS>
__device__ __forceinline__ bool quo_ok(float q) {
    const float omq = 1.0f - q;
    return __float_as_uint(q / omq) == __float_as_uint(seq_quo<RS, QS>(q, omq));
}

__global__ __launch_bounds__(256) void seq_div_test_kernel(int which, int a, int b, uint32_t lo, uint64_t count,
                                                           unsigned long long *__restrict__ bad, uint32_t *__restrict__ first_bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
        const uint32_t bits = lo + (uint32_t)t;
        const float v = __uint_as_float(bits);
        bool ok = true;
        if (which == 0) {
            ok = a == 0 ? rcp_ok<0>(v) : a == 1 ? rcp_ok<1>(v) : a == 2 ? rcp_ok<2>(v) : rcp_ok<3>(v);
        } else {
            if (a == 0) ok = b == 0 ? quo_ok<0, 0>(v) : b == 1 ? quo_ok<0, 1>(v) : b == 2 ? quo_ok<0, 2>(v) : quo_ok<0, 3>(v);
            else if (a == 1) ok = b == 0 ? quo_ok<1, 0>(v) : b == 1 ? quo_ok<1, 1>(v) : b == 2 ? quo_ok<1, 2>(v) : quo_ok<1, 3>(v);
            else ok = b == 0 ? quo_ok<2, 0>(v) : b == 1 ? quo_ok<2, 1>(v) : b == 2 ? quo_ok<2, 2>(v) : quo_ok<2, 3>(v);
        }
        if (!ok) {
            ++mine;
            atomicMin(first_bad, bits);
        }
    }
    if (mine) atomicAdd(bad, mine);
}

}  // namespace

extern "C" {

int kmap_selftest_seq_div(int which, int rcp_steps, int quo_steps, uint32_t lo_bits, uint32_t hi_bits, uint64_t *n_bad,
                          uint32_t *first_bad_bits) {
    KMAP_REQUIRE(which == 0 || which == 1, "selftest_seq_div: which must be 0 (1 / s1) or 1 (q / (1 - q))");
    KMAP_REQUIRE(rcp_steps >= 0 && rcp_steps <= 3 && quo_steps >= 0 && quo_steps <= 3 && (which == 0 || rcp_steps <= 2), "selftest_seq_div: steps out of range");
    KMAP_REQUIRE(hi_bits >= lo_bits && n_bad, "selftest_seq_div: bad range");
    DevBuf d;
    KMAP_TRY(d.alloc(16));
    unsigned long long zero = 0;
    uint32_t ones = 0xFFFFFFFFu;
    KMAP_CHECK_HIP(hipMemcpy(d.p, &zero, 8, hipMemcpyHostToDevice));
    KMAP_CHECK_HIP(hipMemcpy((char *)d.p + 8, &ones, 4, hipMemcpyHostToDevice));
    const uint64_t count = (uint64_t)hi_bits - lo_bits + 1;
    seq_div_test_kernel<<<4096, 256>>>(which, rcp_steps, quo_steps, lo_bits, count, (unsigned long long *)d.p, (uint32_t *)((char *)d.p + 8));
    KMAP_CHECK_HIP(hipGetLastError());
    unsigned long long bad = 0;
    uint32_t fb = 0;
    KMAP_CHECK_HIP(hipMemcpy(&bad, d.p, 8, hipMemcpyDeviceToHost));
    KMAP_CHECK_HIP(hipMemcpy(&fb, (char *)d.p + 8, 4, hipMemcpyDeviceToHost));
    *n_bad = bad;
    if (first_bad_bits) *first_bad_bits = fb;
    return KMAP_OK;
}

}  // extern "C"
