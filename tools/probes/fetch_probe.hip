// fetch_probe.hip -- known-bytes reads for calibrating rocprofv3's FETCH_SIZE on gfx950: each kernel reads
// every byte of a 2-GiB buffer exactly once (far beyond L2 + MALL), with a different access shape.  Run under
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- ./fetch_probe
// and compare the counter (KB) with 2 GiB = 2097152 KB per launch (tools/probes/fetch_probe_summary.py).
//   hipcc --offload-arch=gfx950 -O3 -o fetch_probe fetch_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr size_t BYTES = (size_t)2 << 30;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 16 B per lane, consecutive lanes consecutive (the shape the guide's x2 correction was calibrated on)
__global__ __launch_bounds__(256) void read_b128(const u32x4 *__restrict__ p, size_t n16, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const u32x4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// 4 B per lane, consecutive
__global__ __launch_bounds__(256) void read_b32(const uint32_t *__restrict__ p, size_t n4, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 0x12345678u) *sink = acc;
}
// 1 B per lane, consecutive (64 B per wave instruction)
__global__ __launch_bounds__(256) void read_b8(const uint8_t *__restrict__ p, size_t n1, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n1; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 0x12345678u) *sink = acc;
}
// thread-per-record: every lane walks its own 32-byte record with 4-byte loads (lane stride 32 B: the per-read scan passes' shape)
__global__ __launch_bounds__(256) void read_rec32(const uint32_t *__restrict__ p, size_t n_rec, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < n_rec; r += (size_t)gridDim.x * 256) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += p[r * 8 + j];
    }
    if (acc == 0x12345678u) *sink = acc;
}
// one 4-byte load per 32-byte sector (reads 1/8 of the bytes, touches every sector): what a sector-granular fetch costs
__global__ __launch_bounds__(256) void read_sector_touch(const uint32_t *__restrict__ p, size_t n_sec, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < n_sec; r += (size_t)gridDim.x * 256) acc += p[r * 8];
    if (acc == 0x12345678u) *sink = acc;
}
// one 4-byte load per 128-byte line
__global__ __launch_bounds__(256) void read_line_touch(const uint32_t *__restrict__ p, size_t n_line, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < n_line; r += (size_t)gridDim.x * 256) acc += p[r * 32];
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    void *buf;
    uint32_t *sink;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, BYTES);
    hipDeviceSynchronize();
    const unsigned grid = 256 * 16;
    for (int rep = 0; rep < 2; ++rep) {
        read_b128<<<grid, 256>>>((const u32x4 *)buf, BYTES / 16, sink);
        read_b32<<<grid, 256>>>((const uint32_t *)buf, BYTES / 4, sink);
        read_b8<<<grid, 256>>>((const uint8_t *)buf, BYTES, sink);
        read_rec32<<<grid, 256>>>((const uint32_t *)buf, BYTES / 32, sink);
        read_sector_touch<<<grid, 256>>>((const uint32_t *)buf, BYTES / 32, sink);
        read_line_touch<<<grid, 256>>>((const uint32_t *)buf, BYTES / 128, sink);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
