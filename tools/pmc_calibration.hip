// pmc_calibration.hip -- what do FETCH_SIZE / WRITE_SIZE report for the access patterns of the POA kernels? Three kernels of
// KNOWN byte counts, run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/pmc_passes.sh calibrates with them):
//   cal_stream_write   one wavefront per block writes rows of 512 B (8 B per lane) at a 528-B pitch plus 256-B rows (4 B per
//                      lane) at a 264-B pitch into its own 2.4 MB slab: the score / move rows of the forward pass
//   cal_scatter_read   every lane reads 4 B, 8 B and 16 B records at pseudo-random 16-B-aligned offsets of a 4 GB buffer: the
//                      graph / tile loads of merge, topsort, row table and traceback
//   cal_stream_read    16 B per lane, coalesced: the pattern the microarchitecture guide's x2 FETCH correction was measured on
// Prints the true byte counts as JSON; the factor is true bytes / (counter x 1024).
//   build: hipcc --offload-arch=gfx950 -O2 tools/pmc_calibration.hip -o tools/bin/pmc_calibration
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(64) void cal_stream_write(uint8_t* slabs, size_t per_block, int rows)
{
    uint8_t* s    = slabs + (size_t)blockIdx.x * per_block;
    uint8_t* mv   = s + (size_t)3072 * 528;
    const int lane = threadIdx.x;
    for (int r = 1; r <= rows; r++)
    {
        *reinterpret_cast<uint2*>(s + (size_t)r * 528 + 8 + lane * 8)     = make_uint2(r, lane);
        *reinterpret_cast<uint32_t*>(mv + (size_t)r * 264 + 4 + lane * 4) = (uint32_t)(r ^ lane);
    }
}

__global__ __launch_bounds__(64) void cal_scatter_read(const uint8_t* buf, size_t bytes, int iters, uint32_t* sink)
{
    uint32_t x   = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    const size_t slots = bytes / 16;
    for (int it = 0; it < iters; it++)
    {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const uint8_t* p = buf + (size_t)(x % slots) * 16;
        acc += *reinterpret_cast<const uint32_t*>(p);                 // 4 B
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        p = buf + (size_t)(x % slots) * 16;
        const uint2 b = *reinterpret_cast<const uint2*>(p);           // 8 B
        acc += b.x + b.y;
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        p = buf + (size_t)(x % slots) * 16;
        const uint4 c = *reinterpret_cast<const uint4*>(p);           // 16 B
        acc += c.x + c.y + c.z + c.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void cal_stream_read(const uint4* buf, size_t n, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        const uint4 v = buf[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const int blocks = 1024, rows = 3000, iters = 4096;
    const size_t per_block = (size_t)3072 * 528 + (size_t)3072 * 264 + 4096;
    const size_t big       = (size_t)4 << 30;
    uint8_t *slabs = nullptr, *buf = nullptr;
    uint32_t* sink = nullptr;
    if (hipMalloc(&slabs, per_block * blocks) != hipSuccess || hipMalloc(&buf, big) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, big);
    (void)hipMemset(slabs, 0, per_block * blocks);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(cal_stream_write, dim3(blocks), dim3(64), 0, 0, slabs, per_block, rows);
    hipLaunchKernelGGL(cal_scatter_read, dim3(blocks * 4), dim3(64), 0, 0, buf, big, iters, sink);
    hipLaunchKernelGGL(cal_stream_read, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const uint4*>(buf), big / 16, sink);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    const double w  = (double)blocks * rows * (512 + 256);
    const double sr = (double)blocks * 4 * 64 * iters * (4 + 8 + 16);
    printf("{\"cal_stream_write\": {\"true_bytes_written\": %.0f, \"pattern\": \"512 B rows at 528 B pitch + 256 B rows at 264 B pitch, 1024 slabs\"},\n"
           " \"cal_scatter_read\": {\"true_bytes_read\": %.0f, \"requests\": %.0f, \"pattern\": \"4 / 8 / 16 B records at random 16-B-aligned offsets of 4 GB\"},\n"
           " \"cal_stream_read\": {\"true_bytes_read\": %.0f, \"pattern\": \"16 B per lane, coalesced, 4 GB\"}}\n",
           w, sr, (double)blocks * 4 * 64 * iters * 3, (double)big);
    return 0;
}
