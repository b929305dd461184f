// Cross-wavefront hand-over latency inside one workgroup on gfx950: the cost that decides whether a POA row can be
// split over several waves. A workgroup of W waves (W = 2, 4) runs a dependent chain:
// every iteration each wave writes one LDS word, the group synchronises, each wave reads its left neighbour's word
// and feeds it into the next iteration. Three synchronisation flavours:
//   barrier : ds_write ; s_waitcnt ; s_barrier ; ds_read ; s_waitcnt
//   flag    : producer writes value + sequence number, consumer spins on the sequence number (no s_barrier)
//   chain   : only wave w waits for wave w-1 (pipeline hand-over, what a row split needs for the scan carry)
// Build: hipcc --offload-arch=gfx950 -O2 tools/microbench_handover.hip -o tools/bin/microbench_handover
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int W>
__global__ __launch_bounds__(64 * W) void handover(uint64_t* out, int iters)
{
    __shared__ volatile uint32_t val[W], seq[W];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { val[wave] = wave; seq[wave] = 0; }
    __syncthreads();
    uint32_t x = wave;
    // --- barrier flavour
    uint64_t t0 = clock64();
    for (int it = 0; it < iters; it++)
    {
        if (lane == 0) val[wave] = x + it;
        __syncthreads();
        x = val[(wave + W - 1) % W];
        __syncthreads();
    }
    const uint64_t t_barrier = clock64() - t0;
    __syncthreads();
    // --- flag flavour: all waves exchange, spinning on the neighbour's sequence number
    t0 = clock64();
    for (int it = 1; it <= iters; it++)
    {
        if (lane == 0) { val[wave] = x + it; __threadfence_block(); seq[wave] = it; }
        const int left = (wave + W - 1) % W;
        while (seq[left] < (uint32_t)it) {}
        x = val[left];
    }
    const uint64_t t_flag = clock64() - t0;
    __syncthreads();
    if (lane == 0) seq[wave] = 0;
    __syncthreads();
    // --- chain flavour: wave w consumes wave w-1's value of the same iteration (pipeline across the waves)
    t0 = clock64();
    for (int it = 1; it <= iters; it++)
    {
        if (wave > 0)
        {
            while (seq[wave - 1] < (uint32_t)it) {}
            x += val[wave - 1];
        }
        if (lane == 0) { val[wave] = x; __threadfence_block(); seq[wave] = it; }
    }
    const uint64_t t_chain = clock64() - t0;
    if (lane == 0)
    {
        uint64_t* o = out + ((size_t)blockIdx.x * W + wave) * 4;
        o[0] = t_barrier; o[1] = t_flag; o[2] = t_chain; o[3] = x;
    }
}

template <int W> void run(int blocks, int iters)
{
    uint64_t* d;
    (void)hipMalloc(&d, sizeof(uint64_t) * 4 * W * blocks);
    hipLaunchKernelGGL(handover<W>, dim3(blocks), dim3(64 * W), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> h((size_t)4 * W * blocks);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s[3] = {0, 0, 0};
    for (size_t i = 0; i < (size_t)W * blocks; i++)
        for (int k = 0; k < 3; k++) s[k] += (double)h[i * 4 + k];
    printf("W=%d waves/group, %4d groups: cycles per iteration  barrier(2x) %7.1f   flag exchange %7.1f   chain of %d %7.1f\n", W, blocks,
           s[0] / (W * blocks) / iters, s[1] / (W * blocks) / iters, W, s[2] / (W * blocks) / iters);
    (void)hipFree(d);
}

int main()
{
    const int iters = 2000;
    for (int blocks : {1, 256, 1024})
    {
        run<2>(blocks, iters);
        run<4>(blocks, iters);
    }
    return 0;
}
