// microbench_placement.hip -- where do one-wave blocks with the window kernel's LDS footprint (39 184 B: four per CU) land, SIMD by
// SIMD, and how long does a launch of them take, when a second stream is busy at the same time? (Round 6: two 1024-window
// launches on two streams took 143 ms where a 2048-window launch took 103, and the FIRST of two launches took 98 ms alone.)
// Every block runs a fixed chain of dependent VALU operations (so it takes twice as long when it shares its SIMD), and records
// its HW_ID (XCC, SE, CU, SIMD) and its start / end clock. Scenarios:
//   alone        1024 blocks on stream A
//   fill_before  small fill kernels on stream B, then 1024 blocks on A while they run
//   fill_during  1024 blocks on A, small fill kernels on B 2 ms later
//   two_streams  1024 blocks on A and 1024 on B
//   one_launch   2048 blocks on A
// Output: one JSON line per scenario: kernel ms (events), distribution of blocks per SIMD at the kernel's midpoint.
//   build: hipcc --offload-arch=gfx950 -O2 tools/microbench_placement.hip -o tools/bin/microbench_placement
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <map>
#include <thread>
#include <vector>

#define CHECK(x)                                                                                   \
    do                                                                                             \
    {                                                                                              \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess)                                                                      \
        {                                                                                          \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));         \
            std::exit(1);                                                                          \
        }                                                                                          \
    } while (0)

struct Rec
{
    uint32_t hw_id, xcc_id;
    uint64_t t0, t1;
};

__global__ __launch_bounds__(64) void spin(Rec* out, int iters, int base)
{
    extern __shared__ uint8_t lds[];
    uint32_t hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint64_t t0 = wall_clock64();
    uint32_t x        = threadIdx.x;
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int k = 0; k < 64; k++) asm volatile("v_add_u32_e32 %0, %0, %0" : "+v"(x));
    }
    lds[threadIdx.x] = (uint8_t)x;
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) out[base + blockIdx.x] = Rec{hw, xcc, t0, t1};
}

__global__ void fill(uint32_t* p, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

static void report(const char* name, const std::vector<Rec>& recs, float ms_a, float ms_b)
{
    uint64_t lo = ~0ull, hi = 0;
    for (const Rec& r : recs)
    {
        lo = std::min(lo, r.t0);
        hi = std::max(hi, r.t1);
    }
    // blocks per SIMD: total over the launch, and resident together at 25 % of the span
    const uint64_t probe = lo + (hi - lo) / 4;
    std::map<uint32_t, int> total, live;
    double mean = 0, longest = 0;
    for (const Rec& r : recs)
    {
        // HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
        const uint32_t key = (r.xcc_id & 0xf) << 16 | (r.hw_id & 0xff30u);
        total[key]++;
        if (r.t0 <= probe && probe < r.t1) live[key]++;
        const double d = (double)(r.t1 - r.t0) / 1e5; // wall clock: 100 MHz -> ms
        mean += d;
        longest = std::max(longest, d);
    }
    int hist[8] = {0};
    for (auto& kv : live) hist[std::min(kv.second, 7)]++;
    std::printf("{\"scenario\": \"%s\", \"blocks\": %zu, \"kernel_ms\": [%.2f, %.2f], \"span_ms\": %.2f, \"block_ms_mean\": %.2f, \"block_ms_max\": %.2f, "
                "\"simds_seen\": %zu, \"simds_with_n_blocks_at_quarter_span\": {\"1\": %d, \"2\": %d, \"3\": %d, \"4+\": %d}}\n",
                name, recs.size(), ms_a, ms_b, (double)(hi - lo) / 1e5, mean / recs.size(), longest, total.size(), hist[1], hist[2], hist[3],
                hist[4] + hist[5] + hist[6] + hist[7]);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? std::atoi(argv[1]) : 300000; // ~ 300000 x 64 x 4 cycles = 77 M cycles = ~32 ms
    const size_t lds = 39184;
    hipStream_t a, b;
    CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    Rec* d_rec;
    CHECK(hipMalloc(&d_rec, sizeof(Rec) * 4096));
    uint32_t* d_fill;
    CHECK(hipMalloc(&d_fill, 64 << 20));
    hipEvent_t e[4];
    for (auto& x : e) CHECK(hipEventCreate(&x));
    std::vector<Rec> h(4096);
    auto fills = [&](hipStream_t s, int n) {
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(fill, dim3(65536), dim3(256), 0, s, d_fill, 16 << 20);
    };
    auto run = [&](const char* name, int blocks_a, int blocks_b, int fill_mode) {
        CHECK(hipDeviceSynchronize());
        if (fill_mode == 1) fills(b, 40);
        CHECK(hipEventRecord(e[0], a));
        hipLaunchKernelGGL(spin, dim3(blocks_a), dim3(64), lds, a, d_rec, iters, 0);
        CHECK(hipEventRecord(e[1], a));
        if (fill_mode == 2)
        {
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            fills(b, 40);
        }
        float ms_b = 0;
        if (blocks_b > 0)
        {
            CHECK(hipEventRecord(e[2], b));
            hipLaunchKernelGGL(spin, dim3(blocks_b), dim3(64), lds, b, d_rec, iters, blocks_a);
            CHECK(hipEventRecord(e[3], b));
        }
        CHECK(hipDeviceSynchronize());
        float ms_a = 0;
        CHECK(hipEventElapsedTime(&ms_a, e[0], e[1]));
        if (blocks_b > 0) CHECK(hipEventElapsedTime(&ms_b, e[2], e[3]));
        CHECK(hipMemcpy(h.data(), d_rec, sizeof(Rec) * (blocks_a + blocks_b), hipMemcpyDeviceToHost));
        report(name, std::vector<Rec>(h.begin(), h.begin() + blocks_a + blocks_b), ms_a, ms_b);
    };
    run("warmup", 1024, 0, 0);
    run("alone", 1024, 0, 0);
    run("fill_before", 1024, 0, 1);
    run("fill_during", 1024, 0, 2);
    run("two_streams", 1024, 1024, 0);
    run("one_launch_2048", 2048, 0, 0);
    run("three_streams_740_740_568", 740, 740, 0);
    run("alone_again", 1024, 0, 0);
    return 0;
}
