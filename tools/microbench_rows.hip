// microbench_rows.hip -- what a DP row of the metric kernel's forward pass costs, measured on the production routine itself
// (banded_forward_moves, poa_forward_moves.h) in the production launch geometry: 1024 blocks of one wavefront, 39 KB of LDS
// each (four per CU, one per SIMD), every block writing to its own 2.4 MB slab. The row tables are synthetic so that a run
// consists of ONE row kind (or a stated mix), and GWHIP_DEBUG-style ablation bits remove one piece of the row at a time.
// Results are garbage by construction; only the clock is read.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/microbench_rows.hip -o tools/bin/microbench_rows
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../genomeworks_amd/csrc/poa_device.h"

using namespace gwhip;

constexpr int kRingBytesMb = 8448, kRowInfoRows = 3074, kReadBytes = 2048, kTileBytes = 4096;
constexpr size_t kLdsBytes = kRingBytesMb + kRowInfoRows * 8 + kReadBytes + kTileBytes;

struct MbArgs
{
    int32_t pattern, graph_count, read_length, reps, dbg;
    uint8_t* slabs;
    size_t per_block;
    unsigned long long* cycles;
};

// band start of row r in the synthetic tables: 0 for the first 16 rows, then (pattern-dependent) fixed or moving
__device__ int32_t mb_band_start(int32_t pattern, int32_t r, int32_t max_column)
{
    if (r <= 16) return 0;
    int32_t bs = 4;
    if (pattern == 3) bs = 4 * (r - 16);                 // moves every row
    if (pattern == 4 || pattern == 5) bs = 4 * ((r - 16) / 5 + 1); // moves every fifth row (gradient ~0.8)
    return min(bs, ((max_column - 252) / 4) * 4);
}

__global__ __launch_bounds__(kWave) void rows_kernel(MbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane            = threadIdx.x;
    uint8_t* ring             = smem;
    RowInfo<true>* rowinfo    = reinterpret_cast<RowInfo<true>*>(smem + kRingBytesMb);
    uint8_t* lds_read         = smem + kRingBytesMb + kRowInfoRows * 8;
    uint64_t* xpred           = reinterpret_cast<uint64_t*>(lds_read + kReadBytes);
    uint8_t* slab             = a.slabs + (size_t)blockIdx.x * a.per_block;
    int16_t* scores           = reinterpret_cast<int16_t*>(slab);
    uint8_t* moves            = slab + (size_t)3072 * 264 * 2;
    const int32_t max_column  = a.read_length + 1;
    for (int32_t i = lane; i < kReadBytes; i += kWave) lds_read[i] = "ACGT"[(i * 7 + (i >> 3)) & 3];
    for (int32_t i = lane; i < 256; i += kWave) xpred[i] = 0;
    if (lane == 0) rowinfo[0].set(0, 0, false, 0, 0, 0);
    for (int32_t r = 1 + lane; r <= a.graph_count; r += kWave)
    {
        int32_t cnt = 1, p0 = r - 1, p1 = 0, p2 = 0;
        const int32_t m = r & 15;
        switch (a.pattern)
        {
        case 1: cnt = r >= 2 ? 2 : 1; p1 = r - 2; break;                    // two predecessors from the ring
        case 2: p0 = max(r - 2, 0); break;                                  // one predecessor two rows up
        case 5:                                                             // mix: 7/16 kind 0-1, 3/16 kind 2, 5/16 two, 1/16 three predecessors
            if (m >= 7 && m <= 9) p0 = max(r - 2, 0);
            else if (m >= 10 && m <= 14) { cnt = r >= 2 ? 2 : 1; p1 = r - 2; }
            else if (m == 15) { cnt = r >= 3 ? 3 : 1; p1 = r - 2; p2 = r - 3; }
            break;
        default: break;                                                     // 0, 3, 4: the previous row
        }
        RowInfo<true> ri;
        ri.set("ACGT"[(r * 5 + (r >> 4)) & 3], cnt, r == a.graph_count, p0, p1, p2);
        ri.set_bs(mb_band_start(a.pattern, r, max_column));
        rowinfo[r] = ri;
    }
    GraphView<int16_t> g{};
    wave_sync();
    const unsigned long long t0 = clock64();
    for (int32_t rep = 0; rep < a.reps; rep++)
    {
        banded_forward_moves<int16_t, 256>(g, rowinfo, a.graph_count, lds_read, scores, moves, ring, xpred, max_column, -8, -6, 8, a.dbg, nullptr);
        wave_sync();
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) a.cycles[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 1024;
    const int reps   = argc > 2 ? atoi(argv[2]) : 8;
    MbArgs a{};
    a.graph_count = 1400;
    a.read_length = 1000;
    a.reps        = reps;
    a.per_block   = (size_t)3072 * 264 * 3 + 4096;
    if (hipMalloc(&a.slabs, a.per_block * blocks) != hipSuccess || hipMalloc(&a.cycles, sizeof(unsigned long long) * blocks) != hipSuccess)
    {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    hipMemset(a.slabs, 0, a.per_block * blocks);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    const char* pattern_name[] = {"kind 0 (previous row, band fixed)", "kind 3 (rows r-1 and r-2 from the ring)", "kind 2 (row r-2 from the ring)",
                                  "kind 1 (previous row, band moves every row)", "kinds 0+1 (band moves every 5th row)",
                                  "mix 7/16 kinds 0-1, 3/16 kind 2, 6/16 kind 3"};
    struct Abl { const char* name; int bits; };
    const Abl abl[] = {{"full row", 0},
                       {"no score-row store", 1 << 26},
                       {"no move-row store", 1 << 27},
                       {"no HBM stores", (1 << 26) | (1 << 27)},
                       {"no ring write", 1 << 20},
                       {"no guard write", 1 << 19},
                       {"no LDS writes", (1 << 20) | (1 << 19)},
                       {"no stores at all", (1 << 26) | (1 << 27) | (1 << 20) | (1 << 19)},
                       {"no cross-lane scan", 1 << 18},
                       {"no move bytes", 1 << 17},
                       {"no stores, no scan, no move bytes", (1 << 26) | (1 << 27) | (1 << 20) | (1 << 19) | (1 << 18) | (1 << 17)},
                       {"no rows (classification + setup only)", 1 << 16}};
    std::vector<unsigned long long> h(blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("{\"blocks\": %d, \"rows_per_pass\": %d, \"passes\": %d, \"results\": [\n", blocks, a.graph_count, reps);
    bool first = true;
    for (int p = 0; p < 6; p++)
        for (const Abl& ab : abl)
        {
            if (p != 0 && p != 5 && ab.bits != 0 && ab.bits != ((1 << 26) | (1 << 27)) && ab.bits != (1 << 16)) continue; // full ablation table for kind 0 and the mix
            a.pattern = p;
            a.dbg     = ab.bits ? (ab.bits | (1 << 14)) : 0; // (bit 14 enables the ablation bits, poa_forward_moves.h)
            double best_ms = 1e30, cyc = 0;
            for (int it = 0; it < 3; it++)
            {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(rows_kernel, dim3(blocks), dim3(kWave), kLdsBytes, 0, a);
                hipEventRecord(e1, 0);
                if (hipEventSynchronize(e1) != hipSuccess) { fprintf(stderr, "launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best_ms)
                {
                    best_ms = ms;
                    hipMemcpy(h.data(), a.cycles, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
                    double s = 0;
                    for (int b = 0; b < blocks; b++) s += (double)h[b];
                    cyc = s / blocks / ((double)reps * a.graph_count);
                }
            }
            printf("%s {\"rows\": \"%s\", \"variant\": \"%s\", \"cycles_per_row\": %.1f, \"kernel_ms\": %.3f}", first ? " " : ",\n ", pattern_name[p], ab.name, cyc, best_ms);
            first = false;
        }
    printf("\n]}\n");
    return 0;
}
