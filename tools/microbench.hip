// Lone-wavefront latency calibration for gfx950 (one 64-lane wave per SIMD, like the POA window kernel).
// Prints cycles per operation for dependent chains of the instruction kinds the serial POA phases are made of.
// Build: hipcc --offload-arch=gfx950 -O2 tools/microbench.hip -o gpurun_out/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

enum { T_VALU_DEP, T_VALU_IND, T_SALU_DEP, T_LDS_CHASE, T_LDS_WR_RD, T_BRANCH_TAKEN, T_BRANCH_NOT, T_RFL, T_DPP, T_GLD_CHASE,
       T_GST_WAIT, T_MEMTIME, T_VALU_SALU_MIX, T_LDS_RD_INDEP, T_SBRANCH_LOOP, T_VCC_BRANCH, T_COUNT };
static const char* kNames[T_COUNT] = {"valu_dep(v_add)", "valu_indep(4 chains)", "salu_dep(s_add)", "lds_pointer_chase", "lds_write_then_read",
                                      "uniform_branch_taken", "uniform_branch_not_taken", "valu->readfirstlane->valu", "dpp_dep(row_shr)",
                                      "global_load_chase(L2)", "global_store+vmcnt0", "s_memtime pair", "valu+salu alternating", "lds_read x4 indep + wait",
                                      "s_cmp+s_cbranch loop iter", "v_cmp+vcc branch iter"};

__global__ __launch_bounds__(64) void bench(uint64_t* out, uint32_t* gbuf, int iters)
{
    __shared__ uint32_t lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = ((i * 97 + 13) & 1023) * 4; // byte offsets of a pseudo-random cycle
    __syncthreads();
    uint32_t* g = gbuf + (size_t)blockIdx.x * 4096;
    uint64_t res[T_COUNT];
    uint32_t x = lane, y = lane + 1, z = lane + 2, w = lane + 3;
    uint32_t s = iters;
    // ---- dependent VALU
    {
        uint64_t t0 = clock64();
        for (int it = 0; it < iters; it++) { asm volatile(REP64("v_add_u32 %0, %0, %0\n") : "+v"(x)); }
        res[T_VALU_DEP] = clock64() - t0;
    }
    {
        uint64_t t0 = clock64();
        for (int it = 0; it < iters; it++)
        {
            asm volatile(REP16("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n")
                         : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
        }
        res[T_VALU_IND] = clock64() - t0;
    }
    {
        uint64_t t0 = clock64();
        for (int it = 0; it < iters; it++) { asm volatile(REP64("s_add_u32 %0, %0, %0\n") : "+s"(s)::"scc"); }
        res[T_SALU_DEP] = clock64() - t0;
    }
    {
        uint32_t a = (lane & 1023) * 4;
        uint64_t t0 = clock64();
        for (int it = 0; it < iters; it++)
        {
            asm volatile(REP64("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(a)::"memory");
        }
        res[T_LDS_CHASE] = clock64() - t0;
        x += a;
    }
    {
        uint32_t a = 8192 + lane * 4, v = x;
        uint64_t t0 = clock64();
        for (int it = 0; it < iters; it++)
        {
            asm volatile(REP64("ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(v) : "v"(a) : "memory");
        }
        res[T_LDS_WR_RD] = clock64() - t0;
        x += v;
    }
    {
        // uniform taken branches: 64 forward branches per iteration
        uint64_t t0 = clock64();
        for (int it = 0; it < iters; it++)
        {
            asm volatile(REP64("s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n") ::: "scc");
        }
        res[T_BRANCH_TAKEN] = clock64() - t0;
    }
    {
        uint64_t t0 = clock64();
        for (int it = 0; it < iters; it++)
        {
            asm volatile(REP64("s_cmp_eq_u32 0, 1\n s_cbranch_scc1 1f\n s_nop 0\n1:\n") ::: "scc");
        }
        res[T_BRANCH_NOT] = clock64() - t0;
    }
    {
        uint32_t t;
        uint64_t t0 = clock64();
        for (int it = 0; it < iters; it++)
        {
            asm volatile(REP64("v_readfirstlane_b32 %1, %0\n v_add_u32 %0, %1, %0\n") : "+v"(x), "=s"(t));
        }
        res[T_RFL] = clock64() - t0;
    }
    {
        uint64_t t0 = clock64();
        for (int it = 0; it < iters; it++)
        {
            asm volatile(REP64("s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(x));
        }
        res[T_DPP] = clock64() - t0;
    }
    {
        // global pointer chase within a 16 KB per-block buffer (L2 / MALL resident after the first pass)
        for (int i = lane; i < 4096; i += 64) g[i] = ((i * 97 + 13) & 4095) * 4;
        __syncthreads();
        uint32_t a = lane * 4;
        uint64_t t0 = clock64();
        for (int it = 0; it < iters / 8 + 1; it++)
        {
            asm volatile(REP16("global_load_dword %0, %0, %1\n s_waitcnt vmcnt(0)\n") : "+v"(a) : "s"(g) : "memory");
        }
        res[T_GLD_CHASE] = clock64() - t0;
        x += a;
    }
    {
        uint32_t a = lane * 4;
        uint64_t t0 = clock64();
        for (int it = 0; it < iters / 8 + 1; it++)
        {
            asm volatile(REP16("global_store_dword %0, %1, %2\n s_waitcnt vmcnt(0)\n") ::"v"(a), "v"(x), "s"(g) : "memory");
        }
        res[T_GST_WAIT] = clock64() - t0;
    }
    {
        uint64_t t0 = clock64();
        uint64_t acc = 0;
        for (int it = 0; it < iters; it++) { REP16(acc += clock64();) }
        res[T_MEMTIME] = clock64() - t0;
        x += (uint32_t)acc;
    }
    {
        uint32_t t = 1;
        uint64_t t0 = clock64();
        for (int it = 0; it < iters; it++)
        {
            asm volatile(REP16("v_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1\n v_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1\n") : "+v"(x), "+s"(t)::"scc");
        }
        res[T_VALU_SALU_MIX] = clock64() - t0;
        x += t;
    }
    {
        uint32_t a = (lane & 1023) * 4, b0, b1, b2, b3;
        uint64_t t0 = clock64();
        for (int it = 0; it < iters; it++)
        {
            asm volatile(REP16("ds_read_b32 %1, %0\n ds_read_b32 %2, %0 offset:4\n ds_read_b32 %3, %0 offset:8\n ds_read_b32 %4, %0 offset:12\n s_waitcnt lgkmcnt(0)\n")
                         : "+v"(a), "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3)::"memory");
        }
        res[T_LDS_RD_INDEP] = clock64() - t0;
        x += b0 + b1 + b2 + b3;
    }
    {
        uint32_t c = 0;
        uint64_t t0 = clock64();
        asm volatile("s_mov_b32 %0, 0\n2:\n s_add_u32 %0, %0, 1\n s_cmp_lt_u32 %0, %1\n s_cbranch_scc1 2b\n" : "+s"(c) : "s"((uint32_t)(iters * 64)) : "scc");
        res[T_SBRANCH_LOOP] = clock64() - t0;
        x += c;
    }
    {
        uint32_t c = 0;
        uint64_t t0 = clock64();
        asm volatile("v_mov_b32 %0, 0\n3:\n v_add_u32 %0, %0, 1\n v_cmp_lt_u32 vcc, %0, %1\n s_and_b64 vcc, exec, vcc\n s_cbranch_vccnz 3b\n" : "+v"(c) : "v"((uint32_t)(iters * 64)) : "vcc");
        res[T_VCC_BRANCH] = clock64() - t0;
        x += c;
    }
    if (lane == 0)
        for (int k = 0; k < T_COUNT; k++) out[(size_t)blockIdx.x * T_COUNT + k] = res[k];
    if (x + y + z + w + s == 0x12345678) g[0] = x;
}

int main()
{
    const int iters = 200;
    for (int blocks : {1, 1024, 4096})
    {
        uint64_t* d_out; uint32_t* d_g;
        (void)hipMalloc(&d_out, sizeof(uint64_t) * T_COUNT * blocks);
        (void)hipMalloc(&d_g, sizeof(uint32_t) * 4096 * (size_t)blocks);
        hipLaunchKernelGGL(bench, dim3(blocks), dim3(64), 0, 0, d_out, d_g, iters);
        (void)hipDeviceSynchronize();
        std::vector<uint64_t> h((size_t)T_COUNT * blocks);
        (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        printf("== %d blocks x 64 threads, cycles per op (avg over blocks) ==\n", blocks);
        for (int k = 0; k < T_COUNT; k++)
        {
            double sum = 0;
            for (int b = 0; b < blocks; b++) sum += (double)h[(size_t)b * T_COUNT + k];
            double ops = (double)iters * 64;
            if (k == T_GLD_CHASE || k == T_GST_WAIT) ops = 16.0 * (iters / 8 + 1);
            if (k == T_MEMTIME || k == T_LDS_RD_INDEP) ops = 16.0 * iters;
            printf("  %-32s %8.2f\n", kNames[k], sum / blocks / ops);
        }
        (void)hipFree(d_out); (void)hipFree(d_g);
    }
    return 0;
}
