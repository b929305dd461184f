// microbench_fetch.hip -- is a lone wavefront paced by instructions or by instruction BYTES? Chains of the same operation in its
// 4-byte (VOP2 / SOP2) and 8-byte (VOP3, VOP3P, DPP, 32-bit literal) encodings, one and two wavefronts per SIMD.
//   build: hipcc --offload-arch=gfx950 -O2 tools/microbench_fetch.hip -o tools/bin/microbench_fetch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)

enum { T_VOP2, T_VOP3, T_VOP3P, T_VOP2_LIT, T_SOP2, T_SOP2_LIT, T_VOP2_IND4, T_VOP3P_IND4, T_MIX_PK_SALU, T_DPP, T_SNOP, T_COUNT };
static const char* kNames[T_COUNT] = {"v_add_u32 e32 (4 B) dependent", "v_add_u32 e64 (8 B) dependent", "v_pk_add_u16 (8 B) dependent",
                                      "v_add_u32 + literal (8 B) dependent", "s_add_u32 (4 B) dependent", "s_add_u32 + literal (8 B) dependent",
                                      "v_add_u32 e32, 4 independent chains", "v_pk_add_u16, 4 independent chains", "v_pk_add_u16 + s_add_u32 alternating (12 B per pair)",
                                      "v_max_i32_dpp row_shr:1 (8 B) on 4 independent registers", "s_nop 0 (4 B)"};

__global__ void bench(uint64_t* out, int iters)
{
    const int lane = threadIdx.x;
    uint64_t res[T_COUNT];
    uint32_t x = lane, y = lane + 1, z = lane + 2, w = lane + 3, s = iters;
#define TIME(idx, ...)                                                                     \
    {                                                                                      \
        uint64_t t0 = clock64();                                                           \
        for (int it = 0; it < iters; it++) { asm volatile(__VA_ARGS__); }                  \
        res[idx] = clock64() - t0;                                                         \
    }
    TIME(T_VOP2, REP256("v_add_u32_e32 %0, %0, %0\n") : "+v"(x))
    TIME(T_VOP3, REP256("v_add_u32_e64 %0, %0, %0\n") : "+v"(x))
    TIME(T_VOP3P, REP256("v_pk_add_u16 %0, %0, %0\n") : "+v"(x))
    TIME(T_VOP2_LIT, REP256("v_add_u32_e32 %0, 0x12345, %0\n") : "+v"(x))
    TIME(T_SOP2, REP256("s_add_u32 %0, %0, %0\n") : "+s"(s)::"scc")
    TIME(T_SOP2_LIT, REP256("s_add_u32 %0, %0, 0x12345\n") : "+s"(s)::"scc")
    TIME(T_VOP2_IND4, REP64("v_add_u32_e32 %0, %0, %0\n v_add_u32_e32 %1, %1, %1\n v_add_u32_e32 %2, %2, %2\n v_add_u32_e32 %3, %3, %3\n") : "+v"(x), "+v"(y), "+v"(z), "+v"(w))
    TIME(T_VOP3P_IND4, REP64("v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %1, %1, %1\n v_pk_add_u16 %2, %2, %2\n v_pk_add_u16 %3, %3, %3\n") : "+v"(x), "+v"(y), "+v"(z), "+v"(w))
    TIME(T_MIX_PK_SALU, REP64("v_pk_add_u16 %0, %0, %0\n s_add_u32 %4, %4, %4\n v_pk_add_u16 %1, %1, %1\n s_add_u32 %4, %4, %4\n v_pk_add_u16 %2, %2, %2\n s_add_u32 %4, %4, %4\n v_pk_add_u16 %3, %3, %3\n s_add_u32 %4, %4, %4\n") : "+v"(x), "+v"(y), "+v"(z), "+v"(w), "+s"(s)::"scc")
    TIME(T_DPP, REP64("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(x), "+v"(y), "+v"(z), "+v"(w))
    TIME(T_SNOP, REP256("s_nop 0\n"))
    if ((lane & 63) == 0)
        for (int k = 0; k < T_COUNT; k++) out[((size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x / 64)) * T_COUNT + k] = res[k];
    if (x + y + z + w + s == 0x12345678) out[0] = x;
}

int main()
{
    const int iters = 40;
    printf("{\"cycles_per_instruction\": [\n");
    bool first = true;
    for (int threads : {64, 128, 256})
    {
        const int blocks = 1024, waves = blocks * threads / 64;
        uint64_t* d_out;
        (void)hipMalloc(&d_out, sizeof(uint64_t) * T_COUNT * waves);
        hipLaunchKernelGGL(bench, dim3(blocks), dim3(threads), 0, 0, d_out, iters);
        (void)hipDeviceSynchronize();
        std::vector<uint64_t> h((size_t)T_COUNT * waves);
        (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        for (int k = 0; k < T_COUNT; k++)
        {
            double sum = 0;
            for (int b = 0; b < waves; b++) sum += (double)h[(size_t)b * T_COUNT + k];
            double per = sum / waves / ((double)iters * 256);
            if (k == T_MIX_PK_SALU) per *= 0.5 * 1.0; // per instruction of the pair
            printf("%s {\"waves_per_simd\": %d, \"chain\": \"%s\", \"cycles\": %.2f}", first ? " " : ",\n ", threads / 64, kNames[k], per);
            first = false;
        }
        (void)hipFree(d_out);
    }
    printf("\n]}\n");
    return 0;
}
