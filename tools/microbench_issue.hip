// microbench_issue.hip -- what the instructions of the metric kernel's row loop cost a LONE wavefront (1024 one-wave blocks, one
// per SIMD): dependent chains of the packed / permute / DPP operations the row is made of, the VALU -> SGPR -> SALU round trip of
// the row descriptors (v_readlane), EXEC writes, LDS and HBM store issue, taken branches. Round 4: the kind-0 row issues ~85
// instructions in ~690 cycles, twice what tools/microbench_fetch.hip's 4.1-5.2 cycles per instruction predict.
//   build: hipcc --offload-arch=gfx950 -O2 tools/microbench_issue.hip -o tools/bin/microbench_issue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

enum { T_PKMAX, T_PKMAD, T_PKMIN, T_PERM, T_ALIGNBIT, T_LSHLOR, T_DPP_DEP_NOP, T_DPP_DEP_FILL, T_WAVESHR_NOP, T_READLANE_SALU, T_READLANE_VALU,
       T_SALU_READLANE, T_EXEC, T_LDSW, T_GSTORE, T_BRANCH, T_ROWLIKE, T_COUNT };
static const char* kNames[T_COUNT] = {
    "v_pk_max_i16 dependent", "v_pk_mad_u16 dependent", "v_pk_min_u16 dependent", "v_perm_b32 dependent", "v_alignbit_b32 dependent",
    "v_lshl_or_b32 dependent", "v_max_i32_dpp row_shr:1 dependent + s_nop 1 (per pair)", "v_max_i32_dpp row_shr:1 dependent + 2 independent v_add (per triple)",
    "v_mov_b32_dpp wave_shr:1 dependent + s_nop 1 (per pair)", "v_readlane -> s_add on the result (per pair)", "v_readlane -> v_xor with the SGPR (per pair)",
    "s_add -> v_readlane with that SGPR as lane select (per pair)", "s_mov_b64 exec x2 around one v_add (per triple)",
    "ds_write_b64 + 15 dependent v_add (per 16)", "global_store_dwordx2 + 15 dependent v_add (per 16)", "taken s_cbranch + 15 dependent v_add (per 16)",
    "row-like mix: 2 readlane, 6 salu, 30 pk/perm dependent, 7 dpp, 2 ds_write, 2 global_store, 1 taken branch (per iteration of 50)"};
static const int kPer[T_COUNT] = {1, 1, 1, 1, 1, 1, 2, 3, 2, 2, 2, 2, 3, 16, 16, 16, 50};

__global__ void bench(uint64_t* out, int iters, uint8_t* sink)
{
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x;
    uint64_t res[T_COUNT];
    uint32_t x = lane, y = lane + 1, z = lane + 2, w = lane * 3, s = iters;
    const uint32_t laddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + lane * 8;
    uint8_t* gp = sink + ((size_t)blockIdx.x * 64 + lane) * 8;
#define TIME(idx, ...)                                                                     \
    {                                                                                      \
        uint64_t t0 = clock64();                                                           \
        for (int it = 0; it < iters; it++) { asm volatile(__VA_ARGS__); }                  \
        res[idx] = clock64() - t0;                                                         \
    }
    TIME(T_PKMAX, REP64("v_pk_max_i16 %0, %0, %1\n") : "+v"(x) : "v"(y))
    TIME(T_PKMAD, REP64("v_pk_mad_u16 %0, %0, %1, %2\n") : "+v"(x) : "v"(y), "v"(z))
    TIME(T_PKMIN, REP64("v_pk_min_u16 %0, %0, %1\n") : "+v"(x) : "v"(y))
    TIME(T_PERM, REP64("v_perm_b32 %0, %0, %1, %2\n") : "+v"(x) : "v"(y), "v"(w))
    TIME(T_ALIGNBIT, REP64("v_alignbit_b32 %0, %0, %1, 16\n") : "+v"(x) : "v"(y))
    TIME(T_LSHLOR, REP64("v_lshl_or_b32 %0, %0, 16, %1\n") : "+v"(x) : "v"(y))
    TIME(T_DPP_DEP_NOP, REP64("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n") : "+v"(x))
    TIME(T_DPP_DEP_FILL, REP64("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_e32 %1, %1, %1\n v_add_u32_e32 %2, %2, %2\n") : "+v"(x), "+v"(y), "+v"(z))
    TIME(T_WAVESHR_NOP, REP64("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n") : "+v"(x))
    TIME(T_READLANE_SALU, REP64("v_readlane_b32 %1, %0, 5\n s_add_u32 %1, %1, %1\n") : "+v"(x), "+s"(s)::"scc")
    TIME(T_READLANE_VALU, REP64("v_readlane_b32 %1, %0, 5\n v_xor_b32_e32 %0, %1, %0\n") : "+v"(x), "+s"(s))
    TIME(T_SALU_READLANE, REP64("s_and_b32 %1, %1, 63\n s_nop 3\n v_readlane_b32 %1, %0, %1\n") : "+v"(x), "+s"(s)::"scc")
    TIME(T_EXEC, REP64("s_mov_b64 exec, 0x1ffff\n v_add_u32_e32 %0, %0, %0\n s_mov_b64 exec, -1\n") : "+v"(x))
    TIME(T_LDSW, REP4("ds_write_b64 %1, %2\n" REP4("v_add_u32_e32 %0, %0, %0\n") REP4("v_add_u32_e32 %0, %0, %0\n") REP4("v_add_u32_e32 %0, %0, %0\n") "v_add_u32_e32 %0, %0, %0\n v_add_u32_e32 %0, %0, %0\n v_add_u32_e32 %0, %0, %0\n") : "+v"(x) : "v"(laddr), "v"((uint64_t)y))
    TIME(T_GSTORE, REP4("global_store_dwordx2 %1, %2, off\n" REP4("v_add_u32_e32 %0, %0, %0\n") REP4("v_add_u32_e32 %0, %0, %0\n") REP4("v_add_u32_e32 %0, %0, %0\n") "v_add_u32_e32 %0, %0, %0\n v_add_u32_e32 %0, %0, %0\n v_add_u32_e32 %0, %0, %0\n") : "+v"(x) : "v"(gp), "v"((uint64_t)y) : "memory")
    TIME(T_BRANCH, REP4("s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n" REP4("v_add_u32_e32 %0, %0, %0\n") REP4("v_add_u32_e32 %0, %0, %0\n") REP4("v_add_u32_e32 %0, %0, %0\n") "v_add_u32_e32 %0, %0, %0\n v_add_u32_e32 %0, %0, %0\n") : "+v"(x)::"scc")
    // a row-like iteration: the instruction mix of the kind-0 streak loop, as one dependent chain
    TIME(T_ROWLIKE,
         "v_readlane_b32 %4, %1, 7\n s_and_b32 %4, %4, 7\n s_add_u32 %4, %4, 1\n"
         "v_xor_b32_e32 %0, %4, %0\n v_perm_b32 %0, %0, %1, %3\n v_pk_min_u16 %0, %0, %1\n v_pk_mad_u16 %0, %0, %1, %2\n"
         "v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_alignbit_b32 %0, %0, %2, 16\n v_pk_add_u16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n"
         REP4("v_pk_max_i16 %0, %0, %1\n v_pk_add_u16 %0, %0, %2\n") "v_lshl_or_b32 %0, %0, 16, %1\n v_ashrrev_i32_e32 %0, 16, %0\n s_add_u32 %4, %4, 3\n"
         "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_add_u32 %4, %4, 3\n v_add_u32_e32 %2, %2, %2\n"
         "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_add_u32 %4, %4, 3\n v_add_u32_e32 %2, %2, %2\n"
         "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_add_u32 %4, %4, 3\n v_add_u32_e32 %2, %2, %2\n"
         "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n v_readlane_b32 %4, %1, 9\n s_and_b32 %4, %4, 7\n"
         "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_add_u32 %4, %4, 3\n v_add_u32_e32 %2, %2, %2\n"
         "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_add_u32 %4, %4, 3\n v_add_u32_e32 %2, %2, %2\n"
         "v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_e32 %0, %0, %2\n v_perm_b32 %0, %0, %0, %3\n"
         REP4("v_pk_max_i16 %0, %0, %1\n v_pk_add_u16 %0, %0, %2\n") "global_store_dwordx2 %5, %7, off\n ds_write_b64 %6, %7\n"
         REP4("v_pk_min_u16 %0, %0, %1\n") "ds_write_b64 %6, %7 offset:512\n" REP4("v_pk_mad_u16 %0, %0, %1, %2\n") "v_perm_b32 %0, %0, %1, %3\n"
         "global_store_dword %5, %0, off offset:2048\n s_cmp_eq_u32 0, 0\n s_cbranch_scc1 2f\n s_nop 0\n 2:\n"
         : "+v"(x), "+v"(y), "+v"(z), "+v"(w), "+s"(s) : "v"(gp), "v"(laddr), "v"((uint64_t)y) : "memory", "scc")
    if ((lane & 63) == 0)
        for (int k = 0; k < T_COUNT; k++) out[(size_t)blockIdx.x * T_COUNT + k] = res[k];
    if (x + y + z + w + s == 0x12345678) out[0] = x;
}

int main()
{
    const int iters = 200, blocks = 1024;
    uint64_t* d_out;
    uint8_t* d_sink;
    (void)hipMalloc(&d_out, sizeof(uint64_t) * T_COUNT * blocks);
    (void)hipMalloc(&d_sink, (size_t)blocks * 64 * 8 + 8192);
    hipLaunchKernelGGL(bench, dim3(blocks), dim3(64), 39184, 0, d_out, iters, d_sink);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> h((size_t)T_COUNT * blocks);
    (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    printf("{\"what\": \"cycles per instruction (or per group, as stated) for a lone wavefront: 1024 one-wave blocks with 39 KB of LDS each\", \"results\": [\n");
    for (int k = 0; k < T_COUNT; k++)
    {
        double sum = 0;
        for (int b = 0; b < blocks; b++) sum += (double)h[(size_t)b * T_COUNT + k];
        const int reps = (k <= T_EXEC) ? 64 : (k == T_ROWLIKE ? 1 : 4);
        printf(" {\"chain\": \"%s\", \"cycles\": %.2f}%s\n", kNames[k], sum / blocks / ((double)iters * reps), k + 1 < T_COUNT ? "," : "");
    }
    printf("]}\n");
    return 0;
}
