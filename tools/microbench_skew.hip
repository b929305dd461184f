// microbench_skew.hip -- what a window gains when the 256 columns of a band row are split over NW wavefronts (one per SIMD) that run
// skewed, each a row behind its left neighbour ("run k = min(4, 1024 / W) waves per window on a column-skewed
// schedule ... wave j starts row r when wave j-1 has published row r's carry"). A model of the packed kind-0 row (previous row in
// registers, band fixed): the same dependent chain as poa_forward_moves.h's reg_row -- match / mismatch costs, diagonal / vertical
// candidates with the cell shifted in from the lane (or wavefront) on the left, in-lane maxima, the 6-step DPP prefix maximum of
// u[t] = v[t] - t * gap, the fix-up with the carry-in, move bytes, one ring store to LDS and the score / move rows to HBM -- with
// CPL = 4 / NW cells per lane. Wave w > 0 takes (a) the left wave's last cell of the previous row (for its first diagonal) and
// (b) the left wave's carry of THIS row from an 8-byte hand-over entry in LDS (poa_device.h's MwShared protocol: {value, row},
// one ds_write_b64 behind the row, polled by the reader); it needs (b) only after its own prefix scan.
// Output: cycles per row of the LAST wave (the window's forward pass lasts until it is through), for NW = 1, 2, 4, at W blocks.
//   build: hipcc --offload-arch=gfx950 -O3 tools/microbench_skew.hip -o tools/bin/microbench_skew
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                           \
    do                                                                                     \
    {                                                                                      \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess)                                                              \
        {                                                                                  \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ int32_t wave_shr1(int32_t v, int32_t first)
{
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false); // wave_shr:1, lane 0 keeps `first`
}
__device__ __forceinline__ int32_t wave_inclusive_max(int32_t v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false)); // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false)); // row_bcast:15
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false)); // row_bcast:31
    return v;
}

struct Args
{
    int rows, reps;
    uint8_t* slabs;
    size_t per_block;
    unsigned long long* cycles;
};

template <int NW>
__global__ __launch_bounds__(64 * NW) void skew_rows(Args a)
{
    constexpr int CPL  = 4 / NW;            // cells per lane
    constexpr int REGS = CPL >= 2 ? CPL / 2 : 1;
    __shared__ int32_t hand[NW][8][2];      // per wave and row & 7: {last cell (carry), row}
    __shared__ uint32_t ring[NW][8][64 * REGS];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    uint8_t* slab  = a.slabs + (size_t)blockIdx.x * a.per_block;
    uint32_t* score_row = reinterpret_cast<uint32_t*>(slab) + (size_t)wave * 64 * REGS + (size_t)lane * REGS;
    uint8_t* move_row   = slab + ((size_t)8 << 20);
    move_row += (size_t)wave * 64 * CPL + (size_t)lane * CPL;
    const int32_t gap = -8;
    const uint32_t GAP2 = 0xfff8fff8u, MAT2 = 0x00080008u, DIF2 = 0xfff2fff2u, ONE2 = 0x00010001u, NEG1 = 0xffffffffu, THREE2 = 0x00030003u;
    uint32_t K[REGS], P[REGS], rd[REGS];
    const int c0 = (wave * 64 + lane) * CPL;
    for (int i = 0; i < REGS; i++)
    {
        K[i]  = ((uint32_t)((c0 + 2 * i) * gap) & 0xffffu) | ((uint32_t)((c0 + 2 * i + 1) * gap) << 16);
        P[i]  = K[i];
        rd[i] = 0x01000200u * (uint32_t)(lane + i + 1);
    }
    if (lane < 16 && wave == 0)
        for (int w = 0; w < NW; w++) { hand[w][lane & 7][0] = 0; hand[w][lane & 7][1] = 0; }
    __syncthreads();
    const uint32_t stride_s = 264 * 2 / 4, stride_m = 264;
    int32_t carry_prev = -16384;
    unsigned long long t0 = clock64();
    for (int rep = 0; rep < a.reps; rep++)
    {
        for (int r = 1 + rep * a.rows; r <= (rep + 1) * a.rows; r++)
        {
            const uint32_t base4 = 0x01010101u * (uint32_t)(r & 3);
            // the cell left of the wave's first cell in the previous row = the left wave's carry of row r - 1, which the poll of
            // the previous iteration has brought (one register, no load)
            const int32_t left_prev = carry_prev;
            uint32_t c[REGS], D[REGS], V[REGS], s[REGS];
            // costs
            for (int i = 0; i < REGS; i++)
            {
                const uint32_t x = rd[i] ^ base4;
                c[i] = pk_mad(pk_min_u(__builtin_amdgcn_perm(0u, x, 0x0c010c00u), ONE2), DIF2, MAT2);
            }
            // candidates: diagonal needs the previous row shifted by one cell
            const uint32_t prev_last = (uint32_t)wave_shr1((int32_t)P[REGS - 1], (int32_t)((uint32_t)left_prev << 16));
            for (int i = 0; i < REGS; i++)
            {
                const uint32_t lo = i == 0 ? prev_last : P[i - 1];
                D[i] = pk_add(CPL == 1 ? (prev_last >> 16) : __builtin_amdgcn_alignbit(P[i], lo, 16), c[i]);
                V[i] = pk_add(P[i], GAP2);
                s[i] = pk_max(D[i], V[i]);
            }
            // in-lane maxima of u = s - t * gap
            uint32_t u[REGS], pm[REGS];
            int32_t m;
            if (CPL == 1)
            {
                u[0]  = pk_sub(s[0], K[0]);
                pm[0] = u[0];
                m     = (int32_t)(int16_t)(u[0] & 0xffffu);
            }
            else
            {
                uint32_t best = 0x80008000u;
                for (int i = 0; i < REGS; i++)
                {
                    u[i]  = pk_sub(s[i], K[i]);
                    pm[i] = pk_max(u[i], (u[i] << 16) | 0x8000u);
                    if (i > 0) pm[i] = pk_max(pm[i], __builtin_amdgcn_perm(pm[i - 1], pm[i - 1], 0x03020302u));
                    best = pm[i];
                }
                m = (int32_t)best >> 16;
            }
            const int32_t incl = wave_inclusive_max(m);
            // carry-in: lane 0's exclusive value. Wave 0: the band's left boundary; others: the left wave's row-r carry (polled)
            int32_t cu = -16384 - 2 * 8;
            if (NW > 1 && wave > 0)
            {
                // one 8-byte load per poll: {the left wave's last cell of row r, r}
                unsigned long long e;
                int32_t have;
                do
                {
                    e    = *(volatile unsigned long long*)&hand[wave - 1][r & 7][0];
                    have = __builtin_amdgcn_readfirstlane((int32_t)(e >> 32));
                    if (have < r) __builtin_amdgcn_s_sleep(1);
                } while (have < r);
                cu         = __builtin_amdgcn_readfirstlane((int32_t)(uint32_t)e);
                carry_prev = cu;
            }
            const int32_t excl = max(wave_shr1(incl, cu), cu);
            const uint32_t ex2 = __builtin_amdgcn_perm((uint32_t)excl, (uint32_t)excl, 0x01000100u);
            for (int i = 0; i < REGS; i++) P[i] = pk_add(pk_max(pm[i], ex2), K[i]);
            // publish the wave's last cell (the right neighbour's carry and next row's diagonal operand)
            if (NW > 1 && wave < NW - 1)
            {
                const int32_t last = __builtin_amdgcn_readlane((int32_t)(CPL == 1 ? P[0] << 16 : P[REGS - 1]), 63) >> 16;
                if (lane == 0) // one 8-byte store: the value and the row it belongs to
                {
                    const unsigned long long e = (unsigned long long)(uint32_t)last | ((unsigned long long)(uint32_t)r << 32);
                    *(volatile unsigned long long*)&hand[wave][r & 7][0] = e;
                }
            }
            // move bytes: 3 + [H != D] * (-1 - [H != V])
            uint32_t mv = 0;
            for (int i = 0; i < REGS; i++)
            {
                const uint32_t nd = pk_min_u(pk_sub(P[i], D[i]), ONE2), nv = pk_min_u(pk_sub(P[i], V[i]), ONE2);
                const uint32_t mm = pk_mad(nd, pk_mad(nv, NEG1, NEG1), THREE2);
                mv |= __builtin_amdgcn_perm(0u, mm, 0x0c0c0200u) << (16 * (i & 1));
            }
            // stores: ring (LDS), score row and move row (HBM)
            for (int i = 0; i < REGS; i++) ring[wave][r & 7][lane * REGS + i] = P[i];
            for (int i = 0; i < REGS; i++) __builtin_nontemporal_store(P[i], score_row + i);
            if (CPL == 4) *reinterpret_cast<uint32_t*>(move_row) = mv;
            else if (CPL == 2) *reinterpret_cast<uint16_t*>(move_row) = (uint16_t)mv;
            else *move_row = (uint8_t)mv;
            score_row += stride_s;
            move_row += stride_m;
        }
        score_row -= (size_t)stride_s * a.rows;
        move_row -= (size_t)stride_m * a.rows;
    }
    const unsigned long long dt = clock64() - t0;
    if (lane == 0 && wave == NW - 1) a.cycles[blockIdx.x] = dt;
}

template <int NW> static void run(Args a, int blocks, const char* what)
{
    std::vector<unsigned long long> h(blocks);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    double cyc = 0;
    for (int it = 0; it < 3; it++)
    {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(skew_rows<NW>, dim3(blocks), dim3(64 * NW), 0, 0, a);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best)
        {
            best = ms;
            CHECK(hipMemcpy(h.data(), a.cycles, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
            double s = 0;
            for (int b = 0; b < blocks; b++) s += (double)h[b];
            cyc = s / blocks / ((double)a.reps * a.rows);
        }
    }
    std::printf("{\"waves_per_window\": %d, \"cells_per_lane\": %d, \"windows\": %d, \"what\": \"%s\", \"cycles_per_row\": %.1f, \"kernel_ms\": %.3f}\n", NW, 4 / NW,
                blocks, what, cyc, best);
}

int main()
{
    Args a;
    a.rows = 1400;
    a.reps = 8;
    a.per_block = (size_t)12 << 20;
    const int max_blocks = 1024;
    CHECK(hipMalloc(&a.slabs, a.per_block * max_blocks));
    CHECK(hipMalloc(&a.cycles, sizeof(unsigned long long) * max_blocks));
    CHECK(hipMemset(a.slabs, 0, a.per_block * max_blocks));
    // one wave per window at 1024 windows (one per SIMD) and at 256; two and four waves per window at 256 windows (every wave on
    // a SIMD of its own: 512 / 1024 waves on 1024 SIMDs)
    run<1>(a, 1024, "warm-up (first launch of the process)");
    run<1>(a, 1024, "one wavefront per window, 1024 windows (the production geometry)");
    run<1>(a, 256, "one wavefront per window, 256 windows");
    run<2>(a, 256, "two wavefronts per window (128 columns each), 256 windows");
    run<4>(a, 256, "four wavefronts per window (64 columns each), 256 windows");
    run<2>(a, 512, "two wavefronts per window, 512 windows");
    return 0;
}
