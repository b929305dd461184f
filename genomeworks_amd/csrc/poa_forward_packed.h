// poa_forward_packed.h -- shared pieces of the packed 16-bit forward passes (two score cells per 32-bit register,
// v_pk_add_i16 / v_pk_max_i16): the packed-arithmetic helpers, the LDS ring geometry (8 rows of 512 absolute column
// slots with sentinel cells behind the band end), the side table of predecessors 3..5, and the move codes of the
// long-read passes. The forward pass of the 256-column band itself is poa_forward_moves.h (round 3; the round-1/2 routine
// that lived here -- row classes 0..3, trace codes resolved through the row table -- was retired with its traceback).
#pragma once

namespace gwhip
{

typedef short pk_i16 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pk_i16, a) + __builtin_bit_cast(pk_i16, b));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pk_i16, a) - __builtin_bit_cast(pk_i16, b));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_i16, a), __builtin_bit_cast(pk_i16, b)));
}
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(pk_u16, a), __builtin_bit_cast(pk_u16, b)));
}
__device__ __forceinline__ uint32_t pk_mad_u(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pk_u16, a) * __builtin_bit_cast(pk_u16, b) + __builtin_bit_cast(pk_u16, c));
}
__device__ __forceinline__ uint32_t pk_dup(int32_t v) { return ((uint32_t)v & 0xffffu) | ((uint32_t)v << 16); }
__device__ __forceinline__ uint32_t pk_make(int32_t lo, int32_t hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ int32_t pk_lo(uint32_t v) { return (int32_t)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int32_t pk_hi(uint32_t v) { return (int32_t)v >> 16; }

constexpr int kPkSlots     = 8;    // ring rows
constexpr int kPkSlotBytes = 1024; // 512 column slots x int16
constexpr int kPkMaxDist   = kPkSlots - 1;
constexpr int kPkGuardCols = 60;   // a reader's band may start at most this far right of a ring predecessor's
constexpr int kPkSentinel  = -32768;
constexpr int kClassShift  = 60;   // row class in bits 60..61 of the packed row-table word

// LDS byte address of a pointer into the dynamic shared segment
__device__ __forceinline__ uint32_t lds_addr(const void* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_load_u32(uint32_t addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(addr);
}
__device__ __forceinline__ uint2 lds_load_u64(uint32_t addr)
{
    const u32x2 v = *reinterpret_cast<const __attribute__((address_space(3))) u32x2*>(addr);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void lds_store_u64(uint32_t addr, uint32_t lo, uint32_t hi)
{
    u32x2 v;
    v.x = lo; v.y = hi;
    *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(addr) = v;
}
// The guard store of a ring row: lanes 0..15 the sentinel cells behind the band end, lane 16 the quad that ends in the
// left-boundary slot, lanes 17..63 a quad nobody reads (see guard_off in the forward passes): one plain 8-byte LDS store by
// all lanes. (Rounds 1-3 masked lanes 17..63 off around the store: s_mov_b64 exec twice per row.)
__device__ __forceinline__ void lds_store_guard(uint32_t addr, uint32_t lo, uint32_t hi) { lds_store_u64(addr, lo, hi); }

// ------------------------------------------------------------------------------------------------
// Side table of predecessor rows 3..5 (build_rowinfo, poa_device.h): one 64-bit entry per row & 255
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool xpred_hit(uint64_t e, int32_t row, int32_t cnt)
{
    return (int32_t)(e & 0xfffu) == row && ((e >> 12) & 1u) != 0 && (int32_t)((e >> 13) & 63u) == cnt;
}
__device__ __forceinline__ int32_t xpred_row(uint64_t e, int32_t k) { return (int32_t)((e >> (20 + 12 * (k - 3))) & 0xfffu); } // k = 3..5


__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// keep a loop-invariant in a VGPR (the kernel is SGPR-bound: a uniform constant would otherwise live in an SGPR
// and be spilled / reloaded with v_readlane inside the row loop)
__device__ __forceinline__ uint32_t pin_vgpr(uint32_t v)
{
    asm volatile("" : "+v"(v));
    return v;
}
// lane-0 2-byte global store without a divergent branch (wave-uniform caller, all lanes active)
__device__ __forceinline__ void global_store_u16_lane0(void* p, uint32_t v)
{
    const uint32_t zero = 0;
    asm volatile("s_mov_b64 exec, 1\n\tglobal_store_short %0, %1, %2\n\ts_mov_b64 exec, -1" ::"v"(zero), "v"(v), "s"(p) : "memory");
}

} // namespace gwhip
