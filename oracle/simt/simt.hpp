// simt.hpp -- TEST INFRASTRUCTURE (oracle/): a single-threaded SIMT emulator that lets the reference's CUDA device code run on the
// CPU, compiled by g++ from the sources where they lie under /root/reference (oracle/Makefile.ref, target ref_cudapoa_simt). The
// threads of a block are fibers (ucontext) that the scheduler runs round robin on one OS thread; a warp is 32 consecutive
// fibers; the warp-level primitives (__shfl_sync, __shfl_up_sync, __shfl_down_sync, __any_sync, __all_sync, __ballot_sync,
// __syncwarp) and __syncthreads are rendezvous points among the fibers that have not left the kernel. A kernel launch
// (`k<<<grid, block, shmem, stream>>>(args)`, rewritten by oracle/simt/cuda_to_simt.py into simt::launch(grid, block, [&] { k(args); }))
// runs its blocks one after the other and returns when the last thread has finished: every launch is synchronous.
// Nothing of the product links or includes this: it exists so that tests can compare oracle/poa_oracle.c (and through it the
// HIP kernels) with the reference's own kernels on arbitrary inputs.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <type_traits>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__ static
#define __align__(n) alignas(n)

struct uint3
{
    unsigned x, y, z;
};
struct int2
{
    int x, y;
};
struct int4
{
    int x, y, z, w;
};
inline int2 make_int2(int x, int y) { return int2{x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
struct dim3
{
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
    dim3(uint3 v) : x(v.x), y(v.y), z(v.z) {}
};
inline uint3 threadIdx{0, 0, 0}, blockIdx{0, 0, 0};
inline dim3 blockDim{1, 1, 1}, gridDim{1, 1, 1};
constexpr int warpSize = 32;

namespace simt
{
constexpr size_t kStackBytes = size_t(1) << 20;

struct Fiber
{
    ucontext_t ctx;
    std::vector<char> stack;
    bool done = false;
    unsigned tid = 0;
};

struct Block
{
    std::vector<Fiber> fibers;
    ucontext_t scheduler;
    const std::function<void()>* body = nullptr;
    int current                       = -1;
    // rendezvous state per warp and for the block: arrivals of the current round, a generation that advances when the round is
    // complete, one 64-bit slot per lane for the values exchanged
    struct Rendezvous
    {
        int arrived = 0;
        uint64_t generation = 0;
    };
    std::vector<std::map<unsigned, Rendezvous>> warp_sync; // per warp, per participation mask of the *_sync primitive
    Rendezvous block_sync;
    std::vector<uint64_t> slot;
    uint64_t progress = 0; // (deadlock detection: rounds completed + fibers finished)
};
inline Block* g_block = nullptr;

inline void yield()
{
    Block& b = *g_block;
    Fiber& f = b.fibers[static_cast<size_t>(b.current)];
    swapcontext(&f.ctx, &b.scheduler);
}

// the lanes of [first, last) that have not left the kernel and are named in `mask` (bit i = lane first + i)
inline int live_in_range(const Block& b, int first, int last, unsigned mask = 0xffffffffu)
{
    int n = 0;
    for (int i = first; i < last; ++i)
        if (!b.fibers[static_cast<size_t>(i)].done && (last - first > 32 || ((mask >> (i - first)) & 1u))) n++;
    return n;
}

// the live fibers of [first, last) named in `mask` meet here; fibers that have left the kernel are not waited for (CUDA: exited
// threads do not take part in *_sync primitives)
inline void rendezvous(Block::Rendezvous& r, int first, int last, unsigned mask = 0xffffffffu)
{
    Block& b            = *g_block;
    const uint64_t mine = r.generation;
    r.arrived++;
    for (;;)
    {
        if (r.generation != mine) return;
        if (r.arrived >= live_in_range(b, first, last, mask))
        {
            r.arrived = 0;
            r.generation++;
            b.progress++;
            return;
        }
        yield();
    }
}

// (a warp is 32 consecutive threads of the block in linear order x + y * blockDim.x + z * blockDim.x * blockDim.y)
inline int linear_tid() { return g_block->current; }
inline int lane_id() { return linear_tid() % 32; }
inline int warp_first() { return linear_tid() / 32 * 32; }
inline int warp_last() { return std::min<int>(warp_first() + 32, static_cast<int>(g_block->fibers.size())); }
// the lanes named in `mask` meet (a primitive called with a partial mask is called by exactly those lanes)
inline void warp_rendezvous(unsigned mask = 0xffffffffu)
{
    rendezvous(g_block->warp_sync[static_cast<size_t>(linear_tid() / 32)][mask], warp_first(), warp_last(), mask);
}
/// reconvergence point of a divergent section (cuda_to_simt.py puts one on either side of `if (lane_idx == 0) ...`)
inline void converge()
{
    if (g_block != nullptr) warp_rendezvous();
}

template <typename T>
inline uint64_t to_bits(T v)
{
    static_assert(sizeof(T) <= 8, "values of up to 64 bits are exchanged");
    uint64_t u = 0;
    std::memcpy(&u, &v, sizeof(T));
    return u;
}
template <typename T>
inline T from_bits(uint64_t u)
{
    T v;
    std::memcpy(&v, &u, sizeof(T));
    return v;
}

// every live lane of the warp publishes `v`; returns after all have, `read` picks what this lane wants, and a second rendezvous
// keeps the slots until everyone has read
template <typename T, typename Read>
inline T exchange(unsigned mask, T v, Read read)
{
    Block& b                                  = *g_block;
    b.slot[static_cast<size_t>(linear_tid())] = to_bits(v);
    warp_rendezvous(mask);
    const T out = read(b.slot.data() + warp_first());
    warp_rendezvous(mask);
    return out;
}

inline void run_block(unsigned threads, const std::function<void()>& body)
{
    Block b;
    b.fibers.resize(threads);
    b.warp_sync.assign((threads + 31) / 32, {});
    b.slot.assign(threads, 0);
    b.body  = &body;
    g_block = &b;
    for (unsigned t = 0; t < threads; ++t)
    {
        Fiber& f = b.fibers[t];
        f.tid    = t;
        f.stack.resize(kStackBytes);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp   = f.stack.data();
        f.ctx.uc_stack.ss_size = f.stack.size();
        f.ctx.uc_link          = &b.scheduler;
        makecontext(&f.ctx, reinterpret_cast<void (*)()>(+[]() {
                        Block& blk = *g_block;
                        (*blk.body)();
                        blk.fibers[static_cast<size_t>(blk.current)].done = true;
                        blk.progress++;
                    }),
                    0);
    }
    unsigned remaining = threads;
    while (remaining > 0)
    {
        const uint64_t before = b.progress;
        remaining             = 0;
        // SIMT_ORDER=reverse: the lanes of a round run from the last to the first -- a result that depends on the order depends on a
        // race between lanes that the GPU resolves by executing them in lockstep
        static const bool reverse_order = [] { const char* e = std::getenv("SIMT_ORDER"); return e != nullptr && e[0] == 'r'; }();
        for (unsigned k = 0; k < threads; ++k)
        {
            const unsigned t = reverse_order ? threads - 1 - k : k;
            Fiber& f         = b.fibers[t];
            if (f.done) continue;
            b.current = static_cast<int>(t);
            threadIdx = uint3{t % blockDim.x, t / blockDim.x % blockDim.y, t / (blockDim.x * blockDim.y)};
            swapcontext(&b.scheduler, &f.ctx);
            if (!f.done) remaining++;
        }
        if (remaining > 0 && b.progress == before)
        {
            // one more full round without any rendezvous completing or fiber finishing: the live fibers wait for each other at
            // different primitives (divergent *_sync calls)
            bool waiting_somewhere = false;
            for (const auto& by_mask : b.warp_sync)
                for (const auto& r : by_mask) waiting_somewhere |= r.second.arrived > 0;
            waiting_somewhere |= b.block_sync.arrived > 0;
            if (waiting_somewhere)
            {
                static int stalls = 0;
                if (++stalls > 4)
                {
                    std::fprintf(stderr, "simt: deadlock -- %u live threads wait at rendezvous points that never complete\n", remaining);
                    std::abort();
                }
            }
        }
    }
    g_block = nullptr;
}

/// k<<<grid, block, shared memory, stream>>>(args): blocks one after the other, synchronous
template <typename Body>
inline void launch(dim3 grid, dim3 block, Body body)
{
    const std::function<void()> fn = body;
    gridDim  = grid;
    blockDim = block;
    static const bool trace = std::getenv("SIMT_TRACE") != nullptr;
    if (trace) std::fprintf(stderr, "simt: launch grid (%u, %u, %u) block (%u, %u, %u)\n", grid.x, grid.y, grid.z, block.x, block.y, block.z);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx)
            {
                blockIdx = uint3{bx, by, bz};
                run_block(block.x * block.y * block.z, fn);
            }
}
template <typename Body, typename A>
inline void launch(dim3 grid, dim3 block, A, Body body)
{
    launch(grid, block, body);
}
template <typename Body, typename A, typename B>
inline void launch(dim3 grid, dim3 block, A, B, Body body)
{
    launch(grid, block, body);
}
} // namespace simt

// ---- warp-level and block-level primitives -------------------------------------------------------------------------
inline void __syncwarp(unsigned mask = 0xffffffffu) { simt::warp_rendezvous(mask); }
inline void __syncthreads() { simt::rendezvous(simt::g_block->block_sync, 0, static_cast<int>(simt::g_block->fibers.size())); }
inline void __threadfence() {}
inline void __threadfence_block() {}

template <typename T>
inline T __shfl_sync(unsigned mask, T v, int src_lane, int width = 32)
{
    const int lane = simt::lane_id();
    return simt::exchange(mask, v, [&](const uint64_t* s) { return simt::from_bits<T>(s[(lane & ~(width - 1)) + (src_lane & (width - 1))]); });
}
template <typename T>
inline T __shfl_up_sync(unsigned mask, T v, unsigned delta, int width = 32)
{
    const int lane = simt::lane_id();
    return simt::exchange(mask, v, [&](const uint64_t* s) {
        const int src = lane - static_cast<int>(delta);
        return src >= (lane & ~(width - 1)) ? simt::from_bits<T>(s[src]) : v;
    });
}
template <typename T>
inline T __shfl_down_sync(unsigned mask, T v, unsigned delta, int width = 32)
{
    const int lane = simt::lane_id();
    return simt::exchange(mask, v, [&](const uint64_t* s) {
        const int src = lane + static_cast<int>(delta);
        return src < (lane & ~(width - 1)) + width ? simt::from_bits<T>(s[src]) : v;
    });
}
template <typename T>
inline T __shfl_xor_sync(unsigned mask, T v, int lane_mask, int width = 32)
{
    const int lane = simt::lane_id();
    return simt::exchange(mask, v, [&](const uint64_t* s) {
        const int src = lane ^ lane_mask;
        return src < (lane & ~(width - 1)) + width ? simt::from_bits<T>(s[src]) : v;
    });
}
inline unsigned __ballot_sync(unsigned mask, int predicate)
{
    // lanes that have left the kernel contribute 0 (their slot is cleared when they publish nothing: cleared below)
    simt::Block& b = *simt::g_block;
    const int first = simt::warp_first(), last = simt::warp_last();
    return simt::exchange(mask, static_cast<uint32_t>(predicate != 0), [&](const uint64_t* s) {
        unsigned m = 0;
        for (int i = first; i < last; ++i)
            if (!b.fibers[static_cast<size_t>(i)].done && ((mask >> (i - first)) & 1u) && s[i - first] != 0) m |= 1u << (i - first);
        return m;
    });
}
inline int __any_sync(unsigned mask, int predicate) { return __ballot_sync(mask, predicate) != 0; }
inline int __all_sync(unsigned mask, int predicate)
{
    simt::Block& b = *simt::g_block;
    const int first = simt::warp_first(), last = simt::warp_last();
    unsigned live = 0;
    for (int i = first; i < last; ++i)
        if (!b.fibers[static_cast<size_t>(i)].done && ((mask >> (i - first)) & 1u)) live |= 1u << (i - first);
    return (__ballot_sync(mask, predicate) & live) == live;
}
inline unsigned __activemask() { return 0xffffffffu; }

// ---- arithmetic the device code calls unqualified ----------------------------------------------------------------------
template <typename A, typename B, typename = std::enable_if_t<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>>
inline std::common_type_t<A, B> max(A a, B b)
{
    using C = std::common_type_t<A, B>;
    return static_cast<C>(a) < static_cast<C>(b) ? static_cast<C>(b) : static_cast<C>(a);
}
template <typename A, typename B, typename = std::enable_if_t<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>>
inline std::common_type_t<A, B> min(A a, B b)
{
    using C = std::common_type_t<A, B>;
    return static_cast<C>(b) < static_cast<C>(a) ? static_cast<C>(b) : static_cast<C>(a);
}
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz(static_cast<unsigned>(v)); }
inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll(static_cast<unsigned long long>(v)); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
template <typename T>
inline T __ldg(const T* p)
{
    return *p;
}
template <typename T>
inline T atomicAdd(T* p, T v)
{
    const T old = *p;
    *p          = old + v;
    return old;
}
