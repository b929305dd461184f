// allocator.hpp -- device allocators passed by value through create_batch / create_aligner.
// Public shape follows the reference's utils/allocator.hpp:40-362: a copyable handle with
// allocate(n, streams) / deallocate(p, n) / get_size_of_largest_free_memory_block().
//
// MI355X design: DefaultDeviceAllocator is a handle on a DevicePool -- ONE hipMalloc'd slab (sized for the
// 288 GB part: callers hand most of HBM to a single pool) carved by a first-fit free list. Frees are
// stream-aware: a block returns to the free list only after the events recorded on its associated streams
// have completed, so many Batch/Aligner objects on many host threads can share one pool
// (reference contract: device_preallocated_allocator.cuh:68-175,275-279).
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include <claraparabricks/genomeworks/utils/cudautils.hpp>
#include <claraparabricks/genomeworks/utils/exceptions.hpp>

namespace claraparabricks
{
namespace genomeworks
{

namespace details
{
/// One device slab + first-fit sub-allocation. Thread-safe.
class DevicePool
{
public:
    explicit DevicePool(size_t bytes);
    ~DevicePool();
    DevicePool(const DevicePool&) = delete;
    DevicePool& operator=(const DevicePool&) = delete;

    /// nullptr when no block fits. `streams`: the streams that will touch the block.
    void* allocate(size_t bytes, const std::vector<cudaStream_t>& streams);
    void deallocate(void* p);
    int64_t largest_free_block();
    size_t capacity() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};
} // namespace details

/// Plain hipMalloc / hipFree allocator (reference: CudaMallocAllocator).
template <typename T>
class CudaMallocAllocator
{
public:
    using value_type = T;
    using pointer    = T*;

    explicit CudaMallocAllocator(cudaStream_t default_stream = 0) { (void)default_stream; }
    template <typename U>
    CudaMallocAllocator(const CudaMallocAllocator<U>&) {}

    pointer allocate(std::size_t n, const std::vector<cudaStream_t>& streams = {})
    {
        (void)streams;
        void* p = nullptr;
        if (hipMalloc(&p, n * sizeof(T)) != hipSuccess)
        {
            (void)hipGetLastError();
            throw device_memory_allocation_exception();
        }
        return static_cast<pointer>(p);
    }
    void deallocate(pointer p, std::size_t n)
    {
        (void)n;
        GW_CU_CHECK_ERR(hipFree(p));
    }
    int64_t get_size_of_largest_free_memory_block() const { return cudautils::find_largest_contiguous_device_memory_section(); }
};

/// Handle on a shared DevicePool (reference: CachingDeviceAllocator<T, DevicePreallocatedAllocator>).
template <typename T>
class CachingDeviceAllocator
{
public:
    using value_type = T;
    using pointer    = T*;

    /// Null allocator: allocate() throws.
    CachingDeviceAllocator() = default;
    /// Creates the pool: ONE device allocation of max_cached_bytes.
    explicit CachingDeviceAllocator(size_t max_cached_bytes, cudaStream_t default_stream = 0)
        : pool_(std::make_shared<details::DevicePool>(max_cached_bytes))
        , default_stream_(default_stream)
    {
    }
    template <typename U>
    CachingDeviceAllocator(const CachingDeviceAllocator<U>& rhs)
        : pool_(rhs.memory_resource())
        , default_stream_(rhs.default_stream())
    {
    }
    template <typename U>
    CachingDeviceAllocator& operator=(const CachingDeviceAllocator<U>& rhs)
    {
        pool_           = rhs.memory_resource();
        default_stream_ = rhs.default_stream();
        return *this;
    }

    pointer allocate(std::size_t n, const std::vector<cudaStream_t>& streams = {})
    {
        if (!pool_) throw device_memory_allocation_exception();
        void* p = streams.empty() ? pool_->allocate(n * sizeof(T), {default_stream_}) : pool_->allocate(n * sizeof(T), streams);
        if (p == nullptr) throw device_memory_allocation_exception();
        return static_cast<pointer>(p);
    }
    void deallocate(pointer p, std::size_t n)
    {
        (void)n;
        if (pool_) pool_->deallocate(p);
    }
    int64_t get_size_of_largest_free_memory_block() const { return pool_ ? pool_->largest_free_block() : 0; }

    std::shared_ptr<details::DevicePool> memory_resource() const { return pool_; }
    cudaStream_t default_stream() const { return default_stream_; }

private:
    std::shared_ptr<details::DevicePool> pool_;
    cudaStream_t default_stream_ = 0;
};

/// The reference's default build enables the caching allocator (gw_enable_caching_allocator=ON); so do we.
using DefaultDeviceAllocator = CachingDeviceAllocator<char>;

inline int64_t get_size_of_largest_free_memory_block(DefaultDeviceAllocator const& allocator)
{
    return allocator.get_size_of_largest_free_memory_block();
}

/// Default pool size 2 GiB as in the reference (allocator.hpp:352-362).
inline DefaultDeviceAllocator create_default_device_allocator(std::size_t max_caching_size = 2ull * 1024 * 1024 * 1024,
                                                              cudaStream_t default_stream  = 0)
{
    return DefaultDeviceAllocator(max_caching_size, default_stream);
}

} // namespace genomeworks
} // namespace claraparabricks
