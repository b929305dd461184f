// device_pool.cpp -- DevicePool (utils/allocator.hpp) and the runtime helpers of utils/cudautils.hpp.
#include <claraparabricks/genomeworks/utils/allocator.hpp>
#include <claraparabricks/genomeworks/utils/cudautils.hpp>

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace claraparabricks
{
namespace genomeworks
{

namespace cudautils
{

void gpu_assert(hipError_t code, const char* file, int line)
{
    if (code == hipSuccess) return;
    std::string msg = std::string("GPU Error:: ") + hipGetErrorString(code) + " " + file + " " + std::to_string(line);
    GW_LOG_ERROR(msg.c_str());
    std::fprintf(stderr, "%s\n", msg.c_str());
    std::abort(); // reference behaviour: cudautils.cpp:75-100
}

int64_t find_largest_contiguous_device_memory_section()
{
    // hipMalloc on MI355X can grant nearly all free memory in one piece; probe downwards from `free`.
    size_t free_b = 0, total_b = 0;
    GW_CU_CHECK_ERR(hipMemGetInfo(&free_b, &total_b));
    const size_t step = 256ull << 20;
    size_t want       = free_b > step ? free_b - step : free_b / 2;
    while (want >= step)
    {
        void* p = nullptr;
        if (hipMalloc(&p, want) == hipSuccess)
        {
            GW_CU_CHECK_ERR(hipFree(p));
            return static_cast<int64_t>(want);
        }
        (void)hipGetLastError();
        want -= step;
    }
    return 0;
}

} // namespace cudautils

namespace details
{

struct DevicePool::Impl
{
    struct Pending
    {
        size_t offset, size;
        std::vector<hipEvent_t> events;
    };
    char* base = nullptr;
    size_t size = 0;
    std::mutex mutex;
    std::map<size_t, size_t> free_blocks;                                 // offset -> size
    std::map<size_t, std::pair<size_t, std::vector<hipStream_t>>> used;   // offset -> (size, streams)
    std::vector<Pending> pending;

    void insert_free(size_t off, size_t sz)
    {
        auto it = free_blocks.insert({off, sz}).first;
        auto nx = std::next(it);
        if (nx != free_blocks.end() && it->first + it->second == nx->first)
        {
            it->second += nx->second;
            free_blocks.erase(nx);
        }
        if (it != free_blocks.begin())
        {
            auto pv = std::prev(it);
            if (pv->first + pv->second == it->first)
            {
                pv->second += it->second;
                free_blocks.erase(it);
            }
        }
    }
    // move blocks whose stream events completed (or all, when `wait`) back to the free list
    void reclaim(bool wait)
    {
        for (size_t i = 0; i < pending.size();)
        {
            bool done = true;
            for (hipEvent_t e : pending[i].events)
            {
                if (wait)
                    (void)hipEventSynchronize(e);
                else if (hipEventQuery(e) != hipSuccess)
                {
                    (void)hipGetLastError();
                    done = false;
                    break;
                }
            }
            if (done)
            {
                for (hipEvent_t e : pending[i].events) (void)hipEventDestroy(e);
                insert_free(pending[i].offset, pending[i].size);
                pending[i] = std::move(pending.back());
                pending.pop_back();
            }
            else
                ++i;
        }
    }
    void* take(size_t bytes, const std::vector<hipStream_t>& streams)
    {
        for (auto it = free_blocks.begin(); it != free_blocks.end(); ++it)
        {
            if (it->second >= bytes)
            {
                const size_t off = it->first, sz = it->second;
                free_blocks.erase(it);
                if (sz > bytes) free_blocks.insert({off + bytes, sz - bytes});
                used[off] = {bytes, streams};
                return base + off;
            }
        }
        return nullptr;
    }
};

DevicePool::DevicePool(size_t bytes)
    : impl_(new Impl)
{
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess)
    {
        (void)hipGetLastError();
        throw device_memory_allocation_exception();
    }
    impl_->base = static_cast<char*>(p);
    impl_->size = bytes;
    impl_->free_blocks.insert({0, bytes});
}

DevicePool::~DevicePool()
{
    if (impl_->base != nullptr)
    {
        impl_->reclaim(true);
        (void)hipFree(impl_->base);
    }
}

void* DevicePool::allocate(size_t bytes, const std::vector<cudaStream_t>& streams)
{
    const size_t need = (bytes + 255) & ~size_t(255); // 256-byte granules keep every block cache-line / slab aligned
    std::lock_guard<std::mutex> lock(impl_->mutex);
    impl_->reclaim(false);
    void* p = impl_->take(need ? need : 256, streams);
    if (p == nullptr && !impl_->pending.empty())
    {
        impl_->reclaim(true);
        p = impl_->take(need ? need : 256, streams);
    }
    return p;
}

void DevicePool::deallocate(void* p)
{
    if (p == nullptr) return;
    std::lock_guard<std::mutex> lock(impl_->mutex);
    const size_t off = static_cast<size_t>(static_cast<char*>(p) - impl_->base);
    auto it          = impl_->used.find(off);
    if (it == impl_->used.end()) return;
    Impl::Pending pend{off, it->second.first, {}};
    for (hipStream_t s : it->second.second)
    {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess)
        {
            (void)hipEventRecord(e, s);
            pend.events.push_back(e);
        }
    }
    impl_->used.erase(it);
    impl_->pending.push_back(std::move(pend));
    impl_->reclaim(false);
}

int64_t DevicePool::largest_free_block()
{
    std::lock_guard<std::mutex> lock(impl_->mutex);
    impl_->reclaim(false);
    size_t best = 0;
    for (const auto& kv : impl_->free_blocks) best = kv.second > best ? kv.second : best;
    return static_cast<int64_t>(best);
}

size_t DevicePool::capacity() const { return impl_->size; }

} // namespace details
} // namespace genomeworks
} // namespace claraparabricks
