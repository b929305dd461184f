// runtime.cpp -- device/runtime helpers and synthetic input generators behind include/gw_capi.h.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <random>
#include <string>

#include <claraparabricks/genomeworks/utils/genomeutils.hpp>

#include "../../include/gw_capi.h"
#include "host_common.hpp"

#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "alignment_impl.hpp"

namespace gw = claraparabricks::genomeworks;

// ---- process-wide worker pool (host_common.hpp) --------------------------------------------------------------------
namespace gwhost
{
namespace
{
thread_local bool inside_pool_job = false; // this thread is running tasks of a parallel_tasks call (caller or worker)
struct WorkerPool
{
    std::mutex m;                 // guards the fields below
    std::condition_variable wake; // workers: a job was posted
    std::condition_variable idle; // caller: the last worker left the job
    std::mutex one_caller;        // held for the duration of a job
    const std::function<void(size_t)>* task = nullptr;
    size_t n_tasks = 0, wanted = 0, joined = 0, inside = 0;
    uint64_t job = 0;
    std::atomic<size_t> next{0};
    std::exception_ptr error;
    size_t n_workers = 0;

    void run_tasks(const std::function<void(size_t)>& t, size_t n)
    {
        for (;;)
        {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) return;
            try
            {
                t(i);
            }
            catch (...)
            {
                std::lock_guard<std::mutex> lock(m);
                if (!error) error = std::current_exception();
            }
        }
    }
    void worker()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lock(m);
        for (;;)
        {
            wake.wait(lock, [&] { return job != seen && joined < wanted; });
            seen = job;
            joined++;
            inside++;
            const std::function<void(size_t)>* t = task;
            const size_t n                       = n_tasks;
            lock.unlock();
            inside_pool_job = true;
            run_tasks(*t, n);
            inside_pool_job = false;
            lock.lock();
            if (--inside == 0) idle.notify_all();
        }
    }
    void ensure_workers(size_t n) // under m
    {
        while (n_workers < n)
        {
            std::thread([this] { worker(); }).detach(); // parked for the life of the process; never joined (no work at exit)
            n_workers++;
        }
    }
};
WorkerPool& pool()
{
    static WorkerPool* p = new WorkerPool; // deliberately not destroyed: detached workers may outlive static destructors
    return *p;
}
} // namespace

void parallel_tasks(size_t n_tasks, size_t max_threads, const std::function<void(size_t)>& task)
{
    if (n_tasks == 0) return;
    const size_t hw      = std::max(1u, std::thread::hardware_concurrency());
    const size_t helpers = std::min(std::min(std::max<size_t>(max_threads, 1), hw), n_tasks) - 1;
    // a task that calls parallel_tasks itself (on the caller's thread or on a pool worker) runs its tasks serially: the pool
    // is busy with the outer call, and try_lock on a mutex this thread already holds would be undefined behaviour
    struct InsideFlag
    {
        bool& f;
        bool prev;
        explicit InsideFlag(bool& flag) : f(flag), prev(flag) { f = true; }
        ~InsideFlag() { f = prev; }
    };
    auto serial = [&]() {
        // same contract as the pooled path: every task runs, the first exception is rethrown afterwards
        InsideFlag in(inside_pool_job);
        std::exception_ptr first;
        for (size_t i = 0; i < n_tasks; i++)
        {
            try
            {
                task(i);
            }
            catch (...)
            {
                if (!first) first = std::current_exception();
            }
        }
        if (first) std::rethrow_exception(first);
    };
    if (helpers == 0 || inside_pool_job)
    {
        serial();
        return;
    }
    WorkerPool& p = pool();
    std::unique_lock<std::mutex> caller(p.one_caller, std::try_to_lock);
    if (!caller.owns_lock())
    {
        serial();
        return;
    }
    InsideFlag in(inside_pool_job);
    {
        std::lock_guard<std::mutex> lock(p.m);
        p.ensure_workers(helpers);
        p.task    = &task;
        p.n_tasks = n_tasks;
        p.wanted  = helpers;
        p.joined  = 0;
        p.error   = nullptr;
        p.next.store(0, std::memory_order_relaxed);
        p.job++;
    }
    p.wake.notify_all();
    p.run_tasks(task, n_tasks);
    std::exception_ptr err;
    {
        // every index has been claimed; wait for the workers that are still inside one, and stop latecomers from joining
        std::unique_lock<std::mutex> lock(p.m);
        p.wanted = p.joined;
        p.idle.wait(lock, [&] { return p.inside == 0; });
        p.task = nullptr;
        err    = p.error;
    }
    if (err) std::rethrow_exception(err);
}
} // namespace gwhost

// ---- pinned host staging cache (alignment_impl.hpp) ----------------------------------------------------------------
namespace claraparabricks
{
namespace genomeworks
{
namespace cudaaligner
{
namespace
{
struct PinnedCache
{
    std::mutex m;
    std::vector<std::pair<char*, size_t>> free_list; // deliberately never released at exit (no runtime calls from static destructors)
};
PinnedCache& pinned_cache()
{
    static PinnedCache* c = new PinnedCache;
    return *c;
}
constexpr size_t kPinnedKeep = 16;
} // namespace

char* pinned_acquire(size_t bytes, size_t* capacity)
{
    PinnedCache& c = pinned_cache();
    {
        std::lock_guard<std::mutex> lock(c.m);
        size_t best = c.free_list.size();
        for (size_t i = 0; i < c.free_list.size(); ++i)
            if (c.free_list[i].second >= bytes && (best == c.free_list.size() || c.free_list[i].second < c.free_list[best].second)) best = i;
        if (best != c.free_list.size() && c.free_list[best].second <= 4 * bytes + (1u << 20))
        {
            auto hit = c.free_list[best];
            c.free_list.erase(c.free_list.begin() + static_cast<long>(best));
            *capacity = hit.second;
            return hit.first;
        }
    }
    size_t cap = 1u << 16;
    while (cap < bytes) cap += cap / 2 > (size_t(64) << 20) ? (size_t(64) << 20) : cap; // doubles up to 128 MiB, then +64 MiB steps
    void* p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess)
    {
        (void)hipGetLastError();
        throw std::bad_alloc();
    }
    *capacity = cap;
    return static_cast<char*>(p);
}

void pinned_release(char* p, size_t capacity)
{
    if (p == nullptr) return;
    PinnedCache& c = pinned_cache();
    char* drop     = nullptr;
    {
        std::lock_guard<std::mutex> lock(c.m);
        c.free_list.emplace_back(p, capacity);
        if (c.free_list.size() > kPinnedKeep)
        {
            size_t smallest = 0;
            for (size_t i = 1; i < c.free_list.size(); ++i)
                if (c.free_list[i].second < c.free_list[smallest].second) smallest = i;
            drop = c.free_list[smallest].first;
            c.free_list.erase(c.free_list.begin() + static_cast<long>(smallest));
        }
    }
    if (drop != nullptr) (void)hipHostFree(drop);
    // a cap in bytes as well (cudapoa batches stage GBs): beyond it the LARGEST idle buffers go back to the system
    static const size_t byte_cap = [] {
        const char* e = std::getenv("GW_PINNED_CACHE_BYTES");
        return e ? static_cast<size_t>(std::strtoull(e, nullptr, 10)) : (size_t(24) << 30);
    }();
    for (;;)
    {
        char* big = nullptr;
        {
            std::lock_guard<std::mutex> lock(c.m);
            size_t total = 0, largest = 0;
            for (size_t i = 0; i < c.free_list.size(); ++i)
            {
                total += c.free_list[i].second;
                if (c.free_list[i].second > c.free_list[largest].second) largest = i;
            }
            if (total <= byte_cap || c.free_list.empty()) break;
            big = c.free_list[largest].first;
            c.free_list.erase(c.free_list.begin() + static_cast<long>(largest));
        }
        (void)hipHostFree(big);
    }
}

// ---- pageable host buffers, recycled the same way (the views of a million-pair batch: a recycled buffer is already mapped) ----
namespace
{
PinnedCache& host_cache()
{
    static PinnedCache* c = new PinnedCache;
    return *c;
}
constexpr size_t kHostKeep = 4;
} // namespace

char* host_acquire(size_t bytes, size_t* capacity)
{
    PinnedCache& c = host_cache();
    {
        std::lock_guard<std::mutex> lock(c.m);
        for (size_t i = 0; i < c.free_list.size(); ++i)
            if (c.free_list[i].second >= bytes && c.free_list[i].second <= 4 * bytes + (1u << 20))
            {
                auto hit = c.free_list[i];
                c.free_list.erase(c.free_list.begin() + static_cast<long>(i));
                *capacity = hit.second;
                return hit.first;
            }
    }
    const size_t cap = std::max<size_t>(bytes, 64);
    void* p          = std::malloc(cap);
    if (p == nullptr) throw std::bad_alloc();
    *capacity = cap;
    return static_cast<char*>(p);
}

void host_release(char* p, size_t capacity)
{
    if (p == nullptr) return;
    if (capacity < (size_t(1) << 20)) // small buffers are not worth keeping
    {
        std::free(p);
        return;
    }
    PinnedCache& c = host_cache();
    char* drop     = nullptr;
    {
        std::lock_guard<std::mutex> lock(c.m);
        c.free_list.emplace_back(p, capacity);
        if (c.free_list.size() > kHostKeep)
        {
            drop = c.free_list.front().first;
            c.free_list.erase(c.free_list.begin());
        }
    }
    if (drop != nullptr) std::free(drop);
}
} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks

extern "C" {

const char* gw_last_error(void) { return gwhost::last_error().c_str(); }

int gw_device_count(int* count) { return (int)hipGetDeviceCount(count); }
int gw_set_device(int device) { return (int)hipSetDevice(device); }
int gw_get_device(int* device) { return (int)hipGetDevice(device); }
int gw_mem_info(size_t* free_bytes, size_t* total_bytes) { return (int)hipMemGetInfo(free_bytes, total_bytes); }
int gw_stream_create(void** stream) { return (int)hipStreamCreate((hipStream_t*)stream); }
int gw_stream_sync(void* stream) { return (int)hipStreamSynchronize((hipStream_t)stream); }
int gw_stream_destroy(void* stream) { return (int)hipStreamDestroy((hipStream_t)stream); }

int64_t gw_generate_window(uint32_t seed, int32_t backbone_len, int32_t n_reads, int32_t max_mut, int32_t max_ins,
                           int32_t max_del, char* out, int64_t out_cap, int32_t* lens)
{
    try
    {
        std::minstd_rand rng(seed);
        const std::string backbone = gw::genomeutils::generate_random_genome(backbone_len, rng);
        int64_t off                = 0;
        for (int32_t i = 0; i < n_reads; i++)
        {
            const std::string r = (i == 0) ? backbone
                                           : gw::genomeutils::generate_random_sequence(backbone, rng, max_mut, max_ins, max_del);
            if (off + (int64_t)r.size() > out_cap) return -1;
            std::memcpy(out + off, r.data(), r.size());
            lens[i] = (int32_t)r.size();
            off += (int64_t)r.size();
        }
        return off;
    }
    catch (const std::exception& e)
    {
        gwhost::set_last_error(e.what());
        return -2;
    }
}

int64_t gw_generate_pairs(uint32_t seed, int32_t n_pairs, int32_t len, int32_t max_mut, int32_t max_ins,
                          int32_t max_del, char* out, int64_t out_cap, int32_t* qlens, int32_t* tlens)
{
    try
    {
        std::minstd_rand rng(seed);
        int64_t off = 0;
        for (int32_t i = 0; i < n_pairs; i++)
        {
            const std::string q = gw::genomeutils::generate_random_genome(len, rng);
            const std::string t = gw::genomeutils::generate_random_sequence(q, rng, max_mut, max_ins, max_del);
            if (off + (int64_t)(q.size() + t.size()) > out_cap) return -1;
            std::memcpy(out + off, q.data(), q.size());
            off += (int64_t)q.size();
            std::memcpy(out + off, t.data(), t.size());
            off += (int64_t)t.size();
            qlens[i] = (int32_t)q.size();
            tlens[i] = (int32_t)t.size();
        }
        return off;
    }
    catch (const std::exception& e)
    {
        gwhost::set_last_error(e.what());
        return -2;
    }
}

int64_t gw_generate_random_length_pairs(uint32_t seed, int32_t n_pairs, int32_t max_len, char* out, int64_t out_cap, int32_t* tlens,
                                        int32_t* qlens)
{
    try
    {
        std::minstd_rand rng(seed);
        std::uniform_int_distribution<int> random_length(0, max_len);
        int64_t off = 0;
        for (int32_t i = 0; i < n_pairs; i++)
        {
            const std::string t = gw::genomeutils::generate_random_genome(random_length(rng), rng);
            const int n         = static_cast<int>(t.size());
            const std::string q = gw::genomeutils::generate_random_sequence(t, rng, n, n, n);
            if (off + (int64_t)(q.size() + t.size()) > out_cap) return -1;
            std::memcpy(out + off, t.data(), t.size());
            off += (int64_t)t.size();
            std::memcpy(out + off, q.data(), q.size());
            off += (int64_t)q.size();
            tlens[i] = (int32_t)t.size();
            qlens[i] = (int32_t)q.size();
        }
        return off;
    }
    catch (const std::exception& e)
    {
        gwhost::set_last_error(e.what());
        return -2;
    }
}

} // extern "C"
