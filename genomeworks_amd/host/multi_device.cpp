// multi_device.cpp -- see include/claraparabricks/genomeworks/cudapoa/multi_device.hpp.
#include <claraparabricks/genomeworks/cudapoa/multi_device.hpp>
#include <claraparabricks/genomeworks/utils/allocator.hpp>
#include <claraparabricks/genomeworks/utils/cudautils.hpp>
#include <claraparabricks/genomeworks/utils/signed_integer_utils.hpp>

#include "../../include/gwhip.h"
#include "poa_batch_impl.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <exception>
#include <map>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudapoa
{

namespace
{
// Streams / events owned for the length of a driver call, released on every exit path (device switched per resource).
struct OwnedStreams
{
    std::vector<std::pair<int32_t, cudaStream_t>> items;
    OwnedStreams()                               = default;
    OwnedStreams(const OwnedStreams&)            = delete;
    OwnedStreams& operator=(const OwnedStreams&) = delete;
    cudaStream_t create(int32_t device, bool with_priority = false, int priority = 0)
    {
        scoped_device_switch dev(device);
        cudaStream_t s = nullptr;
        if (with_priority) GW_CU_CHECK_ERR(hipStreamCreateWithPriority(&s, hipStreamDefault, priority));
        else GW_CU_CHECK_ERR(hipStreamCreate(&s));
        items.emplace_back(device, s);
        return s;
    }
    ~OwnedStreams()
    {
        for (auto& it : items)
        {
            scoped_device_switch dev(it.first);
            (void)hipStreamDestroy(it.second);
        }
    }
};
struct OwnedEvents
{
    std::vector<hipEvent_t> items;
    OwnedEvents()                              = default;
    OwnedEvents(const OwnedEvents&)            = delete;
    OwnedEvents& operator=(const OwnedEvents&) = delete;
    hipEvent_t create()
    {
        hipEvent_t e = nullptr;
        GW_CU_CHECK_ERR(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        items.push_back(e);
        return e;
    }
    ~OwnedEvents()
    {
        for (hipEvent_t e : items) (void)hipEventDestroy(e);
    }
};
// joins whatever was started, also when spawning the next thread throws
struct JoinAll
{
    std::vector<std::thread>& threads;
    ~JoinAll()
    {
        for (std::thread& t : threads)
            if (t.joinable()) t.join();
    }
};
} // namespace

namespace
{
struct SharedCursor
{
    std::mutex mutex;
    size_t next = 0;
};

// One worker: fills its batch from the cursor, runs it, stores by global index; until the windows run out.
// Every worker's Batch exists before any of them fills: the clock of the reference's multi-batch benchmark starts there
// (cudapoa/benchmarks/multi_batch.hpp: the batches are created in the constructor, process_batches() is what is timed).
struct CreationGate
{
    int32_t workers = 0;
    std::atomic<int32_t> arrived{0};
    std::atomic<bool> abandoned{false}; // a worker thread could not be started: nobody waits for it
    std::mutex m;
    std::chrono::steady_clock::time_point all_created{};
    std::chrono::steady_clock::time_point last_done{}; // the moment the last worker had stored its last results (before its Batch is destroyed)
    void done()
    {
        const auto t = std::chrono::steady_clock::now();
        std::lock_guard<std::mutex> g(m);
        if (t > last_done) last_done = t;
    }
    void arrive()
    {
        if (arrived.fetch_add(1) + 1 == workers)
        {
            std::lock_guard<std::mutex> g(m);
            all_created = std::chrono::steady_clock::now();
        }
        while (arrived.load() < workers && !abandoned.load()) std::this_thread::yield();
    }
};

void worker_loop(int32_t worker, int32_t device, cudaStream_t stream, DefaultDeviceAllocator allocator, int64_t memory,
                 const BatchConfig& batch_size, const MultiDeviceConfig& config, const std::vector<std::vector<std::string>>& windows,
                 SharedCursor& cursor, MultiDeviceOutput& out, std::atomic<int32_t>& launches, CreationGate& gate, bool& arrived)
{
    scoped_device_switch dev(device);
    std::unique_ptr<Batch> batch = create_batch(device, stream, allocator, memory, config.output_mask, batch_size, config.gap_score,
                                                config.mismatch_score, config.match_score);
    arrived = true;
    gate.arrive();
    const bool want_msa = (config.output_mask & OutputType::msa) != 0;
    // A fill stops at a whole number of device rounds: the device runs `resident` windows side by side (one wavefront per SIMD;
    // 0 = a configuration that is not one wavefront per window), and a launch of 1400 windows lasts two rounds like one of 2048
    // -- the 376 windows of its second round would have filled the SIMDs together with another worker's. Measured
    // (profiles/r06_multibatch_timeline.txt): 2048 full-band windows over two batches of 16 GB, 1400 + 648: 145 ms; 1024 + 1024: 1xx ms.
    int32_t fill_cap = INT32_MAX;
    {
        const gwhip_poa_config dc = make_device_config(batch_size, config.output_mask, config.gap_score, config.mismatch_score, config.match_score);
        const int32_t resident    = gwhip_poa_resident_windows(&dc);
        if (const char* e = std::getenv("GW_POA_FILL_ROUNDS"); e != nullptr && e[0] == '0') {}
        else if (resident > 0) fill_cap = resident; // (a batch that holds less fills up as before)
    }
    std::vector<size_t> in_batch;
    for (;;)
    {
        batch->reset();
        in_batch.clear();
        {
            // the cursor only moves under the lock: a window is taken by exactly one worker
            std::lock_guard<std::mutex> guard(cursor.mutex);
            while (cursor.next < windows.size() && batch->get_total_poas() < fill_cap)
            {
                const std::vector<std::string>& window = windows[cursor.next];
                Group group;
                group.reserve(window.size());
                for (const std::string& read : window) group.push_back(Entry{read.c_str(), nullptr, get_size<int32_t>(read)});
                std::vector<StatusType> per_read;
                const StatusType st = batch->add_poa_group(per_read, group);
                if (st == StatusType::exceeded_maximum_poas)
                {
                    if (in_batch.empty()) throw std::runtime_error("a batch of this configuration cannot hold a single window");
                    break;
                }
                out.worker_of_window[cursor.next] = worker;
                if (st == StatusType::success)
                    in_batch.push_back(cursor.next);
                else if (st == StatusType::empty_poa_group && !window.empty())
                {
                    // every read was refused after the batch opened a POA for the group (cudapoa_batch.cuh:122-150): the
                    // empty POA owns an output slot of this launch; its window reports the add status
                    out.status[cursor.next] = st;
                    in_batch.push_back(windows.size()); // placeholder slot
                }
                else
                    out.status[cursor.next] = st;
                cursor.next++;
            }
        }
        if (batch->get_total_poas() == 0)
        {
            // everything this worker took is stored: the timed region of the reference's multi-batch benchmark ends here (its
            // batches outlive process_batches(); releasing a Batch -- its pinned staging block above all -- is not part of it)
            gate.done();
            break;
        }
        batch->generate_poa();
        launches++;
        std::vector<StatusType> status;
        if (want_msa)
        {
            std::vector<std::vector<std::string>> msa;
            batch->get_msa(msa, status);
            if (msa.size() != in_batch.size()) throw std::runtime_error("MSA count does not match the windows of the batch");
            for (size_t k = 0; k < in_batch.size(); k++)
                if (in_batch[k] < windows.size())
                {
                    out.msa[in_batch[k]]    = std::move(msa[k]);
                    out.status[in_batch[k]] = status[k];
                }
        }
        else
        {
            std::vector<std::string> consensus;
            std::vector<std::vector<uint16_t>> coverage;
            batch->get_consensus(consensus, coverage, status);
            if (consensus.size() != in_batch.size()) throw std::runtime_error("consensus count does not match the windows of the batch");
            for (size_t k = 0; k < in_batch.size(); k++)
                if (in_batch[k] < windows.size())
                {
                    out.consensus[in_batch[k]] = std::move(consensus[k]);
                    out.coverage[in_batch[k]]  = std::move(coverage[k]);
                    out.status[in_batch[k]]    = status[k];
                }
        }
    }
}
} // namespace

void process_windows_multi_device(MultiDeviceOutput& out, const std::vector<std::vector<std::string>>& windows,
                                  const BatchConfig& batch_size, const MultiDeviceConfig& config)
{
    if (config.devices.empty()) throw std::invalid_argument("at least one device is needed");
    if (config.batches_per_device < 1) throw std::invalid_argument("batches_per_device has to be at least 1");
    int32_t n_devices = 0;
    GW_CU_CHECK_ERR(hipGetDeviceCount(&n_devices));
    std::map<int32_t, int32_t> entries_of_device;
    for (int32_t d : config.devices)
    {
        if (d < 0 || d >= n_devices) throw std::invalid_argument("device id out of range: " + std::to_string(d));
        entries_of_device[d]++;
    }
    const size_t n = windows.size();
    out            = MultiDeviceOutput{};
    out.status.assign(n, StatusType::success);
    out.worker_of_window.assign(n, -1);
    if (config.output_mask & OutputType::msa)
        out.msa.resize(n);
    else
    {
        out.consensus.resize(n);
        out.coverage.resize(n);
    }
    if (n == 0) return;

    // one allocator per entry of `devices`, shared by that entry's batches (as multi_batch.hpp:52-60 shares one)
    struct Group
    {
        int32_t device;
        int64_t memory;
        DefaultDeviceAllocator allocator;
    };
    std::vector<Group> groups;
    for (int32_t d : config.devices)
    {
        scoped_device_switch dev(d);
        int64_t memory = config.memory_per_device;
        if (memory < 0)
        {
            size_t free_bytes = 0, total_bytes = 0;
            GW_CU_CHECK_ERR(hipMemGetInfo(&free_bytes, &total_bytes));
            // free memory is read before any group of this device allocates: the entries naming the device share it
            memory = static_cast<int64_t>(config.memory_fraction * static_cast<double>(free_bytes)) / entries_of_device[d];
        }
        groups.push_back(Group{d, memory, DefaultDeviceAllocator()});
    }
    // allocate after all the free-memory readings
    for (Group& g : groups)
    {
        scoped_device_switch dev(g.device);
        g.allocator = DefaultDeviceAllocator(static_cast<size_t>(g.memory), nullptr);
    }

    SharedCursor cursor;
    std::atomic<int32_t> launches{0};
    const auto t_begin = std::chrono::steady_clock::now();
    // every stream exists before the first worker starts (a failing hipStreamCreate must not leave joinable threads behind)
    OwnedStreams streams;
    for (Group& g : groups)
        for (int32_t b = 0; b < config.batches_per_device; b++) streams.create(g.device);
    std::vector<std::exception_ptr> errors(groups.size() * static_cast<size_t>(config.batches_per_device));
    std::vector<std::thread> threads;
    CreationGate gate;
    gate.workers = static_cast<int32_t>(groups.size()) * config.batches_per_device;
    threads.reserve(static_cast<size_t>(gate.workers));
    {
        JoinAll join{threads};
        int32_t worker = 0;
        try
        {
            for (Group& g : groups)
                for (int32_t b = 0; b < config.batches_per_device; b++, worker++)
                {
                    cudaStream_t stream  = streams.items[static_cast<size_t>(worker)].second;
                    const int64_t memory = g.memory / config.batches_per_device;
                    threads.emplace_back([&, worker, stream, memory, device = g.device, allocator = g.allocator]() {
                        bool arrived = false;
                        try
                        {
                            worker_loop(worker, device, stream, allocator, memory, batch_size, config, windows, cursor, out, launches, gate, arrived);
                        }
                        catch (...)
                        {
                            errors[static_cast<size_t>(worker)] = std::current_exception();
                            if (!arrived) gate.arrive(); // a failed creation releases the others
                        }
                    });
                }
        }
        catch (...)
        {
            gate.abandoned.store(true); // std::thread could not start a worker: the started ones must not wait for it
            throw;
        }
    }
    const auto t_end = std::chrono::steady_clock::now();
    out.seconds      = std::chrono::duration<double>(t_end - t_begin).count();
    if (gate.arrived.load() == gate.workers)
        out.seconds_after_creation = std::chrono::duration<double>((gate.last_done > gate.all_created ? gate.last_done : t_end) - gate.all_created).count();
    out.launches = launches.load();
    for (const std::exception_ptr& e : errors)
        if (e) std::rethrow_exception(e);
}

// ---- size classes -------------------------------------------------------------------------------------------------------
void plan_size_classes(SizeClassPlan& plan, const std::vector<int32_t>& longest, const std::vector<int32_t>& reads, bool msa_flag,
                       int32_t band_width, BandMode band_mode, float adaptive_storage_factor, float graph_length_factor,
                       int32_t max_pred_distance, int32_t mismatch_score, int32_t gap_score, int32_t match_score)
{
    if (longest.size() != reads.size()) throw std::invalid_argument("one read count per window");
    plan = SizeClassPlan{};
    int32_t top = 0;
    for (int32_t l : longest) top = std::max(top, l);
    if (top <= 0) return;
    // class k: longest read in (top / 2^(k+1), top / 2^k]; everything below top / 64 shares the last class
    constexpr int kClasses = 6;
    std::vector<std::vector<int32_t>> members(kClasses);
    for (size_t w = 0; w < longest.size(); ++w)
    {
        int k = 0;
        while (k + 1 < kClasses && static_cast<int64_t>(longest[w]) * (int64_t(2) << k) <= top) ++k;
        members[static_cast<size_t>(k)].push_back(static_cast<int32_t>(w));
    }
    for (const std::vector<int32_t>& m : members)
    {
        if (m.empty()) continue;
        int32_t len = 0, most = 0;
        for (int32_t w : m)
        {
            len  = std::max(len, longest[static_cast<size_t>(w)]);
            most = std::max(most, reads[static_cast<size_t>(w)]);
        }
        // a band cannot be wider than the reads it is laid over (batch.cu:96-97): short classes keep the requested width only if they can
        const BatchConfig cfg(std::max(len, band_width), most, band_width, band_mode, adaptive_storage_factor, graph_length_factor, max_pred_distance);
        const gwhip_poa_config dc = make_device_config(cfg, static_cast<int8_t>(msa_flag ? OutputType::msa : OutputType::consensus), gap_score,
                                                       mismatch_score, match_score);
        int64_t per_poa = 0, per_matrix = 0;
        gwhip_poa_bytes_per_window(&dc, &per_poa, &per_matrix);
        plan.configs.push_back(cfg);
        plan.groups.push_back(m);
        plan.bytes_per_window.push_back(per_poa + per_matrix);
        plan.total_bytes += static_cast<int64_t>(m.size()) * (per_poa + per_matrix);
    }
}

std::vector<int32_t> size_class_admission_gates(const SizeClassPlan& plan, int32_t compute_units)
{
    const size_t classes = plan.groups.size();
    std::vector<int32_t> gate_on(classes, -1);
    const int64_t cus  = compute_units > 0 ? compute_units : 256;
    const int64_t room = cus + cus / 4;
    int64_t in_group   = 0;
    int32_t prev_last = -1, last = -1;
    for (size_t k = 0; k < classes; ++k)
    {
        if (plan.groups[k].empty()) continue;
        const int64_t w = static_cast<int64_t>(plan.groups[k].size());
        if (last >= 0 && in_group + w > room)
        {
            prev_last = last;
            in_group  = 0;
        }
        gate_on[k] = prev_last;
        in_group += w;
        last = static_cast<int32_t>(k);
    }
    return gate_on;
}

void process_windows_size_classes(MultiDeviceOutput& out, const std::vector<std::vector<std::string>>& windows,
                                  const SizeClassPlan& plan, int32_t device, int64_t memory_budget, int8_t output_mask,
                                  int16_t gap_score, int16_t mismatch_score, int16_t match_score, double* compute_seconds)
{
    const size_t n = windows.size();
    out            = MultiDeviceOutput{};
    out.status.assign(n, StatusType::success);
    out.worker_of_window.assign(n, -1);
    const bool want_msa = (output_mask & OutputType::msa) != 0;
    if (want_msa)
        out.msa.resize(n);
    else
    {
        out.consensus.resize(n);
        out.coverage.resize(n);
    }
    if (compute_seconds) *compute_seconds = 0;
    const size_t classes = plan.configs.size();
    if (n == 0 || classes == 0) return;
    scoped_device_switch dev(device);
    size_t active_classes = 0;
    for (size_t k = 0; k < classes; ++k) active_classes += plan.groups[k].empty() ? 0 : 1;
    if (active_classes == 0) return;
    // every class gets its planned bytes (+ slack for alignment and one spare window), scaled down when the plan exceeds the budget
    const int64_t slack_total = static_cast<int64_t>(classes) * (int64_t(64) << 20);
    std::vector<int64_t> share(classes);
    double scale = 1.0;
    {
        int64_t want = slack_total;
        for (size_t k = 0; k < classes; ++k) want += (static_cast<int64_t>(plan.groups[k].size()) + 1) * plan.bytes_per_window[k];
        if (want > memory_budget) scale = static_cast<double>(memory_budget - slack_total) / static_cast<double>(want - slack_total);
        for (size_t k = 0; k < classes; ++k)
        {
            const int64_t planned = (static_cast<int64_t>(plan.groups[k].size()) + 1) * plan.bytes_per_window[k];
            share[k] = std::max<int64_t>(2 * plan.bytes_per_window[k], static_cast<int64_t>(scale * static_cast<double>(planned))) + (int64_t(64) << 20);
        }
    }
    std::atomic<int32_t> launches{0}, filled{0};
    // The classes' first launches are submitted in plan order (longest reads first) and on streams whose priority falls in
    // the same order: a window is one chain of dependent steps on one CU, the set lasts at least as long as its heaviest
    // window, so that window's class must own its CUs from the first moment instead of queueing behind hundreds of light
    // blocks that happened to be submitted a millisecond earlier.
    std::atomic<int32_t> launch_turn{0};
    std::atomic<int64_t> results_done_ns{0}; // when the last class handed over its last results, relative to t_begin
    std::vector<int32_t> launch_rank(classes, 0);
    {
        int32_t rank = 0;
        for (size_t k = 0; k < classes; ++k)
            if (!plan.groups[k].empty()) launch_rank[k] = rank++;
    }
    int priority_least = 0, priority_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&priority_least, &priority_greatest);
    // Admission by residency: a window occupies a CU for its whole life, so the device holds about one window per CU at
    // a time. Classes are admitted in plan order while their windows (a quarter more than there are CUs: the first to
    // finish make room at once) fit; the next group of classes is gated, on the device, on the end of the lightest class
    // of the group before it. Admitting everything at once only makes the long chains of the heavy classes queue for CUs
    // behind light windows -- and those chains are what the set waits for at the end.
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    const std::vector<int32_t> gate_on = size_class_admission_gates(plan, cus);
    OwnedEvents class_events; // released on every exit path
    for (size_t k = 0; k < classes; ++k) class_events.create();
    const std::vector<hipEvent_t>& class_done = class_events.items;
    // streams of the class workers, created up front for the same reason
    OwnedStreams class_streams;
    std::vector<cudaStream_t> stream_of(classes, nullptr);
    for (size_t k = 0; k < classes; ++k)
        if (!plan.groups[k].empty())
            // numerically lower = more urgent; the range is narrow (three levels on this hardware), later classes share the last
            stream_of[k] = class_streams.create(device, true, std::min(priority_least, priority_greatest + launch_rank[k]));
    std::mutex start_mutex;
    std::chrono::steady_clock::time_point compute_begin{}, fill_begin{};
    std::atomic<int32_t> created{0}, create_arrived{0};
    std::atomic<bool> spawn_failed{false}; // a worker thread could not be started: the barriers below must not wait for it
    std::vector<std::exception_ptr> errors(classes);
    std::vector<std::thread> threads;
    JoinAll join_on_exit{threads};
    const auto t_begin = std::chrono::steady_clock::now();
    threads.reserve(classes);
    try
    {
    for (size_t k = 0; k < classes; ++k)
    {
        if (plan.groups[k].empty()) continue;
        threads.emplace_back([&, k]() {
            bool counted = false;
            auto arrive  = [&] { // first fill of this worker is done (or it failed): the compute clock starts when all have arrived
                if (counted) return;
                counted = true;
                if (filled.fetch_add(1) + 1 == static_cast<int32_t>(active_classes))
                {
                    std::lock_guard<std::mutex> g(start_mutex);
                    compute_begin = std::chrono::steady_clock::now();
                }
                while (filled.load() < static_cast<int32_t>(active_classes) && !spawn_failed.load()) std::this_thread::yield();
            };
            bool counted_created = false;
            auto arrive_created  = [&](bool ok) { // this worker's Batch exists (or its creation failed): barrier before any filling
                if (counted_created) return;
                counted_created = true;
                if (ok) created.fetch_add(1);
                if (create_arrived.fetch_add(1) + 1 == static_cast<int32_t>(active_classes))
                {
                    std::lock_guard<std::mutex> g(start_mutex);
                    fill_begin = std::chrono::steady_clock::now();
                }
                while (create_arrived.load() < static_cast<int32_t>(active_classes) && !spawn_failed.load()) std::this_thread::yield();
            };
            try
            {
                scoped_device_switch d(device);
                cudaStream_t stream = stream_of[k];
                {
                    DefaultDeviceAllocator allocator(static_cast<size_t>(share[k]), stream);
                    std::unique_ptr<Batch> batch = create_batch(device, stream, allocator, share[k], output_mask, plan.configs[k], gap_score,
                                                                mismatch_score, match_score);
                    // every class's Batch exists: the fill-inclusive clock (the reference's multi-batch region,
                    // cudapoa/benchmarks/multi_batch.hpp:72-177, creates all batches first and times all of the filling)
                    // starts, and no class fills before that point -- a class that was created early would otherwise do
                    // its filling outside the clock
                    arrive_created(true);
                    // heaviest windows first: blocks are dispatched in window order, and a class that does not fit the free CUs
                    // at once should not keep its long chains for the end
                    std::vector<int32_t> mine = plan.groups[k];
                    {
                        auto bases = [&](int32_t w) {
                            int64_t b = 0;
                            for (const std::string& read : windows[static_cast<size_t>(w)]) b += static_cast<int64_t>(read.size());
                            return b;
                        };
                        std::vector<std::pair<int64_t, int32_t>> keyed;
                        keyed.reserve(mine.size());
                        for (int32_t w : mine) keyed.emplace_back(-bases(w), w);
                        std::stable_sort(keyed.begin(), keyed.end());
                        for (size_t i = 0; i < mine.size(); i++) mine[i] = keyed[i].second;
                    }
                    bool first_launch = true;
                    size_t next = 0;
                    std::vector<size_t> in_batch;
                    while (next < mine.size())
                    {
                        batch->reset();
                        in_batch.clear();
                        while (next < mine.size())
                        {
                            const size_t w = static_cast<size_t>(mine[next]);
                            Group group;
                            for (const std::string& read : windows[w]) group.push_back(Entry{read.c_str(), nullptr, get_size<int32_t>(read)});
                            std::vector<StatusType> per_read;
                            const StatusType st = batch->add_poa_group(per_read, group);
                            if (st == StatusType::exceeded_maximum_poas)
                            {
                                if (in_batch.empty()) throw std::runtime_error("a batch of this size class cannot hold a single window");
                                break;
                            }
                            out.worker_of_window[w] = static_cast<int32_t>(k);
                            if (st == StatusType::success)
                                in_batch.push_back(w);
                            else
                            {
                                out.status[w] = st;
                                if (st == StatusType::empty_poa_group && !windows[w].empty()) in_batch.push_back(n); // its empty POA owns a slot
                            }
                            next++;
                        }
                        arrive();
                        if (first_launch) // submission order of the classes' first launches
                        {
                            while (launch_turn.load() < launch_rank[k] && !spawn_failed.load()) std::this_thread::yield();
                            // (the gate's event was recorded before its class passed the turn on)
                            if (gate_on[k] >= 0) GW_CU_CHECK_ERR(hipStreamWaitEvent(stream, class_done[static_cast<size_t>(gate_on[k])], 0));
                        }
                        const bool trace = std::getenv("GW_SIZE_CLASS_TRACE") != nullptr; // debugging: host-side timeline on stderr
                        auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
                        if (trace) std::fprintf(stderr, "[size classes] class %zu: generate_poa() called at %.1f ms\n", k, since());
                        if (batch->get_total_poas() > 0) batch->generate_poa();
                        if (trace) std::fprintf(stderr, "[size classes] class %zu: generate_poa() returned at %.1f ms\n", k, since());
                        // the gate of later class groups: (re-)recorded behind EVERY launch of this class, so a waiter that
                        // arrives late waits for the class's last submitted launch, not only for its first
                        GW_CU_CHECK_ERR(hipEventRecord(class_done[k], stream));
                        if (first_launch)
                        {
                            first_launch = false;
                            launch_turn.fetch_add(1);
                        }
                        if (batch->get_total_poas() == 0) continue;
                        launches++;
                        std::vector<StatusType> status;
                        if (want_msa)
                        {
                            std::vector<std::vector<std::string>> msa;
                            if (trace)
                            {
                                GW_CU_CHECK_ERR(hipStreamSynchronize(stream));
                                std::fprintf(stderr, "[size classes] class %zu: kernels done at %.1f ms\n", k, since());
                            }
                            batch->get_msa(msa, status);
                            if (trace) std::fprintf(stderr, "[size classes] class %zu: get_msa() returned at %.1f ms\n", k, since());
                            for (size_t i = 0; i < in_batch.size(); i++)
                                if (in_batch[i] < n)
                                {
                                    out.msa[in_batch[i]]    = std::move(msa[i]);
                                    out.status[in_batch[i]] = status[i];
                                }
                        }
                        else
                        {
                            std::vector<std::string> consensus;
                            std::vector<std::vector<uint16_t>> coverage;
                            batch->get_consensus(consensus, coverage, status);
                            for (size_t i = 0; i < in_batch.size(); i++)
                                if (in_batch[i] < n)
                                {
                                    out.consensus[in_batch[i]] = std::move(consensus[i]);
                                    out.coverage[in_batch[i]]  = std::move(coverage[i]);
                                    out.status[in_batch[i]]    = status[i];
                                }
                        }
                        // the compute clock stops when the last results have been handed over: releasing the slabs (hundreds
                        // of GB for a long-read set, most of a second) is not part of generate_poa() + get_msa()
                        {
                            const int64_t now_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count();
                            int64_t seen         = results_done_ns.load();
                            while (seen < now_ns && !results_done_ns.compare_exchange_weak(seen, now_ns)) {}
                        }
                    }
                    arrive();
                }
            }
            catch (...)
            {
                errors[k] = std::current_exception();
                arrive_created(false); // releases the creation barrier on the error path too
                arrive();
                // a class that failed before its first launch still passes the turn on
                while (launch_turn.load() < launch_rank[k] && !spawn_failed.load()) std::this_thread::yield();
                int32_t mine_turn = launch_rank[k];
                launch_turn.compare_exchange_strong(mine_turn, launch_rank[k] + 1);
            }
        });
    }
    }
    catch (...)
    {
        // std::thread could not start a worker (std::system_error): the started ones must not spin at the barriers for it;
        // JoinAll joins them on the way out and the error surfaces
        spawn_failed.store(true);
        throw;
    }
    for (std::thread& t : threads) t.join();
    const auto t_end = std::chrono::steady_clock::now();
    out.seconds      = std::chrono::duration<double>(t_end - t_begin).count();
    if (compute_seconds)
    {
        const auto t_results = t_begin + std::chrono::nanoseconds(results_done_ns.load());
        *compute_seconds     = results_done_ns.load() > 0 ? std::chrono::duration<double>(t_results - compute_begin).count()
                                                          : std::chrono::duration<double>(t_end - compute_begin).count();
    }
    out.launches = launches.load();
    if (created.load() == static_cast<int32_t>(active_classes) && results_done_ns.load() > 0)
        out.seconds_after_creation = std::chrono::duration<double>(t_begin + std::chrono::nanoseconds(results_done_ns.load()) - fill_begin).count();
    for (const std::exception_ptr& e : errors)
        if (e) std::rethrow_exception(e);
}

} // namespace cudapoa
} // namespace genomeworks
} // namespace claraparabricks
