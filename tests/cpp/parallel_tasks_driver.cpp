// Driver for gwhost::parallel_tasks (genomeworks_amd/host/host_common.hpp): every index exactly once, for task counts around
// the thread count; several callers at once (one gets the pool, the others run their tasks themselves); an exception from a
// task reaches the caller after the other tasks have finished, and the pool works afterwards. Prints "ok" or the failure.
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "host_common.hpp"

static int fail(const std::string& what)
{
    std::printf("FAIL %s\n", what.c_str());
    return 1;
}

int main()
{
    for (size_t threads : {size_t(1), size_t(2), size_t(8), size_t(64)})
        for (size_t n : {size_t(0), size_t(1), size_t(2), size_t(7), size_t(8), size_t(9), size_t(1000)})
        {
            std::vector<std::atomic<int>> hits(n);
            for (auto& h : hits) h = 0;
            gwhost::parallel_tasks(n, threads, [&](size_t i) { hits[i]++; });
            for (size_t i = 0; i < n; i++)
                if (hits[i] != 1) return fail("index " + std::to_string(i) + " of " + std::to_string(n) + " ran " + std::to_string(hits[i]) + " times");
        }
    // callers at once
    {
        std::atomic<long> total{0};
        std::vector<std::thread> callers;
        for (int c = 0; c < 6; c++)
            callers.emplace_back([&] {
                for (int rep = 0; rep < 200; rep++) gwhost::parallel_tasks(50, 8, [&](size_t i) { total += static_cast<long>(i); });
            });
        for (std::thread& t : callers) t.join();
        if (total != 6L * 200L * (49L * 50L / 2)) return fail("concurrent callers: sum " + std::to_string(total));
    }
    // an exception from one task
    {
        std::atomic<int> ran{0};
        bool caught = false;
        try
        {
            gwhost::parallel_tasks(100, 8, [&](size_t i) {
                ran++;
                if (i == 37) throw std::runtime_error("task 37");
            });
        }
        catch (const std::runtime_error& e)
        {
            caught = std::string(e.what()) == "task 37";
        }
        if (!caught) return fail("exception not delivered");
        if (ran != 100) return fail("tasks after the failing one were dropped: " + std::to_string(ran));
        std::atomic<int> again{0};
        gwhost::parallel_tasks(64, 8, [&](size_t) { again++; });
        if (again != 64) return fail("pool unusable after an exception");
    }
    // the serial paths keep the same contract: one thread (no helpers), max_threads == 0 (clamped to 1, must not spawn
    // without bound), and a call nested inside a task (on the caller's thread and on a pool worker)
    for (size_t threads : {size_t(0), size_t(1)})
    {
        std::atomic<int> ran{0};
        bool caught = false;
        try
        {
            gwhost::parallel_tasks(100, threads, [&](size_t i) {
                ran++;
                if (i == 3) throw std::runtime_error("task 3");
            });
        }
        catch (const std::runtime_error&)
        {
            caught = true;
        }
        if (!caught || ran != 100) return fail("serial path: caught " + std::to_string(caught) + ", ran " + std::to_string(ran));
    }
    {
        std::atomic<int> inner{0};
        gwhost::parallel_tasks(16, 8, [&](size_t) { gwhost::parallel_tasks(10, 8, [&](size_t) { inner++; }); });
        if (inner != 160) return fail("nested calls: " + std::to_string(inner));
    }
    std::printf("ok\n");
    return 0;
}
