// host_common.hpp -- small shared helpers of the host library.
#pragma once
#include <cstddef>
#include <functional>
#include <string>

namespace gwhost
{
inline std::string& last_error()
{
    static thread_local std::string e;
    return e;
}
inline void set_last_error(const std::string& s) { last_error() = s; }

/// Runs task(0) .. task(n_tasks - 1), each exactly once, on the calling thread and up to max_threads - 1 workers of a
/// process-wide pool (created on first use, parked on a condition variable in between; runtime.cpp). The pool serves one
/// caller at a time: a second caller that arrives meanwhile runs its tasks itself. An exception thrown by a task is
/// rethrown here once every task has finished. The un-reversal of a batch's results takes a few hundred microseconds; starting and joining
/// std::threads for it cost as much as the work.
void parallel_tasks(size_t n_tasks, size_t max_threads, const std::function<void(size_t)>& task);
} // namespace gwhost
