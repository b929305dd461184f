// logging.cpp -- implementation of logging/logging.hpp.
#include <claraparabricks/genomeworks/logging/logging.hpp>

#include <atomic>
#include <cstdio>
#include <mutex>

namespace claraparabricks
{
namespace genomeworks
{
namespace logging
{
namespace
{
std::mutex g_mutex;
FILE* g_sink                = nullptr; // nullptr = stderr
std::atomic<int> g_level{static_cast<int>(LogLevel::error)};
bool g_initialized = false;
const char* level_name(LogLevel l)
{
    switch (l)
    {
    case LogLevel::critical: return "CRITICAL";
    case LogLevel::error: return "ERROR";
    case LogLevel::warn: return "WARN";
    case LogLevel::info: return "INFO";
    default: return "DEBUG";
    }
}
} // namespace

LoggingStatus initialize_logger(LogLevel level, const char* filename)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    g_level = static_cast<int>(level);
    if (g_initialized) return LoggingStatus::success; // first initialisation wins, like the reference
    g_initialized = true;
    if (filename != nullptr)
    {
        g_sink = std::fopen(filename, "a");
        if (g_sink == nullptr) return LoggingStatus::cannot_open_file;
    }
    return LoggingStatus::success;
}

LoggingStatus set_logging_level(LogLevel level)
{
    g_level = static_cast<int>(level);
    return LoggingStatus::success;
}

void log(LogLevel level, const char* file, int line, const char* msg)
{
    if (static_cast<int>(level) > g_level.load()) return;
    std::lock_guard<std::mutex> lock(g_mutex);
    FILE* out = g_sink ? g_sink : stderr;
    std::fprintf(out, "[%s %s:%d] %s\n", level_name(level), file, line, msg);
    std::fflush(out);
}

} // namespace logging
} // namespace genomeworks
} // namespace claraparabricks
