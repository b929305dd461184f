// logging.hpp -- "[LEVEL file:line] message" logging to stderr or a file.
// API-compatible with the reference's logging/logging.hpp (initialize_logger, set_logging_level, GW_LOG_*).
#pragma once
#include <string>

namespace claraparabricks
{
namespace genomeworks
{
namespace logging
{

enum class LoggingStatus
{
    success = 0,
    cannot_open_file,
    cannot_open_stdout
};

enum LogLevel
{
    critical = 0,
    error,
    warn,
    info,
    debug
};

/// Initialise once per process. `filename` == nullptr logs to stderr.
LoggingStatus initialize_logger(LogLevel level, const char* filename = nullptr);
LoggingStatus set_logging_level(LogLevel level);
void log(LogLevel level, const char* file, int line, const char* msg);

} // namespace logging
} // namespace genomeworks
} // namespace claraparabricks

#define GW_LOG_DEBUG(msg) ::claraparabricks::genomeworks::logging::log(::claraparabricks::genomeworks::logging::LogLevel::debug, __FILE__, __LINE__, msg)
#define GW_LOG_INFO(msg) ::claraparabricks::genomeworks::logging::log(::claraparabricks::genomeworks::logging::LogLevel::info, __FILE__, __LINE__, msg)
#define GW_LOG_WARN(msg) ::claraparabricks::genomeworks::logging::log(::claraparabricks::genomeworks::logging::LogLevel::warn, __FILE__, __LINE__, msg)
#define GW_LOG_ERROR(msg) ::claraparabricks::genomeworks::logging::log(::claraparabricks::genomeworks::logging::LogLevel::error, __FILE__, __LINE__, msg)
#define GW_LOG_CRITICAL(msg) ::claraparabricks::genomeworks::logging::log(::claraparabricks::genomeworks::logging::LogLevel::critical, __FILE__, __LINE__, msg)
