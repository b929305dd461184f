// ref_support.cpp -- TEST INFRASTRUCTURE (oracle/simt): the four functions of the reference's common library that its cudapoa
// sources call and whose own definitions need CUDA (cudautils.cpp) or spdlog (logging.cpp, an absent submodule).
#include <claraparabricks/genomeworks/logging/logging.hpp>
#include <claraparabricks/genomeworks/utils/cudautils.hpp>

#include <cstdio>
#include <cstdlib>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudautils
{
void print_error_and_abort(cudaError_t code, const char* file, int line)
{
    std::fprintf(stderr, "simt stub: CUDA error %d at %s:%d\n", static_cast<int>(code), file, line);
    std::abort();
}
std::size_t find_largest_contiguous_device_memory_section() { return std::size_t(2) << 30; }
} // namespace cudautils
namespace logging
{
void initialize_logger(LogLevel, const char*) {}
void log(LogLevel level, const char* file, int line, const char* msg)
{
    if (std::getenv("SIMT_REF_LOG") != nullptr) std::fprintf(stderr, "[ref log %d] %s:%d %s\n", static_cast<int>(level), file, line, msg);
}
} // namespace logging
} // namespace genomeworks
} // namespace claraparabricks
