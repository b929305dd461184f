// exceptions.hpp -- exception types thrown by the host library (reference: utils/exceptions.hpp).
#pragma once
#include <exception>

namespace claraparabricks
{
namespace genomeworks
{

/// Thrown when device memory cannot be obtained from the allocator.
class device_memory_allocation_exception : public std::exception
{
public:
    const char* what() const noexcept override { return "Could not allocate device memory!"; }
};

} // namespace genomeworks
} // namespace claraparabricks
