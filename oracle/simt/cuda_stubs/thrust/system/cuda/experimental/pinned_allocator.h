// thrust/system/cuda/experimental/pinned_allocator.h -- TEST INFRASTRUCTURE (oracle/simt): pinned host memory is host memory
#pragma once
#include <memory>
namespace thrust
{
namespace system
{
namespace cuda
{
namespace experimental
{
template <typename T>
struct pinned_allocator : std::allocator<T>
{
    pinned_allocator() = default;
    template <typename U>
    pinned_allocator(const pinned_allocator<U>&) {}
    template <typename U>
    struct rebind
    {
        typedef pinned_allocator<U> other;
    };
};
} // namespace experimental
} // namespace cuda
} // namespace system
} // namespace thrust
