// cub/util_type.cuh -- TEST INFRASTRUCTURE (oracle/simt): cub::DoubleBuffer
#pragma once
namespace cub
{
template <typename T>
struct DoubleBuffer
{
    T* d_buffers[2];
    int selector;
    DoubleBuffer() : d_buffers{nullptr, nullptr}, selector(0) {}
    DoubleBuffer(T* current, T* alternate) : d_buffers{current, alternate}, selector(0) {}
    T* Current() { return d_buffers[selector]; }
    T* Alternate() { return d_buffers[selector ^ 1]; }
};
} // namespace cub
