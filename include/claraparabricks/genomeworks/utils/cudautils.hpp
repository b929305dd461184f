// cudautils.hpp -- runtime helpers named as in the reference's utils/cudautils.hpp, implemented on the HIP
// runtime for MI355X. The only CUDA-named symbol we define is the stream alias below: the public
// cudapoa / cudaaligner signatures spell `cudaStream_t` (batch.hpp:176-204, aligner.hpp:138,183-219), so one
// alias keeps callers source-compatible. No CUDA headers, no dual path: everything here is HIP.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <memory>
#include <string>

#include <claraparabricks/genomeworks/logging/logging.hpp>

/// The single compatibility alias (see file comment).
using cudaStream_t = hipStream_t;

namespace claraparabricks
{
namespace genomeworks
{
namespace cudautils
{

/// Logs and aborts on a HIP runtime failure (reference behaviour: cudautils.cpp:75-100).
void gpu_assert(hipError_t code, const char* file, int line);

/// Power-of-two round-up used by BatchConfig (observable through the x4 / x128 roundings).
template <typename IntType, int32_t boundary>
inline IntType align(const IntType& value)
{
    static_assert((boundary & (boundary - 1)) == 0, "Boundary for align must be power of 2");
    return (value + boundary - 1) & ~(boundary - 1);
}

/// Largest allocation the device would currently grant, minus a safety margin.
int64_t find_largest_contiguous_device_memory_section();

template <typename Type>
Type get_value_from_device(const Type* d_ptr, cudaStream_t stream = 0)
{
    Type val;
    gpu_assert(hipMemcpyAsync(&val, d_ptr, sizeof(Type), hipMemcpyDeviceToHost, stream), __FILE__, __LINE__);
    gpu_assert(hipStreamSynchronize(stream), __FILE__, __LINE__);
    return val;
}

template <typename Type>
void set_device_value(Type* d_ptr, const Type& value, cudaStream_t stream = 0)
{
    gpu_assert(hipMemcpyAsync(d_ptr, &value, sizeof(Type), hipMemcpyHostToDevice, stream), __FILE__, __LINE__);
    gpu_assert(hipStreamSynchronize(stream), __FILE__, __LINE__);
}

template <typename Type>
void device_copy_n_async(const Type* src, size_t n, Type* dst, hipMemcpyKind kind, cudaStream_t stream)
{
    gpu_assert(hipMemcpyAsync(dst, src, n * sizeof(Type), kind, stream), __FILE__, __LINE__);
}

} // namespace cudautils

#define GW_CU_CHECK_ERR(ans) ::claraparabricks::genomeworks::cudautils::gpu_assert((ans), __FILE__, __LINE__)

/// RAII: make `device_id` current for the scope, restore the previous device afterwards.
class scoped_device_switch
{
public:
    explicit scoped_device_switch(int32_t device_id)
    {
        GW_CU_CHECK_ERR(hipGetDevice(&previous_));
        GW_CU_CHECK_ERR(hipSetDevice(device_id));
    }
    ~scoped_device_switch() { (void)hipSetDevice(previous_); }
    scoped_device_switch(const scoped_device_switch&) = delete;
    scoped_device_switch& operator=(const scoped_device_switch&) = delete;

private:
    int32_t previous_ = 0;
};

/// Owning stream handle (reference: CudaStream / make_cuda_stream, cudautils.hpp:186-220).
class CudaStream
{
public:
    CudaStream(const CudaStream&) = delete;
    CudaStream& operator=(const CudaStream&) = delete;
    CudaStream(CudaStream&& rhs) noexcept : stream_(rhs.stream_) { rhs.stream_ = nullptr; }
    CudaStream& operator=(CudaStream&& rhs) noexcept
    {
        std::swap(stream_, rhs.stream_);
        return *this;
    }
    ~CudaStream()
    {
        if (stream_ != nullptr) (void)hipStreamDestroy(stream_);
    }
    cudaStream_t get() const { return stream_; }
    friend CudaStream make_cuda_stream();

private:
    explicit CudaStream(cudaStream_t s) : stream_(s) {}
    cudaStream_t stream_ = nullptr;
};

inline CudaStream make_cuda_stream()
{
    cudaStream_t s = nullptr;
    GW_CU_CHECK_ERR(hipStreamCreate(&s));
    return CudaStream(s);
}

/// Profiling range (reference: GW_NVTX_RANGE). Compiled to nothing unless GW_PROFILING is defined.
#ifdef GW_PROFILING
struct roctx_range
{
    explicit roctx_range(const char* label);
    ~roctx_range();
};
#define GW_NVTX_RANGE(varname, label) ::claraparabricks::genomeworks::roctx_range varname(label)
#else
#define GW_NVTX_RANGE(varname, label)
#endif

} // namespace genomeworks
} // namespace claraparabricks
