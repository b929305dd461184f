// cuda_runtime_api.h -- TEST INFRASTRUCTURE (oracle/simt): the slice of the CUDA runtime API that the reference's host code calls,
// on host memory. "Device" memory is malloc'ed, copies are memcpy, streams and events are tokens, every call is synchronous
// (kernel launches run to completion inside simt::launch).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../simt.hpp"

typedef int cudaError_t;
constexpr cudaError_t cudaSuccess             = 0;
constexpr cudaError_t cudaErrorMemoryAllocation = 2;
constexpr cudaError_t cudaErrorInvalidValue   = 1;
typedef struct simt_stream* cudaStream_t;
typedef struct simt_event* cudaEvent_t;
enum cudaMemcpyKind
{
    cudaMemcpyHostToHost,
    cudaMemcpyHostToDevice,
    cudaMemcpyDeviceToHost,
    cudaMemcpyDeviceToDevice,
    cudaMemcpyDefault
};
constexpr unsigned cudaStreamNonBlocking = 1, cudaStreamDefault = 0, cudaEventDisableTiming = 2, cudaEventDefault = 0;
struct cudaDeviceProp
{
    char name[256];
    size_t totalGlobalMem;
    int major, minor, multiProcessorCount, warpSize, maxThreadsPerBlock;
};
inline const char* cudaGetErrorString(cudaError_t e) { return e == cudaSuccess ? "no error" : "simt stub error"; }
inline const char* cudaGetErrorName(cudaError_t e) { return e == cudaSuccess ? "cudaSuccess" : "cudaErrorStub"; }
inline cudaError_t cudaGetLastError() { return cudaSuccess; }
inline cudaError_t cudaPeekAtLastError() { return cudaSuccess; }
inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
inline cudaError_t cudaStreamSynchronize(cudaStream_t) { return cudaSuccess; }
inline cudaError_t cudaStreamCreate(cudaStream_t* s) { *s = nullptr; return cudaSuccess; }
inline cudaError_t cudaStreamCreateWithFlags(cudaStream_t* s, unsigned) { *s = nullptr; return cudaSuccess; }
inline cudaError_t cudaStreamDestroy(cudaStream_t) { return cudaSuccess; }
inline cudaError_t cudaEventCreate(cudaEvent_t* e) { *e = nullptr; return cudaSuccess; }
inline cudaError_t cudaEventCreateWithFlags(cudaEvent_t* e, unsigned) { *e = nullptr; return cudaSuccess; }
inline cudaError_t cudaEventDestroy(cudaEvent_t) { return cudaSuccess; }
inline cudaError_t cudaEventRecord(cudaEvent_t, cudaStream_t = nullptr) { return cudaSuccess; }
inline cudaError_t cudaEventSynchronize(cudaEvent_t) { return cudaSuccess; }
inline cudaError_t cudaStreamWaitEvent(cudaStream_t, cudaEvent_t, unsigned = 0) { return cudaSuccess; }
inline cudaError_t cudaGetDevice(int* d) { *d = 0; return cudaSuccess; }
inline cudaError_t cudaSetDevice(int) { return cudaSuccess; }
inline cudaError_t cudaGetDeviceCount(int* n) { *n = 1; return cudaSuccess; }
// SIMT_MALLOC_FILL=<byte>: fresh "device" memory is filled with that byte (default: zero, what a fresh CUDA context hands out);
// results that depend on the fill read memory the reference never wrote
inline cudaError_t cudaMalloc(void** p, size_t bytes)
{
    static const int fill = [] { const char* e = std::getenv("SIMT_MALLOC_FILL"); return e ? std::atoi(e) : 0; }();
    *p = std::malloc(bytes ? bytes : 1);
    if (*p) std::memset(*p, fill, bytes ? bytes : 1);
    return *p ? cudaSuccess : cudaErrorMemoryAllocation;
}
template <typename T>
inline cudaError_t cudaMalloc(T** p, size_t bytes) { return cudaMalloc(reinterpret_cast<void**>(p), bytes); }
inline cudaError_t cudaFree(void* p) { std::free(p); return cudaSuccess; }
inline cudaError_t cudaMallocHost(void** p, size_t bytes) { return cudaMalloc(p, bytes); }
template <typename T>
inline cudaError_t cudaMallocHost(T** p, size_t bytes) { return cudaMalloc(reinterpret_cast<void**>(p), bytes); }
inline cudaError_t cudaHostAlloc(void** p, size_t bytes, unsigned) { return cudaMalloc(p, bytes); }
template <typename T>
inline cudaError_t cudaHostAlloc(T** p, size_t bytes, unsigned f) { return cudaMalloc(reinterpret_cast<void**>(p), bytes); }
constexpr unsigned cudaHostAllocDefault = 0;
inline cudaError_t cudaFreeHost(void* p) { std::free(p); return cudaSuccess; }
inline cudaError_t cudaMemcpy(void* dst, const void* src, size_t n, cudaMemcpyKind) { std::memcpy(dst, src, n); return cudaSuccess; }
inline cudaError_t cudaMemcpyAsync(void* dst, const void* src, size_t n, cudaMemcpyKind, cudaStream_t = nullptr) { std::memcpy(dst, src, n); return cudaSuccess; }
inline cudaError_t cudaMemset(void* p, int v, size_t n) { std::memset(p, v, n); return cudaSuccess; }
inline cudaError_t cudaMemsetAsync(void* p, int v, size_t n, cudaStream_t = nullptr) { std::memset(p, v, n); return cudaSuccess; }
inline cudaError_t cudaMemGetInfo(size_t* free_bytes, size_t* total)
{
    *free_bytes = size_t(8) << 30;
    *total      = size_t(8) << 30;
    return cudaSuccess;
}
enum cudaDeviceAttr
{
    cudaDevAttrMultiProcessorCount = 16,
    cudaDevAttrMaxThreadsPerBlock  = 1,
    cudaDevAttrWarpSize            = 10
};
inline cudaError_t cudaDeviceGetAttribute(int* value, cudaDeviceAttr attr, int)
{
    *value = attr == cudaDevAttrMultiProcessorCount ? 2 : (attr == cudaDevAttrWarpSize ? 32 : 1024);
    return cudaSuccess;
}
template <typename F>
inline cudaError_t cudaOccupancyMaxActiveBlocksPerMultiprocessor(int* blocks, F, int, size_t)
{
    *blocks = 2;
    return cudaSuccess;
}
enum cudaFuncCache
{
    cudaFuncCachePreferNone,
    cudaFuncCachePreferShared,
    cudaFuncCachePreferL1,
    cudaFuncCachePreferEqual
};
inline cudaError_t cudaDeviceSetCacheConfig(cudaFuncCache) { return cudaSuccess; }
inline cudaError_t cudaGetDeviceProperties(cudaDeviceProp* p, int)
{
    std::memset(p, 0, sizeof(*p));
    std::strcpy(p->name, "simt emulator");
    p->totalGlobalMem = size_t(8) << 30;
    p->major = 7, p->minor = 0, p->multiProcessorCount = 1, p->warpSize = 32, p->maxThreadsPerBlock = 1024;
    return cudaSuccess;
}
