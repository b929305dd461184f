// nvToolsExt.h -- TEST INFRASTRUCTURE (oracle/simt): profiler ranges are no-ops
#pragma once
#include <cstdint>
typedef struct
{
    uint16_t version, size;
    uint32_t category;
    int32_t colorType;
    uint32_t color;
    int32_t payloadType, reserved0;
    union { uint64_t ullValue; int64_t llValue; double doubleValue; } payload;
    int32_t messageType;
    union { const char* ascii; const wchar_t* unicode; } message;
} nvtxEventAttributes_t;
#define NVTX_VERSION 2
#define NVTX_EVENT_ATTRIB_STRUCT_SIZE ((uint16_t)(sizeof(nvtxEventAttributes_t)))
#define NVTX_COLOR_ARGB 1
#define NVTX_MESSAGE_TYPE_ASCII 1
inline int nvtxRangePushEx(const nvtxEventAttributes_t*) { return 0; }
inline int nvtxRangePushA(const char*) { return 0; }
inline int nvtxRangePop() { return 0; }
