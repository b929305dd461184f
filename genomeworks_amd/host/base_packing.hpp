// base_packing.hpp -- the banded aligner's upload format: two bases per byte (include/gwhip.h, gwhip_unpack_bases).
// Base i of the batch sits in bits 4 (i & 1) .. 4 (i & 1) + 3 of byte i >> 1. A query base keeps what the kernels can tell apart
// -- 'A', 'C', 'T', 'G' (myers_gpu.cu:196-208 compares with exactly these) or "anything else" (code 4) -- a target base its
// pattern index (c >> 1) & 3 (myers_gpu.cu:210-241). Host only; tests/cpp/base_packing_driver.cpp checks it against a
// restatement of the device side.
#pragma once
#include <cstdint>

namespace gwhost
{

struct QueryCodes
{
    uint8_t of[256];
    QueryCodes()
    {
        for (int c = 0; c < 256; c++) of[c] = 4;
        of['A'] = 0, of['C'] = 1, of['T'] = 2, of['G'] = 3;
    }
};
inline const QueryCodes kQueryCodes{};
inline uint8_t query_code(char c) { return kQueryCodes.of[static_cast<unsigned char>(c)]; }
inline uint8_t target_code(char c) { return static_cast<uint8_t>((static_cast<unsigned char>(c) >> 1) & 3u); }

/// appends the codes of bases[0 .. n) at base index `first` of the packed array; an odd `first` shares its byte with the base
/// before it, which an earlier call has written (low nibble kept)
template <typename Code>
void pack_bases(uint8_t* packed, int64_t first, const char* bases, int32_t n, Code code)
{
    int64_t i = first;
    int32_t k = 0;
    if (n > 0 && (i & 1))
    {
        packed[i >> 1] = static_cast<uint8_t>((packed[i >> 1] & 0x0f) | (code(bases[0]) << 4));
        ++i;
        ++k;
    }
    for (; k + 2 <= n; k += 2, i += 2) packed[i >> 1] = static_cast<uint8_t>(code(bases[k]) | (code(bases[k + 1]) << 4));
    if (k < n) packed[i >> 1] = code(bases[k]);
}

} // namespace gwhost
