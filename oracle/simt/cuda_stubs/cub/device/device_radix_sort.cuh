// cub/device/device_radix_sort.cuh -- TEST INFRASTRUCTURE (oracle/simt): cub::DeviceRadixSort::SortPairs on host memory: a stable
// sort by the key bits [begin_bit, end_bit), ascending, results in the buffers that become Current() (as cub does)
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "../util_type.cuh"
#include <cuda_runtime_api.h>
namespace cub
{
struct DeviceRadixSort
{
    template <typename K, typename V>
    static cudaError_t SortPairs(void* temp, size_t& temp_bytes, DoubleBuffer<K>& keys, DoubleBuffer<V>& values, int n, int begin_bit = 0,
                                 int end_bit = sizeof(K) * 8, cudaStream_t = nullptr)
    {
        if (temp == nullptr)
        {
            temp_bytes = 16;
            return cudaSuccess;
        }
        using U          = typename std::make_unsigned<K>::type;
        const int bits   = end_bit - begin_bit;
        const uint64_t m = bits >= 64 ? ~uint64_t(0) : ((uint64_t(1) << bits) - 1);
        auto key_of      = [&](int i) {
            // radix order of a signed key: the sign bit flipped (cub's twiddling)
            U u = static_cast<U>(keys.Current()[i]);
            if (std::is_signed<K>::value) u ^= U(1) << (sizeof(K) * 8 - 1);
            return (static_cast<uint64_t>(u) >> begin_bit) & m;
        };
        std::vector<int> idx(static_cast<size_t>(n));
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key_of(a) < key_of(b); });
        for (int i = 0; i < n; ++i)
        {
            keys.Alternate()[i]   = keys.Current()[idx[static_cast<size_t>(i)]];
            values.Alternate()[i] = values.Current()[idx[static_cast<size_t>(i)]];
        }
        keys.selector ^= 1;
        values.selector ^= 1;
        return cudaSuccess;
    }
};
} // namespace cub
