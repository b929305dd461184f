// multi_device.hpp -- POA windows over several devices and several batches per device (MI355X addition; the reference
// has no such class: its tools run one worker thread per device pulling whole batches, cudamapper/src/main.cu:577-592,
// and its multi-batch benchmark runs several Batch objects on host threads that share one device and one allocator,
// cudapoa/benchmarks/multi_batch.hpp:41-61,72-177 -- this is both patterns behind one call).
//
// Windows are independent, so there is no device-to-device traffic and no collective: every worker (one host thread
// with its own stream and its own Batch, on its device through scoped_device_switch) pulls the next windows from a
// shared cursor until its batch reports exceeded_maximum_poas, runs generate_poa + get_consensus / get_msa, and stores
// the results by GLOBAL window index -- the output does not depend on the number of devices or workers.
#pragma once

#include <claraparabricks/genomeworks/cudapoa/batch.hpp>

#include <cstdint>
#include <string>
#include <vector>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudapoa
{

struct MultiDeviceConfig
{
    /// device id of every device worker group; an id may appear more than once (logical shards of one device)
    std::vector<int32_t> devices{0};
    /// Batch objects (= host threads, streams) per entry of `devices`; they share that entry's allocator
    int32_t batches_per_device = 1;
    /// device memory of one entry of `devices`, split evenly over its batches: bytes, or -1 for
    /// memory_fraction x free memory at the time of the call divided by the number of entries that name the device
    int64_t memory_per_device = -1;
    double memory_fraction    = 0.9;
    int8_t output_mask        = OutputType::consensus;
    int16_t gap_score = -8, mismatch_score = -6, match_score = 8;
};

struct MultiDeviceOutput
{
    std::vector<std::string> consensus;             ///< [window] (output_mask & consensus)
    std::vector<std::vector<uint16_t>> coverage;    ///< [window]
    std::vector<std::vector<std::string>> msa;      ///< [window][read] (output_mask & msa)
    std::vector<StatusType> status;                 ///< [window] add_poa_group / kernel status
    std::vector<int32_t> worker_of_window;          ///< [window] which worker ran it (diagnostics; not deterministic)
    int32_t launches = 0;                           ///< generate_poa() calls over all workers
    double seconds   = 0;                           ///< wall time from the first worker's start to the last worker's end
                                                    ///< (batch creation, filling, kernels, result unpacking)
};

/// Runs every window (a window = its reads) under `batch_size`. Throws what create_batch / Batch throw.
void process_windows_multi_device(MultiDeviceOutput& out, const std::vector<std::vector<std::string>>& windows,
                                  const BatchConfig& batch_size, const MultiDeviceConfig& config);

} // namespace cudapoa
} // namespace genomeworks
} // namespace claraparabricks
