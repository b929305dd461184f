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
    double seconds_after_creation = 0;              ///< from the moment every worker's / class's Batch exists (process_windows_multi_device: to the end of the call) to the
                                                    ///< last results -- filling + generate_poa() + get_*(), the region the reference's
                                                    ///< multi-batch benchmark times (cudapoa/benchmarks/multi_batch.hpp:72-177)
};

/// Windows binned by size so that every bin's batch is resident at the same time (MI355X addition). get_multi_batch_sizes
/// (utils.hpp:36-69) bins windows to run the bins one after the other, and folds smaller bins into a larger one's batch
/// whenever they fit -- on a 288 GB device a whole long-read set then shares the shape of its longest window, needs several
/// fills, and every fill lasts as long as its heaviest window. Here the classes are geometric in the longest read (class k:
/// longest read in (L / 2^(k+1), L / 2^k]), each with the BatchConfig of its own largest member, so the set's slabs add up to a
/// fraction of the single-shape plan and all classes can run concurrently (process_windows_size_classes): the wall time is
/// the heaviest class's, not the sum of the fills.
struct SizeClassPlan
{
    std::vector<BatchConfig> configs;            ///< one per non-empty class, largest reads first
    std::vector<std::vector<int32_t>> groups;    ///< window indices of each class (ascending)
    std::vector<int64_t> bytes_per_window;       ///< device bytes of one window under configs[k]
    int64_t total_bytes = 0;                     ///< sum over classes of windows x bytes_per_window
};

/// Host-only (no device query). longest / reads: per window its longest read and its number of reads.
void plan_size_classes(SizeClassPlan& plan, const std::vector<int32_t>& longest, const std::vector<int32_t>& reads, bool msa_flag,
                       int32_t band_width = 256, BandMode band_mode = BandMode::adaptive_band, float adaptive_storage_factor = 2.0f,
                       float graph_length_factor = 3.0f, int32_t max_pred_distance = 0, int32_t mismatch_score = -6,
                       int32_t gap_score = -8, int32_t match_score = 8);

/// Admission by residency (process_windows_size_classes): a long-read window occupies a compute unit for its whole life, so the
/// device holds about one window per unit at a time. Classes are admitted in plan order while their windows (a quarter more than
/// there are units: the first to finish make room at once) fit; the classes of the next group wait, on the device, for the end
/// of the last class of the group before them. Returns, per class, the class it waits for (-1: admitted at once; empty classes: -1).
std::vector<int32_t> size_class_admission_gates(const SizeClassPlan& plan, int32_t compute_units);

/// One worker (host thread, stream, allocator slice, Batch) per class of the plan on `device`, all at once; a class whose
/// windows do not fit its slice takes several fills. memory_budget: device bytes for all classes together (each class gets
/// its share of the plan's total, scaled down if the total exceeds the budget). out.seconds = wall time of the workers
/// including batch creation, filling and the release of the slabs; *compute_seconds (optional) = from the moment every worker
/// has filled its first batch to the moment the last worker has handed over its last results (generate_poa + get_consensus /
/// get_msa and any further fills; not the destruction of the batches).
void process_windows_size_classes(MultiDeviceOutput& out, const std::vector<std::vector<std::string>>& windows,
                                  const SizeClassPlan& plan, int32_t device, int64_t memory_budget, int8_t output_mask,
                                  int16_t gap_score = -8, int16_t mismatch_score = -6, int16_t match_score = 8,
                                  double* compute_seconds = nullptr);

/// Runs every window (a window = its reads) under `batch_size`. Throws what create_batch / Batch throw.
void process_windows_multi_device(MultiDeviceOutput& out, const std::vector<std::vector<std::string>>& windows,
                                  const BatchConfig& batch_size, const MultiDeviceConfig& config);

} // namespace cudapoa
} // namespace genomeworks
} // namespace claraparabricks
