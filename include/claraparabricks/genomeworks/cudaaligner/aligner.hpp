// aligner.hpp -- batched pairwise aligner interface (source-compatible with the reference's
// cudaaligner/aligner.hpp:41-219).
#pragma once

#include <claraparabricks/genomeworks/cudaaligner/cudaaligner.hpp>
#include <claraparabricks/genomeworks/utils/allocator.hpp>
#include <claraparabricks/genomeworks/utils/cudautils.hpp>

#include <cstdint>
#include <memory>
#include <vector>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudaaligner
{

class Alignment;

/// Device-resident results: packed run-length encoded alignments. Valid until reset() / destruction of the
/// aligner; consumers must order their work after the aligner's stream (get_stream()).
struct DeviceAlignmentsPtrs
{
    const int8_t* cigar_operations;  ///< [total_length] AlignmentState per run, each alignment stored back to front
    const int32_t* cigar_runlengths; ///< [total_length] repetitions of the operation at the same position
    const int32_t* cigar_offsets;    ///< [n_alignments + 1] begin / end of alignment i in the two arrays above
    const uint32_t* metadata;        ///< [n_alignments] bit 31: is_optimal, bits 26-0: index of the alignment as added
    int64_t total_length;
    int32_t n_alignments;
    static constexpr uint32_t index_mask = (1u << 27) - 1;
};

class Aligner
{
public:
    virtual ~Aligner() = default;
    /// Launch all queued alignments asynchronously on the aligner's stream.
    virtual StatusType align_all() = 0;
    /// Wait for the results and materialise them on the host.
    virtual StatusType sync_alignments() = 0;
    /// Queue one pair (sequences are copied). `exceeded_max_alignments`: run the batch, reset(), then retry.
    virtual StatusType add_alignment(const char* query, int32_t query_length, const char* target, int32_t target_length,
                                     bool reverse_complement_query = false, bool reverse_complement_target = false) = 0;
    virtual const std::vector<std::shared_ptr<Alignment>>& get_alignments() const = 0;
    virtual DeviceAlignmentsPtrs get_alignments_device() const                    = 0;
    virtual void reset()                                                          = 0;
    virtual void free_temporary_device_buffers()                                  = 0;
    virtual int32_t num_alignments() const                                        = 0;
    virtual cudaStream_t get_stream() const                                       = 0;
    virtual int32_t get_device() const                                            = 0;
    virtual DefaultDeviceAllocator get_device_allocator() const                   = 0;
};

/// Aligner whose band is capped per alignment (banded Myers with Ukkonen band doubling).
class FixedBandAligner : public Aligner
{
public:
    virtual void reset_max_bandwidth(int32_t max_bandwidth) = 0;
    using Aligner::add_alignment;
    virtual StatusType add_alignment(int32_t max_bandwidth, const char* query, int32_t query_length, const char* target,
                                     int32_t target_length, bool reverse_complement_query = false,
                                     bool reverse_complement_target = false) = 0;
};

std::unique_ptr<Aligner> create_aligner(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments,
                                        AlignmentType type, DefaultDeviceAllocator allocator, cudaStream_t stream,
                                        int32_t device_id);
std::unique_ptr<Aligner> create_aligner(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments,
                                        AlignmentType type, cudaStream_t stream, int32_t device_id,
                                        int64_t max_device_memory_allocator_caching_size = -1);
std::unique_ptr<FixedBandAligner> create_aligner(AlignmentType type, int32_t max_bandwidth, cudaStream_t stream,
                                                 int32_t device_id, DefaultDeviceAllocator allocator,
                                                 int64_t max_device_memory);
std::unique_ptr<FixedBandAligner> create_aligner(AlignmentType type, int32_t max_bandwidth, cudaStream_t stream,
                                                 int32_t device_id, int64_t max_device_memory = -1);

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
