// aligner_global.hpp -- the fixed-limit global aligners (reference: cudaaligner/src/aligner_global.hpp and its three
// subclasses). AlignerGlobal holds what they share -- limits, statuses, packing, the device block, the host-side
// reversal of the kernels' back-to-front paths (aligner_global.cpp:48-190) -- and each algorithm supplies its
// workspace size and its launch through the C-ABI of libgwhip:
//   AlignerGlobalHirschbergMyers  the default of create_aligner(max_query, max_target, max_alignments, ...)
//                                 (aligner_global_hirschberg_myers.cpp)             -> gwhip_hirschberg_myers
//   AlignerGlobalUkkonen          band parameter p = 100, int16 scores (aligner_global_ukkonen.cpp) -> gwhip_ukkonen
//   AlignerGlobalMyers            full bit-vector matrices (aligner_global_myers.cpp): the banded kernel with a band
//                                 that covers the whole query                          -> gwhip_myers_banded
#pragma once
#include "pinned_vector.hpp"
#include <claraparabricks/genomeworks/cudaaligner/aligner.hpp>
#include <claraparabricks/genomeworks/cudaaligner/alignment.hpp>

#include <vector>

#include "aligner_impl.hpp"

namespace claraparabricks
{
namespace genomeworks
{
namespace cudaaligner
{

class AlignerGlobal : public Aligner
{
public:
    AlignerGlobal(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments,
                  DefaultDeviceAllocator allocator, cudaStream_t stream, int32_t device_id);
    ~AlignerGlobal() override;

    StatusType align_all() override;
    StatusType sync_alignments() override;
    StatusType add_alignment(const char* query, int32_t query_length, const char* target, int32_t target_length,
                             bool reverse_complement_query = false, bool reverse_complement_target = false) override;
    const std::vector<std::shared_ptr<Alignment>>& get_alignments() const override { return alignments_; }
    DeviceAlignmentsPtrs get_alignments_device() const override;
    void reset() override;
    void free_temporary_device_buffers() override {}
    int32_t num_alignments() const override { return static_cast<int32_t>(alignments_.size()); }
    cudaStream_t get_stream() const override { return stream_; }
    int32_t get_device() const override { return device_id_; }
    DefaultDeviceAllocator get_device_allocator() const override { return allocator_; }

    /// measurement aid: the kernels of the last align_all() once more on the inputs still resident in HBM, timed with HIP
    /// events on the aligner's stream; < 0 when nothing is resident
    float relaunch_resident_timed();

    int32_t get_max_query_length() const { return max_query_length_; }
    int32_t get_max_target_length() const { return max_target_length_; }

protected:
    /// device bytes the algorithm needs for the queued pairs (sequence_starts: host, [2n+1])
    virtual size_t workspace_bytes(int32_t n, const int64_t* sequence_starts) const = 0;
    /// launch on stream_; results: back-to-front states of pair i in [starts[2i], starts[2i+2]); 0 or a hipError_t
    virtual int run_alignment(int32_t n, const char* sequences_d, const int64_t* sequence_starts_d,
                              const int64_t* sequence_starts_h, int8_t* results_d, int32_t* result_lengths_d,
                              void* workspace_d, size_t workspace_size) = 0;

private:
    void free_device();

    int32_t max_query_length_, max_target_length_, max_alignments_;
    DefaultDeviceAllocator allocator_;
    cudaStream_t stream_;
    int32_t device_id_;
    // staging arrays in pinned memory (process-wide cache of pinned buffers): the copies of align_all() are true async DMA
    PinnedVector<char> seq_h_;
    PinnedVector<int64_t> seq_starts_h_;
    std::vector<std::shared_ptr<Alignment>> alignments_;
    PinnedVector<int8_t> results_h_;
    PinnedVector<int32_t> result_lengths_h_;
    char* device_block_        = nullptr;
    size_t device_block_bytes_ = 0;
    char* d_seq_               = nullptr;
    int64_t* d_starts_         = nullptr;
    char* d_ws_                = nullptr;
    size_t ws_bytes_           = 0;
    int8_t* d_results_         = nullptr;
    int32_t* d_result_lengths_ = nullptr;
    bool launched_             = false;
};

class AlignerGlobalHirschbergMyers : public AlignerGlobal
{
public:
    using AlignerGlobal::AlignerGlobal;

protected:
    size_t workspace_bytes(int32_t n, const int64_t* sequence_starts) const override;
    int run_alignment(int32_t n, const char* sequences_d, const int64_t* sequence_starts_d, const int64_t* sequence_starts_h,
                      int8_t* results_d, int32_t* result_lengths_d, void* workspace_d, size_t workspace_size) override;
};
using HirschbergAligner = AlignerGlobalHirschbergMyers;

class AlignerGlobalUkkonen : public AlignerGlobal
{
public:
    AlignerGlobalUkkonen(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments,
                         DefaultDeviceAllocator allocator, cudaStream_t stream, int32_t device_id);
    /// additionally: exceeded_max_alignment_difference when |query - target| > 10 % of max_target_length
    StatusType add_alignment(const char* query, int32_t query_length, const char* target, int32_t target_length,
                             bool reverse_complement_query = false, bool reverse_complement_target = false) override;

protected:
    size_t workspace_bytes(int32_t n, const int64_t* sequence_starts) const override;
    int run_alignment(int32_t n, const char* sequences_d, const int64_t* sequence_starts_d, const int64_t* sequence_starts_h,
                      int8_t* results_d, int32_t* result_lengths_d, void* workspace_d, size_t workspace_size) override;

private:
    int32_t ukkonen_p_;
};

/// Full-matrix Myers. Its backtrace rule is the banded class's (myers_gpu.cu:240-315 vs :392-442), so it is the banded
/// aligner with a band that always covers the whole query, behind AlignerGlobal's fixed limits and per-state results.
class AlignerGlobalMyers : public BandedAligner
{
public:
    AlignerGlobalMyers(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments,
                       DefaultDeviceAllocator allocator, cudaStream_t stream, int32_t device_id);
};

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
