// hirschberg_aligner.hpp -- the default global aligner (create_aligner(max_query, max_target, max_alignments, ...)):
// Hirschberg + Myers on the GPU, with the fixed limits and statuses of the reference's AlignerGlobal
// (aligner_global.cpp:48-190) and AlignerGlobalHirschbergMyers (aligner_global_hirschberg_myers.cpp).
#pragma once
#include <claraparabricks/genomeworks/cudaaligner/aligner.hpp>
#include <claraparabricks/genomeworks/cudaaligner/alignment.hpp>

#include <vector>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudaaligner
{

class HirschbergAligner : public Aligner
{
public:
    HirschbergAligner(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments,
                      DefaultDeviceAllocator allocator, cudaStream_t stream, int32_t device_id);
    ~HirschbergAligner() override;

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

private:
    void free_device();

    int32_t max_query_length_, max_target_length_, max_alignments_;
    DefaultDeviceAllocator allocator_;
    cudaStream_t stream_;
    int32_t device_id_;
    std::vector<char> seq_h_;
    std::vector<int64_t> seq_starts_h_{0};
    std::vector<std::shared_ptr<Alignment>> alignments_;
    std::vector<int8_t> results_h_;
    std::vector<int32_t> result_lengths_h_;
    char* device_block_        = nullptr;
    size_t device_block_bytes_ = 0;
    int8_t* d_results_         = nullptr;
    int32_t* d_result_lengths_ = nullptr;
    bool launched_             = false;
};

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
