// aligner_impl.hpp -- concrete banded / unbanded Myers aligner for MI355X (see cudaaligner.cpp).
#pragma once
#include <claraparabricks/genomeworks/cudaaligner/aligner.hpp>
#include <claraparabricks/genomeworks/cudaaligner/alignment.hpp>

#include <string>
#include <vector>

#include "pinned_vector.hpp"

namespace claraparabricks
{
namespace genomeworks
{
namespace cudaaligner
{

class BandedAligner : public FixedBandAligner
{
public:
    /// expand_results: materialise per-position states (the fixed-stride factories) instead of run lengths.
    /// max_query_length >= 0 switches on the fixed-stride limits (exceeded_max_length / exceeded_max_alignments).
    BandedAligner(int64_t max_device_memory, int32_t max_bandwidth, DefaultDeviceAllocator allocator, cudaStream_t stream,
                  int32_t device_id, bool expand_results, int32_t max_query_length, int32_t max_target_length,
                  int32_t max_alignments);
    ~BandedAligner() override;

    StatusType align_all() override;
    StatusType sync_alignments() override;
    StatusType add_alignment(const char* query, int32_t query_length, const char* target, int32_t target_length,
                             bool reverse_complement_query = false, bool reverse_complement_target = false) override;
    StatusType add_alignment(int32_t max_bandwidth, const char* query, int32_t query_length, const char* target,
                             int32_t target_length, bool reverse_complement_query = false,
                             bool reverse_complement_target = false) override;
    const std::vector<std::shared_ptr<Alignment>>& get_alignments() const override { return alignments_; }
    DeviceAlignmentsPtrs get_alignments_device() const override;
    void reset() override;
    void free_temporary_device_buffers() override;
    int32_t num_alignments() const override { return static_cast<int32_t>(seq_starts_h_.size() / 2); } // [2n + 1] offsets
    cudaStream_t get_stream() const override { return stream_; }
    int32_t get_device() const override { return device_id_; }
    DefaultDeviceAllocator get_device_allocator() const override { return allocator_; }
    void reset_max_bandwidth(int32_t max_bandwidth) override;

    // benchmark helpers (not part of the reference interface)
    void relaunch_resident();     ///< run the kernels again on the inputs already resident in HBM
    float relaunch_resident_timed(); ///< same; returns the kernels' time in ms (HIP events on the aligner's stream)
    bool expands_results() const { return expand_results_; }
    uint64_t total_band_cells();  ///< 32 * band words * target length, summed over band attempts and pairs

private:
    void reset_data();
    void free_device();
    struct Chunk;
    void launch(void* event_before = nullptr, void* event_after = nullptr); ///< every chunk's kernels again + fetch_head()
    void launch_chunk(const Chunk& c, int32_t phases = 0);
    /// every chunk's kernels, pipelined over the aligner's stream and the side stream; `order` (host, pinned) goes up chunk by
    /// chunk when given, and workspaces are allocated when `allocate`
    void run_chunks(const int32_t* order, bool allocate);
    void enqueue_inputs(size_t k); ///< chunk k's bases, offsets and band widths on the upload stream, and its event
    void prepare_head();                                  ///< pinned [result_starts | metadata] of the launched batch
    void fetch_head_slice(const Chunk& c, void* stream);  ///< the chunk's offsets and metadata follow its kernels to the host
    void drain_streams();                                 ///< host waits for the aligner's, the upload and the side stream
    void join_side_stream();                              ///< stream_ continues after everything queued on the side stream

    cudaStream_t stream_;
    int32_t device_id_;
    DefaultDeviceAllocator allocator_;
    int32_t max_bandwidth_;
    int64_t max_device_memory_;
    bool expand_results_;
    int32_t max_query_length_, max_target_length_, max_alignments_;

    // staging arrays in pinned host memory: uploaded asynchronously at link speed
    PinnedVector<char> seq_h_;
    /// the same bases two per byte (include/gwhip.h, gwhip_unpack_bases): what align_all() uploads -- half the bytes over the
    /// link; the characters above stay for the Alignment objects
    PinnedVector<uint8_t> packed_h_;
    PinnedVector<int64_t> seq_starts_h_;
    PinnedVector<int32_t> max_bandwidths_h_;
    PinnedVector<int32_t> order_h_;
    std::vector<std::shared_ptr<Alignment>> alignments_;
    size_t workspace_bytes_estimate_ = 0;
    size_t largest_wave_ws_          = 0;
    int32_t longest_query_           = 0;
    int64_t longest_pair_            = 0; ///< query + target of the longest pair (buckets of the counting sort in align_all())
    int32_t widest_band_             = 0;
    bool launched_                   = false;
    bool uploads_in_flight_          = false;
    int64_t total_length_h_          = 0;
    int32_t n_last_                  = 0;
    char* mirror_                    = nullptr; ///< pinned: operations[mirror_runs_] | run lengths[mirror_runs_] written by the chunks' kernels
    size_t mirror_cap_               = 0;       ///< bytes (pinned_acquire)
    int64_t mirror_runs_             = 0;       ///< capacity in runs; the run lengths start at byte up64(mirror_runs_)
    bool raw_upload_                 = false;   ///< GW_ALIGNER_RAW_UPLOAD (A/B switch): characters instead of packed bases over the link
    char* head_                      = nullptr; ///< pinned: result_starts[n + 1] | metadata[n] of the last launch
    size_t head_cap_                 = 0;
    int32_t n_head_                  = 0;

    char* device_block_        = nullptr; ///< inputs and outputs (allocated first: the uploads start before the batch is sorted)
    size_t device_block_bytes_ = 0;
    /// A batch is processed as one chunk, or -- large batches -- as several chunks of consecutive pairs whose uploads run on a
    /// stream of their own: the upload of chunk k + 1 overlaps the kernels of chunk k (include/gwhip.h, gwhip_myers_args).
    /// Every chunk has its own workspace (sized once its processing order is known).
    struct Chunk
    {
        int32_t lo = 0, hi = 0;
        int64_t first_offset = 0, span = 0; ///< sequence offset of the first pair, bytes of the chunk's sequences
        char* workspace        = nullptr;
        size_t workspace_bytes = 0, block_bytes = 0;
        void* uploaded         = nullptr; ///< hipEvent_t on the upload stream (null: uploaded on the aligner's own stream)
    };
    std::vector<Chunk> chunks_;
    int64_t launched_total_length_ = 0;   ///< bases of the launched batch (the host arrays move to the views at sync_alignments())
    void* upload_stream_ = nullptr;       ///< hipStream_t, created with the first chunked batch
    void* side_stream_   = nullptr;       ///< hipStream_t of a chunked batch's sizing / compaction kernels and result offsets (gwhip_myers_args::side_stream)
    std::vector<void*> upload_events_;    ///< hipEvent_t pool (timing disabled)
    char* d_seq_               = nullptr;
    uint8_t* d_packed_         = nullptr; ///< upload staging of packed_h_ (unpacked into d_seq_ on the device)
    int64_t* d_starts_         = nullptr;
    int32_t* d_bw_             = nullptr;
    int32_t* d_order_          = nullptr;
    int8_t* d_results_         = nullptr;
    int32_t* d_result_counts_  = nullptr;
    int32_t* d_result_starts_  = nullptr;
    uint32_t* d_metadata_      = nullptr;
    uint64_t* d_cells_         = nullptr;
};

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
