// poa_batch_impl.hpp -- concrete cudapoa::Batch for MI355X (declaration; see cudapoa_batch.cpp).
#pragma once
#include <claraparabricks/genomeworks/cudapoa/batch.hpp>

#include "../../include/gwhip.h"

namespace claraparabricks
{
namespace genomeworks
{
namespace cudapoa
{

gwhip_poa_config make_device_config(const BatchConfig& b, int8_t output_mask, int32_t gap, int32_t mismatch, int32_t match);

/// The binning rule of get_multi_batch_sizes on its own (cudapoa_utils.cpp; exposed to the tests through the C API).
void bin_poa_groups(std::vector<BatchConfig>& list_of_batch_sizes, std::vector<std::vector<int32_t>>& list_of_groups_per_batch,
                    const std::vector<int32_t>& capacity, const std::vector<int32_t>& longest, const std::vector<int32_t>& reads,
                    int32_t band_width, BandMode band_mode, float adaptive_storage_factor, float graph_length_factor,
                    int32_t max_pred_distance, const std::vector<int32_t>* bins_capacity);

class PoaBatch : public Batch
{
public:
    PoaBatch(int32_t device_id, cudaStream_t stream, DefaultDeviceAllocator allocator, int64_t max_mem, int8_t output_mask,
             const BatchConfig& batch_size, int32_t gap_score = -8, int32_t mismatch_score = -6, int32_t match_score = 8);
    ~PoaBatch() override;

    StatusType add_poa_group(std::vector<StatusType>& per_seq_status, const Group& poa_group) override;
    int32_t get_total_poas() const override { return poa_count_; }
    void generate_poa() override;
    StatusType get_consensus(std::vector<std::string>& consensus, std::vector<std::vector<uint16_t>>& coverage,
                             std::vector<StatusType>& output_status) override;
    StatusType get_msa(std::vector<std::vector<std::string>>& msa, std::vector<StatusType>& output_status) override;
    void get_graphs(std::vector<DirectedGraph>& graphs, std::vector<StatusType>& output_status) override;
    int32_t batch_id() const override { return bid_; }
    void reset() override;

    // extensions used by the benchmark / tests (not part of the reference interface)
    /// get_consensus() that overwrites its arguments (resized to the batch's windows) instead of appending to them, so that a
    /// caller who passes the vectors of its previous call gets the results without a heap allocation per window.
    StatusType get_consensus_in_place(std::vector<std::string>& consensus, std::vector<std::vector<uint16_t>>& coverage,
                                      std::vector<StatusType>& output_status);
    int32_t max_poas() const { return max_poas_; }
    uint64_t total_cells();    ///< DP cells computed by the last generate_poa() (device counters)
    void relaunch_resident();  ///< re-run the kernels on the inputs already resident in HBM (no H2D)
    /// Same, timed with HIP events on the batch's stream: graph-build kernel and output kernel (milliseconds).
    void relaunch_resident_timed(float* graph_build_ms, float* output_ms);
    /// Profiling aid: relaunch with per-phase cycle accounting; out[6] = mean ticks per window of
    /// {row table, NW forward, sink+traceback, graph merge, topsort, other}.
    void profile_phases(double out[6]);
    void profile_phases_per_window(std::vector<uint64_t>& ticks); // six phase counters per window, in window order
    const gwhip_poa_config& device_config() const { return cfg_; }
    cudaStream_t stream() const { return stream_; }

private:
    void debug_message(const std::string& message);
    bool reserve_buf(int32_t max_seq_length);
    StatusType add_poa();
    StatusType add_seq_to_poa(const char* seq, const int8_t* weights, int32_t seq_len);
    void upload_inputs();
    void launch(void* event_after_graph_build = nullptr, uint64_t* phase_cycles = nullptr);
    gwhip_poa_args kernel_args() const;
    void log_kernel_error(StatusType error_type, std::vector<StatusType>& output_status);
    void fetch_consensus(std::string* consensus, std::vector<uint16_t>* coverage, StatusType* output_status, bool presize);
    size_t plan(int32_t n_poas, size_t* offsets) const;

    int32_t max_sequences_per_poa_ = 0;
    int32_t device_id_             = 0;
    cudaStream_t stream_           = nullptr;
    int8_t output_mask_            = 0;
    BatchConfig batch_size_;
    DefaultDeviceAllocator allocator_;
    gwhip_poa_config cfg_{};
    int32_t bid_      = 0;
    int32_t max_poas_ = 0;

    // host-side counters (reset() zeroes them)
    int32_t poa_count_              = 0;
    int32_t num_nucleotides_copied_ = 0;
    bool unit_weights_only_         = true; ///< no read of the batch carried base weights so far
    int32_t global_sequence_idx_    = 0;
    size_t avail_buf_mem_           = 0;
    size_t next_scores_offset_      = 0;
    size_t score_buffer_bytes_      = 0;

    // device block
    char* device_block_        = nullptr;
    size_t device_block_bytes_ = 0;
    size_t workspace_bytes_    = 0;
    size_t input_capacity_     = 0;
    uint8_t* d_sequences_      = nullptr;
    int8_t* d_weights_         = nullptr;
    int32_t* d_seq_lens_       = nullptr;
    gwhip_window_details* d_windows_ = nullptr;
    uint8_t* d_consensus_      = nullptr;
    uint16_t* d_coverage_      = nullptr;
    uint8_t* d_msa_            = nullptr;
    uint64_t* d_cells_         = nullptr;
    uint32_t* d_work_counters_ = nullptr; // two zeroed words behind the cell counters: the window counter of a persistent launch
    char* d_workspace_         = nullptr;

    // pinned staging block
    char* host_block_        = nullptr;
    size_t host_block_bytes_ = 0;
    size_t host_block_capacity_ = 0; // what the pinned cache handed out (pinned_release wants it back)
    uint8_t* h_sequences_    = nullptr;
    int8_t* h_weights_       = nullptr;
    int32_t* h_seq_lens_     = nullptr;
    gwhip_window_details* h_windows_ = nullptr;
    uint8_t* h_consensus_    = nullptr;
    uint16_t* h_coverage_    = nullptr;
    uint8_t* h_msa_          = nullptr;
    uint64_t* h_cells_       = nullptr;
};

} // namespace cudapoa
} // namespace genomeworks
} // namespace claraparabricks
