// batch.hpp -- the cudapoa::Batch interface: add windows ("POA groups"), run them on the GPU, read back
// consensus / MSA / graphs. Source-compatible with the reference's cudapoa/batch.hpp:46-204.
#pragma once

#include <claraparabricks/genomeworks/cudapoa/cudapoa.hpp>
#include <claraparabricks/genomeworks/utils/allocator.hpp>
#include <claraparabricks/genomeworks/utils/cudautils.hpp>
#include <claraparabricks/genomeworks/utils/graph.hpp>
#include <claraparabricks/genomeworks/utils/signed_integer_utils.hpp>

#include <cstdint>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudapoa
{

/// One read of a window. `seq` / `weights` are only borrowed during add_poa_group().
struct Entry
{
    const char* seq;       ///< bases (not NUL terminated)
    const int8_t* weights; ///< per-base weights, or nullptr for all-ones
    int32_t length;
};

/// The reads of one window.
typedef std::vector<Entry> Group;

/// Upper limits a batch is sized for.
struct BatchConfig
{
    int32_t max_sequence_size;         ///< longest read
    int32_t max_consensus_size;        ///< longest consensus / MSA row
    int32_t max_nodes_per_graph;       ///< node capacity of one POA graph
    int32_t matrix_sequence_dimension; ///< score-matrix row length
    int32_t alignment_band_width;      ///< band width (multiple of 128)
    int32_t max_sequences_per_poa;     ///< reads per window
    BandMode band_mode;
    int32_t max_banded_pred_distance;  ///< score rows kept in the traceback modes

    /// Derives the remaining limits from a minimal set.
    BatchConfig(int32_t max_seq_sz = 1024, int32_t max_seq_per_poa = 100, int32_t band_width = 256,
                BandMode banding = BandMode::full_band, float adapive_storage_factor = 2.0,
                float graph_length_factor = 3.0, int32_t max_pred_dist = 0);

    /// Sets every limit explicitly.
    BatchConfig(int32_t max_seq_sz, int32_t max_consensus_sz, int32_t max_nodes_per_poa, int32_t band_width,
                int32_t max_seq_per_poa, int32_t matrix_seq_dim, BandMode banding, int32_t max_pred_dist);
};

class Batch
{
public:
    virtual ~Batch() = default;

    /// Adds one window. per_seq_status receives one status per entry; the return value is the group status
    /// (exceeded_maximum_poas: run the batch, reset(), then retry this group).
    virtual StatusType add_poa_group(std::vector<StatusType>& per_seq_status, const Group& poa_group) = 0;

    virtual int32_t get_total_poas() const = 0;

    /// Uploads the windows and launches the kernels asynchronously on the batch's stream.
    virtual void generate_poa() = 0;

    /// Blocks until done; one consensus (+ per-base coverage) per window, status per window.
    virtual StatusType get_consensus(std::vector<std::string>& consensus, std::vector<std::vector<uint16_t>>& coverage,
                                     std::vector<genomeworks::cudapoa::StatusType>& output_status) = 0;

    /// Blocks until done; msa[window][read].
    virtual StatusType get_msa(std::vector<std::vector<std::string>>& msa, std::vector<StatusType>& output_status) = 0;

    /// Blocks until done; the POA graph of every window.
    virtual void get_graphs(std::vector<DirectedGraph>& graphs, std::vector<StatusType>& output_status) = 0;

    virtual int32_t batch_id() const = 0;

    /// Forget all windows (device buffers are reused as they are).
    virtual void reset() = 0;
};

std::unique_ptr<Batch> create_batch(int32_t device_id, cudaStream_t stream, DefaultDeviceAllocator allocator,
                                    int64_t max_gpu_mem, int8_t output_mask, const BatchConfig& batch_size,
                                    int16_t gap_score, int16_t mismatch_score, int16_t match_score);

/// max_gpu_mem == -1: use all available device memory.
std::unique_ptr<Batch> create_batch(int32_t device_id, cudaStream_t stream, int64_t max_gpu_mem, int8_t output_mask,
                                    const BatchConfig& batch_size, int16_t gap_score, int16_t mismatch_score,
                                    int16_t match_score);

} // namespace cudapoa
} // namespace genomeworks
} // namespace claraparabricks
