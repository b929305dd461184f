// utils.hpp -- callers' helpers around the POA batch: splitting a set of POA groups into the fewest batch shapes
// that fit the GPU, and the readers for the two on-disk window formats.
// Source-compatible with the reference's cudapoa/utils.hpp:36-187 (same names, parameters, defaults, and the same
// binning rule as cudapoa/src/utils.cu:30-146); implemented in genomeworks_amd/host/cudapoa_utils.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "batch.hpp"
#include "cudapoa.hpp"

namespace claraparabricks
{
namespace genomeworks
{
namespace cudapoa
{

/// Creates the batch shapes needed to process `poa_groups` and assigns every group to one of them.
/// Groups are binned by how many POAs of their size fit on the device (power-of-two bins 1, 2, 4 ... unless
/// `bins_capacity` is given); one BatchConfig is emitted per non-empty bin (sized for the bin's longest read and
/// largest group), and the groups of later (smaller) bins are folded into it while they fit its capacity.
/// \param[out] list_of_batch_sizes      one BatchConfig per batch to create
/// \param[out] list_of_groups_per_batch for each of them, the indices (into poa_groups) it has to process
void get_multi_batch_sizes(std::vector<BatchConfig>& list_of_batch_sizes,
                           std::vector<std::vector<int32_t>>& list_of_groups_per_batch,
                           const std::vector<Group>& poa_groups,
                           bool msa_flag                       = false,
                           int32_t band_width                  = 256,
                           BandMode band_mode                  = BandMode::adaptive_band,
                           float adaptive_storage_factor       = 2.0f,
                           float graph_length_factor           = 3.0f,
                           int32_t max_pred_distance           = 0,
                           std::vector<int32_t>* bins_capacity = nullptr,
                           float gpu_memory_usage_quota        = 0.9,
                           int32_t mismatch_score              = -6,
                           int32_t gap_score                   = -8,
                           int32_t match_score                 = 8);

/// How many POAs of this shape fit into `gpu_memory_usage_quota` of the device memory that is free right now
/// (the reference's BatchBlock::estimate_max_poas, allocate_block.hpp:329-376, with this engine's byte counts).
int32_t estimate_max_poas(const BatchConfig& batch_size, bool msa_flag = false, float gpu_memory_usage_quota = 0.9,
                          int32_t mismatch_score = -6, int32_t gap_score = -8, int32_t match_score = 8);

/// Truncates `windows` to `total_windows`, or repeats the windows read so far until there are that many
/// (total_windows < 0: leave as is).
void resize_windows(std::vector<std::vector<std::string>>& windows, int32_t total_windows);

/// Reads a cudapoa window file: a line with the number of sequences of a window, then that many sequence lines, ...
void parse_cudapoa_file(std::vector<std::vector<std::string>>& windows, const std::string& filename, int32_t total_windows);

/// Reads one window per FASTA file (all records of file i become the sequences of window i).
void parse_fasta_files(std::vector<std::vector<std::string>>& windows, const std::vector<std::string>& input_paths,
                       int32_t total_windows);

/// First line of a golden-value (expected genome) file.
std::string parse_golden_value_file(const std::string& filename);

} // namespace cudapoa
} // namespace genomeworks
} // namespace claraparabricks
