// cudapoa_utils.cpp -- batch-shape planning and window-file readers (include/.../cudapoa/utils.hpp).
// Behaviour follows the reference's cudapoa/src/utils.cu:30-146 (binning) and cudapoa/utils.hpp:77-187 (readers);
// FASTA is read with a small reader of our own (the reference goes through kseq++, which is not vendored).
#include <claraparabricks/genomeworks/cudapoa/utils.hpp>
#include <claraparabricks/genomeworks/utils/signed_integer_utils.hpp>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "../../include/gwhip.h"
#include "poa_batch_impl.hpp"

namespace claraparabricks
{
namespace genomeworks
{
namespace cudapoa
{

int32_t estimate_max_poas(const BatchConfig& batch_size, bool msa_flag, float gpu_memory_usage_quota, int32_t mismatch_score,
                          int32_t gap_score, int32_t match_score)
{
    size_t free_bytes = 0, total_bytes = 0;
    if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess) throw std::runtime_error("hipMemGetInfo failed");
    const int64_t budget = static_cast<int64_t>(static_cast<double>(gpu_memory_usage_quota) * static_cast<double>(free_bytes));
    const gwhip_poa_config cfg =
        make_device_config(batch_size, static_cast<int8_t>(msa_flag ? OutputType::msa : OutputType::consensus), gap_score,
                           mismatch_score, match_score);
    int64_t per_poa = 0, per_matrix = 0;
    gwhip_poa_bytes_per_window(&cfg, &per_poa, &per_matrix);
    const int64_t per_window = per_poa + per_matrix;
    if (budget < per_window)
        throw std::runtime_error("Requires at least " + std::to_string(per_window) +
                                 " bytes of device memory per CUDAPOA batch to process correctly.");
    return static_cast<int32_t>(std::min<int64_t>(budget / per_window, INT32_MAX));
}

// The binning rule on its own (no device query): capacity[i] = POAs of group i's shape that fit the device,
// longest[i] / reads[i] = its longest read and number of reads. Exposed to the tests through the C API.
void bin_poa_groups(std::vector<BatchConfig>& list_of_batch_sizes, std::vector<std::vector<int32_t>>& list_of_groups_per_batch,
                    const std::vector<int32_t>& capacity, const std::vector<int32_t>& longest, const std::vector<int32_t>& reads,
                    int32_t band_width, BandMode band_mode, float adaptive_storage_factor, float graph_length_factor,
                    int32_t max_pred_distance, const std::vector<int32_t>* bins_capacity)
{
    const int32_t num_groups = get_size<int32_t>(capacity);
    // bins: capacity thresholds 1, 2, 4, ... (20 of them) unless the caller supplies its own
    std::vector<int32_t> default_bins;
    if (bins_capacity == nullptr)
    {
        default_bins.resize(20);
        for (size_t j = 0; j < default_bins.size(); j++) default_bins[j] = 1 << j;
        bins_capacity = &default_bins;
    }
    const int32_t num_bins = get_size<int32_t>(*bins_capacity);
    struct Bin
    {
        int32_t count = 0, longest_read = 0, most_reads = 0;
        std::vector<int32_t> groups;
    };
    std::vector<Bin> bins(static_cast<size_t>(num_bins));
    for (int32_t i = 0; i < num_groups; i++)
    {
        // first bin whose capacity covers the group's; the last bin takes everything beyond
        int32_t j = 0;
        while (j < num_bins - 1 && capacity[i] > bins_capacity->at(j)) j++;
        Bin& b = bins[j];
        b.count++;
        b.groups.push_back(i);
        b.longest_read = std::max(b.longest_read, longest[i]);
        b.most_reads   = std::max(b.most_reads, reads[i]);
    }

    // one batch per non-empty bin; a batch built for bin j holds up to capacity(j) POAs of its (larger) shape, so the
    // groups of the following bins ride along as long as each of those bins fits that capacity as a whole
    for (int32_t j = 0; j < num_bins; j++)
    {
        if (bins[j].count == 0) continue;
        list_of_batch_sizes.emplace_back(bins[j].longest_read, bins[j].most_reads, band_width, band_mode,
                                         adaptive_storage_factor, graph_length_factor, max_pred_distance);
        list_of_groups_per_batch.push_back(bins[j].groups);
        std::vector<int32_t>& merged = list_of_groups_per_batch.back();
        for (int32_t k = j + 1; k < num_bins; k++)
        {
            if (bins[k].count == 0) continue;
            if (bins_capacity->at(j) < bins[k].count) break;
            merged.insert(merged.end(), bins[k].groups.begin(), bins[k].groups.end());
            bins[k].count = 0;
        }
    }
}

void get_multi_batch_sizes(std::vector<BatchConfig>& list_of_batch_sizes,
                           std::vector<std::vector<int32_t>>& list_of_groups_per_batch, const std::vector<Group>& poa_groups,
                           bool msa_flag, int32_t band_width, BandMode band_mode, float adaptive_storage_factor,
                           float graph_length_factor, int32_t max_pred_distance, std::vector<int32_t>* bins_capacity,
                           float gpu_memory_usage_quota, int32_t mismatch_score, int32_t gap_score, int32_t match_score)
{
    const int32_t num_groups = get_size<int32_t>(poa_groups);
    // capacity (POAs per batch), longest read and read count of every group, each sized on its own
    std::vector<int32_t> capacity(num_groups), longest(num_groups), reads(num_groups);
    for (int32_t i = 0; i < num_groups; i++)
    {
        int32_t len = 0;
        for (const Entry& e : poa_groups[i]) len = std::max(len, e.length);
        longest[i] = len;
        reads[i]   = get_size<int32_t>(poa_groups[i]);
        const BatchConfig shape(len, reads[i], band_width, band_mode, adaptive_storage_factor, graph_length_factor,
                                max_pred_distance);
        capacity[i] = estimate_max_poas(shape, msa_flag, gpu_memory_usage_quota, mismatch_score, gap_score, match_score);
    }
    bin_poa_groups(list_of_batch_sizes, list_of_groups_per_batch, capacity, longest, reads, band_width, band_mode,
                   adaptive_storage_factor, graph_length_factor, max_pred_distance, bins_capacity);
}

void resize_windows(std::vector<std::vector<std::string>>& windows, const int32_t total_windows)
{
    if (total_windows < 0) return;
    const size_t want = static_cast<size_t>(total_windows);
    if (windows.size() > want)
        windows.resize(want);
    else if (windows.size() < want && !windows.empty())
    {
        const size_t period = windows.size(); // cycle through the windows that were read
        windows.reserve(want);
        while (windows.size() < want) windows.push_back(windows[windows.size() - period]);
    }
}

void parse_cudapoa_file(std::vector<std::vector<std::string>>& windows, const std::string& filename, int32_t total_windows)
{
    std::ifstream in(filename);
    if (!in.good()) throw std::runtime_error("Cannot read file " + filename);
    std::string line;
    int32_t remaining = 0;
    while (std::getline(in, line))
    {
        if (remaining == 0)
        {
            std::istringstream header(line);
            header >> remaining;
            windows.emplace_back();
        }
        else
        {
            windows.back().push_back(line);
            remaining--;
        }
    }
    resize_windows(windows, total_windows);
}

namespace
{
// all records of a FASTA file, sequence lines joined; '>' starts a record, blank lines and '\r' are ignored
std::vector<std::string> read_fasta_records(const std::string& path)
{
    std::ifstream in(path);
    if (!in.good()) throw std::runtime_error("Cannot read file " + path);
    std::vector<std::string> records;
    std::string line;
    bool open = false;
    while (std::getline(in, line))
    {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>')
        {
            records.emplace_back();
            open = true;
        }
        else if (open)
            records.back() += line;
        else
            throw std::runtime_error("Invalid FASTA file (sequence before the first header): " + path);
    }
    return records;
}
} // namespace

void parse_fasta_files(std::vector<std::vector<std::string>>& windows, const std::vector<std::string>& input_paths,
                       const int32_t total_windows)
{
    windows.resize(input_paths.size());
    for (size_t i = 0; i < input_paths.size(); i++)
        for (std::string& record : read_fasta_records(input_paths[i])) windows[i].push_back(std::move(record));
    resize_windows(windows, total_windows);
}

std::string parse_golden_value_file(const std::string& filename)
{
    std::ifstream in(filename);
    if (!in.good()) throw std::runtime_error("Cannot read file " + filename);
    std::string line;
    std::getline(in, line);
    return line;
}

} // namespace cudapoa
} // namespace genomeworks
} // namespace claraparabricks
