// ref_cudapoa_capi.cpp -- TEST INFRASTRUCTURE (oracle/simt): a flat C interface over the REFERENCE's cudapoa library -- its own
// batch.cu / cudapoa_batch.cuh / cudapoa_kernels.cuh ... compiled by g++ from /root/reference where they lie and run on the CPU
// by the SIMT emulator of simt.hpp (oracle/Makefile.ref, target ref_cudapoa_simt -> oracle/_ref/libref_cudapoa_simt.so). This
// file only calls the reference's public API (cudapoa/include/.../batch.hpp): create_batch, add_poa_group, generate_poa,
// get_consensus, get_msa. Used by tests/ref_cudapoa.py to check oracle/poa_oracle.c and to write tests/golden/reference_simt_*.
#include <claraparabricks/genomeworks/cudapoa/batch.hpp>
#include <claraparabricks/genomeworks/cudapoa/cudapoa.hpp>
#include <claraparabricks/genomeworks/cudapoa/utils.hpp>

#include "allocate_block.hpp"

#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace claraparabricks::genomeworks;
using namespace claraparabricks::genomeworks::cudapoa;

struct RefPoa
{
    std::unique_ptr<Batch> batch;
    std::vector<std::string> consensus;
    std::vector<std::vector<uint16_t>> coverage;
    std::vector<StatusType> status;
    std::vector<std::vector<std::string>> msa;
    std::vector<StatusType> msa_status;
};

#pragma GCC visibility push(default)
extern "C" {

// BatchConfig(max_seq_sz, max_seq_per_poa, band_width, banding, adaptive_storage_factor, graph_length_factor, max_pred_dist) of batch.hpp
void* ref_poa_create(int max_sequence_size, int max_sequences_per_poa, int band_width, int band_mode, float adaptive_storage_factor,
                     float graph_length_factor, int max_pred_distance, int output_mask, int gap, int mismatch, int match, long long max_mem)
{
    try
    {
        BatchConfig cfg(max_sequence_size, max_sequences_per_poa, band_width, static_cast<BandMode>(band_mode), adaptive_storage_factor, graph_length_factor,
                        max_pred_distance);
        auto* h  = new RefPoa;
        h->batch = create_batch(0, nullptr, max_mem, static_cast<int8_t>(output_mask), cfg, static_cast<int16_t>(gap), static_cast<int16_t>(mismatch),
                                static_cast<int16_t>(match));
        return h;
    }
    catch (...)
    {
        return nullptr;
    }
}

void ref_poa_config(int max_sequence_size, int max_sequences_per_poa, int band_width, int band_mode, float adaptive_storage_factor, float graph_length_factor,
                    int max_pred_distance, int* out8)
{
    BatchConfig c(max_sequence_size, max_sequences_per_poa, band_width, static_cast<BandMode>(band_mode), adaptive_storage_factor, graph_length_factor,
                  max_pred_distance);
    out8[0] = c.max_sequence_size, out8[1] = c.max_consensus_size, out8[2] = c.max_nodes_per_graph, out8[3] = c.matrix_sequence_dimension;
    out8[4] = c.alignment_band_width, out8[5] = c.max_sequences_per_poa, out8[6] = static_cast<int>(c.band_mode), out8[7] = c.max_banded_pred_distance;
}

void ref_poa_destroy(void* handle) { delete static_cast<RefPoa*>(handle); }

// BatchBlock<int32_t, int32_t, int16_t>::estimate_max_poas (allocate_block.hpp:403) for the BatchConfig of a group of `reads` sequences
// whose longest has `longest` bases -- what get_multi_batch_sizes() computes per group (utils.cu:52-58); the "free device memory" is
// what the stub's cudaMemGetInfo() reports
long long ref_poa_estimate_max_poas(int longest, int reads, int band_width, int band_mode, float adaptive_storage_factor, float graph_length_factor,
                                    int max_pred_distance, int msa_flag, float quota, int mismatch, int gap, int match)
{
    BatchConfig cfg(longest, reads, band_width, static_cast<BandMode>(band_mode), adaptive_storage_factor, graph_length_factor, max_pred_distance);
    return BatchBlock<int32_t, int32_t, int16_t>::estimate_max_poas(cfg, msa_flag != 0, quota, mismatch, gap, match);
}

// get_multi_batch_sizes (utils.cu:30-135) for groups given by their longest sequence and number of sequences (all it looks at).
// Out: n_batches; per batch eight BatchConfig fields in cfg8[8 b ..], its number of groups in per_batch[b], the group ids of all
// batches back to back in ids[].
int ref_poa_multi_batch_sizes(int n_groups, const int* longest, const int* reads, int msa_flag, int band_width, int band_mode,
                              float adaptive_storage_factor, float graph_length_factor, int max_pred_distance, float quota, int mismatch, int gap,
                              int match, int* cfg8, int* per_batch, int* ids)
{
    std::vector<Group> groups(static_cast<size_t>(n_groups));
    for (int g = 0; g < n_groups; ++g)
        for (int r = 0; r < reads[g]; ++r)
        {
            Entry e{};
            e.length = r == 0 ? longest[g] : std::max(1, longest[g] / 2);
            groups[static_cast<size_t>(g)].push_back(e);
        }
    std::vector<BatchConfig> cfgs;
    std::vector<std::vector<int32_t>> per;
    get_multi_batch_sizes(cfgs, per, groups, msa_flag != 0, band_width, static_cast<BandMode>(band_mode), adaptive_storage_factor, graph_length_factor,
                          max_pred_distance, nullptr, quota, mismatch, gap, match);
    int at = 0;
    for (size_t b = 0; b < cfgs.size(); ++b)
    {
        const BatchConfig& c = cfgs[b];
        int* o               = cfg8 + 8 * b;
        o[0] = c.max_sequence_size, o[1] = c.max_consensus_size, o[2] = c.max_nodes_per_graph, o[3] = c.matrix_sequence_dimension;
        o[4] = c.alignment_band_width, o[5] = c.max_sequences_per_poa, o[6] = static_cast<int>(c.band_mode), o[7] = c.max_banded_pred_distance;
        per_batch[b] = static_cast<int>(per[b].size());
        for (int32_t g : per[b]) ids[at++] = g;
    }
    return static_cast<int>(cfgs.size());
}

// -> StatusType of add_poa_group; per-read statuses into read_status[n_reads]; weights may be NULL (or weights[i] NULL)
int ref_poa_add_group(void* handle, int n_reads, const char** reads, const int* lengths, const signed char** weights, int* read_status)
{
    RefPoa* h = static_cast<RefPoa*>(handle);
    Group g;
    for (int i = 0; i < n_reads; ++i)
    {
        Entry e{};
        e.seq     = reads[i];
        e.weights = weights ? reinterpret_cast<const int8_t*>(weights[i]) : nullptr;
        e.length  = lengths[i];
        g.push_back(e);
    }
    std::vector<StatusType> st;
    try
    {
        const StatusType s = h->batch->add_poa_group(st, g);
        for (int i = 0; i < n_reads && i < static_cast<int>(st.size()); ++i) read_status[i] = static_cast<int>(st[i]);
        return static_cast<int>(s);
    }
    catch (...)
    {
        return -1;
    }
}

int ref_poa_total_poas(void* handle) { return static_cast<RefPoa*>(handle)->batch->get_total_poas(); }
void ref_poa_generate(void* handle) { static_cast<RefPoa*>(handle)->batch->generate_poa(); }
void ref_poa_reset(void* handle) { static_cast<RefPoa*>(handle)->batch->reset(); }

int ref_poa_fetch_consensus(void* handle)
{
    RefPoa* h = static_cast<RefPoa*>(handle);
    h->consensus.clear(), h->coverage.clear(), h->status.clear();
    return static_cast<int>(h->batch->get_consensus(h->consensus, h->coverage, h->status));
}
int ref_poa_consensus_length(void* handle, int w) { return static_cast<int>(static_cast<RefPoa*>(handle)->consensus[static_cast<size_t>(w)].size()); }
int ref_poa_window_status(void* handle, int w) { return static_cast<int>(static_cast<RefPoa*>(handle)->status[static_cast<size_t>(w)]); }
void ref_poa_consensus(void* handle, int w, char* bases, unsigned short* coverage)
{
    RefPoa* h = static_cast<RefPoa*>(handle);
    std::memcpy(bases, h->consensus[static_cast<size_t>(w)].data(), h->consensus[static_cast<size_t>(w)].size());
    std::memcpy(coverage, h->coverage[static_cast<size_t>(w)].data(), h->coverage[static_cast<size_t>(w)].size() * 2);
}

int ref_poa_fetch_msa(void* handle)
{
    RefPoa* h = static_cast<RefPoa*>(handle);
    h->msa.clear(), h->msa_status.clear();
    return static_cast<int>(h->batch->get_msa(h->msa, h->msa_status));
}
int ref_poa_msa_status(void* handle, int w) { return static_cast<int>(static_cast<RefPoa*>(handle)->msa_status[static_cast<size_t>(w)]); }
int ref_poa_msa_rows(void* handle, int w) { return static_cast<int>(static_cast<RefPoa*>(handle)->msa[static_cast<size_t>(w)].size()); }
int ref_poa_msa_row_length(void* handle, int w, int r) { return static_cast<int>(static_cast<RefPoa*>(handle)->msa[static_cast<size_t>(w)][static_cast<size_t>(r)].size()); }
void ref_poa_msa_row(void* handle, int w, int r, char* out)
{
    const std::string& s = static_cast<RefPoa*>(handle)->msa[static_cast<size_t>(w)][static_cast<size_t>(r)];
    std::memcpy(out, s.data(), s.size());
}
}
#pragma GCC visibility pop
