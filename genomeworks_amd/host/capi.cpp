// capi.cpp -- object-level C-ABI (include/gw_capi.h) over the host C++ classes: what a Cython / ctypes / cgo
// binding of cudapoa::Batch and cudaaligner::Aligner binds. Exceptions never cross the boundary.
#include <claraparabricks/genomeworks/cudapoa/batch.hpp>
#include <claraparabricks/genomeworks/cudapoa/cudapoa.hpp>
#include <claraparabricks/genomeworks/cudapoa/utils.hpp>

#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/gw_capi.h"
#include "host_common.hpp"
#include "poa_batch_impl.hpp"
#include "aligner_impl.hpp"
#include "aligner_global.hpp"
#include "alignment_impl.hpp"
#include <claraparabricks/genomeworks/cudapoa/multi_device.hpp>
#include <claraparabricks/genomeworks/utils/cudautils.hpp>
#include <stdexcept>
#include <claraparabricks/genomeworks/cudaaligner/aligner.hpp>
#include <claraparabricks/genomeworks/cudaaligner/alignment.hpp>

namespace gw  = claraparabricks::genomeworks;
namespace poa = claraparabricks::genomeworks::cudapoa;
namespace aln = claraparabricks::genomeworks::cudaaligner;

struct gw_poa_batch
{
    std::unique_ptr<poa::Batch> batch;
    poa::PoaBatch* impl = nullptr;
    std::vector<std::string> consensus;
    std::vector<std::vector<uint16_t>> coverage;
    std::vector<poa::StatusType> status;
    std::vector<std::vector<std::string>> msa;
    std::vector<gw::DirectedGraph> graphs;
    std::vector<std::vector<std::pair<gw::Graph::edge_t, gw::Graph::edge_weight_t>>> graph_edges;
    std::vector<int32_t> graph_nodes;
};

struct gw_aligner
{
    std::unique_ptr<aln::Aligner> aligner;
    aln::BandedAligner* impl = nullptr;
    std::string cigar; // last string handed out
};

#define GW_TRY try {
#define GW_CATCH(ret)                                                                                                  \
    }                                                                                                                  \
    catch (const std::exception& e)                                                                                    \
    {                                                                                                                  \
        gwhost::set_last_error(e.what());                                                                              \
        return ret;                                                                                                    \
    }                                                                                                                  \
    catch (...)                                                                                                        \
    {                                                                                                                  \
        gwhost::set_last_error("unknown exception");                                                                   \
        return ret;                                                                                                    \
    }

static void fill(gw_poa_batch_config* out, const poa::BatchConfig& c)
{
    out->max_sequence_size         = c.max_sequence_size;
    out->max_consensus_size        = c.max_consensus_size;
    out->max_nodes_per_graph       = c.max_nodes_per_graph;
    out->matrix_sequence_dimension = c.matrix_sequence_dimension;
    out->alignment_band_width      = c.alignment_band_width;
    out->max_sequences_per_poa     = c.max_sequences_per_poa;
    out->band_mode                 = static_cast<int32_t>(c.band_mode);
    out->max_banded_pred_distance  = c.max_banded_pred_distance;
}

extern "C" {

int gw_poa_batch_config_default(gw_poa_batch_config* out, int32_t max_seq_sz, int32_t max_seq_per_poa, int32_t band_width,
                                int32_t band_mode, float adaptive_storage_factor, float graph_length_factor,
                                int32_t max_pred_dist)
{
    GW_TRY
    poa::BatchConfig c(max_seq_sz, max_seq_per_poa, band_width, static_cast<poa::BandMode>(band_mode),
                       adaptive_storage_factor, graph_length_factor, max_pred_dist);
    fill(out, c);
    return 0;
    GW_CATCH(-1)
}

int gw_poa_batch_config_full(gw_poa_batch_config* out, int32_t max_seq_sz, int32_t max_consensus_sz,
                             int32_t max_nodes_per_poa, int32_t band_width, int32_t max_seq_per_poa,
                             int32_t matrix_seq_dim, int32_t band_mode, int32_t max_pred_distance)
{
    GW_TRY
    poa::BatchConfig c(max_seq_sz, max_consensus_sz, max_nodes_per_poa, band_width, max_seq_per_poa, matrix_seq_dim,
                       static_cast<poa::BandMode>(band_mode), max_pred_distance);
    fill(out, c);
    return 0;
    GW_CATCH(-1)
}

gw_poa_batch* gw_poa_create_batch(int32_t device_id, void* stream, int64_t max_mem, int8_t output_mask,
                                  const gw_poa_batch_config* cfg, int16_t gap_score, int16_t mismatch_score,
                                  int16_t match_score)
{
    GW_TRY
    // the fields were validated when the config was built; rebuild through the explicit ctor
    poa::BatchConfig c(cfg->max_sequence_size, cfg->max_consensus_size, cfg->max_nodes_per_graph,
                       cfg->alignment_band_width, cfg->max_sequences_per_poa, cfg->matrix_sequence_dimension,
                       static_cast<poa::BandMode>(cfg->band_mode), cfg->max_banded_pred_distance);
    auto h   = std::make_unique<gw_poa_batch>();
    h->batch = poa::create_batch(device_id, static_cast<cudaStream_t>(stream), max_mem, output_mask, c, gap_score,
                                 mismatch_score, match_score);
    h->impl  = dynamic_cast<poa::PoaBatch*>(h->batch.get());
    return h.release();
    GW_CATCH(nullptr)
}

void gw_poa_destroy_batch(gw_poa_batch* b) { delete b; }

int gw_poa_add_poa_group(gw_poa_batch* b, int32_t n, const char* const* seqs, const int8_t* const* weights,
                         const int32_t* lengths, int32_t* per_seq_status)
{
    GW_TRY
    poa::Group g;
    g.reserve(static_cast<size_t>(n));
    for (int32_t i = 0; i < n; i++) g.push_back(poa::Entry{seqs[i], weights ? weights[i] : nullptr, lengths[i]});
    std::vector<poa::StatusType> st;
    poa::StatusType r = b->batch->add_poa_group(st, g);
    if (per_seq_status)
        for (size_t i = 0; i < st.size() && i < static_cast<size_t>(n); i++) per_seq_status[i] = static_cast<int32_t>(st[i]);
    return static_cast<int>(r);
    GW_CATCH(-1)
}

int32_t gw_poa_get_total_poas(gw_poa_batch* b) { return b->batch->get_total_poas(); }
int32_t gw_poa_batch_id(gw_poa_batch* b) { return b->batch->batch_id(); }
int32_t gw_poa_max_poas(gw_poa_batch* b) { return b->impl ? b->impl->max_poas() : -1; }

int gw_poa_generate_poa(gw_poa_batch* b)
{
    GW_TRY
    b->batch->generate_poa();
    return 0;
    GW_CATCH(-1)
}

int gw_poa_reset(gw_poa_batch* b)
{
    GW_TRY
    b->batch->reset();
    return 0;
    GW_CATCH(-1)
}

// What a source-compatible caller of cudapoa::Batch does (cudapoa/benchmarks/single_batch.hpp:86-93): three fresh vectors,
// the public virtual get_consensus(), and the previous call's results destroyed. The handle keeps the new vectors for the
// accessors below. On any failure (status or exception) the handle shows no results at all.
int gw_poa_get_consensus(gw_poa_batch* b, int32_t* n_out)
{
    if (n_out) *n_out = 0;
    try
    {
        std::vector<std::string> consensus;
        std::vector<std::vector<uint16_t>> coverage;
        std::vector<poa::StatusType> status;
        poa::StatusType r = b->batch->get_consensus(consensus, coverage, status);
        if (r != poa::StatusType::success)
        {
            consensus.clear();
            coverage.clear();
            status.clear();
        }
        b->consensus = std::move(consensus); // the old vectors die here, as the benchmark's do at the end of its scope
        b->coverage  = std::move(coverage);
        b->status    = std::move(status);
        if (n_out) *n_out = static_cast<int32_t>(b->consensus.size());
        return static_cast<int>(r);
    }
    catch (const std::exception& e)
    {
        gwhost::set_last_error(e.what());
    }
    catch (...)
    {
        gwhost::set_last_error("unknown exception");
    }
    b->consensus.clear();
    b->coverage.clear();
    b->status.clear();
    return -1;
}

// EXTENSION (not part of cudapoa::Batch): the same fetch into the strings and vectors of the handle's previous call, whose
// storage is reused -- no heap traffic in a steady-state loop. bench.py reports it next to the metric, never as the metric.
int gw_poa_get_consensus_in_place(gw_poa_batch* b, int32_t* n_out)
{
    if (n_out) *n_out = 0;
    try
    {
        poa::StatusType r = b->impl ? b->impl->get_consensus_in_place(b->consensus, b->coverage, b->status)
                                    : (b->consensus.clear(), b->coverage.clear(), b->status.clear(),
                                       b->batch->get_consensus(b->consensus, b->coverage, b->status));
        if (r != poa::StatusType::success) // nothing was fetched: the handle must not keep showing the previous call's results
        {
            b->consensus.clear();
            b->coverage.clear();
            b->status.clear();
        }
        if (n_out) *n_out = static_cast<int32_t>(b->consensus.size());
        return static_cast<int>(r);
    }
    catch (const std::exception& e)
    {
        gwhost::set_last_error(e.what());
    }
    catch (...)
    {
        gwhost::set_last_error("unknown exception");
    }
    b->consensus.clear(); // a throwing fetch leaves the vectors resized and partly overwritten: show nothing
    b->coverage.clear();
    b->status.clear();
    return -1;
}

// Accessors of the last get_*() call: an index outside it sets the error string and returns null / -1 (no exception
// crosses the C ABI).
const char* gw_poa_consensus_str(gw_poa_batch* b, int32_t poa_idx, int32_t* length)
{
    GW_TRY
    const std::string& s = b->consensus.at(static_cast<size_t>(poa_idx));
    if (length) *length = static_cast<int32_t>(s.size());
    return s.c_str();
    GW_CATCH(nullptr)
}

const uint16_t* gw_poa_consensus_coverage(gw_poa_batch* b, int32_t poa_idx, int32_t* length)
{
    GW_TRY
    const auto& v = b->coverage.at(static_cast<size_t>(poa_idx));
    if (length) *length = static_cast<int32_t>(v.size());
    return v.data();
    GW_CATCH(nullptr)
}

int32_t gw_poa_output_status(gw_poa_batch* b, int32_t poa_idx)
{
    GW_TRY
    return static_cast<int32_t>(b->status.at(static_cast<size_t>(poa_idx)));
    GW_CATCH(-1)
}

int gw_poa_get_msa(gw_poa_batch* b, int32_t* n_out)
{
    GW_TRY
    b->msa.clear();
    b->status.clear();
    poa::StatusType r = b->batch->get_msa(b->msa, b->status);
    if (n_out) *n_out = static_cast<int32_t>(b->msa.size());
    return static_cast<int>(r);
    GW_CATCH(-1)
}

int32_t gw_poa_msa_rows(gw_poa_batch* b, int32_t poa_idx)
{
    GW_TRY
    return static_cast<int32_t>(b->msa.at(static_cast<size_t>(poa_idx)).size());
    GW_CATCH(-1)
}

const char* gw_poa_msa_row(gw_poa_batch* b, int32_t poa_idx, int32_t row, int32_t* length)
{
    GW_TRY
    const std::string& s = b->msa.at(static_cast<size_t>(poa_idx)).at(static_cast<size_t>(row));
    if (length) *length = static_cast<int32_t>(s.size());
    return s.c_str();
    GW_CATCH(nullptr)
}

int gw_poa_get_graphs(gw_poa_batch* b, int32_t* n_out)
{
    GW_TRY
    b->graphs.clear();
    b->status.clear();
    b->batch->get_graphs(b->graphs, b->status);
    b->graph_edges.clear();
    b->graph_nodes.clear();
    for (const auto& g : b->graphs)
    {
        b->graph_edges.push_back(g.get_edges());
        int32_t n = 0;
        while (!g.get_node_label(n).empty()) n++;
        b->graph_nodes.push_back(n);
    }
    if (n_out) *n_out = static_cast<int32_t>(b->graphs.size());
    return 0;
    GW_CATCH(-1)
}

int32_t gw_poa_graph_num_nodes(gw_poa_batch* b, int32_t poa_idx)
{
    GW_TRY
    return b->graph_nodes.at(static_cast<size_t>(poa_idx));
    GW_CATCH(-1)
}
int32_t gw_poa_graph_num_edges(gw_poa_batch* b, int32_t poa_idx)
{
    GW_TRY
    return static_cast<int32_t>(b->graph_edges.at(static_cast<size_t>(poa_idx)).size());
    GW_CATCH(-1)
}

int gw_poa_graph_copy(gw_poa_batch* b, int32_t poa_idx, char* node_labels, int32_t* edge_src, int32_t* edge_dst, int32_t* edge_weight)
{
    GW_TRY
    const auto& g = b->graphs.at(static_cast<size_t>(poa_idx));
    const int32_t n = b->graph_nodes.at(static_cast<size_t>(poa_idx));
    for (int32_t i = 0; i < n; i++) node_labels[i] = g.get_node_label(i)[0];
    const auto& e = b->graph_edges.at(static_cast<size_t>(poa_idx));
    for (size_t i = 0; i < e.size(); i++)
    {
        edge_src[i]    = e[i].first.first;
        edge_dst[i]    = e[i].first.second;
        edge_weight[i] = e[i].second;
    }
    return 0;
    GW_CATCH(-1)
}

int gw_poa_total_cells(gw_poa_batch* b, uint64_t* cells)
{
    GW_TRY
    if (!b->impl) return -1;
    *cells = b->impl->total_cells();
    return 0;
    GW_CATCH(-1)
}

int gw_poa_relaunch(gw_poa_batch* b)
{
    GW_TRY
    if (!b->impl) return -1;
    b->impl->relaunch_resident();
    return 0;
    GW_CATCH(-1)
}

int gw_poa_relaunch_timed(gw_poa_batch* b, float* graph_build_ms, float* output_ms)
{
    GW_TRY
    if (!b->impl) return -1;
    b->impl->relaunch_resident_timed(graph_build_ms, output_ms);
    return 0;
    GW_CATCH(-1)
}

int gw_poa_profile_phases(gw_poa_batch* b, double* out6)
{
    GW_TRY
    if (!b->impl) return -1;
    b->impl->profile_phases(out6);
    return 0;
    GW_CATCH(-1)
}

int gw_poa_profile_phases_per_window(gw_poa_batch* b, uint64_t* out, int32_t capacity_windows)
{
    GW_TRY
    if (!b->impl) return -1;
    std::vector<uint64_t> ticks;
    b->impl->profile_phases_per_window(ticks);
    const int32_t n = std::min<int32_t>(static_cast<int32_t>(ticks.size() / 6), capacity_windows);
    std::copy(ticks.begin(), ticks.begin() + static_cast<size_t>(n) * 6, out);
    return n;
    GW_CATCH(-1)
}

// ---- cudaaligner --------------------------------------------------------------------------------------
gw_aligner* gw_aligner_create_banded(int32_t max_bandwidth, void* stream, int32_t device_id, int64_t max_device_memory)
{
    GW_TRY
    auto h     = std::make_unique<gw_aligner>();
    h->aligner = aln::create_aligner(aln::AlignmentType::global_alignment, max_bandwidth, static_cast<cudaStream_t>(stream),
                                     device_id, max_device_memory);
    h->impl    = dynamic_cast<aln::BandedAligner*>(h->aligner.get());
    return h.release();
    GW_CATCH(nullptr)
}

gw_aligner* gw_aligner_create(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments, void* stream,
                              int32_t device_id, int64_t max_device_memory)
{
    GW_TRY
    auto h     = std::make_unique<gw_aligner>();
    h->aligner = aln::create_aligner(max_query_length, max_target_length, max_alignments, aln::AlignmentType::global_alignment,
                                     static_cast<cudaStream_t>(stream), device_id, max_device_memory);
    h->impl    = dynamic_cast<aln::BandedAligner*>(h->aligner.get());
    return h.release();
    GW_CATCH(nullptr)
}

gw_aligner* gw_aligner_create_algorithm(const char* algorithm, int32_t max_query_length, int32_t max_target_length,
                                        int32_t max_alignments, void* stream, int32_t device_id, int64_t max_device_memory)
{
    GW_TRY
    const std::string algo = algorithm ? algorithm : "default";
    if (algo == "default")
        return gw_aligner_create(max_query_length, max_target_length, max_alignments, stream, device_id, max_device_memory);
    gw::scoped_device_switch device(device_id);
    if (max_device_memory < -1) throw std::invalid_argument("max_device_memory has to be -1 (all available memory) or >= 0.");
    if (max_device_memory == -1) max_device_memory = gw::cudautils::find_largest_contiguous_device_memory_section();
    gw::DefaultDeviceAllocator allocator(static_cast<size_t>(max_device_memory), static_cast<cudaStream_t>(stream));
    auto h = std::make_unique<gw_aligner>();
    if (algo == "hirschberg_myers")
        h->aligner = std::make_unique<aln::AlignerGlobalHirschbergMyers>(max_query_length, max_target_length, max_alignments, allocator,
                                                                          static_cast<cudaStream_t>(stream), device_id);
    else if (algo == "ukkonen")
        h->aligner = std::make_unique<aln::AlignerGlobalUkkonen>(max_query_length, max_target_length, max_alignments, allocator,
                                                                  static_cast<cudaStream_t>(stream), device_id);
    else if (algo == "myers")
        h->aligner = std::make_unique<aln::AlignerGlobalMyers>(max_query_length, max_target_length, max_alignments, allocator,
                                                                static_cast<cudaStream_t>(stream), device_id);
    else
        throw std::invalid_argument("unknown aligner algorithm '" + algo + "' (default, hirschberg_myers, ukkonen, myers)");
    h->impl = dynamic_cast<aln::BandedAligner*>(h->aligner.get());
    return h.release();
    GW_CATCH(nullptr)
}

void gw_aligner_destroy(gw_aligner* a) { delete a; }

int gw_aligner_add_alignment(gw_aligner* a, const char* query, int32_t query_length, const char* target,
                             int32_t target_length, int reverse_complement_query, int reverse_complement_target)
{
    GW_TRY
    return static_cast<int>(a->aligner->add_alignment(query, query_length, target, target_length, reverse_complement_query != 0,
                                                      reverse_complement_target != 0));
    GW_CATCH(-1)
}

int gw_aligner_align_all(gw_aligner* a)
{
    GW_TRY
    return static_cast<int>(a->aligner->align_all());
    GW_CATCH(-1)
}

int gw_aligner_sync_alignments(gw_aligner* a)
{
    GW_TRY
    return static_cast<int>(a->aligner->sync_alignments());
    GW_CATCH(-1)
}

int32_t gw_aligner_num_alignments(gw_aligner* a)
{
    GW_TRY
    return static_cast<int32_t>(a->aligner->get_alignments().size());
    GW_CATCH(-1)
}

int gw_aligner_reset(gw_aligner* a)
{
    GW_TRY
    a->aligner->reset();
    return 0;
    GW_CATCH(-1)
}

// Per-alignment accessors: an index outside get_alignments() sets the error string and returns -1 / null.
int32_t gw_alignment_status(gw_aligner* a, int32_t i)
{
    GW_TRY
    return static_cast<int32_t>(a->aligner->get_alignments().at(static_cast<size_t>(i))->get_status());
    GW_CATCH(-1)
}
int32_t gw_alignment_is_optimal(gw_aligner* a, int32_t i)
{
    GW_TRY
    return a->aligner->get_alignments().at(static_cast<size_t>(i))->is_optimal() ? 1 : 0;
    GW_CATCH(-1)
}
int32_t gw_alignment_edit_distance(gw_aligner* a, int32_t i)
{
    GW_TRY
    return a->aligner->get_alignments().at(static_cast<size_t>(i))->get_edit_distance();
    GW_CATCH(-1)
}

const char* gw_alignment_cigar(gw_aligner* a, int32_t i, int32_t extended, int32_t* length)
{
    GW_TRY
    a->cigar = a->aligner->get_alignments().at(static_cast<size_t>(i))->convert_to_cigar(extended ? aln::CigarFormat::extended : aln::CigarFormat::basic);
    if (length) *length = static_cast<int32_t>(a->cigar.size());
    return a->cigar.c_str();
    GW_CATCH(nullptr)
}

int32_t gw_alignment_states(gw_aligner* a, int32_t i, int8_t* out, int32_t cap)
{
    GW_TRY
    const auto& al = *a->aligner->get_alignments().at(static_cast<size_t>(i));
    // per-position states from whichever form the aligner filled
    int32_t n = 0;
    if (!al.get_actions().empty())
    {
        for (size_t k = 0; k < al.get_actions().size(); ++k)
            for (int32_t r = 0; r < al.get_runlengths()[k]; ++r, ++n)
                if (out && n < cap) out[n] = al.get_actions()[k];
    }
    else
        for (auto s : al.get_alignment())
        {
            if (out && n < cap) out[n] = static_cast<int8_t>(s);
            ++n;
        }
    return n;
    GW_CATCH(-1)
}

int64_t gw_aligner_get_runs(gw_aligner* a, int64_t* offsets, int8_t* ops, int32_t* counts, int64_t capacity, int32_t* status,
                            int32_t* optimal)
{
    GW_TRY
    const auto& all = a->aligner->get_alignments();
    int64_t total   = 0;
    for (size_t i = 0; i < all.size(); ++i)
    {
        const aln::Alignment& al = *all[i];
        if (offsets) offsets[i] = total;
        if (status) status[i] = static_cast<int32_t>(al.get_status());
        if (optimal) optimal[i] = al.is_optimal() ? 1 : 0;
        auto emit = [&](int8_t op, int32_t count) {
            if (ops && counts && total < capacity)
            {
                ops[total]    = op;
                counts[total] = count;
            }
            ++total;
        };
        if (const auto* pk = dynamic_cast<const aln::PackedAlignment*>(&al))
        {
            if (a->impl && a->impl->expands_results())
            {
                // per-position results: runs of equal states
                int8_t last = -1;
                int32_t acc = 0;
                for (int32_t k = 0; k < pk->num_runs(); ++k)
                {
                    if (pk->op(k) == last) acc += pk->count(k);
                    else
                    {
                        if (last >= 0) emit(last, acc);
                        last = pk->op(k);
                        acc  = pk->count(k);
                    }
                }
                if (last >= 0) emit(last, acc);
            }
            else
                for (int32_t k = 0; k < pk->num_runs(); ++k) emit(pk->op(k), pk->count(k));
        }
        else if (!al.get_actions().empty())
            for (size_t k = 0; k < al.get_actions().size(); ++k) emit(al.get_actions()[k], al.get_runlengths()[k]);
        else
        {
            int8_t last = -1;
            int32_t acc = 0;
            for (auto st : al.get_alignment())
            {
                if (static_cast<int8_t>(st) == last) ++acc;
                else
                {
                    if (last >= 0) emit(last, acc);
                    last = static_cast<int8_t>(st);
                    acc  = 1;
                }
            }
            if (last >= 0) emit(last, acc);
        }
    }
    if (offsets) offsets[all.size()] = total;
    return total;
    GW_CATCH(-1)
}

int gw_aligner_device_alignments(gw_aligner* a, int32_t* n_alignments, int64_t* total_length)
{
    GW_TRY
    gw::scoped_device_switch dev(a->aligner->get_device());
    GW_CU_CHECK_ERR(hipStreamSynchronize(a->aligner->get_stream()));
    const aln::DeviceAlignmentsPtrs d = a->aligner->get_alignments_device();
    if (n_alignments) *n_alignments = d.cigar_operations ? d.n_alignments : 0;
    if (total_length) *total_length = d.cigar_operations ? d.total_length : 0;
    return d.cigar_operations ? 0 : 1;
    GW_CATCH(-1)
}

int gw_aligner_copy_device_alignments(gw_aligner* a, int8_t* cigar_operations, int32_t* cigar_runlengths, int32_t* cigar_offsets,
                                      uint32_t* metadata)
{
    GW_TRY
    gw::scoped_device_switch dev(a->aligner->get_device());
    hipStream_t stream = a->aligner->get_stream();
    GW_CU_CHECK_ERR(hipStreamSynchronize(stream));
    const aln::DeviceAlignmentsPtrs d = a->aligner->get_alignments_device();
    if (!d.cigar_operations) return 1;
    const size_t n = static_cast<size_t>(d.n_alignments), total = static_cast<size_t>(d.total_length);
    if (cigar_operations && total) GW_CU_CHECK_ERR(hipMemcpyAsync(cigar_operations, d.cigar_operations, total, hipMemcpyDeviceToHost, stream));
    if (cigar_runlengths && total) GW_CU_CHECK_ERR(hipMemcpyAsync(cigar_runlengths, d.cigar_runlengths, total * 4, hipMemcpyDeviceToHost, stream));
    if (cigar_offsets) GW_CU_CHECK_ERR(hipMemcpyAsync(cigar_offsets, d.cigar_offsets, (n + 1) * 4, hipMemcpyDeviceToHost, stream));
    if (metadata && n) GW_CU_CHECK_ERR(hipMemcpyAsync(metadata, d.metadata, n * 4, hipMemcpyDeviceToHost, stream));
    GW_CU_CHECK_ERR(hipStreamSynchronize(stream));
    return 0;
    GW_CATCH(-1)
}

int gw_aligner_relaunch(gw_aligner* a)
{
    GW_TRY
    if (!a->impl) return -1;
    a->impl->relaunch_resident();
    return 0;
    GW_CATCH(-1)
}

int gw_aligner_relaunch_timed(gw_aligner* a, float* kernels_ms)
{
    GW_TRY
    if (a->impl)
        *kernels_ms = a->impl->relaunch_resident_timed();
    else if (auto* global = dynamic_cast<aln::AlignerGlobal*>(a->aligner.get()))
        *kernels_ms = global->relaunch_resident_timed();
    else
    {
        gwhost::set_last_error("gw_aligner_relaunch_timed: this aligner class keeps no resident batch");
        return -1;
    }
    if (*kernels_ms < 0.f)
    {
        gwhost::set_last_error("gw_aligner_relaunch_timed: no batch resident (call align_all() first)");
        return -1;
    }
    return 0;
    GW_CATCH(-1)
}

int gw_aligner_band_cells(gw_aligner* a, uint64_t* cells)
{
    GW_TRY
    if (!a->impl) return -1;
    *cells = a->impl->total_band_cells();
    return 0;
    GW_CATCH(-1)
}


// ---- cudapoa/multi_device.hpp ---------------------------------------------------------------------------------

struct gw_poa_multi
{
    poa::MultiDeviceOutput out;
};

gw_poa_multi* gw_poa_multi_device_run(int32_t n_windows, const int32_t* reads_per_window, const char* const* seqs,
                                      const int32_t* lengths, const gw_poa_batch_config* cfg, const int32_t* devices,
                                      int32_t n_devices, int32_t batches_per_device, int64_t memory_per_device, int8_t output_mask,
                                      int16_t gap_score, int16_t mismatch_score, int16_t match_score)
{
    GW_TRY
    std::vector<std::vector<std::string>> windows(static_cast<size_t>(n_windows));
    size_t at = 0;
    for (int32_t w = 0; w < n_windows; ++w)
        for (int32_t r = 0; r < reads_per_window[w]; ++r, ++at) windows[static_cast<size_t>(w)].emplace_back(seqs[at], static_cast<size_t>(lengths[at]));
    const poa::BatchConfig c(cfg->max_sequence_size, cfg->max_consensus_size, cfg->max_nodes_per_graph, cfg->alignment_band_width,
                             cfg->max_sequences_per_poa, cfg->matrix_sequence_dimension, static_cast<poa::BandMode>(cfg->band_mode),
                             cfg->max_banded_pred_distance);
    poa::MultiDeviceConfig mc;
    mc.devices.assign(devices, devices + n_devices);
    mc.batches_per_device = batches_per_device;
    mc.memory_per_device  = memory_per_device;
    mc.output_mask        = output_mask;
    mc.gap_score          = gap_score;
    mc.mismatch_score     = mismatch_score;
    mc.match_score        = match_score;
    auto h = std::make_unique<gw_poa_multi>();
    poa::process_windows_multi_device(h->out, windows, c, mc);
    return h.release();
    GW_CATCH(nullptr)
}

void gw_poa_multi_destroy(gw_poa_multi* h) { delete h; }
int32_t gw_poa_multi_launches(gw_poa_multi* h) { return h->out.launches; }
double gw_poa_multi_seconds(gw_poa_multi* h) { return h->out.seconds; }
double gw_poa_multi_seconds_after_creation(gw_poa_multi* h) { return h->out.seconds_after_creation; }
// Accessors by window index: an index out of range, or asking for an output the run did not produce (consensus of an
// MSA-only run and vice versa), is an error return with gw_last_error() set -- never an exception across the C ABI.
int32_t gw_poa_multi_status(gw_poa_multi* h, int32_t w)
{
    GW_TRY
    return static_cast<int32_t>(h->out.status.at(static_cast<size_t>(w)));
    GW_CATCH(-1)
}
int32_t gw_poa_multi_worker(gw_poa_multi* h, int32_t w)
{
    GW_TRY
    return h->out.worker_of_window.at(static_cast<size_t>(w));
    GW_CATCH(-1)
}
const char* gw_poa_multi_consensus(gw_poa_multi* h, int32_t w, int32_t* length)
{
    GW_TRY
    const std::string& s = h->out.consensus.at(static_cast<size_t>(w));
    if (length) *length = static_cast<int32_t>(s.size());
    return s.c_str();
    GW_CATCH(nullptr)
}
const uint16_t* gw_poa_multi_coverage(gw_poa_multi* h, int32_t w, int32_t* length)
{
    GW_TRY
    const auto& v = h->out.coverage.at(static_cast<size_t>(w));
    if (length) *length = static_cast<int32_t>(v.size());
    return v.data();
    GW_CATCH(nullptr)
}
int32_t gw_poa_multi_msa_rows(gw_poa_multi* h, int32_t w)
{
    GW_TRY
    return static_cast<int32_t>(h->out.msa.at(static_cast<size_t>(w)).size());
    GW_CATCH(-1)
}
const char* gw_poa_multi_msa_row(gw_poa_multi* h, int32_t w, int32_t row, int32_t* length)
{
    GW_TRY
    const std::string& s = h->out.msa.at(static_cast<size_t>(w)).at(static_cast<size_t>(row));
    if (length) *length = static_cast<int32_t>(s.size());
    return s.c_str();
    GW_CATCH(nullptr)
}

struct gw_poa_size_plan
{
    poa::SizeClassPlan plan;
};

gw_poa_size_plan* gw_poa_plan_size_classes(int32_t n_windows, const int32_t* longest, const int32_t* reads, int32_t msa_flag,
                                           int32_t band_width, int32_t band_mode, float adaptive_storage_factor, float graph_length_factor,
                                           int32_t max_pred_distance, int32_t mismatch_score, int32_t gap_score, int32_t match_score)
{
    GW_TRY
    auto h = std::make_unique<gw_poa_size_plan>();
    poa::plan_size_classes(h->plan, std::vector<int32_t>(longest, longest + n_windows), std::vector<int32_t>(reads, reads + n_windows),
                           msa_flag != 0, band_width, static_cast<poa::BandMode>(band_mode), adaptive_storage_factor, graph_length_factor,
                           max_pred_distance, mismatch_score, gap_score, match_score);
    return h.release();
    GW_CATCH(nullptr)
}
void gw_poa_size_plan_destroy(gw_poa_size_plan* p) { delete p; }
int32_t gw_poa_size_plan_classes(gw_poa_size_plan* p) { return static_cast<int32_t>(p->plan.configs.size()); }
int64_t gw_poa_size_plan_total_bytes(gw_poa_size_plan* p) { return p->plan.total_bytes; }
int gw_poa_size_plan_class(gw_poa_size_plan* p, int32_t k, gw_poa_batch_config* cfg, int64_t* bytes_per_window, int32_t* n_windows)
{
    GW_TRY
    const poa::BatchConfig& c = p->plan.configs.at(static_cast<size_t>(k));
    if (cfg)
    {
        cfg->max_sequence_size         = c.max_sequence_size;
        cfg->max_consensus_size        = c.max_consensus_size;
        cfg->max_nodes_per_graph       = c.max_nodes_per_graph;
        cfg->matrix_sequence_dimension = c.matrix_sequence_dimension;
        cfg->alignment_band_width      = c.alignment_band_width;
        cfg->max_sequences_per_poa     = c.max_sequences_per_poa;
        cfg->band_mode                 = static_cast<int32_t>(c.band_mode);
        cfg->max_banded_pred_distance  = c.max_banded_pred_distance;
    }
    if (bytes_per_window) *bytes_per_window = p->plan.bytes_per_window.at(static_cast<size_t>(k));
    if (n_windows) *n_windows = static_cast<int32_t>(p->plan.groups.at(static_cast<size_t>(k)).size());
    return 0;
    GW_CATCH(-1)
}
int gw_poa_size_plan_windows(gw_poa_size_plan* p, int32_t k, int32_t* window_ids)
{
    GW_TRY
    const auto& g = p->plan.groups.at(static_cast<size_t>(k));
    std::copy(g.begin(), g.end(), window_ids);
    return 0;
    GW_CATCH(-1)
}

int gw_poa_size_plan_keep(gw_poa_size_plan* p, const uint8_t* keep, int32_t n_windows)
{
    GW_TRY
    p->plan.total_bytes = 0;
    for (size_t k = 0; k < p->plan.groups.size(); ++k)
    {
        std::vector<int32_t> kept;
        for (int32_t w : p->plan.groups[k])
            if (w < n_windows && keep[w]) kept.push_back(w);
        p->plan.groups[k] = std::move(kept);
        p->plan.total_bytes += static_cast<int64_t>(p->plan.groups[k].size()) * p->plan.bytes_per_window[k];
    }
    return 0;
    GW_CATCH(-1)
}

int gw_poa_size_plan_admission_gates(gw_poa_size_plan* p, int32_t compute_units, int32_t* gates)
{
    GW_TRY
    const std::vector<int32_t> g = poa::size_class_admission_gates(p->plan, compute_units);
    for (size_t k = 0; k < g.size(); ++k) gates[k] = g[k];
    return 0;
    GW_CATCH(-1)
}

gw_poa_multi* gw_poa_size_classes_run(int32_t n_windows, const int32_t* reads_per_window, const char* const* seqs, const int32_t* lengths,
                                      gw_poa_size_plan* plan, int32_t device, int64_t memory_budget, int8_t output_mask, int16_t gap_score,
                                      int16_t mismatch_score, int16_t match_score, double* compute_seconds)
{
    GW_TRY
    std::vector<std::vector<std::string>> windows(static_cast<size_t>(n_windows));
    size_t at = 0;
    for (int32_t w = 0; w < n_windows; ++w)
        for (int32_t r = 0; r < reads_per_window[w]; ++r, ++at) windows[static_cast<size_t>(w)].emplace_back(seqs[at], static_cast<size_t>(lengths[at]));
    auto h = std::make_unique<gw_poa_multi>();
    poa::process_windows_size_classes(h->out, windows, plan->plan, device, memory_budget, output_mask, gap_score, mismatch_score, match_score,
                                      compute_seconds);
    return h.release();
    GW_CATCH(nullptr)
}

// ---- cudapoa/utils.hpp: batch-shape planning and window-file readers -----------------------------------------

struct gw_windows
{
    std::vector<std::vector<std::string>> windows;
};

static int emit_plan(const std::vector<poa::BatchConfig>& shapes, const std::vector<std::vector<int32_t>>& groups,
                     int32_t n_groups, int32_t* n_batches, gw_poa_batch_config* batch_cfgs, int32_t* groups_per_batch,
                     int32_t* group_ids)
{
    *n_batches = static_cast<int32_t>(shapes.size());
    int32_t pos = 0;
    for (size_t b = 0; b < shapes.size(); b++)
    {
        const poa::BatchConfig& c = shapes[b];
        batch_cfgs[b] = gw_poa_batch_config{c.max_sequence_size, c.max_consensus_size, c.max_nodes_per_graph,
                                            c.matrix_sequence_dimension, c.alignment_band_width, c.max_sequences_per_poa,
                                            static_cast<int32_t>(c.band_mode), c.max_banded_pred_distance};
        groups_per_batch[b] = static_cast<int32_t>(groups[b].size());
        for (int32_t id : groups[b])
        {
            if (pos >= n_groups) return -1;
            group_ids[pos++] = id;
        }
    }
    return 0;
}

int gw_poa_bin_groups(int32_t n_groups, const int32_t* capacity, const int32_t* longest, const int32_t* reads,
                      int32_t band_width, int32_t band_mode, float adaptive_storage_factor, float graph_length_factor,
                      int32_t max_pred_distance, const int32_t* bins_capacity, int32_t n_bins, int32_t* n_batches,
                      gw_poa_batch_config* batch_cfgs, int32_t* groups_per_batch, int32_t* group_ids)
{
    GW_TRY
    std::vector<poa::BatchConfig> shapes;
    std::vector<std::vector<int32_t>> groups;
    std::vector<int32_t> bins(bins_capacity ? bins_capacity : nullptr, bins_capacity ? bins_capacity + n_bins : nullptr);
    poa::bin_poa_groups(shapes, groups, std::vector<int32_t>(capacity, capacity + n_groups),
                            std::vector<int32_t>(longest, longest + n_groups), std::vector<int32_t>(reads, reads + n_groups),
                            band_width, static_cast<poa::BandMode>(band_mode), adaptive_storage_factor,
                            graph_length_factor, max_pred_distance, bins_capacity ? &bins : nullptr);
    return emit_plan(shapes, groups, n_groups, n_batches, batch_cfgs, groups_per_batch, group_ids);
    GW_CATCH(-1)
}

int gw_poa_get_multi_batch_sizes(int32_t n_groups, const int32_t* longest, const int32_t* reads, int32_t msa_flag,
                                 int32_t band_width, int32_t band_mode, float adaptive_storage_factor,
                                 float graph_length_factor, int32_t max_pred_distance, float gpu_memory_usage_quota,
                                 int32_t mismatch_score, int32_t gap_score, int32_t match_score, int32_t* n_batches,
                                 gw_poa_batch_config* batch_cfgs, int32_t* groups_per_batch, int32_t* group_ids)
{
    GW_TRY
    // groups are described by (longest read, number of reads): only the lengths matter to the planner
    std::vector<poa::Group> poa_groups(static_cast<size_t>(n_groups));
    for (int32_t i = 0; i < n_groups; i++)
        poa_groups[static_cast<size_t>(i)].assign(static_cast<size_t>(reads[i]), poa::Entry{nullptr, nullptr, longest[i]});
    std::vector<poa::BatchConfig> shapes;
    std::vector<std::vector<int32_t>> groups;
    poa::get_multi_batch_sizes(shapes, groups, poa_groups, msa_flag != 0, band_width, static_cast<poa::BandMode>(band_mode),
                                   adaptive_storage_factor, graph_length_factor, max_pred_distance, nullptr,
                                   gpu_memory_usage_quota, mismatch_score, gap_score, match_score);
    return emit_plan(shapes, groups, n_groups, n_batches, batch_cfgs, groups_per_batch, group_ids);
    GW_CATCH(-1)
}

int32_t gw_poa_estimate_max_poas(const gw_poa_batch_config* cfg, int32_t msa_flag, float gpu_memory_usage_quota,
                                 int32_t mismatch_score, int32_t gap_score, int32_t match_score)
{
    GW_TRY
    const poa::BatchConfig c(cfg->max_sequence_size, cfg->max_consensus_size, cfg->max_nodes_per_graph,
                                 cfg->alignment_band_width, cfg->max_sequences_per_poa, cfg->matrix_sequence_dimension,
                                 static_cast<poa::BandMode>(cfg->band_mode), cfg->max_banded_pred_distance);
    return poa::estimate_max_poas(c, msa_flag != 0, gpu_memory_usage_quota, mismatch_score, gap_score, match_score);
    GW_CATCH(-1)
}

int64_t gw_poa_window_device_bytes(const gw_poa_batch_config* cfg, int32_t msa_flag, int32_t mismatch_score, int32_t gap_score,
                                   int32_t match_score)
{
    GW_TRY
    const poa::BatchConfig c(cfg->max_sequence_size, cfg->max_consensus_size, cfg->max_nodes_per_graph,
                             cfg->alignment_band_width, cfg->max_sequences_per_poa, cfg->matrix_sequence_dimension,
                             static_cast<poa::BandMode>(cfg->band_mode), cfg->max_banded_pred_distance);
    const gwhip_poa_config dc = poa::make_device_config(c, static_cast<int8_t>(msa_flag ? poa::OutputType::msa : poa::OutputType::consensus),
                                                        gap_score, mismatch_score, match_score);
    int64_t per_poa = 0, per_matrix = 0;
    gwhip_poa_bytes_per_window(&dc, &per_poa, &per_matrix);
    return per_poa + per_matrix;
    GW_CATCH(-1)
}

gw_windows* gw_windows_parse(const char* const* paths, int32_t n_paths, int32_t fasta, int32_t total_windows)
{
    GW_TRY
    std::unique_ptr<gw_windows> w(new gw_windows());
    if (fasta)
        poa::parse_fasta_files(w->windows, std::vector<std::string>(paths, paths + n_paths), total_windows);
    else
        poa::parse_cudapoa_file(w->windows, paths[0], total_windows);
    return w.release();
    GW_CATCH(nullptr)
}
void gw_windows_destroy(gw_windows* w) { delete w; }
int32_t gw_windows_count(const gw_windows* w) { return static_cast<int32_t>(w->windows.size()); }
int32_t gw_windows_num_sequences(const gw_windows* w, int32_t window)
{
    GW_TRY
    return static_cast<int32_t>(w->windows.at(static_cast<size_t>(window)).size());
    GW_CATCH(-1)
}
const char* gw_windows_sequence(const gw_windows* w, int32_t window, int32_t seq, int32_t* length)
{
    GW_TRY
    const std::string& s = w->windows.at(static_cast<size_t>(window)).at(static_cast<size_t>(seq));
    if (length) *length = static_cast<int32_t>(s.size());
    return s.data();
    GW_CATCH(nullptr)
}

} // extern "C"
