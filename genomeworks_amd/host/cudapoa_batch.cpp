// cudapoa_batch.cpp -- host side of cudapoa::Batch on MI355X.
//
// Behavioural contract (status codes, quirks, error conventions) follows the reference host code:
//   BatchConfig ctors        cudapoa/src/batch.cu:34-104
//   type selection           cudapoa/src/cudapoa_limits.hpp:34-59
//   create_batch             cudapoa/src/batch.cu:107-232
//   CudapoaBatch methods     cudapoa/src/cudapoa_batch.cuh:103-570
// The memory plan is ours: one device block from the allocator = [inputs | outputs | kernel workspace], with
// the workspace layout owned by libgwhip (gwhip_poa_workspace_bytes); one pinned host block for staging.
// All device work goes through the C-ABI in include/gwhip.h.
#include <claraparabricks/genomeworks/cudapoa/batch.hpp>
#include <claraparabricks/genomeworks/cudapoa/cudapoa.hpp>
#include <claraparabricks/genomeworks/logging/logging.hpp>

#include <algorithm>
#include <thread>
#include <atomic>
#include <climits>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <iomanip>
#include <stdexcept>
#include <tuple>

#include "../../include/gwhip.h"
#include "alignment_impl.hpp" // pinned_acquire / pinned_release (the process-wide cache of pinned buffers, runtime.cpp)
#include "host_common.hpp"
#include "poa_batch_impl.hpp"

namespace claraparabricks
{
namespace genomeworks
{
namespace cudapoa
{

namespace
{
constexpr int32_t kCellsPerThread = 4;   // CUDAPOA_CELLS_PER_THREAD
constexpr int32_t kMinBandWidth   = 128; // CUDAPOA_MIN_BAND_WIDTH (API constant, not the hardware wave size)
constexpr int32_t kRightPadding   = 8;   // CUDAPOA_BANDED_MATRIX_RIGHT_PADDING
constexpr uint8_t kKernelError    = 0xFF;
std::atomic<int32_t> g_batch_ids{0};     // one process-wide atomic counter (the reference has six racy ones)
} // namespace

StatusType Init()
{
    logging::initialize_logger(logging::LogLevel::warn);
    return StatusType::success;
}

void decode_error(StatusType error_type, std::string& error_message, std::string& error_hint)
{
    struct Text { const char* msg; const char* hint; };
    static const Text table[] = {
        /* success */ {nullptr, nullptr},
        {"Kernel Error: Number of groups per batch exceeded maximum POAs.",
         "Suggestion  : Evaluate maximum number of groups per batch using BatchBlock::estimate_max_poas()."},
        {"Kernel Error: Input read length or output consensus/MSA sequence length exceeded max sequence size.",
         "Suggestion  : Check BatchConfig::max_sequence_size and BatchConfig::max_consensus_size, increase if necessary."},
        {"Kernel Error: Exceeded maximum number of reads per POA.",
         "Suggestion  : Check BatchConfig::max_sequences_per_poa and increase if necessary."},
        {"Kernel Error: Node count exceeded maximum nodes per POA graph.",
         "Suggestion  : Check BatchConfig::max_nodes_per_graph and increase if necessary."},
        {"Kernel Error: Edge count exceeded maximum edges per graph.",
         "Suggestion  : Check default value of CUDAPOA_MAX_NODE_EDGES, note that increasing this macro would increase memory usage per POA."},
        {"Kernel Error: Allocated buffer for score/traceback matrix in adaptive banding is not large enough.",
         "Suggestion  : Check BatchConfig::matrix_sequence_dimension and increase if necessary."},
        {"Kernel Error: Set value for maximum predecessor distance in Needleman-Wunsch algorithm with traceback buffer is not large enough.",
         "Suggestion  : Check BatchConfig::max_banded_pred_distance and increase if necessary."},
        {"Kernel Error: Traceback in Needleman-Wunsch algorithm failed.", "Suggestion  : You may retry with a different banding mode."},
        {"Kernel Error: Output type not available.", "Suggestion  : Check MSA/Consensus selection for output type."},
        {"Error      : Input sequence has all zero base weights.", "Suggestion : Check base weights of POA group sequences on host."},
        {"Error      : No sequence was added to POA group.", "Suggestion : Check status vector in Batch::add_poa_group()."},
        {"Unknown error.", ""},
    };
    const int idx = static_cast<int>(error_type);
    if (idx <= 0 || idx > static_cast<int>(StatusType::generic_error)) throw std::runtime_error("Unknown error type detected.");
    error_message = table[idx].msg;
    error_hint    = table[idx].hint;
}

// ---- BatchConfig (batch.cu:34-104) ----------------------------------------------------------------
BatchConfig::BatchConfig(int32_t max_seq_sz, int32_t max_seq_per_poa, int32_t band_width, BandMode banding,
                         float adapive_storage_factor, float graph_length_factor, int32_t max_pred_dist)
    : max_sequence_size(max_seq_sz)
    , max_consensus_size(2 * max_seq_sz)
    , alignment_band_width(cudautils::align<int32_t, kMinBandWidth>(band_width))
    , max_sequences_per_poa(max_seq_per_poa)
    , band_mode(banding)
    , max_banded_pred_distance(max_pred_dist > 0 ? max_pred_dist : 2 * cudautils::align<int32_t, kMinBandWidth>(band_width))
{
    max_nodes_per_graph = cudautils::align<int32_t, kCellsPerThread>(static_cast<int32_t>(graph_length_factor * max_sequence_size));
    if (banding == BandMode::full_band)
        matrix_sequence_dimension = cudautils::align<int32_t, kCellsPerThread>(max_sequence_size);
    else if (banding == BandMode::static_band || banding == BandMode::static_band_traceback)
        matrix_sequence_dimension = cudautils::align<int32_t, kCellsPerThread>(alignment_band_width + kRightPadding);
    else
        matrix_sequence_dimension =
            cudautils::align<int32_t, kCellsPerThread>(static_cast<int32_t>(adapive_storage_factor * (alignment_band_width + kRightPadding)));

    throw_on_negative(max_seq_sz, "max_sequence_size cannot be negative.");
    throw_on_negative(max_seq_per_poa, "max_sequences_per_poa cannot be negative.");
    throw_on_negative(band_width, "alignment_band_width cannot be negative.");
    throw_on_negative(max_nodes_per_graph, "max_nodes_per_graph cannot be negative.");
    if (alignment_band_width != band_width)
        std::cerr << "Band-width should be multiple of 128. The input was changed from " << band_width << " to "
                  << alignment_band_width << std::endl;
}

BatchConfig::BatchConfig(int32_t max_seq_sz, int32_t max_consensus_sz, int32_t max_nodes_per_poa, int32_t band_width,
                         int32_t max_seq_per_poa, int32_t matrix_seq_dim, BandMode banding, int32_t max_pred_distance)
    : max_sequence_size(max_seq_sz)
    , max_consensus_size(max_consensus_sz)
    , max_nodes_per_graph(cudautils::align<int32_t, kCellsPerThread>(max_nodes_per_poa))
    , matrix_sequence_dimension(cudautils::align<int32_t, kCellsPerThread>(matrix_seq_dim))
    , alignment_band_width(cudautils::align<int32_t, kMinBandWidth>(band_width))
    , max_sequences_per_poa(max_seq_per_poa)
    , band_mode(banding)
    , max_banded_pred_distance(max_pred_distance)
{
    throw_on_negative(max_seq_sz, "max_sequence_size cannot be negative.");
    throw_on_negative(max_consensus_sz, "max_consensus_size cannot be negative.");
    throw_on_negative(max_nodes_per_poa, "max_nodes_per_graph cannot be negative.");
    throw_on_negative(max_seq_per_poa, "max_sequences_per_poa cannot be negative.");
    throw_on_negative(band_width, "alignment_band_width cannot be negative.");
    throw_on_negative(max_pred_distance, "max_banded_pred_distance cannot be negative.");
    if (max_nodes_per_graph < max_sequence_size)
        throw std::invalid_argument("max_nodes_per_graph should be greater than or equal to max_sequence_size.");
    if (max_consensus_size < max_sequence_size)
        throw std::invalid_argument("max_consensus_size should be greater than or equal to max_sequence_size.");
    if (max_sequence_size < alignment_band_width)
        throw std::invalid_argument("alignment_band_width should not be greater than max_sequence_size.");
    if (alignment_band_width != band_width)
        std::cerr << "Band-width should be multiple of 128. The input was changed from " << band_width << " to "
                  << alignment_band_width << std::endl;
}

// ---- type selection (cudapoa_limits.hpp:34-59) ----------------------------------------------------
gwhip_poa_config make_device_config(const BatchConfig& b, int8_t output_mask, int32_t gap, int32_t mismatch, int32_t match)
{
    gwhip_poa_config c{};
    c.max_sequence_size         = b.max_sequence_size;
    c.max_consensus_size        = b.max_consensus_size;
    c.max_nodes_per_graph       = b.max_nodes_per_graph;
    c.matrix_sequence_dimension = b.matrix_sequence_dimension;
    c.alignment_band_width      = b.alignment_band_width;
    c.max_sequences_per_poa     = b.max_sequences_per_poa;
    c.band_mode                 = static_cast<int32_t>(b.band_mode);
    c.max_banded_pred_distance  = b.max_banded_pred_distance;
    c.gap_score                 = gap;
    c.mismatch_score            = mismatch;
    c.match_score               = match;
    c.output_mask               = output_mask;
    const int32_t upper_bound   = b.max_sequence_size * match;
    const int32_t lower_bound   = b.max_sequence_size * std::max(gap, mismatch) + (b.max_nodes_per_graph - b.max_sequence_size) * gap;
    c.score32                   = (upper_bound > INT16_MAX || (-lower_bound) > (INT16_MAX + 1)) ? 1 : 0;
    int32_t max_length          = std::max(std::max(b.max_consensus_size, b.max_nodes_per_graph), b.matrix_sequence_dimension);
    // "if ScoreT is 16-bit, then it's safe to assume SizeT is 16-bit" (batch.cu:173): size32 only with score32
    c.size32                    = (c.score32 && max_length > INT16_MAX) ? 1 : 0;
    c.trace16                   = (b.max_banded_pred_distance > INT8_MAX) ? 1 : 0;
#ifdef SPOA_ACCURATE
    c.spoa_accurate = 1;
#endif
    // the reference selects the racon-style topological order at build time (-Dspoa_accurate=ON, cudapoa_kernels.cuh:
    // 516-531); the kernels here take it as a run-time flag, so one build serves both -- GW_SPOA_ACCURATE=1 turns it on
    if (const char* e = std::getenv("GW_SPOA_ACCURATE")) c.spoa_accurate = (e[0] == '1') ? 1 : c.spoa_accurate;
    return c;
}

// Batch objects alive per device: with more than one, kernels of different batches may share the device
// (gwhip_poa_args::shared_device).
namespace
{
std::atomic<int32_t>& live_batches(int32_t device)
{
    static std::atomic<int32_t> table[64];
    return table[static_cast<uint32_t>(device) & 63u];
}
} // namespace

// ---- PoaBatch ------------------------------------------------------------------------------------------
PoaBatch::PoaBatch(int32_t device_id, cudaStream_t stream, DefaultDeviceAllocator allocator, int64_t max_mem,
                   int8_t output_mask, const BatchConfig& batch_size, int32_t gap_score, int32_t mismatch_score,
                   int32_t match_score)
    : max_sequences_per_poa_(throw_on_negative(batch_size.max_sequences_per_poa, "Maximum sequences per POA has to be non-negative"))
    , device_id_(throw_on_negative(device_id, "Device ID has to be non-negative"))
    , stream_(stream)
    , output_mask_(output_mask)
    , batch_size_(batch_size)
    , allocator_(allocator)
{
    scoped_device_switch dev(device_id_);
    cfg_ = make_device_config(batch_size_, output_mask_, gap_score, mismatch_score, match_score);
    bid_ = g_batch_ids++;

    // capacity: max_poas = avail / (per_poa + per_matrix)  (allocate_block.hpp:64-89, with our byte counts)
    int64_t per_poa = 0, per_matrix = 0;
    gwhip_poa_bytes_per_window(&cfg_, &per_poa, &per_matrix);
    const int64_t avail_mem = std::min(get_size_of_largest_free_memory_block(allocator_), max_mem);
    const int64_t minimum   = per_poa + per_matrix;
    if (avail_mem < minimum)
    {
        std::string msg = std::string("Requires at least ").append(std::to_string(minimum)).append(
            " bytes of device memory per CUDAPOA batch to process correctly.");
        throw std::runtime_error(msg);
    }
    // ---- carve the device block: [sequences | weights | lengths | windows | consensus | coverage | msa | cells | workspace]
    int64_t guess = std::min<int64_t>(avail_mem / (per_poa + per_matrix), INT32_MAX);
    // window_details.seq_starts and the host write offset are int32 (as in the reference, cudapoa_batch.cuh:456-537): a
    // 288 GB device could otherwise hold more input bases than they can index
    const int64_t per_poa_input = std::max<int64_t>(1, static_cast<int64_t>(max_sequences_per_poa_) *
                                                           cudautils::align<int32_t, 4>(batch_size_.max_sequence_size));
    guess = std::min<int64_t>(guess, (static_cast<int64_t>(INT32_MAX) - 8192) / per_poa_input);
    size_t o[10];
    max_poas_ = static_cast<int32_t>(std::max<int64_t>(guess, 1));
    while (max_poas_ > 1 && static_cast<int64_t>(plan(max_poas_, o)) > avail_mem) max_poas_--;
    device_block_bytes_ = plan(max_poas_, o);
    if (static_cast<int64_t>(device_block_bytes_) > avail_mem)
    {
        std::string msg = std::string("Requires at least ").append(std::to_string(device_block_bytes_)).append(
            " bytes of device memory per CUDAPOA batch to process correctly.");
        throw std::runtime_error(msg);
    }
    score_buffer_bytes_ = static_cast<size_t>(per_matrix) * static_cast<size_t>(max_poas_);
    workspace_bytes_    = gwhip_poa_workspace_bytes(&cfg_, max_poas_, 0);
    const size_t o_seq = o[0], o_w = o[1], o_len = o[2], o_wd = o[3], o_cons = o[4], o_cov = o[5], o_msa = o[6], o_cells = o[7], o_ws = o[8];
    const size_t seq_bytes = o_w - o_seq, len_bytes = o_wd - o_len, wd_bytes = o_cons - o_wd, cons_bytes = o_cov - o_cons;
    const size_t cov_bytes = o_msa - o_cov, msa_bytes = o_cells - o_msa, cell_bytes = o_ws - o_cells;
    device_block_ = allocator_.allocate(device_block_bytes_, {stream_});
    d_sequences_  = reinterpret_cast<uint8_t*>(device_block_ + o_seq);
    d_weights_    = reinterpret_cast<int8_t*>(device_block_ + o_w);
    d_seq_lens_   = reinterpret_cast<int32_t*>(device_block_ + o_len);
    d_windows_    = reinterpret_cast<gwhip_window_details*>(device_block_ + o_wd);
    d_consensus_  = reinterpret_cast<uint8_t*>(device_block_ + o_cons);
    d_coverage_   = reinterpret_cast<uint16_t*>(device_block_ + o_cov);
    d_msa_        = msa_bytes ? reinterpret_cast<uint8_t*>(device_block_ + o_msa) : nullptr;
    d_cells_      = reinterpret_cast<uint64_t*>(device_block_ + o_cells);
    d_work_counters_ = reinterpret_cast<uint32_t*>(device_block_ + o_ws - 256);
    d_workspace_  = device_block_ + o_ws;
    input_capacity_ = seq_bytes - 4096;
    // the kernels read up to 2 KiB past a read (never consumed): keep that slack zero
    GW_CU_CHECK_ERR(hipMemsetAsync(d_sequences_, 0, seq_bytes, stream_));
    GW_CU_CHECK_ERR(hipMemsetAsync(d_weights_, 0, seq_bytes, stream_));
    GW_CU_CHECK_ERR(hipMemsetAsync(d_work_counters_, 0, 256, stream_)); // (a launch leaves them at zero again)

    // ---- pinned staging block ----
    host_block_bytes_ = seq_bytes * 2 + len_bytes + wd_bytes + cons_bytes + cov_bytes + msa_bytes + cell_bytes;
    // from the process-wide cache of pinned buffers (runtime.cpp): pinning and unpinning 1.2 GB took 200 and 100 ms of every
    // construction and destruction of a BatchConfig(1024, 200) batch with 32 GB of device memory
    host_block_ = cudaaligner::pinned_acquire(host_block_bytes_, &host_block_capacity_);
    size_t h          = 0;
    auto htake        = [&](size_t b) { size_t o = h; h += b; return o; };
    h_sequences_      = reinterpret_cast<uint8_t*>(host_block_ + htake(seq_bytes));
    h_weights_        = reinterpret_cast<int8_t*>(host_block_ + htake(seq_bytes));
    h_seq_lens_       = reinterpret_cast<int32_t*>(host_block_ + htake(len_bytes));
    h_windows_        = reinterpret_cast<gwhip_window_details*>(host_block_ + htake(wd_bytes));
    h_consensus_      = reinterpret_cast<uint8_t*>(host_block_ + htake(cons_bytes));
    h_coverage_       = reinterpret_cast<uint16_t*>(host_block_ + htake(cov_bytes));
    h_msa_            = msa_bytes ? reinterpret_cast<uint8_t*>(host_block_ + htake(msa_bytes)) : nullptr;
    h_cells_          = reinterpret_cast<uint64_t*>(host_block_ + htake(cell_bytes));
    // (the staging arrays are not cleared: an upload covers exactly the bytes add_seq_to_poa() wrote, padding included, and
    // the read-ahead slack is zeroed on the device -- clearing 2 x 600 MB of pinned memory was most of the construction time of
    // a BatchConfig(1024, 200) batch with tens of GB of device memory)

    live_batches(device_id_).fetch_add(1);
    debug_message(" Initializing batch on device ");
    reset();
}

PoaBatch::~PoaBatch()
{
    debug_message(" Destroyed buffers on device ");
    scoped_device_switch dev(device_id_);
    (void)hipStreamSynchronize(stream_);
    live_batches(device_id_).fetch_sub(1);
    if (host_block_ != nullptr) cudaaligner::pinned_release(host_block_, host_block_capacity_);
    if (device_block_ != nullptr) allocator_.deallocate(device_block_, device_block_bytes_);
}

void PoaBatch::debug_message(const std::string& message)
{
    std::string msg = std::string(static_cast<size_t>(bid_), '\t') + " " + std::to_string(bid_) + " " + message + " " + std::to_string(device_id_);
    GW_LOG_DEBUG(msg.c_str());
}

// Byte plan of the device block for n windows; fills the 9 section offsets, returns the total.
size_t PoaBatch::plan(int32_t n_poas, size_t* o) const
{
    auto up          = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t n   = static_cast<size_t>(n_poas);
    // every read is padded to a multiple of 4 bytes (add_seq_to_poa)
    const size_t seq = up(n * max_sequences_per_poa_ * static_cast<size_t>(cudautils::align<int32_t, 4>(batch_size_.max_sequence_size)) + 4096);
    size_t off       = 0;
    auto take        = [&](size_t b) { size_t at = off; off += b; return at; };
    o[0] = take(seq);
    o[1] = take(seq);
    o[2] = take(up(n * max_sequences_per_poa_ * sizeof(int32_t)));
    o[3] = take(up(n * sizeof(gwhip_window_details)));
    o[4] = take(up(n * batch_size_.max_consensus_size));
    o[5] = take(up(n * batch_size_.max_consensus_size * sizeof(uint16_t)));
    o[6] = take((output_mask_ & OutputType::msa) ? up(n * max_sequences_per_poa_ * batch_size_.max_consensus_size) : 0);
    o[7] = take(up(n * sizeof(uint64_t)) + 256); // + the two work counters of a persistent launch (gwhip_poa_args::work_counters)
    o[8] = take(up(gwhip_poa_workspace_bytes(&cfg_, n_poas, 0)));
    return off;
}

void PoaBatch::reset()
{
    poa_count_              = 0;
    num_nucleotides_copied_ = 0;
    global_sequence_idx_    = 0;
    next_scores_offset_     = 0;
    avail_buf_mem_          = score_buffer_bytes_;
    unit_weights_only_      = true;
}

// reserve_buf, cudapoa_batch.cuh:545-570
bool PoaBatch::reserve_buf(int32_t max_seq_length)
{
    const int32_t matrix_height = batch_size_.max_nodes_per_graph;
    const int32_t matrix_width  = (batch_size_.band_mode != BandMode::full_band)
                                      ? batch_size_.matrix_sequence_dimension
                                      : cudautils::align<int32_t, 4>(max_seq_length + 1 + kCellsPerThread);
    size_t required_size        = static_cast<size_t>(matrix_width) * static_cast<size_t>(matrix_height);
    const bool tb               = batch_size_.band_mode == static_band_traceback || batch_size_.band_mode == adaptive_band_traceback;
    // (full band with int16 scores: + one move byte per cell, as gwhip_poa_bytes_per_window counts it)
    required_size *= tb ? (cfg_.trace16 ? 2 : 1) : (cfg_.score32 ? 4 : (batch_size_.band_mode == BandMode::full_band ? 3 : 2));
    if (required_size > avail_buf_mem_)
    {
        if (get_total_poas() == 0)
        {
            std::cout << "Memory available " << std::fixed << std::setprecision(2)
                      << (static_cast<double>(avail_buf_mem_)) / 1024. / 1024. / 1024.;
            std::cout << "GB, Memory required " << (static_cast<double>(required_size)) / 1024. / 1024. / 1024.;
            std::cout << "GB (sequence length " << max_seq_length << ", graph length " << matrix_height << ")" << std::endl;
        }
        return false;
    }
    avail_buf_mem_ -= required_size;
    return true;
}

// add_poa, cudapoa_batch.cuh:456-472 (poa_count_ is bumped before any sequence is accepted: reference quirk kept)
StatusType PoaBatch::add_poa()
{
    if (poa_count_ == max_poas_) return StatusType::exceeded_maximum_poas;
    gwhip_window_details wd{};
    wd.seq_len_buffer_offset = global_sequence_idx_;
    wd.seq_starts            = num_nucleotides_copied_;
    wd.scores_width          = 0;
    wd.scores_offset         = next_scores_offset_;
    h_windows_[poa_count_]   = wd;
    poa_count_++;
    return StatusType::success;
}

// add_seq_to_poa, cudapoa_batch.cuh:475-542
StatusType PoaBatch::add_seq_to_poa(const char* seq, const int8_t* weights, int32_t seq_len)
{
    if (seq_len > batch_size_.max_sequence_size) return StatusType::exceeded_maximum_sequence_size;
    if (weights != nullptr)
    {
        bool all_zero = true;
        for (int32_t i = 0; i < seq_len; i++)
        {
            throw_on_negative(weights[i], "Base weights need to be non-negative");
            if (weights[i] > 0) all_zero = false;
        }
        if (all_zero) return StatusType::zero_weighted_poa_sequence;
    }
    gwhip_window_details* wd = &h_windows_[poa_count_ - 1];
    const int32_t scores_width = cudautils::align<int32_t, 4>(seq_len + 1 + kCellsPerThread);
    if (scores_width > wd->scores_width)
    {
        next_scores_offset_ += static_cast<size_t>(scores_width - wd->scores_width);
        wd->scores_width = scores_width;
    }
    if (static_cast<int32_t>(wd->num_seqs) >= max_sequences_per_poa_) return StatusType::exceeded_maximum_sequences_per_poa;
    wd->num_seqs++;
    std::memcpy(&h_sequences_[num_nucleotides_copied_], seq, static_cast<size_t>(seq_len));
    if (weights == nullptr)
        std::memset(&h_weights_[num_nucleotides_copied_], 1, static_cast<size_t>(seq_len));
    else
    {
        std::memcpy(&h_weights_[num_nucleotides_copied_], weights, static_cast<size_t>(seq_len));
        unit_weights_only_ = false;
    }
    // padding to the 4-byte boundary: the reference leaves stale bytes there; we zero them (never consumed)
    const int32_t padded = cudautils::align<int32_t, 4>(seq_len);
    for (int32_t i = seq_len; i < padded; i++)
    {
        h_sequences_[num_nucleotides_copied_ + i] = 0;
        h_weights_[num_nucleotides_copied_ + i]   = 0;
    }
    h_seq_lens_[global_sequence_idx_] = seq_len;
    num_nucleotides_copied_ += padded;
    global_sequence_idx_++;
    return StatusType::success;
}

StatusType PoaBatch::add_poa_group(std::vector<StatusType>& per_seq_status, const Group& poa_group)
{
    // The reference dereferences max_element of an empty group (UB, cudapoa_batch.cuh:108-113); we report it.
    if (poa_group.empty())
    {
        per_seq_status.clear();
        return StatusType::empty_poa_group;
    }
    const auto longest = std::max_element(poa_group.begin(), poa_group.end(),
                                          [](const Entry& a, const Entry& b) { return a.length < b.length; });
    // input bytes this group can take (reads that will be rejected are counted too: an upper bound)
    {
        int64_t bytes = 0;
        int32_t taken = 0;
        for (const auto& entry : poa_group)
            if (entry.length <= batch_size_.max_sequence_size && taken < max_sequences_per_poa_)
            {
                bytes += cudautils::align<int32_t, 4>(entry.length);
                taken++;
            }
        if (static_cast<int64_t>(num_nucleotides_copied_) + bytes > static_cast<int64_t>(input_capacity_)) return StatusType::exceeded_maximum_poas;
    }
    if (!reserve_buf(longest->length)) return StatusType::exceeded_maximum_poas;
    per_seq_status.clear();
    StatusType status = add_poa();
    if (status != StatusType::success) return status;
    bool poa_empty = true;
    for (const auto& entry : poa_group)
    {
        StatusType entry_status = add_seq_to_poa(entry.seq, entry.weights, entry.length);
        if (entry_status == StatusType::success) poa_empty = false;
        per_seq_status.push_back(entry_status);
    }
    return poa_empty ? StatusType::empty_poa_group : StatusType::success;
}

gwhip_poa_args PoaBatch::kernel_args() const
{
    gwhip_poa_args a{};
    a.cfg              = cfg_;
    a.total_windows    = poa_count_;
    a.sequences        = d_sequences_;
    a.base_weights     = d_weights_;
    a.sequence_lengths = d_seq_lens_;
    a.window_details   = d_windows_;
    a.consensus        = d_consensus_;
    a.coverage         = d_coverage_;
    a.msa              = d_msa_;
    a.workspace        = d_workspace_;
    a.workspace_bytes  = workspace_bytes_;
    a.cells            = d_cells_;
    a.work_counters    = d_work_counters_;
    a.shared_device    = live_batches(device_id_).load(std::memory_order_relaxed) > 1 ? 1 : 0;
    return a;
}

void PoaBatch::upload_inputs()
{
    GW_CU_CHECK_ERR(hipMemcpyAsync(d_sequences_, h_sequences_, static_cast<size_t>(num_nucleotides_copied_), hipMemcpyHostToDevice, stream_));
    // a batch whose reads all came without base weights (the common case, and the benchmark's) needs no weight upload:
    // every consumed byte is 1 (padding bytes are never read), so the device array is filled in place
    if (unit_weights_only_)
        GW_CU_CHECK_ERR(hipMemsetAsync(d_weights_, 1, static_cast<size_t>(num_nucleotides_copied_), stream_));
    else
        GW_CU_CHECK_ERR(hipMemcpyAsync(d_weights_, h_weights_, static_cast<size_t>(num_nucleotides_copied_), hipMemcpyHostToDevice, stream_));
    // zero the 2 KiB read-ahead slack behind the last read (it may hold an older batch's bases)
    GW_CU_CHECK_ERR(hipMemsetAsync(d_sequences_ + num_nucleotides_copied_, 0, 2048, stream_));
    GW_CU_CHECK_ERR(hipMemcpyAsync(d_windows_, h_windows_, static_cast<size_t>(poa_count_) * sizeof(gwhip_window_details), hipMemcpyHostToDevice, stream_));
    GW_CU_CHECK_ERR(hipMemcpyAsync(d_seq_lens_, h_seq_lens_, static_cast<size_t>(global_sequence_idx_) * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
}

void PoaBatch::launch(void* event_after_graph_build, uint64_t* phase_cycles)
{
    gwhip_poa_args a          = kernel_args();
    a.event_after_graph_build = event_after_graph_build;
    a.phase_cycles            = phase_cycles;
    // (A launch gate that kept concurrent batches from overlapping on the device beyond one wavefront per SIMD was built and
    // measured in round 6 -- it loses: 2048 windows over four batches 154 ms gated, 108-126 ms with the launches left to the
    // hardware; what did matter is the size of a fill, host/multi_device.cpp.)
    const int rc     = gwhip_poa_generate(&a, stream_);
    if (rc != 0)
    {
        char buf[512];
        gwhip_last_error_string(buf, sizeof(buf));
        GW_LOG_ERROR(buf);
        GW_CU_CHECK_ERR(static_cast<hipError_t>(rc));
    }
}

void PoaBatch::generate_poa()
{
    scoped_device_switch dev(device_id_);
    if (poa_count_ == 0)
    {
        debug_message(" No POA was added to compute! ");
        return;
    }
    upload_inputs();
    debug_message(" Launching kernel for " + std::to_string(poa_count_) + " on device ");
    launch();
    debug_message(" Launched kernel on device ");
}

void PoaBatch::relaunch_resident()
{
    scoped_device_switch dev(device_id_);
    if (poa_count_ == 0) return;
    // sequence_lengths[first read of each window] was overwritten with the node count: restore lengths only
    GW_CU_CHECK_ERR(hipMemcpyAsync(d_seq_lens_, h_seq_lens_, static_cast<size_t>(global_sequence_idx_) * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    launch();
}

void PoaBatch::relaunch_resident_timed(float* graph_build_ms, float* output_ms)
{
    scoped_device_switch dev(device_id_);
    if (poa_count_ == 0) return;
    hipEvent_t e0, e1, e2;
    GW_CU_CHECK_ERR(hipEventCreate(&e0));
    GW_CU_CHECK_ERR(hipEventCreate(&e1));
    GW_CU_CHECK_ERR(hipEventCreate(&e2));
    GW_CU_CHECK_ERR(hipMemcpyAsync(d_seq_lens_, h_seq_lens_, static_cast<size_t>(global_sequence_idx_) * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    GW_CU_CHECK_ERR(hipEventRecord(e0, stream_));
    launch(e1);
    GW_CU_CHECK_ERR(hipEventRecord(e2, stream_));
    GW_CU_CHECK_ERR(hipEventSynchronize(e2));
    GW_CU_CHECK_ERR(hipEventElapsedTime(graph_build_ms, e0, e1));
    GW_CU_CHECK_ERR(hipEventElapsedTime(output_ms, e1, e2));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(e2);
}

void PoaBatch::profile_phases(double out[6])
{
    for (int k = 0; k < 6; k++) out[k] = 0;
    if (poa_count_ == 0) return;
    std::vector<uint64_t> h;
    profile_phases_per_window(h);
    for (size_t i = 0; i < h.size(); i++) out[i % 6] += static_cast<double>(h[i]);
    for (int k = 0; k < 6; k++) out[k] /= poa_count_;
}

void PoaBatch::profile_phases_per_window(std::vector<uint64_t>& ticks)
{
    scoped_device_switch dev(device_id_);
    const size_t n = static_cast<size_t>(poa_count_) * 6;
    ticks.assign(n, 0);
    if (n == 0) return;
    uint64_t* d = nullptr;
    GW_CU_CHECK_ERR(hipMalloc(reinterpret_cast<void**>(&d), n * sizeof(uint64_t)));
    GW_CU_CHECK_ERR(hipMemcpyAsync(d_seq_lens_, h_seq_lens_, static_cast<size_t>(global_sequence_idx_) * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    launch(nullptr, d);
    GW_CU_CHECK_ERR(hipMemcpyAsync(ticks.data(), d, n * sizeof(uint64_t), hipMemcpyDeviceToHost, stream_));
    GW_CU_CHECK_ERR(hipStreamSynchronize(stream_));
    GW_CU_CHECK_ERR(hipFree(d));
}

void PoaBatch::log_kernel_error(StatusType error_type, std::vector<StatusType>& output_status)
{
    std::string message, hint;
    decode_error(error_type, message, hint);
    message += " in batch " + std::to_string(bid_) + "\n" + hint;
    GW_LOG_WARN(message.c_str());
    output_status.emplace_back(error_type);
}

namespace
{
// dst[0 .. n) = src[n-1 .. 0]: the kernels write consensus and coverage back to front, as the reference's do
// (cudapoa_generate_consensus.cuh:245-283). Eight bytes / four counters per step through a 64-bit word.
void reversed_copy_u8(char* dst, const char* src, size_t n)
{
    size_t i = 0;
    for (; i + 8 <= n; i += 8)
    {
        uint64_t v;
        std::memcpy(&v, src + n - i - 8, 8);
        v = __builtin_bswap64(v);
        std::memcpy(dst + i, &v, 8);
    }
    for (; i < n; i++) dst[i] = src[n - 1 - i];
}
void reversed_copy_u16(uint16_t* dst, const uint16_t* src, size_t n)
{
    size_t i = 0;
    for (; i + 4 <= n; i += 4)
    {
        uint64_t v;
        std::memcpy(&v, src + n - i - 4, 8);
        v = (v >> 48) | ((v >> 16) & 0xffff0000ull) | ((v << 16) & 0xffff00000000ull) | (v << 48);
        std::memcpy(dst + i, &v, 8);
    }
    for (; i < n; i++) dst[i] = src[n - 1 - i];
}
} // namespace

// D2H of consensus + coverage, then un-reversal into count = poa_count_ slots (strings / vectors whose storage is reused
// when they have any); statuses of failed windows are logged in window order.
void PoaBatch::fetch_consensus(std::string* consensus, std::vector<uint16_t>* coverage, StatusType* output_status, bool presize)
{
    scoped_device_switch dev(device_id_);
    // D2H of the windows actually in the batch (the reference copies the whole capacity: SURVEY Appendix C.4). The two
    // arrays are neighbours on both sides with the same spacing: one copy when the unused tail of the first is small.
    const size_t n        = static_cast<size_t>(poa_count_) * batch_size_.max_consensus_size;
    const size_t cons_cap = static_cast<size_t>(reinterpret_cast<uint8_t*>(d_coverage_) - d_consensus_);
    if (n > 0)
    {
        if (cons_cap - n <= n / 4 && reinterpret_cast<uint8_t*>(h_coverage_) - h_consensus_ == static_cast<std::ptrdiff_t>(cons_cap))
            GW_CU_CHECK_ERR(hipMemcpyAsync(h_consensus_, d_consensus_, cons_cap + n * sizeof(uint16_t), hipMemcpyDeviceToHost, stream_));
        else
        {
            GW_CU_CHECK_ERR(hipMemcpyAsync(h_consensus_, d_consensus_, n, hipMemcpyDeviceToHost, stream_));
            GW_CU_CHECK_ERR(hipMemcpyAsync(h_coverage_, d_coverage_, n * sizeof(uint16_t), hipMemcpyDeviceToHost, stream_));
        }
    }
    const size_t count = static_cast<size_t>(poa_count_);
    const size_t row   = static_cast<size_t>(batch_size_.max_consensus_size);
    if (presize)
    {
        // Fresh (empty) strings and vectors of the caller: their heap blocks are requested NOW, while the kernels and the
        // copies above are still in flight and this thread would only wait -- a consensus is about as long as the window's
        // longest read (an estimate: the resize below corrects it either way).
        for (size_t poa = 0; poa < count; poa++)
        {
            const gwhip_window_details& wd = h_windows_[poa];
            int32_t longest = 0;
            for (int32_t k = 0; k < static_cast<int32_t>(wd.num_seqs); k++) longest = std::max(longest, h_seq_lens_[wd.seq_len_buffer_offset + k]);
            const size_t hint = std::min(row, static_cast<size_t>(longest) + static_cast<size_t>(longest) / 16 + 16);
            consensus[poa].reserve(hint);
            coverage[poa].reserve(hint);
        }
    }
    GW_CU_CHECK_ERR(hipStreamSynchronize(stream_));
    // The staging block is pinned memory, which the host reads slowly (measured: 3 GB/s in 8-byte steps from the far end of a
    // row, 5 GB/s with memcpy): a row goes to a small cached buffer in one forward memcpy and is reversed from there.
    constexpr size_t kLocalRow = 4096;
    auto unpack = [&](size_t first, size_t last) {
        alignas(64) char local_c[kLocalRow];
        alignas(64) uint16_t local_v[kLocalRow];
        for (size_t poa = first; poa < last; poa++)
        {
            const char* c      = reinterpret_cast<const char*>(&h_consensus_[poa * row]);
            const uint16_t* cv = &h_coverage_[poa * row];
            if (static_cast<uint8_t>(c[0]) == kKernelError)
            {
                output_status[poa] = static_cast<StatusType>(c[1]); // logged below, in window order
                consensus[poa].clear();
                coverage[poa].clear();
                continue;
            }
            output_status[poa] = StatusType::success;
            size_t len;
            if (row <= kLocalRow)
            {
                std::memcpy(local_c, c, row);
                len = ::strnlen(local_c, row);
                std::memcpy(local_v, cv, len * sizeof(uint16_t));
                c  = local_c;
                cv = local_v;
            }
            else
                len = ::strnlen(c, row);
            consensus[poa].resize(len);
            reversed_copy_u8(&consensus[poa][0], c, len);
            coverage[poa].resize(len);
            reversed_copy_u16(coverage[poa].data(), cv, len);
        }
    };
    // the windows are independent: 16 at a time on the library's worker pool
    constexpr size_t kPerTask = 16;
    gwhost::parallel_tasks((count + kPerTask - 1) / kPerTask, count >= 256 ? 8 : 1,
                           [&](size_t t) { unpack(t * kPerTask, std::min(count, (t + 1) * kPerTask)); });
    for (size_t poa = 0; poa < count; poa++)
        if (output_status[poa] != StatusType::success)
        {
            std::vector<StatusType> sink; // log_kernel_error appends the code; the status slot is already filled
            log_kernel_error(output_status[poa], sink);
        }
}

StatusType PoaBatch::get_consensus(std::vector<std::string>& consensus, std::vector<std::vector<uint16_t>>& coverage,
                                   std::vector<StatusType>& output_status)
{
    if (!(OutputType::consensus & output_mask_)) return StatusType::output_type_unavailable;
    // the results are appended behind whatever the caller's vectors already hold, as in the reference (cudapoa_batch.cuh:186-249)
    const size_t base_c = consensus.size(), base_v = coverage.size(), base_s = output_status.size();
    const size_t count  = static_cast<size_t>(poa_count_);
    consensus.resize(base_c + count);
    coverage.resize(base_v + count);
    output_status.resize(base_s + count, StatusType::success);
    fetch_consensus(consensus.data() + base_c, coverage.data() + base_v, output_status.data() + base_s, true);
    return StatusType::success;
}

StatusType PoaBatch::get_consensus_in_place(std::vector<std::string>& consensus, std::vector<std::vector<uint16_t>>& coverage,
                                            std::vector<StatusType>& output_status)
{
    if (!(OutputType::consensus & output_mask_)) return StatusType::output_type_unavailable;
    const size_t count = static_cast<size_t>(poa_count_);
    consensus.resize(count); // strings and vectors of an earlier call keep their storage
    coverage.resize(count);
    output_status.resize(count);
    fetch_consensus(consensus.data(), coverage.data(), output_status.data(), false);
    return StatusType::success;
}

StatusType PoaBatch::get_msa(std::vector<std::vector<std::string>>& msa, std::vector<StatusType>& output_status)
{
    if (!(OutputType::msa & output_mask_)) return StatusType::output_type_unavailable;
    scoped_device_switch dev(device_id_);
    const size_t row = static_cast<size_t>(batch_size_.max_consensus_size);
    if (poa_count_ > 0)
    {
        GW_CU_CHECK_ERR(hipMemcpyAsync(h_msa_, d_msa_, static_cast<size_t>(poa_count_) * max_sequences_per_poa_ * row, hipMemcpyDeviceToHost, stream_));
        GW_CU_CHECK_ERR(hipMemcpyAsync(h_consensus_, d_consensus_, static_cast<size_t>(poa_count_) * row, hipMemcpyDeviceToHost, stream_));
    }
    GW_CU_CHECK_ERR(hipStreamSynchronize(stream_));
    // rows are unpacked behind whatever the caller's vectors already hold; long-read MSAs are hundreds of megabytes of
    // rows, so the windows are split over a few host threads (statuses are logged afterwards, in window order)
    const size_t base_m = msa.size(), base_s = output_status.size();
    const size_t count  = static_cast<size_t>(poa_count_);
    msa.resize(base_m + count);
    output_status.resize(base_s + count, StatusType::success);
    auto unpack = [&](size_t first, size_t last) {
        for (size_t poa = first; poa < last; poa++)
        {
            const char* c = reinterpret_cast<const char*>(&h_consensus_[poa * row]);
            if (static_cast<uint8_t>(c[0]) == kKernelError)
            {
                output_status[base_s + poa] = static_cast<StatusType>(c[1]);
                continue;
            }
            const uint16_t num_seqs = h_windows_[poa].num_seqs;
            std::vector<std::string>& rows = msa[base_m + poa];
            rows.reserve(num_seqs);
            for (int32_t i = 0; i < num_seqs; i++)
                rows.emplace_back(reinterpret_cast<const char*>(&h_msa_[(poa * max_sequences_per_poa_ + static_cast<size_t>(i)) * row]));
        }
    };
    // long-read MSAs are hundreds of megabytes of rows: the windows go over the library's worker pool in chunks (joined and
    // exception-safe: a bad_alloc in one chunk is rethrown here once every chunk has finished)
    const size_t bytes     = count * static_cast<size_t>(max_sequences_per_poa_) * row;
    const size_t n_threads = bytes >= (size_t(8) << 20) ? std::min<size_t>(8, std::max<size_t>(1, count)) : 1;
    const size_t per_task  = std::max<size_t>(1, (count + 4 * n_threads - 1) / (4 * n_threads));
    gwhost::parallel_tasks((count + per_task - 1) / per_task, n_threads,
                           [&](size_t t) { unpack(t * per_task, std::min(count, (t + 1) * per_task)); });
    for (size_t poa = 0; poa < count; poa++)
        if (output_status[base_s + poa] != StatusType::success)
        {
            std::vector<StatusType> sink; // log_kernel_error appends the code; the status slot is already filled
            log_kernel_error(output_status[base_s + poa], sink);
        }
    return StatusType::success;
}

void PoaBatch::get_graphs(std::vector<DirectedGraph>& graphs, std::vector<StatusType>& output_status)
{
    scoped_device_switch dev(device_id_);
    graphs.resize(poa_count_);
    if (poa_count_ == 0) return;
    const size_t mn = static_cast<size_t>(batch_size_.max_nodes_per_graph);
    const size_t W  = static_cast<size_t>(poa_count_);
    auto up         = [](size_t v) { return (v + 255) & ~size_t(255); };
    // Temporaries in the reference's array layout (303 bytes per node slot), filled by the export kernel. The batch's
    // pool is normally fully committed to the batch block, so they come straight from the runtime -- in chunks of
    // windows of at most ~1 GiB, so that a batch filled to device capacity can still be exported.
    const size_t per_window   = mn * (1 + 2 + GWHIP_MAX_NODE_EDGES * (4 + 2)) + 4 * 256;
    size_t chunk_budget       = size_t(1) << 30;
    if (const char* e = std::getenv("GW_GRAPH_EXPORT_CHUNK_BYTES")) chunk_budget = std::max<size_t>(1, std::strtoull(e, nullptr, 10)); // tests
    const size_t chunk        = std::max<size_t>(1, std::min(W, chunk_budget / per_window));
    const size_t b_nodes = chunk * mn, b_cnt = chunk * mn * 2, b_edges = chunk * mn * GWHIP_MAX_NODE_EDGES * 4, b_w = chunk * mn * GWHIP_MAX_NODE_EDGES * 2;
    const size_t total = up(b_nodes) + up(b_cnt) + up(b_edges) + up(b_w);
    char* d_tmp = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d_tmp), total) != hipSuccess)
    {
        (void)hipGetLastError();
        throw device_memory_allocation_exception();
    }
    uint8_t* d_nodes   = reinterpret_cast<uint8_t*>(d_tmp);
    uint16_t* d_cnt    = reinterpret_cast<uint16_t*>(d_tmp + up(b_nodes));
    int32_t* d_edges   = reinterpret_cast<int32_t*>(d_tmp + up(b_nodes) + up(b_cnt));
    uint16_t* d_w      = reinterpret_cast<uint16_t*>(d_tmp + up(b_nodes) + up(b_cnt) + up(b_edges));
    gwhip_poa_args a   = kernel_args();
    std::vector<uint8_t> nodes(b_nodes);
    std::vector<uint16_t> cnt(chunk * mn), w(chunk * mn * GWHIP_MAX_NODE_EDGES);
    std::vector<int32_t> edges(chunk * mn * GWHIP_MAX_NODE_EDGES);
    std::vector<int32_t> lens(static_cast<size_t>(global_sequence_idx_));
    GW_CU_CHECK_ERR(hipMemcpyAsync(lens.data(), d_seq_lens_, lens.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    GW_CU_CHECK_ERR(hipMemcpyAsync(h_consensus_, d_consensus_, W * batch_size_.max_consensus_size, hipMemcpyDeviceToHost, stream_));
    for (size_t first = 0; first < W; first += chunk)
    {
        const size_t n_here = std::min(chunk, W - first);
        const int rc = gwhip_poa_export_graphs_range(&a, static_cast<int32_t>(first), static_cast<int32_t>(n_here), d_nodes, d_edges,
                                                     d_w, d_cnt, nullptr, nullptr, stream_);
        if (rc != 0)
        {
            (void)hipFree(d_tmp);
            GW_CU_CHECK_ERR(static_cast<hipError_t>(rc));
        }
        GW_CU_CHECK_ERR(hipMemcpyAsync(nodes.data(), d_nodes, n_here * mn, hipMemcpyDeviceToHost, stream_));
        GW_CU_CHECK_ERR(hipMemcpyAsync(cnt.data(), d_cnt, n_here * mn * 2, hipMemcpyDeviceToHost, stream_));
        GW_CU_CHECK_ERR(hipMemcpyAsync(edges.data(), d_edges, n_here * mn * GWHIP_MAX_NODE_EDGES * 4, hipMemcpyDeviceToHost, stream_));
        GW_CU_CHECK_ERR(hipMemcpyAsync(w.data(), d_w, n_here * mn * GWHIP_MAX_NODE_EDGES * 2, hipMemcpyDeviceToHost, stream_));
        GW_CU_CHECK_ERR(hipStreamSynchronize(stream_));
        for (size_t k = 0; k < n_here; k++)
        {
            const size_t poa = first + k;
            const char* c    = reinterpret_cast<const char*>(&h_consensus_[poa * batch_size_.max_consensus_size]);
            if (static_cast<uint8_t>(c[0]) == kKernelError)
            {
                log_kernel_error(static_cast<StatusType>(c[1]), output_status);
                continue;
            }
            output_status.emplace_back(StatusType::success);
            DirectedGraph& graph    = graphs[poa];
            const int32_t num_nodes = lens[static_cast<size_t>(h_windows_[poa].seq_len_buffer_offset)];
            for (int32_t n = 0; n < num_nodes; n++)
            {
                graph.set_node_label(n, std::string(1, static_cast<char>(nodes[k * mn + n])));
                const uint16_t ne = cnt[k * mn + n];
                for (int32_t e = 0; e < ne; e++)
                {
                    const size_t idx = (k * mn + n) * GWHIP_MAX_NODE_EDGES + e;
                    graph.add_edge(edges[idx], n, w[idx]);
                }
            }
        }
    }
    GW_CU_CHECK_ERR(hipFree(d_tmp));
}

uint64_t PoaBatch::total_cells()
{
    scoped_device_switch dev(device_id_);
    if (poa_count_ == 0) return 0;
    GW_CU_CHECK_ERR(hipMemcpyAsync(h_cells_, d_cells_, static_cast<size_t>(poa_count_) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream_));
    GW_CU_CHECK_ERR(hipStreamSynchronize(stream_));
    uint64_t t = 0;
    for (int32_t i = 0; i < poa_count_; i++) t += h_cells_[i];
    return t;
}

// ---- factories (batch.cu:107-232) ----------------------------------------------------------------------
std::unique_ptr<Batch> create_batch(int32_t device_id, cudaStream_t stream, DefaultDeviceAllocator allocator,
                                    int64_t max_mem, int8_t output_mask, const BatchConfig& batch_size,
                                    int16_t gap_score, int16_t mismatch_score, int16_t match_score)
{
    return std::make_unique<PoaBatch>(device_id, stream, allocator, max_mem, output_mask, batch_size, gap_score,
                                      mismatch_score, match_score);
}

std::unique_ptr<Batch> create_batch(int32_t device_id, cudaStream_t stream, int64_t max_mem, int8_t output_mask,
                                    const BatchConfig& batch_size, int16_t gap_score, int16_t mismatch_score,
                                    int16_t match_score)
{
    if (max_mem < -1)
        throw std::invalid_argument("max_mem has to be either -1 (=all available GPU memory) or greater or equal than 0.");
    scoped_device_switch dev(device_id);
    if (max_mem == -1)
    {
        max_mem = cudautils::find_largest_contiguous_device_memory_section();
        if (max_mem == 0) throw std::runtime_error("No memory available for caching");
    }
    if (max_mem == 0)
    {
        // BatchBlock throws "Requires at least N bytes" for a zero budget (Test_CudapoaBatch.cu:70-97)
        gwhip_poa_config c = make_device_config(batch_size, output_mask, gap_score, mismatch_score, match_score);
        int64_t per_poa = 0, per_matrix = 0;
        gwhip_poa_bytes_per_window(&c, &per_poa, &per_matrix);
        throw std::runtime_error(std::string("Requires at least ").append(std::to_string(per_poa + per_matrix)).append(
            " bytes of device memory per CUDAPOA batch to process correctly."));
    }
    DefaultDeviceAllocator allocator(static_cast<size_t>(max_mem), stream);
    return create_batch(device_id, stream, allocator, max_mem, output_mask, batch_size, gap_score, mismatch_score, match_score);
}

} // namespace cudapoa
} // namespace genomeworks
} // namespace claraparabricks
