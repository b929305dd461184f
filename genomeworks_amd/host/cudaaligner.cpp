// cudaaligner.cpp -- host side of cudaaligner::Aligner on MI355X.
//
// Behavioural contract from the reference host code:
//   factories                         cudaaligner/src/aligner.cpp:31-126
//   AlignerGlobalMyersBanded          cudaaligner/src/aligner_global_myers_banded.cpp:127-543
//   AlignerGlobal (fixed stride)      cudaaligner/src/aligner_global.cpp:50-192
// Memory plan is ours: the kernels take one lane per pair, so the device workspace is per pair
// (gwhip_myers_banded_workspace_bytes); a batch is refused with exceeded_max_alignments when it would not fit
// max_device_memory, exactly where the reference's fits_device_memory() refuses.
#include <claraparabricks/genomeworks/cudaaligner/aligner.hpp>
#include <claraparabricks/genomeworks/cudaaligner/alignment.hpp>
#include <claraparabricks/genomeworks/logging/logging.hpp>
#include <claraparabricks/genomeworks/utils/genomeutils.hpp>
#include <claraparabricks/genomeworks/utils/signed_integer_utils.hpp>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <new>
#include <numeric>
#include <cstdlib>
#include <exception>
#include <stdexcept>
#include <thread>

#include "../../include/gwhip.h"
#include "aligner_impl.hpp"
#include "base_packing.hpp"
#include "chunk_order.hpp"
#include "host_common.hpp"
#include "aligner_global.hpp"
#include "alignment_impl.hpp"

namespace claraparabricks
{
namespace genomeworks
{
namespace cudaaligner
{

StatusType Init()
{
    logging::initialize_logger(logging::LogLevel::warn);
    return StatusType::success;
}

namespace
{
constexpr int32_t kWordSize = 32;
size_t up256(size_t v) { return (v + 255) & ~size_t(255); }
using gwhost::pack_bases;
using gwhost::query_code;
using gwhost::target_code;
// GW_ALIGNER_TRACE=1: host-side timeline of align_all() / sync_alignments() on stderr (debugging aid)
struct Tracer
{
    bool on = std::getenv("GW_ALIGNER_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char* what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[aligner] %-44s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
} // namespace

BandedAligner::BandedAligner(int64_t max_device_memory, int32_t max_bandwidth, DefaultDeviceAllocator allocator,
                             cudaStream_t stream, int32_t device_id, bool expand_results, int32_t max_query_length,
                             int32_t max_target_length, int32_t max_alignments)
    : stream_(stream)
    , device_id_(device_id)
    , allocator_(allocator)
    , max_bandwidth_(throw_on_negative(max_bandwidth, "max_bandwidth cannot be negative."))
    , max_device_memory_(max_device_memory < 0 ? get_size_of_largest_free_memory_block(allocator) : max_device_memory)
    , expand_results_(expand_results)
    , max_query_length_(max_query_length)
    , max_target_length_(max_target_length)
    , max_alignments_(max_alignments)
{
    seq_starts_h_.assign(1, 0);
    reset_max_bandwidth(max_bandwidth);
}

BandedAligner::~BandedAligner()
{
    scoped_device_switch dev(device_id_);
    (void)hipStreamSynchronize(stream_);
    if (upload_stream_ != nullptr)
    {
        (void)hipStreamSynchronize(static_cast<hipStream_t>(upload_stream_));
        (void)hipStreamDestroy(static_cast<hipStream_t>(upload_stream_));
    }
    if (side_stream_ != nullptr)
    {
        (void)hipStreamSynchronize(static_cast<hipStream_t>(side_stream_));
        (void)hipStreamDestroy(static_cast<hipStream_t>(side_stream_));
    }
    for (void* e : upload_events_) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
    free_device();
    if (head_ != nullptr) pinned_release(head_, head_cap_);
    if (mirror_ != nullptr) pinned_release(mirror_, mirror_cap_);
}

void BandedAligner::reset_max_bandwidth(int32_t max_bandwidth)
{
    throw_on_negative(max_bandwidth, "max_bandwidth cannot be negative.");
    if (max_bandwidth % kWordSize == 1)
        throw std::invalid_argument("Invalid max_bandwidth. max_bandwidth % 32 == 1 is not allowed. Please change it by +/-1.");
    reset();
    max_bandwidth_ = max_bandwidth;
}

void BandedAligner::free_device()
{
    if (device_block_ != nullptr)
    {
        allocator_.deallocate(device_block_, device_block_bytes_);
        device_block_       = nullptr;
        device_block_bytes_ = 0;
    }
    for (Chunk& c : chunks_)
        if (c.workspace != nullptr) allocator_.deallocate(c.workspace, c.block_bytes);
    chunks_.clear();
}

void BandedAligner::reset_data()
{
    seq_h_.clear();
    packed_h_.clear();
    seq_starts_h_.assign(1, 0);
    max_bandwidths_h_.clear();
    workspace_bytes_estimate_ = 0;
    largest_wave_ws_          = 0;
    longest_query_            = 0;
    longest_pair_             = 0;
    widest_band_              = 0;
    launched_                 = false;
}

void BandedAligner::reset()
{
    scoped_device_switch dev(device_id_);
    (void)hipStreamSynchronize(stream_);
    if (upload_stream_ != nullptr) (void)hipStreamSynchronize(static_cast<hipStream_t>(upload_stream_));
    if (side_stream_ != nullptr) (void)hipStreamSynchronize(static_cast<hipStream_t>(side_stream_));
    uploads_in_flight_ = false;
    reset_data();
    free_device();
    alignments_.clear();
}

void BandedAligner::free_temporary_device_buffers()
{
    // everything but the packed results could go; we keep one block per batch, so this is a no-op until reset()
}

void BandedAligner::drain_streams()
{
    // the aligner's stream has waited for the other two by the end of align_all(); after a call that threw in between it has not
    (void)hipStreamSynchronize(stream_);
    if (upload_stream_ != nullptr) (void)hipStreamSynchronize(static_cast<hipStream_t>(upload_stream_));
    if (side_stream_ != nullptr) (void)hipStreamSynchronize(static_cast<hipStream_t>(side_stream_));
}

StatusType BandedAligner::add_alignment(const char* query, int32_t query_length, const char* target, int32_t target_length,
                                        bool reverse_complement_query, bool reverse_complement_target)
{
    return add_alignment(max_bandwidth_, query, query_length, target, target_length, reverse_complement_query,
                         reverse_complement_target);
}

StatusType BandedAligner::add_alignment(int32_t max_bandwidth, const char* query, int32_t query_length, const char* target,
                                        int32_t target_length, bool reverse_complement_query, bool reverse_complement_target)
{
    GW_NVTX_RANGE(profiler, "BandedAligner::add_alignment");
    if (uploads_in_flight_)
    {
        // the staging arrays may be read by copies queued in align_all(): let them drain before the arrays can move
        scoped_device_switch dev(device_id_);
        GW_CU_CHECK_ERR(hipStreamSynchronize(stream_));
        drain_streams();
        uploads_in_flight_ = false;
    }
    if (max_bandwidth < 0 || query_length < 0 || target_length < 0 || query == nullptr || target == nullptr)
        return StatusType::generic_error;
    if (max_query_length_ >= 0)
    {
        // fixed-stride flavour (AlignerGlobal::add_alignment, aligner_global.cpp:80-112)
        if (query_length > max_query_length_ || target_length > max_target_length_) return StatusType::exceeded_max_length;
        if (num_alignments() >= max_alignments_) return StatusType::exceeded_max_alignments;
    }
    if (expand_results_)
    {
        // full-matrix flavour (AlignerGlobalMyers): the band always covers the whole query, and the kernel's
        // "max_bandwidth - 1 >= |target - query|" admission test must pass for every pair inside the limits
        // (a target of twice the query's length is a legal input of the reference class, aligner_global_myers.cpp)
        max_bandwidth = std::max(query_length, std::abs(target_length - query_length) + 2);
        if (max_bandwidth % kWordSize == 1) max_bandwidth += 1;
    }
    else if (max_bandwidth > query_length) // keep max_bandwidth % 32 != 1 (aligner_global_myers_banded.cpp:174-178)
        max_bandwidth = (query_length % kWordSize == 1 ? query_length + 1 : query_length);

    const int32_t n_alignments = num_alignments();
    const int64_t new_len_sum  = seq_starts_h_.back() + query_length + target_length;
    if ((static_cast<uint32_t>(n_alignments + 1) & (~DeviceAlignmentsPtrs::index_mask)) != 0u ||
        new_len_sum > static_cast<int64_t>(INT32_MAX))
    {
        if (n_alignments == 0) throw std::runtime_error("Could not fit alignment into device or host memory.");
        return StatusType::exceeded_max_alignments;
    }
    // device bytes if this pair joins the batch: its matrices + its share of the fixed arrays
    const int64_t starts2[3] = {0, query_length, static_cast<int64_t>(query_length) + target_length};
    // the workspace interleaves the 64 pairs of a wave and pads them to the largest of the wave: with the pairs
    // sorted by length that costs at most about one wave of the largest pair on top of the sum
    const size_t wave_ws      = gwhip_myers_banded_workspace_bytes(1, starts2, &max_bandwidth); // a whole wave of this pair
    const size_t pair_ws      = wave_ws / 64 + 64;
    largest_wave_ws_          = std::max(largest_wave_ws_, wave_ws);
    // per base: sequences 1 + packed results 1 + 4 in the device block, per-pair result slots 1 + 4 inside the kernel
    // workspace (plan_fixed, gwhip_myers.hip); per pair: starts, band width, order, result start, metadata, cell and
    // run counters on both sides
    const size_t per_pair_io  = static_cast<size_t>(query_length + target_length) * (1 + 1 + 4 + 1 + 4) + (query_length + target_length) / 2 + 160;
    const size_t new_estimate = workspace_bytes_estimate_ + pair_ws + pair_ws / 4 + per_pair_io;
    if (static_cast<int64_t>(new_estimate + largest_wave_ws_) + (1 << 20) >= max_device_memory_)
    {
        if (n_alignments == 0) throw std::runtime_error("Could not fit alignment into device or host memory.");
        return StatusType::exceeded_max_alignments;
    }
    const int64_t seq_start = seq_starts_h_.back();
    seq_h_.resize(static_cast<size_t>(new_len_sum));
    genomeutils::copy_sequence(query, query_length, seq_h_.data() + seq_start, reverse_complement_query);
    genomeutils::copy_sequence(target, target_length, seq_h_.data() + seq_start + query_length, reverse_complement_target);
    packed_h_.resize(static_cast<size_t>(new_len_sum + 1) / 2);
    pack_bases(packed_h_.data(), seq_start, seq_h_.data() + seq_start, query_length, query_code);
    pack_bases(packed_h_.data(), seq_start + query_length, seq_h_.data() + seq_start + query_length, target_length, target_code);
    seq_starts_h_.push_back(seq_start + query_length);
    seq_starts_h_.push_back(new_len_sum);
    max_bandwidths_h_.push_back(max_bandwidth);
    longest_query_            = std::max(longest_query_, query_length);
    longest_pair_             = std::max<int64_t>(longest_pair_, static_cast<int64_t>(query_length) + target_length);
    widest_band_              = std::max(widest_band_, max_bandwidth);
    workspace_bytes_estimate_ = new_estimate;
    return StatusType::success;
}

StatusType BandedAligner::align_all()
{
    GW_NVTX_RANGE(profiler, "BandedAligner::align_all");
    const int32_t n = num_alignments();
    if (n == 0) return StatusType::success;
    scoped_device_switch dev(device_id_);
    const int64_t total_len = seq_starts_h_.back();
    Tracer trace;
    // Inputs and outputs first: the sequences (the bulk of the upload: 300 MB for a million 150-bp pairs), their offsets and the
    // band widths do not depend on the processing order, so their copies are under way while the host sorts the batch and
    // sizes the workspace below (aligner_global_myers_banded.cpp:364-371 uploads after sorting; the result is the same).
    size_t off       = 0;
    auto take        = [&](size_t b) { size_t o = off; off += up256(b); return o; };
    const size_t o_seq = take(static_cast<size_t>(total_len) + 16), o_starts = take((2 * static_cast<size_t>(n) + 1) * 8);
    const size_t o_bw = take(static_cast<size_t>(n) * 4), o_order = take(static_cast<size_t>(n) * 4);
    const size_t o_res = take(static_cast<size_t>(total_len) + 16), o_cnt = take((static_cast<size_t>(total_len) + 16) * 4);
    const size_t o_rs = take((static_cast<size_t>(n) + 1) * 4), o_meta = take(static_cast<size_t>(n) * 4);
    const size_t o_cells = take(static_cast<size_t>(n) * 8);
    const size_t o_packed = take(static_cast<size_t>(total_len) / 2 + 64);
    // a previous align_all() without a sync in between may still be uploading from / computing on what this replaces
    if (uploads_in_flight_)
    {
        GW_CU_CHECK_ERR(hipStreamSynchronize(stream_));
        drain_streams();
    }
    // from here until launch() the object describes NO finished run: if an allocation below throws, a later
    // sync_alignments() must not read the previous run's offsets against the fresh, never-written result block
    launched_ = false;
    n_head_   = 0;
    free_device();
    device_block_bytes_ = off;
    device_block_       = allocator_.allocate(device_block_bytes_, {stream_});
    d_seq_              = device_block_ + o_seq;
    d_starts_           = reinterpret_cast<int64_t*>(device_block_ + o_starts);
    d_bw_               = reinterpret_cast<int32_t*>(device_block_ + o_bw);
    d_order_            = reinterpret_cast<int32_t*>(device_block_ + o_order);
    d_results_          = reinterpret_cast<int8_t*>(device_block_ + o_res);
    d_result_counts_    = reinterpret_cast<int32_t*>(device_block_ + o_cnt);
    d_result_starts_    = reinterpret_cast<int32_t*>(device_block_ + o_rs);
    d_metadata_         = reinterpret_cast<uint32_t*>(device_block_ + o_meta);
    d_cells_            = reinterpret_cast<uint64_t*>(device_block_ + o_cells);
    d_packed_           = reinterpret_cast<uint8_t*>(device_block_ + o_packed);

    // Chunks of consecutive pairs for large batches (a million short reads: the upload is 2 / 3 of the call): the upload of
    // chunk k + 1 runs on a stream of its own under the kernels of chunk k. The pairs' results do not depend on how the
    // batch is cut (scheduling order and workspace are per chunk, the packed runs are appended in input order).
    int32_t n_chunks = (n >= 131072 && total_len >= (int64_t(32) << 20)) ? 6 : 1; // (2 .. 24 measured: profiles/r05_d_aligner_chunk_sweep.txt)
    if (const char* e = std::getenv("GW_ALIGNER_CHUNKS")) n_chunks = std::max(1, std::min(std::atoi(e), std::max(1, n / 64)));
    chunks_.resize(static_cast<size_t>(n_chunks));
    for (int32_t k = 0; k < n_chunks; ++k)
    {
        chunks_[static_cast<size_t>(k)].lo = static_cast<int32_t>(static_cast<int64_t>(n) * k / n_chunks);
        chunks_[static_cast<size_t>(k)].hi = static_cast<int32_t>(static_cast<int64_t>(n) * (k + 1) / n_chunks);
        Chunk& c       = chunks_[static_cast<size_t>(k)];
        c.first_offset = seq_starts_h_[2 * static_cast<size_t>(c.lo)];
        c.span         = seq_starts_h_[2 * static_cast<size_t>(c.hi)] - c.first_offset;
    }
    launched_total_length_ = total_len;
    raw_upload_            = std::getenv("GW_ALIGNER_RAW_UPLOAD") != nullptr;
    // a chunked batch's runs also arrive in a pinned mirror as the chunks finish (gwhip_myers_args::results_host): sized for 16 runs
    // per pair or one per 16 bases; sync_alignments() copies for itself when a batch has more
    if (n_chunks > 1)
    {
        int64_t want     = std::min<int64_t>(total_len, std::max<int64_t>(16 * static_cast<int64_t>(n), total_len / 16));
        const char* runs = std::getenv("GW_ALIGNER_MIRROR_RUNS"); // tests: a capacity the batch exceeds
        if (runs != nullptr) want = std::max<int64_t>(1, std::min<int64_t>(std::atoll(runs), total_len));
        if (mirror_ == nullptr || mirror_runs_ < want || runs != nullptr)
        {
            if (mirror_ != nullptr) pinned_release(mirror_, mirror_cap_);
            mirror_      = nullptr;
            mirror_runs_ = 0;
            mirror_      = pinned_acquire(static_cast<size_t>((want + 63) & ~int64_t(63)) + static_cast<size_t>(want) * 4 + 64, &mirror_cap_);
            mirror_runs_ = want;
        }
    }
    hipStream_t up = stream_;
    if (n_chunks > 1)
    {
        if (upload_stream_ == nullptr)
        {
            hipStream_t s = nullptr;
            GW_CU_CHECK_ERR(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            upload_stream_ = s;
        }
        if (side_stream_ == nullptr)
        {
            hipStream_t s = nullptr;
            GW_CU_CHECK_ERR(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            side_stream_ = s;
        }
        while (upload_events_.size() < 2 * static_cast<size_t>(n_chunks) + 2)
        {
            hipEvent_t e = nullptr;
            GW_CU_CHECK_ERR(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            upload_events_.push_back(e);
        }
        up = static_cast<hipStream_t>(upload_stream_);
        // the fresh block may be memory that earlier work on the aligner's stream still uses: the uploads start behind it
        hipEvent_t begin = static_cast<hipEvent_t>(upload_events_[static_cast<size_t>(n_chunks)]);
        GW_CU_CHECK_ERR(hipEventRecord(begin, stream_));
        GW_CU_CHECK_ERR(hipStreamWaitEvent(up, begin, 0));
    }
    PinnedVector<int32_t> order;
    order.resize(static_cast<size_t>(n));
    // The first two chunks' inputs are queued on the upload stream right away (pinned sources: the calls return at once); the link
    // runs while the host orders and sizes the chunks -- on host threads: for a million short pairs that work (5 ms on one
    // thread) would otherwise be longer than the uploads and the kernels. run_chunks() queues the inputs of chunk k + 2 when it
    // has queued chunk k + 1's processing order: copies of other streams are then never submitted behind uploads they do not need
    // (copies of different streams can share an engine, which runs them in submission order).
    for (int32_t k = 0; k < std::min(n_chunks, 2); ++k) enqueue_inputs(static_cast<size_t>(k));
    uploads_in_flight_ = true;
    trace.mark("align_all: device block, uploads enqueued");
    // processing order (longest pairs first within a chunk, aligner_global_myers_banded.cpp:306-309) and workspace words of every
    // chunk, on host threads (chunk_order.hpp)
    {
        std::vector<gwhost::PairRange> ranges;
        for (const Chunk& c : chunks_) ranges.push_back(gwhost::PairRange{c.lo, c.hi});
        const size_t host_threads      = std::min<size_t>(32, std::max(1u, std::thread::hardware_concurrency()));
        const std::vector<int64_t> words = gwhost::order_and_size_chunks(ranges, seq_starts_h_.data(), max_bandwidths_h_.data(), longest_pair_, host_threads,
                                                                         order.data(), &gwhip_myers_banded_workspace_words);
        for (size_t k = 0; k < chunks_.size(); ++k)
            chunks_[k].workspace_bytes = gwhip_myers_banded_workspace_bytes_of_words(chunks_[k].hi - chunks_[k].lo, chunks_[k].span, words[k]);
    }
    trace.mark("align_all: chunks ordered and sized");
    try
    {
        run_chunks(order.data(), true);
    }
    catch (...)
    {
        // copies queued so far read `order` and the staging arrays; kernels queued so far use the device block
        drain_streams();
        throw;
    }
    order_h_ = std::move(order);
    trace.mark("align_all: chunks launched");
    launched_ = true;
    return StatusType::success;
}

void BandedAligner::enqueue_inputs(size_t k)
{
    Chunk& c         = chunks_[k];
    hipStream_t up   = chunks_.size() > 1 ? static_cast<hipStream_t>(upload_stream_) : stream_;
    const int64_t b0 = seq_starts_h_[2 * static_cast<size_t>(c.lo)], b1 = seq_starts_h_[2 * static_cast<size_t>(c.hi)];
    const size_t m   = static_cast<size_t>(c.hi - c.lo);
    if (b1 > b0 && raw_upload_)
        GW_CU_CHECK_ERR(hipMemcpyAsync(d_seq_ + b0, seq_h_.data() + b0, static_cast<size_t>(b1 - b0), hipMemcpyHostToDevice, up));
    else if (b1 > b0)
    {
        // the bases go up two per byte -- half the bytes over the link -- and are expanded on the device (run_chunks())
        const int64_t p0 = b0 >> 1, p1 = (b1 + 1) >> 1;
        GW_CU_CHECK_ERR(hipMemcpyAsync(d_packed_ + p0, packed_h_.data() + p0, static_cast<size_t>(p1 - p0), hipMemcpyHostToDevice, up));
    }
    GW_CU_CHECK_ERR(hipMemcpyAsync(d_starts_ + 2 * static_cast<size_t>(c.lo), seq_starts_h_.data() + 2 * static_cast<size_t>(c.lo), (2 * m + 1) * 8,
                                   hipMemcpyHostToDevice, up));
    GW_CU_CHECK_ERR(hipMemcpyAsync(d_bw_ + c.lo, max_bandwidths_h_.data() + c.lo, m * 4, hipMemcpyHostToDevice, up));
    if (chunks_.size() > 1)
    {
        c.uploaded = upload_events_[k];
        GW_CU_CHECK_ERR(hipEventRecord(static_cast<hipEvent_t>(c.uploaded), up));
    }
}

void BandedAligner::run_chunks(const int32_t* order, bool allocate)
{
    // A chunked batch keeps three streams busy: the uploads; the alignment kernels, back to back on the aligner's stream; and on
    // the side stream what surrounds them (gwhip_myers_args::side_stream, ::phases) -- the processing order going up and the
    // workspace sizing of chunk k + 1 (queued ahead of) the run-offset scan and compaction of chunk k and its offsets' way to
    // the host. Events: upload_events_[k] = chunk k's inputs are up, [n + 2 + k] = its workspace is sized.
    const size_t n_chunks = chunks_.size();
    prepare_head();
    // the chunk's bases from two per byte to characters (gwhip_unpack_bases), on the stream that has waited for its upload
    auto unpack_chunk = [&](const Chunk& c, hipStream_t s) {
        if (raw_upload_ || c.span == 0) return;
        const int rc = gwhip_unpack_bases(d_packed_, d_seq_, c.first_offset, c.first_offset + c.span, s);
        if (rc != 0) GW_CU_CHECK_ERR(static_cast<hipError_t>(rc));
    };
    if (n_chunks <= 1)
    {
        for (Chunk& c : chunks_)
        {
            if (order != nullptr)
            {
                GW_CU_CHECK_ERR(hipMemcpyAsync(d_order_ + c.lo, order + c.lo, static_cast<size_t>(c.hi - c.lo) * 4, hipMemcpyHostToDevice, stream_));
                unpack_chunk(c, stream_);
            }
            if (allocate)
            {
                c.block_bytes = up256(c.workspace_bytes);
                c.workspace   = allocator_.allocate(c.block_bytes, {stream_});
            }
            launch_chunk(c);
            fetch_head_slice(c, stream_);
        }
        return;
    }
    hipStream_t side = static_cast<hipStream_t>(side_stream_);
    auto size_chunk  = [&](size_t k) {
        Chunk& c = chunks_[k];
        if (c.uploaded != nullptr && order != nullptr) GW_CU_CHECK_ERR(hipStreamWaitEvent(side, static_cast<hipEvent_t>(c.uploaded), 0));
        // (the copy reads the caller's vector, whose storage order_h_ takes over at the end of align_all())
        if (order != nullptr)
        {
            GW_CU_CHECK_ERR(hipMemcpyAsync(d_order_ + c.lo, order + c.lo, static_cast<size_t>(c.hi - c.lo) * 4, hipMemcpyHostToDevice, side));
            unpack_chunk(c, side);
        }
        if (allocate)
        {
            c.block_bytes = up256(c.workspace_bytes);
            c.workspace   = allocator_.allocate(c.block_bytes, {stream_});
        }
        launch_chunk(c, GWHIP_MYERS_SIZING);
        GW_CU_CHECK_ERR(hipEventRecord(static_cast<hipEvent_t>(upload_events_[n_chunks + 2 + k]), side));
    };
    size_chunk(0);
    for (size_t k = 0; k < n_chunks; ++k)
    {
        if (k + 1 < n_chunks) size_chunk(k + 1);
        if (order != nullptr && k + 2 < n_chunks) enqueue_inputs(k + 2);
        Chunk& c = chunks_[k];
        if (c.uploaded != nullptr && order != nullptr) GW_CU_CHECK_ERR(hipStreamWaitEvent(stream_, static_cast<hipEvent_t>(c.uploaded), 0));
        GW_CU_CHECK_ERR(hipStreamWaitEvent(stream_, static_cast<hipEvent_t>(upload_events_[n_chunks + 2 + k]), 0));
        launch_chunk(c, GWHIP_MYERS_ALIGN);
    }
    join_side_stream();
}

void BandedAligner::launch_chunk(const Chunk& c, int32_t phases)
{
    gwhip_myers_args a{};
    const size_t lo         = static_cast<size_t>(c.lo);
    a.n_alignments          = c.hi - c.lo;
    a.sequences             = d_seq_;
    a.sequence_starts       = d_starts_ + 2 * lo;
    a.max_bandwidths        = d_bw_ + lo;
    a.results               = d_results_;
    a.result_counts         = d_result_counts_;
    a.result_starts         = d_result_starts_ + lo;
    a.result_metadata       = d_metadata_ + lo;
    a.results_capacity      = launched_total_length_;
    a.workspace             = c.workspace;
    a.workspace_bytes       = c.workspace_bytes;
    a.total_sequence_length = c.span;
    a.first_sequence_offset = c.first_offset;
    a.index_base            = c.lo;
    a.result_starts_base    = c.lo > 0 ? d_result_starts_ + lo : nullptr; // written by the chunk before this one
    a.scheduling_index      = d_order_ + lo;
    a.band_cells            = d_cells_ + lo;
    a.side_stream           = chunks_.size() > 1 ? side_stream_ : nullptr;
    a.phases                = phases;
    if (chunks_.size() > 1)
    {
        // the chunk's offsets and metadata reach the pinned head by a kernel of the call (no copies queued behind the uploads)
        a.result_starts_host   = reinterpret_cast<int32_t*>(head_) + lo;
        a.result_metadata_host = reinterpret_cast<uint32_t*>(head_) + static_cast<size_t>(n_head_) + 1 + lo;
    }
    if (chunks_.size() > 1 && mirror_ != nullptr)
    {
        a.results_host          = reinterpret_cast<int8_t*>(mirror_);
        a.result_counts_host    = reinterpret_cast<int32_t*>(mirror_ + static_cast<size_t>((mirror_runs_ + 63) & ~int64_t(63)));
        a.results_host_capacity = mirror_runs_;
    }
    // hints for the LDS-cached kernel variant: longest query and widest band of this batch
    a.max_query_length   = longest_query_;
    a.max_bandwidth_hint = widest_band_;
    const int rc         = gwhip_myers_banded(&a, stream_);
    if (rc != 0)
    {
        char buf[512];
        gwhip_last_error_string(buf, sizeof(buf));
        GW_LOG_ERROR(buf);
        GW_CU_CHECK_ERR(static_cast<hipError_t>(rc));
    }
}

void BandedAligner::launch(void* event_before, void* event_after)
{
    if (event_before != nullptr) GW_CU_CHECK_ERR(hipEventRecord(static_cast<hipEvent_t>(event_before), stream_));
    if (chunks_.size() > 1) // the side stream's work of this round starts behind what the aligner's stream holds
    {
        hipEvent_t begin = static_cast<hipEvent_t>(upload_events_[chunks_.size()]);
        GW_CU_CHECK_ERR(hipEventRecord(begin, stream_));
        GW_CU_CHECK_ERR(hipStreamWaitEvent(static_cast<hipStream_t>(side_stream_), begin, 0));
    }
    try
    {
        run_chunks(nullptr, false);
    }
    catch (...)
    {
        // a HIP call failed in the middle of a chunked batch: the side / upload streams hold work nobody joins, and head_ and the
        // pinned mirror may still be written; drain all three streams and take the batch out of its launched state, as align_all() does
        drain_streams();
        launched_ = false;
        throw;
    }
    if (event_after != nullptr) GW_CU_CHECK_ERR(hipEventRecord(static_cast<hipEvent_t>(event_after), stream_));
}

void BandedAligner::join_side_stream()
{
    if (chunks_.size() <= 1) return;
    hipEvent_t done = static_cast<hipEvent_t>(upload_events_[chunks_.size() + 1]);
    GW_CU_CHECK_ERR(hipEventRecord(done, static_cast<hipStream_t>(side_stream_)));
    GW_CU_CHECK_ERR(hipStreamWaitEvent(stream_, done, 0));
}

void BandedAligner::prepare_head()
{
    // result offsets and metadata follow the kernels to the host (pinned), as the reference's align_all() does with its
    // result_starts (aligner_global_myers_banded.cpp:372-374): sync_alignments() and get_alignments_device() read
    // them after the stream has drained
    const size_t un = chunks_.empty() ? 0 : static_cast<size_t>(chunks_.back().hi);
    if (head_ == nullptr || head_cap_ < (2 * un + 1) * 4) // (sync_alignments() hands the buffer to the views' block)
    {
        if (head_ != nullptr) pinned_release(head_, head_cap_);
        head_ = pinned_acquire((2 * un + 1) * 4, &head_cap_);
    }
    n_head_ = static_cast<int32_t>(un);
}

void BandedAligner::fetch_head_slice(const Chunk& c, void* stream)
{
    const size_t un = static_cast<size_t>(n_head_), lo = static_cast<size_t>(c.lo), m = static_cast<size_t>(c.hi - c.lo);
    hipStream_t s   = static_cast<hipStream_t>(stream);
    // entry hi of the offsets is the number of runs up to the chunk's end (the batch's total for the last chunk)
    GW_CU_CHECK_ERR(hipMemcpyAsync(head_ + lo * 4, d_result_starts_ + lo, (m + 1) * 4, hipMemcpyDeviceToHost, s));
    GW_CU_CHECK_ERR(hipMemcpyAsync(head_ + (un + 1 + lo) * 4, d_metadata_ + lo, m * 4, hipMemcpyDeviceToHost, s));
}

void BandedAligner::relaunch_resident()
{
    if (!launched_) return;
    scoped_device_switch dev(device_id_);
    launch();
}

float BandedAligner::relaunch_resident_timed()
{
    if (!launched_) return 0.f;
    scoped_device_switch dev(device_id_);
    hipEvent_t e0, e1;
    GW_CU_CHECK_ERR(hipEventCreate(&e0));
    GW_CU_CHECK_ERR(hipEventCreate(&e1));
    launch(e0, e1);
    GW_CU_CHECK_ERR(hipEventSynchronize(e1));
    float ms = 0.f;
    GW_CU_CHECK_ERR(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms;
}

uint64_t BandedAligner::total_band_cells()
{
    const int32_t n = num_alignments();
    if (!launched_ || n == 0) return 0;
    scoped_device_switch dev(device_id_);
    std::vector<uint64_t> h(static_cast<size_t>(n));
    GW_CU_CHECK_ERR(hipMemcpyAsync(h.data(), d_cells_, h.size() * 8, hipMemcpyDeviceToHost, stream_));
    GW_CU_CHECK_ERR(hipStreamSynchronize(stream_));
    uint64_t t = 0;
    for (uint64_t v : h) t += v;
    return t;
}

StatusType BandedAligner::sync_alignments()
{
    GW_NVTX_RANGE(profiler, "BandedAligner::sync");
    const int32_t n = num_alignments();
    alignments_.clear();
    if (n == 0) return StatusType::success;
    scoped_device_switch dev(device_id_);
    // One block per sync: the packed runs arrive in pinned memory that the block keeps; the Alignment objects are views into it
    // (alignment_impl.hpp). A view holds nothing but its index: it reads its run range and flags from the block's copy of
    // [result_starts | metadata] when asked -- so the million views of a short-read batch are laid out and published on host
    // threads WHILE the uploads and kernels of align_all() still run, and only the runs themselves wait for the device.
    auto block            = std::make_shared<PackedAlignmentBlock>();
    block->expand_states  = expand_results_;
    const size_t un       = static_cast<size_t>(n);
    if (!launched_ || n_head_ != n || head_ == nullptr) throw std::runtime_error("sync_alignments() called before align_all()");
    Tracer trace;
    // From here on the batch's pinned arrays belong to the block while copies from / into them may be in flight: any exception
    // below (bad_alloc, a HIP error) must first drain the stream -- the block's destructor hands the pinned buffers back to the
    // process-wide cache -- and leave the aligner in its empty, consistent state.
    try
    {
        block->sequences_buffer  = seq_h_.detach(&block->sequences_bytes);
        block->sequences         = block->sequences_buffer;
        block->seq_starts_buffer = reinterpret_cast<char*>(seq_starts_h_.detach(&block->seq_starts_bytes));
        block->seq_starts        = reinterpret_cast<const int64_t*>(block->seq_starts_buffer);
        block->head_buffer       = head_;
        block->head_bytes        = head_cap_;
        block->run_starts        = reinterpret_cast<const int32_t*>(head_);
        block->metadata          = reinterpret_cast<const uint32_t*>(head_) + un + 1; // the kernels write metadata[i] = i | flags
        head_                    = nullptr;
        head_cap_                = 0;
        block->allocate_views(un);
        alignments_.resize(un);
        trace.mark("sync: view storage");
        // the shared_ptr of every view aliases the block (no allocation per alignment); big batches are split over host threads,
        // each with its own copy of the owner so that the reference count is not one contended cache line
        auto bind_range = [&](size_t first, size_t last) {
            // a control block of this thread's own that keeps the block alive: every aliasing shared_ptr below bumps ITS
            // reference count, so the threads do not fight over one cache line (copies of `block` would share one counter:
            // a million contended atomic increments were 9 of the 10 ms this loop took)
            const std::shared_ptr<PackedAlignmentBlock> owner(block.get(), [keep = block](PackedAlignmentBlock*) {});
            for (size_t i = first; i < last; ++i)
            {
                new (&block->alignments[i]) PackedAlignment(block.get(), static_cast<int32_t>(i));
                alignments_[i] = std::shared_ptr<Alignment>(owner, &block->alignments[i]);
            }
        };
        const size_t n_threads = un >= 65536 ? std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())) : 1;
        const size_t share = (un + n_threads - 1) / n_threads;
        gwhost::parallel_tasks(n_threads, n_threads, [&](size_t t) { bind_range(std::min(un, t * share), std::min(un, (t + 1) * share)); });
        block->n_alignments = un;
        trace.mark("sync: views bound and published");
        GW_CU_CHECK_ERR(hipStreamSynchronize(stream_)); // uploads, kernels and the offsets / metadata copy queued by align_all()
        uploads_in_flight_ = false;
        trace.mark("sync: stream drained (H2D + kernels)");
        const size_t total = static_cast<size_t>(block->run_starts[un]);
        if (chunks_.size() > 1 && mirror_ != nullptr && static_cast<int64_t>(total) <= mirror_runs_)
        {
            // the chunks' kernels have put the runs there already: the block takes the buffer over
            block->pinned       = mirror_;
            block->pinned_bytes = mirror_cap_;
            block->ops          = reinterpret_cast<const int8_t*>(mirror_);
            block->counts       = reinterpret_cast<const int32_t*>(mirror_ + static_cast<size_t>((mirror_runs_ + 63) & ~int64_t(63)));
            mirror_             = nullptr;
            mirror_cap_         = 0;
            mirror_runs_        = 0;
        }
        else
        {
            const size_t counts_at = (total + 63) & ~size_t(63);
            block->pinned          = pinned_acquire(counts_at + total * 4 + 64, &block->pinned_bytes);
            block->ops             = reinterpret_cast<const int8_t*>(block->pinned);
            block->counts          = reinterpret_cast<const int32_t*>(block->pinned + counts_at);
            if (total > 0)
            {
                GW_CU_CHECK_ERR(hipMemcpyAsync(block->pinned, d_results_, total, hipMemcpyDeviceToHost, stream_));
                GW_CU_CHECK_ERR(hipMemcpyAsync(block->pinned + counts_at, d_result_counts_, total * 4, hipMemcpyDeviceToHost, stream_));
                GW_CU_CHECK_ERR(hipStreamSynchronize(stream_));
            }
        }
        trace.mark("sync: runs on the host");
        total_length_h_ = static_cast<int64_t>(total);
    }
    catch (...)
    {
        drain_streams(); // all three: the side stream's mirror kernels write the pinned buffers the block is about to release
        uploads_in_flight_ = false;
        alignments_.clear();
        reset_data();
        throw;
    }
    // keep the device block (device-resident results stay valid until reset()); host queues are cleared like the reference
    n_last_ = n;
    reset_data();
    return StatusType::success;
}

DeviceAlignmentsPtrs BandedAligner::get_alignments_device() const
{
    DeviceAlignmentsPtrs r{};
    r.cigar_operations = d_results_;
    r.cigar_runlengths = d_result_counts_;
    r.cigar_offsets    = d_result_starts_;
    r.metadata         = d_metadata_;
    // after align_all() + a stream synchronisation the offsets are on the host; after sync_alignments() the total is kept
    r.total_length     = (launched_ && head_ != nullptr) ? reinterpret_cast<const int32_t*>(head_)[static_cast<size_t>(n_head_)] : total_length_h_;
    r.n_alignments     = launched_ ? num_alignments() : n_last_;
    return r;
}

// ---- factories (aligner.cpp:31-126) ------------------------------------------------------------------------
std::unique_ptr<Aligner> create_aligner(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments,
                                        AlignmentType type, DefaultDeviceAllocator allocator, cudaStream_t stream,
                                        int32_t device_id)
{
    if (type != AlignmentType::global_alignment) throw std::runtime_error("Aligner for specified type not implemented yet.");
    throw_on_negative(max_query_length, "max_query_length must be non-negative.");
    throw_on_negative(max_target_length, "max_target_length must be non-negative.");
    throw_on_negative(max_alignments, "max_alignments must be non-negative.");
    // the reference's default: Hirschberg + Myers (aligner.cpp:39-43)
    return std::make_unique<HirschbergAligner>(max_query_length, max_target_length, max_alignments, allocator, stream, device_id);
}

std::unique_ptr<Aligner> create_aligner(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments,
                                        AlignmentType type, cudaStream_t stream, int32_t device_id, int64_t max_mem)
{
    scoped_device_switch device(device_id);
    if (max_mem < -1)
        throw std::invalid_argument("max_device_memory_allocator_caching_size has to be either -1 (=all available GPU memory) or greater or equal than 0.");
    if (max_mem == -1)
    {
        max_mem = cudautils::find_largest_contiguous_device_memory_section();
        if (max_mem == 0) throw std::runtime_error("No memory available for caching");
    }
    DefaultDeviceAllocator allocator(static_cast<size_t>(max_mem), stream);
    return create_aligner(max_query_length, max_target_length, max_alignments, type, allocator, stream, device_id);
}

std::unique_ptr<FixedBandAligner> create_aligner(AlignmentType type, int32_t max_bandwidth, cudaStream_t stream,
                                                 int32_t device_id, DefaultDeviceAllocator allocator, int64_t max_device_memory)
{
    if (type != AlignmentType::global_alignment) throw std::runtime_error("Aligner for specified type not implemented yet.");
    return std::make_unique<BandedAligner>(max_device_memory, max_bandwidth, allocator, stream, device_id, false, -1, -1, -1);
}

std::unique_ptr<FixedBandAligner> create_aligner(AlignmentType type, int32_t max_bandwidth, cudaStream_t stream,
                                                 int32_t device_id, int64_t max_device_memory)
{
    scoped_device_switch device(device_id);
    if (max_device_memory < -1)
        throw std::invalid_argument("max_device_memory has to be either -1 (=all available GPU memory), or greater than or equal to 0.");
    if (max_device_memory == -1)
    {
        max_device_memory = cudautils::find_largest_contiguous_device_memory_section();
        if (max_device_memory == 0) throw std::runtime_error("No memory available for caching");
    }
    DefaultDeviceAllocator allocator(static_cast<size_t>(max_device_memory), stream);
    return create_aligner(type, max_bandwidth, stream, device_id, allocator, -1);
}

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
