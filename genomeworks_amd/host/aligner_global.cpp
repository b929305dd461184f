// aligner_global.cpp -- host side of the fixed-limit global aligners (see aligner_global.hpp).
// The sequences are packed back to back (q0 t0 q1 t1 ...) instead of the reference's fixed 2 * max_length stride;
// limits, statuses and the host-side reversal of the kernels' back-to-front paths follow aligner_global.cpp.
#include <cstring>
#include <exception>
#include <thread>
#include "aligner_global.hpp"

#include <claraparabricks/genomeworks/utils/cudautils.hpp>
#include <claraparabricks/genomeworks/utils/genomeutils.hpp>
#include <claraparabricks/genomeworks/utils/signed_integer_utils.hpp>
#include <claraparabricks/genomeworks/logging/logging.hpp>

#include <algorithm>
#include <stdexcept>

#include "../../include/gwhip.h"
#include "alignment_impl.hpp"
#include "host_common.hpp"

namespace claraparabricks
{
namespace genomeworks
{
namespace cudaaligner
{

AlignerGlobal::AlignerGlobal(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments,
                                     DefaultDeviceAllocator allocator, cudaStream_t stream, int32_t device_id)
    : max_query_length_(throw_on_negative(max_query_length, "max_query_length must be non-negative."))
    , max_target_length_(throw_on_negative(max_target_length, "max_target_length must be non-negative."))
    , max_alignments_(throw_on_negative(max_alignments, "max_alignments must be non-negative."))
    , allocator_(allocator)
    , stream_(stream)
    , device_id_(device_id)
{
    if (max_alignments < 1) throw std::runtime_error("Max alignments must be at least 1.");
    seq_starts_h_.assign(1, 0);
}

AlignerGlobal::~AlignerGlobal()
{
    scoped_device_switch dev(device_id_);
    (void)hipStreamSynchronize(stream_);
    free_device();
}

void AlignerGlobal::free_device()
{
    if (device_block_)
    {
        allocator_.deallocate(device_block_, device_block_bytes_);
        device_block_       = nullptr;
        device_block_bytes_ = 0;
    }
}

StatusType AlignerGlobal::add_alignment(const char* query, int32_t query_length, const char* target, int32_t target_length,
                                            bool reverse_complement_query, bool reverse_complement_target)
{
    if (query_length < 0 || target_length < 0)
    {
        GW_LOG_DEBUG("Negative target or query length is not allowed.");
        return StatusType::generic_error;
    }
    if (num_alignments() >= max_alignments_) return StatusType::exceeded_max_alignments;
    if (query_length > max_query_length_ || target_length > max_target_length_) return StatusType::exceeded_max_length;
    if (launched_) // the staging arrays are pinned: a batch still in flight reads them by DMA
    {
        scoped_device_switch dev(device_id_);
        GW_CU_CHECK_ERR(hipStreamSynchronize(stream_));
    }
    const int64_t begin = seq_starts_h_.back();
    seq_h_.resize(static_cast<size_t>(begin + query_length + target_length));
    genomeutils::copy_sequence(query, query_length, seq_h_.data() + begin, reverse_complement_query);
    genomeutils::copy_sequence(target, target_length, seq_h_.data() + begin + query_length, reverse_complement_target);
    seq_starts_h_.push_back(begin + query_length);
    seq_starts_h_.push_back(begin + query_length + target_length);
    // the Alignment object exists from now on (status uninitialized) and is filled in place by sync_alignments(), as in
    // the reference (aligner_global.cpp:100-109): a caller may hold the vector from before the sync -- its Python
    // binding does (cudaaligner.pyx:243-247)
    auto alignment = std::make_shared<AlignmentImpl>(seq_h_.data() + begin, query_length, seq_h_.data() + begin + query_length, target_length);
    alignment->set_alignment_type(AlignmentType::global_alignment);
    alignments_.push_back(std::move(alignment));
    launched_ = false;
    return StatusType::success;
}

StatusType AlignerGlobal::align_all()
{
    const int32_t n = num_alignments();
    if (n == 0) return StatusType::success;
    scoped_device_switch dev(device_id_);
    const int64_t total = seq_starts_h_.back();
    const size_t ws_bytes = workspace_bytes(n, seq_starts_h_.data());
    size_t off = 0;
    auto take  = [&](size_t b) { size_t o = off; off += (b + 255) / 256 * 256; return o; };
    const size_t o_seq = take(static_cast<size_t>(total) + 16), o_starts = take((2 * static_cast<size_t>(n) + 1) * 8);
    const size_t o_res = take(static_cast<size_t>(total) + 16), o_len = take(static_cast<size_t>(n) * 4), o_ws = take(ws_bytes);
    free_device();
    device_block_bytes_ = off;
    device_block_       = allocator_.allocate(device_block_bytes_, {stream_});
    char* d_seq         = device_block_ + o_seq;
    int64_t* d_starts   = reinterpret_cast<int64_t*>(device_block_ + o_starts);
    d_results_          = reinterpret_cast<int8_t*>(device_block_ + o_res);
    d_result_lengths_   = reinterpret_cast<int32_t*>(device_block_ + o_len);
    GW_CU_CHECK_ERR(hipMemcpyAsync(d_seq, seq_h_.data(), static_cast<size_t>(total), hipMemcpyHostToDevice, stream_));
    GW_CU_CHECK_ERR(hipMemcpyAsync(d_starts, seq_starts_h_.data(), seq_starts_h_.size() * 8, hipMemcpyHostToDevice, stream_));
    d_seq_    = d_seq;
    d_starts_ = d_starts;
    d_ws_     = device_block_ + o_ws;
    ws_bytes_ = ws_bytes;
    const int rc = run_alignment(n, d_seq, d_starts, seq_starts_h_.data(), d_results_, d_result_lengths_, device_block_ + o_ws, ws_bytes);
    if (rc != 0)
    {
        char buf[512];
        gwhip_last_error_string(buf, sizeof(buf));
        GW_LOG_ERROR(buf);
        GW_CU_CHECK_ERR(static_cast<hipError_t>(rc));
    }
    results_h_.resize(static_cast<size_t>(total) + 16);
    result_lengths_h_.resize(static_cast<size_t>(n));
    GW_CU_CHECK_ERR(hipMemcpyAsync(results_h_.data(), d_results_, static_cast<size_t>(total), hipMemcpyDeviceToHost, stream_));
    GW_CU_CHECK_ERR(hipMemcpyAsync(result_lengths_h_.data(), d_result_lengths_, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost, stream_));
    launched_ = true;
    return StatusType::success;
}

StatusType AlignerGlobal::sync_alignments()
{
    scoped_device_switch dev(device_id_);
    GW_CU_CHECK_ERR(hipStreamSynchronize(stream_));
    const size_t n = static_cast<size_t>(num_alignments());
    if (!launched_ || n == 0) return StatusType::success;
    // the device writes every path back to front; one reversed copy per alignment, handed over without a second one.
    // Big batches are split over a few host threads (the alignments are independent objects).
    auto fill_range = [&](size_t first, size_t last) {
        for (size_t i = first; i < last; ++i)
        {
            const int64_t qlen = seq_starts_h_[2 * i + 1] - seq_starts_h_[2 * i];
            const int64_t tlen = seq_starts_h_[2 * i + 2] - seq_starts_h_[2 * i + 1];
            AlignmentImpl* alignment = dynamic_cast<AlignmentImpl*>(alignments_[i].get());
            const int32_t len        = result_lengths_h_[i];
            const size_t count       = static_cast<size_t>(std::abs(len));
            const int8_t* r_begin    = results_h_.data() + seq_starts_h_[2 * i];
            std::vector<AlignmentState> states(count);
            static_assert(sizeof(AlignmentState) == 1, "the device writes one byte per state");
            {
                // reversed copy, eight states at a time (one byte-swapped 64-bit word), the tail byte by byte
                uint8_t* dst       = reinterpret_cast<uint8_t*>(states.data());
                const uint8_t* src = reinterpret_cast<const uint8_t*>(r_begin);
                size_t k           = 0;
                for (; k + 8 <= count; k += 8)
                {
                    uint64_t v;
                    std::memcpy(&v, src + count - 8 - k, 8);
                    v = __builtin_bswap64(v);
                    std::memcpy(dst + k, &v, 8);
                }
                for (; k < count; ++k) dst[k] = src[count - 1 - k];
            }
            if (count != 0 || (qlen == 0 && tlen == 0))
            {
                alignment->set_alignment(std::move(states), len >= 0);
                alignment->set_status(StatusType::success);
            }
        }
    };
    const size_t total     = static_cast<size_t>(seq_starts_h_.back());
    const size_t n_threads = (n >= 256 && total >= (size_t(1) << 20)) ? std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (n_threads <= 1)
        fill_range(0, n);
    else
    {
        const size_t chunk = (n + n_threads - 1) / n_threads;
        std::vector<std::thread> workers;
        std::vector<std::exception_ptr> errors(n_threads);
        {
            // joins whatever was started on every exit path (a throwing emplace_back or the caller's own range included);
            // a worker's exception is carried back to the caller instead of ending the process
            struct JoinAll
            {
                std::vector<std::thread>& threads;
                ~JoinAll()
                {
                    for (std::thread& t : threads)
                        if (t.joinable()) t.join();
                }
            } join_on_exit{workers};
            workers.reserve(n_threads);
            for (size_t t = 1; t < n_threads; ++t)
                workers.emplace_back([&, t] {
                    try
                    {
                        fill_range(std::min(n, t * chunk), std::min(n, (t + 1) * chunk));
                    }
                    catch (...)
                    {
                        errors[t] = std::current_exception();
                    }
                });
            fill_range(0, std::min(n, chunk));
        }
        for (const std::exception_ptr& e : errors)
            if (e) std::rethrow_exception(e);
    }
    return StatusType::success;
}

float AlignerGlobal::relaunch_resident_timed()
{
    if (!launched_ || device_block_ == nullptr) return -1.f;
    scoped_device_switch dev(device_id_);
    struct EventPair // destroyed on every exit path
    {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~EventPair()
        {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } ev;
    GW_CU_CHECK_ERR(hipEventCreate(&ev.e0));
    GW_CU_CHECK_ERR(hipEventCreate(&ev.e1));
    GW_CU_CHECK_ERR(hipEventRecord(ev.e0, stream_));
    const int rc = run_alignment(num_alignments(), d_seq_, d_starts_, seq_starts_h_.data(), d_results_, d_result_lengths_, d_ws_, ws_bytes_);
    GW_CU_CHECK_ERR(hipEventRecord(ev.e1, stream_));
    GW_CU_CHECK_ERR(hipEventSynchronize(ev.e1));
    float ms = 0.f;
    GW_CU_CHECK_ERR(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    return rc == 0 ? ms : -1.f;
}

DeviceAlignmentsPtrs AlignerGlobal::get_alignments_device() const
{
    // the packed run-length form of DeviceAlignmentsPtrs is only produced by the banded aligner; as in the reference
    // (aligner_global.hpp:54-64, "TODO implement for other aligners") the others hand back null pointers
    return DeviceAlignmentsPtrs{};
}

void AlignerGlobal::reset()
{
    scoped_device_switch dev(device_id_);
    (void)hipStreamSynchronize(stream_);
    alignments_.clear();
    seq_h_.clear();
    seq_starts_h_.assign(1, 0);
    launched_ = false;
    free_device();
}

// ---- Hirschberg + Myers (the default) ----------------------------------------------------------------------------
size_t AlignerGlobalHirschbergMyers::workspace_bytes(int32_t n, const int64_t* sequence_starts) const
{
    return gwhip_hirschberg_myers_workspace_bytes(n, sequence_starts, get_max_query_length());
}

int AlignerGlobalHirschbergMyers::run_alignment(int32_t n, const char* sequences_d, const int64_t* sequence_starts_d,
                                                const int64_t*, int8_t* results_d, int32_t* result_lengths_d,
                                                void* workspace_d, size_t workspace_size)
{
    gwhip_hirschberg_args a{};
    a.n_alignments     = n;
    a.sequences        = sequences_d;
    a.sequence_starts  = sequence_starts_d;
    a.max_query_length = get_max_query_length();
    a.results          = results_d;
    a.result_lengths   = result_lengths_d;
    a.workspace        = workspace_d;
    a.workspace_bytes  = workspace_size;
    return gwhip_hirschberg_myers(&a, get_stream());
}

// ---- Ukkonen ----------------------------------------------------------------------------------------------------
namespace
{
constexpr float max_target_query_length_difference = 0.1f; // query has to be >= 90 % of target length (aligner_global_ukkonen.cpp:30)
} // namespace

AlignerGlobalUkkonen::AlignerGlobalUkkonen(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments,
                                           DefaultDeviceAllocator allocator, cudaStream_t stream, int32_t device_id)
    : AlignerGlobal(max_query_length, max_target_length, max_alignments, allocator, stream, device_id)
    , ukkonen_p_(100) // aligner_global_ukkonen.cpp:35
{
}

StatusType AlignerGlobalUkkonen::add_alignment(const char* query, int32_t query_length, const char* target, int32_t target_length,
                                               bool reverse_complement_query, bool reverse_complement_target)
{
    // int * float, truncated -- as aligner_global_ukkonen.cpp:53-54
    const int32_t allocated_max_length_difference = static_cast<int32_t>(get_max_target_length() * max_target_query_length_difference);
    if (std::abs(query_length - target_length) > allocated_max_length_difference)
    {
        GW_LOG_DEBUG(("Exceeded maximum length difference b/w target and query allowed : " + std::to_string(allocated_max_length_difference)).c_str());
        return StatusType::exceeded_max_alignment_difference;
    }
    return AlignerGlobal::add_alignment(query, query_length, target, target_length, reverse_complement_query, reverse_complement_target);
}

size_t AlignerGlobalUkkonen::workspace_bytes(int32_t n, const int64_t* sequence_starts) const
{
    return gwhip_ukkonen_workspace_bytes(n, sequence_starts, ukkonen_p_);
}

int AlignerGlobalUkkonen::run_alignment(int32_t n, const char* sequences_d, const int64_t* sequence_starts_d,
                                        const int64_t* sequence_starts_h, int8_t* results_d, int32_t* result_lengths_d,
                                        void* workspace_d, size_t workspace_size)
{
    gwhip_ukkonen_args a{};
    for (int32_t i = 0; i < n; ++i) // aligner_global_ukkonen.cpp:66-72
    {
        const int32_t q = static_cast<int32_t>(sequence_starts_h[2 * i + 1] - sequence_starts_h[2 * i]);
        const int32_t t = static_cast<int32_t>(sequence_starts_h[2 * i + 2] - sequence_starts_h[2 * i + 1]);
        a.max_length_difference = std::max(a.max_length_difference, std::abs(q - t));
        a.max_sequence_length   = std::max(a.max_sequence_length, std::max(q, t));
    }
    a.n_alignments    = n;
    a.sequences       = sequences_d;
    a.sequence_starts = sequence_starts_d;
    a.ukkonen_p       = ukkonen_p_;
    a.results         = results_d;
    a.result_lengths  = result_lengths_d;
    a.workspace       = workspace_d;
    a.workspace_bytes = workspace_size;
    return gwhip_ukkonen(&a, get_stream());
}

// ---- full-matrix Myers ------------------------------------------------------------------------------------------
namespace
{
// wider than any query: add_alignment clamps it to the query length (aligner_global_myers_banded.cpp:174-178), which
// makes the kernel take its whole-matrix path; a multiple of 32, so the "% 32 == 1" restriction never triggers
constexpr int32_t covers_every_query = 1 << 30;
} // namespace

AlignerGlobalMyers::AlignerGlobalMyers(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments,
                                       DefaultDeviceAllocator allocator, cudaStream_t stream, int32_t device_id)
    : BandedAligner(-1, covers_every_query, allocator, stream, device_id, true,
                    throw_on_negative(max_query_length, "max_query_length must be non-negative."),
                    throw_on_negative(max_target_length, "max_target_length must be non-negative."),
                    throw_on_negative(max_alignments, "max_alignments must be non-negative."))
{
    if (max_alignments < 1) throw std::runtime_error("Max alignments must be at least 1.");
}

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
