// alignment_impl.hpp -- concrete cudaaligner::Alignment (behaviour of the reference's alignment_impl.cpp:30-278).
#pragma once
#include <claraparabricks/genomeworks/cudaaligner/alignment.hpp>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudaaligner
{

class AlignmentImpl : public Alignment
{
public:
    AlignmentImpl(const char* query, int32_t query_length, const char* target, int32_t target_length);

    const std::string& get_query_sequence() const override { return query_; }
    const std::string& get_target_sequence() const override { return target_; }
    std::string convert_to_cigar(CigarFormat format = CigarFormat::basic) const override;
    AlignmentType get_alignment_type() const override { return type_; }
    bool is_optimal() const override { return is_optimal_; }
    StatusType get_status() const override { return status_; }
    const std::vector<AlignmentState>& get_alignment() const override { return alignment_; }
    const std::vector<int8_t>& get_actions() const override { return action_; }
    const std::vector<int32_t>& get_runlengths() const override { return runlength_; }
    int32_t get_edit_distance() const override;
    FormattedAlignment format_alignment(int32_t maximal_line_length = 80) const override;

    void set_alignment_type(AlignmentType type) { type_ = type; }
    void set_status(StatusType status) { status_ = status; }
    /// per-position form
    void set_alignment(const std::vector<AlignmentState>& alignment, bool is_optimal)
    {
        alignment_  = alignment;
        is_optimal_ = is_optimal;
    }
    void set_alignment(std::vector<AlignmentState>&& alignment, bool is_optimal)
    {
        alignment_  = std::move(alignment);
        is_optimal_ = is_optimal;
    }
    /// run-length encoded form
    void set_alignment(std::vector<int8_t>&& action, std::vector<int32_t>&& runlength, bool is_optimal)
    {
        action_     = std::move(action);
        runlength_  = std::move(runlength);
        is_optimal_ = is_optimal;
    }

private:
    std::string query_;
    std::string target_;
    StatusType status_  = StatusType::uninitialized;
    AlignmentType type_ = AlignmentType::unset;
    std::vector<AlignmentState> alignment_;
    std::vector<int8_t> action_;
    std::vector<int32_t> runlength_;
    bool is_optimal_ = false;
};

/// Results of one sync_alignments() of the banded aligner, kept the way they left the device: the sequences of the
/// batch back to back, the packed run-length results (each alignment back to front) and their offsets. The
/// Alignment objects handed to the caller are views into this block (PackedAlignment) -- a batch of a million
/// short-read alignments is materialised by one D2H copy and one pass over two index arrays, not by a million
/// string / vector constructions (the reference builds every AlignmentImpl eagerly,
/// aligner_global_myers_banded.cpp:402-427, which is most of its sync time on short reads). Whatever a caller asks an
/// Alignment for is produced on first use from the block and is identical to what the eager construction held.
class PackedAlignment;
struct PackedAlignmentBlock
{
    ~PackedAlignmentBlock();
    /// q0 t0 q1 t1 ... and the [2n + 1] offsets into it: the batch's own (pinned) staging arrays, handed over by the
    /// aligner at sync time; returned to the pinned cache with the block
    const char* sequences     = nullptr;
    const int64_t* seq_starts = nullptr;
    char* sequences_buffer    = nullptr;
    size_t sequences_bytes    = 0;
    char* seq_starts_buffer   = nullptr;
    size_t seq_starts_bytes   = 0;
    std::vector<char> sequences_owned;      ///< tests / callers without pinned buffers
    std::vector<int64_t> seq_starts_owned;
    /// pinned host buffer [ops (total) | pad | counts (total x int32)] from the runtime's pinned cache
    char* pinned          = nullptr;
    size_t pinned_bytes   = 0;
    const int8_t* ops     = nullptr;
    const int32_t* counts = nullptr;
    /// [n + 1] offsets of the alignments' runs in ops / counts and [n] metadata words (bit 31 = optimal), by index of
    /// add_alignment: what the kernels wrote (DeviceAlignmentsPtrs::cigar_offsets / metadata). A view reads its range from
    /// these arrays when it is asked, so the views can be laid out while the kernels still run.
    const int32_t* run_starts = nullptr;
    const uint32_t* metadata  = nullptr;
    char* head_buffer         = nullptr;     ///< pinned [run_starts | metadata], handed over by the aligner
    size_t head_bytes         = 0;
    std::vector<int32_t> run_starts_owned;   ///< tests / callers without pinned buffers
    std::vector<uint32_t> metadata_owned;
    PackedAlignment* alignments = nullptr;   ///< [n_alignments] by index of add_alignment (raw storage, see allocate_views)
    size_t n_alignments         = 0;
    size_t alignments_bytes     = 0;         ///< capacity of the views' storage (from the host-buffer cache)
    bool expand_states          = false;     ///< AlignerGlobalMyers: per-position states instead of run lengths
    void allocate_views(size_t n);           ///< uninitialised storage for n views (constructed by the binder threads)
};

class PackedAlignment final : public Alignment
{
public:
    PackedAlignment(const PackedAlignmentBlock* block, int32_t index)
        : block_(block)
        , index_(index)
    {
    }
    ~PackedAlignment() override;

    const std::string& get_query_sequence() const override;
    const std::string& get_target_sequence() const override;
    std::string convert_to_cigar(CigarFormat format = CigarFormat::basic) const override;
    AlignmentType get_alignment_type() const override { return AlignmentType::global_alignment; }
    bool is_optimal() const override { return has_result() && (block_->metadata[index_] >> 31) != 0; }
    StatusType get_status() const override { return has_result() ? StatusType::success : StatusType::uninitialized; }
    const std::vector<AlignmentState>& get_alignment() const override;
    const std::vector<int8_t>& get_actions() const override;
    const std::vector<int32_t>& get_runlengths() const override;
    int32_t get_edit_distance() const override;
    FormattedAlignment format_alignment(int32_t maximal_line_length = 80) const override;

    int32_t run_begin() const { return block_->run_starts[index_]; }
    int32_t run_end() const { return block_->run_starts[index_ + 1]; }
    int32_t num_runs() const { return run_end() - run_begin(); }
    /// a pair the kernels reported nothing for has no runs -- unless both sequences are empty (an empty alignment is a result)
    bool has_result() const
    {
        const int64_t* st = block_->seq_starts + 2 * static_cast<size_t>(index_);
        return run_begin() != run_end() || (st[0] == st[1] && st[1] == st[2]);
    }
    /// run k in forward order (the device stores an alignment back to front)
    int8_t op(int32_t k) const { return block_->ops[run_end() - 1 - k]; }
    int32_t count(int32_t k) const { return block_->counts[run_end() - 1 - k]; }
    bool materialised() const { return lazy_.load(std::memory_order_acquire) != nullptr; }

private:
    /// what the accessors that return references need to own; built on first use, one allocation per alignment that is
    /// actually inspected (a view itself is 32 bytes: a million-pair batch costs 32 MB, not a million string pairs)
    struct Lazy
    {
        std::once_flag seq_once, runs_once;
        std::string query, target;
        std::vector<AlignmentState> alignment;
        std::vector<int8_t> action;
        std::vector<int32_t> runlength;
    };
    Lazy& lazy() const;
    void materialise_sequences() const;
    void materialise_runs() const;

    const PackedAlignmentBlock* block_;
    int32_t index_;
    mutable std::atomic<Lazy*> lazy_{nullptr};
};

/// pinned host staging buffers, recycled process-wide (hipHostMalloc / hipHostFree cost far more than the copies
/// they serve); thread-safe
char* pinned_acquire(size_t bytes, size_t* capacity);
void pinned_release(char* p, size_t capacity);
/// pageable host buffers of a megabyte and more, recycled process-wide (a recycled buffer is already mapped); thread-safe
char* host_acquire(size_t bytes, size_t* capacity);
void host_release(char* p, size_t capacity);

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
