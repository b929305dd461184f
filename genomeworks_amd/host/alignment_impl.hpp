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
    PackedAlignment* alignments = nullptr;   ///< [n_alignments] by index of add_alignment (raw storage, see allocate_views)
    size_t n_alignments         = 0;
    bool expand_states          = false;     ///< AlignerGlobalMyers: per-position states instead of run lengths
    void allocate_views(size_t n);           ///< uninitialised storage for n views (constructed by the binder threads)
};

class PackedAlignment final : public Alignment
{
public:
    PackedAlignment(const PackedAlignmentBlock* block, int32_t index, int32_t run_begin, int32_t run_end, bool has_result, bool is_optimal)
        : block_(block)
        , index_(index)
        , run_begin_(run_begin)
        , run_end_(run_end)
        , has_result_(has_result)
        , is_optimal_(has_result && is_optimal)
    {
    }
    ~PackedAlignment() override;

    const std::string& get_query_sequence() const override;
    const std::string& get_target_sequence() const override;
    std::string convert_to_cigar(CigarFormat format = CigarFormat::basic) const override;
    AlignmentType get_alignment_type() const override { return AlignmentType::global_alignment; }
    bool is_optimal() const override { return is_optimal_; }
    StatusType get_status() const override { return has_result_ ? StatusType::success : StatusType::uninitialized; }
    const std::vector<AlignmentState>& get_alignment() const override;
    const std::vector<int8_t>& get_actions() const override;
    const std::vector<int32_t>& get_runlengths() const override;
    int32_t get_edit_distance() const override;
    FormattedAlignment format_alignment(int32_t maximal_line_length = 80) const override;

    int32_t num_runs() const { return run_end_ - run_begin_; }
    /// run k in forward order (the device stores an alignment back to front)
    int8_t op(int32_t k) const { return block_->ops[run_end_ - 1 - k]; }
    int32_t count(int32_t k) const { return block_->counts[run_end_ - 1 - k]; }
    bool materialised() const { return lazy_.load(std::memory_order_acquire) != nullptr; }

private:
    /// what the accessors that return references need to own; built on first use, one allocation per alignment that is
    /// actually inspected (a view itself is 40 bytes: a million-pair batch costs 40 MB, not a million string pairs)
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
    int32_t index_, run_begin_, run_end_;
    bool has_result_, is_optimal_;
    mutable std::atomic<Lazy*> lazy_{nullptr};
};

/// pinned host staging buffers, recycled process-wide (hipHostMalloc / hipHostFree cost far more than the copies
/// they serve); thread-safe
char* pinned_acquire(size_t bytes, size_t* capacity);
void pinned_release(char* p, size_t capacity);

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
