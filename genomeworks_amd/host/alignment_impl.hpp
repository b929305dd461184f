// alignment_impl.hpp -- concrete cudaaligner::Alignment (behaviour of the reference's alignment_impl.cpp:30-278).
#pragma once
#include <claraparabricks/genomeworks/cudaaligner/alignment.hpp>

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

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
