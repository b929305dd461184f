// alignment.hpp -- result of one pairwise alignment (source-compatible with the reference's
// cudaaligner/alignment.hpp:37-112).
#pragma once

#include <claraparabricks/genomeworks/cudaaligner/cudaaligner.hpp>

#include <cstdint>
#include <memory>
#include <ostream>
#include <string>
#include <vector>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudaaligner
{

/// Three-line rendering of an alignment ("|" match, "x" mismatch, " " gap).
typedef struct FormattedAlignment
{
    std::string query;
    std::string pairing;
    std::string target;
    uint32_t linebreak_after = 80; ///< 0 = no line breaks
} FormattedAlignment;

std::ostream& operator<<(std::ostream& os, const FormattedAlignment& formatted_alignment);

class Alignment
{
public:
    virtual ~Alignment() = default;
    virtual const std::string& get_query_sequence() const                           = 0;
    virtual const std::string& get_target_sequence() const                          = 0;
    virtual std::string convert_to_cigar(CigarFormat format = CigarFormat::basic) const = 0;
    virtual AlignmentType get_alignment_type() const                                = 0;
    /// false when the band was clipped by max_bandwidth / memory and the result may be sub-optimal
    virtual bool is_optimal() const                                                 = 0;
    /// `uninitialized` when the aligner produced no result for this pair
    virtual StatusType get_status() const                                           = 0;
    /// per-position states (filled by the fixed-stride aligners)
    virtual const std::vector<AlignmentState>& get_alignment() const                = 0;
    /// run-length encoded form (filled by the banded aligner)
    virtual const std::vector<int8_t>& get_actions() const                          = 0;
    virtual const std::vector<int32_t>& get_runlengths() const                      = 0;
    virtual int32_t get_edit_distance() const                                       = 0;
    virtual FormattedAlignment format_alignment(int32_t maximal_line_length = 80) const = 0;
};

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
