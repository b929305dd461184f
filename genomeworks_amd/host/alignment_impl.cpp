// alignment_impl.cpp -- CIGAR / edit distance / pretty printing of an alignment.
// Output formats follow the reference (cudaaligner/src/alignment_impl.cpp:30-278, alignment.cpp:30-43):
// basic CIGAR merges match+mismatch into M; extended uses = and X; the run-length form takes precedence when
// present; format_alignment renders the per-position form only.
#include "alignment_impl.hpp"

#include <claraparabricks/genomeworks/utils/signed_integer_utils.hpp>

#include <algorithm>
#include <stdexcept>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudaaligner
{

namespace
{
char cigar_symbol(int8_t s, bool extended)
{
    switch (s)
    {
    case AlignmentState::match: return extended ? '=' : 'M';
    case AlignmentState::mismatch: return extended ? 'X' : 'M';
    case AlignmentState::insertion: return 'I';
    case AlignmentState::deletion: return 'D';
    default: return '!';
    }
}

// runs of equal symbols -> "<count><symbol>"...
template <typename SymbolAt, typename CountAt>
std::string build_cigar(int32_t n, SymbolAt symbol_at, CountAt count_at, bool merge_equal)
{
    std::string cigar;
    if (n < 1) return cigar;
    cigar.reserve(3 * static_cast<size_t>(n));
    char last     = symbol_at(0);
    int64_t count = count_at(0);
    for (int32_t i = 1; i < n; ++i)
    {
        const char c = symbol_at(i);
        if (merge_equal && c == last)
            count += count_at(i);
        else
        {
            cigar += std::to_string(count);
            cigar += last;
            last  = c;
            count = count_at(i);
        }
    }
    cigar += std::to_string(count);
    cigar += last;
    return cigar;
}
} // namespace

AlignmentImpl::AlignmentImpl(const char* query, int32_t query_length, const char* target, int32_t target_length)
    : query_(query, query + throw_on_negative(query_length, "query_length has to be non-negative."))
    , target_(target, target + throw_on_negative(target_length, "target_length has to be non-negative."))
{
}

std::string AlignmentImpl::convert_to_cigar(CigarFormat format) const
{
    const bool ext = (format == CigarFormat::extended);
    if (!action_.empty())
    {
        // the extended form of a run-length result prints every run as is; the basic form merges M runs
        return build_cigar(
            get_size<int32_t>(action_), [&](int32_t i) { return cigar_symbol(action_[i], ext); },
            [&](int32_t i) { return static_cast<int64_t>(runlength_[i]); }, !ext);
    }
    return build_cigar(
        get_size<int32_t>(alignment_), [&](int32_t i) { return cigar_symbol(alignment_[i], ext); }, [](int32_t) { return int64_t(1); }, true);
}

int32_t AlignmentImpl::get_edit_distance() const
{
    if (!action_.empty())
    {
        int32_t d = 0;
        for (size_t i = 0; i < action_.size(); ++i)
            if (action_[i] != static_cast<int8_t>(AlignmentState::match)) d += runlength_[i];
        return d;
    }
    return static_cast<int32_t>(std::count_if(alignment_.begin(), alignment_.end(), [](AlignmentState s) { return s != AlignmentState::match; }));
}

FormattedAlignment AlignmentImpl::format_alignment(int32_t maximal_line_length) const
{
    FormattedAlignment out;
    out.linebreak_after = maximal_line_length < 0 ? 0 : static_cast<uint32_t>(maximal_line_length);
    int64_t t = 0, q = 0;
    for (AlignmentState s : alignment_)
    {
        switch (s)
        {
        case AlignmentState::match:
        case AlignmentState::mismatch:
            out.target += target_[t++];
            out.query += query_[q++];
            out.pairing += (s == AlignmentState::match ? '|' : 'x');
            break;
        case AlignmentState::deletion:
            out.target += '-';
            out.query += query_[q++];
            out.pairing += ' ';
            break;
        case AlignmentState::insertion:
            out.target += target_[t++];
            out.query += '-';
            out.pairing += ' ';
            break;
        default: throw std::runtime_error("Unknown alignment state");
        }
    }
    return out;
}

std::ostream& operator<<(std::ostream& os, const FormattedAlignment& f)
{
    const std::size_t line = (f.linebreak_after == 0) ? f.query.size() : f.linebreak_after;
    for (std::size_t i = 0; i < f.query.size(); i += line)
        os << f.query.substr(i, line) << '\n' << f.pairing.substr(i, line) << '\n' << f.target.substr(i, line) << '\n';
    os << std::endl;
    return os;
}

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
