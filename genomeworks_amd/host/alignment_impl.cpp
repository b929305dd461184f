// alignment_impl.cpp -- CIGAR / edit distance / pretty printing of an alignment.
// Output formats follow the reference (cudaaligner/src/alignment_impl.cpp:30-278, alignment.cpp:30-43):
// basic CIGAR merges match+mismatch into M; extended uses = and X; the run-length form takes precedence when
// present; format_alignment renders the per-position form only.
#include "alignment_impl.hpp"

#include <claraparabricks/genomeworks/utils/signed_integer_utils.hpp>

#include <algorithm>
#include <mutex>
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
FormattedAlignment format_states(const std::string& query, const std::string& target, const std::vector<AlignmentState>& states,
                                 int32_t maximal_line_length)
{
    FormattedAlignment out;
    out.linebreak_after = maximal_line_length < 0 ? 0 : static_cast<uint32_t>(maximal_line_length);
    int64_t t = 0, q = 0;
    for (AlignmentState s : states)
    {
        switch (s)
        {
        case AlignmentState::match:
        case AlignmentState::mismatch:
            out.target += target[t++];
            out.query += query[q++];
            out.pairing += (s == AlignmentState::match ? '|' : 'x');
            break;
        case AlignmentState::deletion:
            out.target += '-';
            out.query += query[q++];
            out.pairing += ' ';
            break;
        case AlignmentState::insertion:
            out.target += target[t++];
            out.query += '-';
            out.pairing += ' ';
            break;
        default: throw std::runtime_error("Unknown alignment state");
        }
    }
    return out;
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
    return format_states(query_, target_, alignment_, maximal_line_length);
}

// ---- PackedAlignment: a view into the block one sync_alignments() produced -------------------------------------
PackedAlignmentBlock::~PackedAlignmentBlock()
{
    if (alignments != nullptr)
    {
        for (size_t i = 0; i < n_alignments; ++i) alignments[i].~PackedAlignment();
        host_release(reinterpret_cast<char*>(alignments), alignments_bytes);
    }
    if (pinned != nullptr) pinned_release(pinned, pinned_bytes);
    if (sequences_buffer != nullptr) pinned_release(sequences_buffer, sequences_bytes);
    if (seq_starts_buffer != nullptr) pinned_release(seq_starts_buffer, seq_starts_bytes);
    if (head_buffer != nullptr) pinned_release(head_buffer, head_bytes);
}

void PackedAlignmentBlock::allocate_views(size_t n)
{
    // from the process-wide cache of host buffers: recycled storage is already mapped (a fresh 32-MB allocation per sync pays
    // for its page faults on the binder threads)
    alignments   = reinterpret_cast<PackedAlignment*>(host_acquire(std::max<size_t>(n, 1) * sizeof(PackedAlignment), &alignments_bytes));
    n_alignments = 0; // raised by the caller once the views are constructed
}

PackedAlignment::~PackedAlignment()
{
    delete lazy_.load(std::memory_order_acquire);
}

PackedAlignment::Lazy& PackedAlignment::lazy() const
{
    Lazy* l = lazy_.load(std::memory_order_acquire);
    if (l == nullptr)
    {
        Lazy* fresh = new Lazy;
        if (lazy_.compare_exchange_strong(l, fresh, std::memory_order_acq_rel))
            l = fresh;
        else
            delete fresh; // another thread was first; l now holds its object
    }
    return *l;
}

void PackedAlignment::materialise_sequences() const
{
    Lazy& l = lazy();
    std::call_once(l.seq_once, [&] {
        const int64_t* st = block_->seq_starts + 2 * static_cast<size_t>(index_);
        l.query.assign(block_->sequences + st[0], block_->sequences + st[1]);
        l.target.assign(block_->sequences + st[1], block_->sequences + st[2]);
    });
}

void PackedAlignment::materialise_runs() const
{
    Lazy& l = lazy();
    std::call_once(l.runs_once, [&] {
        const int32_t n = num_runs();
        if (block_->expand_states)
        {
            size_t total = 0;
            for (int32_t k = 0; k < n; ++k) total += static_cast<size_t>(count(k));
            l.alignment.reserve(total);
            for (int32_t k = 0; k < n; ++k) l.alignment.insert(l.alignment.end(), static_cast<size_t>(count(k)), static_cast<AlignmentState>(op(k)));
        }
        else
        {
            l.action.resize(static_cast<size_t>(n));
            l.runlength.resize(static_cast<size_t>(n));
            for (int32_t k = 0; k < n; ++k)
            {
                l.action[static_cast<size_t>(k)]    = op(k);
                l.runlength[static_cast<size_t>(k)] = count(k);
            }
        }
    });
}

const std::string& PackedAlignment::get_query_sequence() const
{
    materialise_sequences();
    return lazy().query;
}

const std::string& PackedAlignment::get_target_sequence() const
{
    materialise_sequences();
    return lazy().target;
}

const std::vector<AlignmentState>& PackedAlignment::get_alignment() const
{
    materialise_runs();
    return lazy().alignment;
}

const std::vector<int8_t>& PackedAlignment::get_actions() const
{
    materialise_runs();
    return lazy().action;
}

const std::vector<int32_t>& PackedAlignment::get_runlengths() const
{
    materialise_runs();
    return lazy().runlength;
}

std::string PackedAlignment::convert_to_cigar(CigarFormat format) const
{
    const bool ext = (format == CigarFormat::extended);
    if (block_->expand_states)
    {
        // per-position form: equal symbols merge in both formats (AlignmentImpl::convert_to_cigar on alignment_)
        return build_cigar(
            num_runs(), [&](int32_t i) { return cigar_symbol(op(i), ext); }, [&](int32_t i) { return static_cast<int64_t>(count(i)); }, true);
    }
    return build_cigar(
        num_runs(), [&](int32_t i) { return cigar_symbol(op(i), ext); }, [&](int32_t i) { return static_cast<int64_t>(count(i)); }, !ext);
}

int32_t PackedAlignment::get_edit_distance() const
{
    int32_t d = 0;
    for (int32_t k = 0; k < num_runs(); ++k)
        if (op(k) != static_cast<int8_t>(AlignmentState::match)) d += count(k);
    return d;
}

FormattedAlignment PackedAlignment::format_alignment(int32_t maximal_line_length) const
{
    materialise_sequences();
    materialise_runs();
    const Lazy& l = lazy();
    return format_states(l.query, l.target, l.alignment, maximal_line_length); // renders the per-position form only, like AlignmentImpl
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
