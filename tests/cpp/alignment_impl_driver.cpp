// alignment_impl_driver.cpp -- test driver for the host Alignment classes (AlignmentImpl and the PackedAlignment views
// of sync_alignments()). Reads cases from stdin, one per line: "<query> <target> <states as digits 0-3> <is_optimal 0|1>"
// ('-' for an empty string) and prints what each class reports, tab separated. Compiled and run by
// tests/test_alignment_impl.py against the vectors of the reference's Test_AlignmentImpl.cpp:36-204. No device calls.
#include <iostream>
#include <memory>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "alignment_impl.hpp"

using namespace claraparabricks::genomeworks::cudaaligner;

static void report(const char* cls, const Alignment& a)
{
    const FormattedAlignment f = a.format_alignment();
    std::string states;
    for (AlignmentState s : a.get_alignment()) states += static_cast<char>('0' + static_cast<int>(s));
    std::ostringstream wrapped;
    wrapped << a.format_alignment(3);
    std::string w = wrapped.str();
    for (char& c : w)
        if (c == '\n') c = '/';
    std::cout << cls << '\t' << a.get_query_sequence() << '\t' << a.get_target_sequence() << '\t' << states << '\t'
              << (a.is_optimal() ? 1 : 0) << '\t' << static_cast<int>(a.get_status()) << '\t' << static_cast<int>(a.get_alignment_type())
              << '\t' << a.convert_to_cigar(CigarFormat::basic) << '\t' << a.convert_to_cigar(CigarFormat::extended) << '\t'
              << a.get_edit_distance() << '\t' << f.query << '\t' << f.pairing << '\t' << f.target << '\t' << w << '\n';
}

int main()
{
    {
        // TestAlignmentImplIndividual.Status / .Type
        AlignmentImpl a("A", 1, "T", 1);
        std::cout << "initial\t" << static_cast<int>(a.get_status()) << '\t' << static_cast<int>(a.get_alignment_type()) << '\n';
        a.set_status(StatusType::success);
        a.set_alignment_type(AlignmentType::global_alignment);
        std::cout << "after_set\t" << static_cast<int>(a.get_status()) << '\t' << static_cast<int>(a.get_alignment_type()) << '\n';
    }
    std::string line;
    while (std::getline(std::cin, line))
    {
        std::istringstream in(line);
        std::string q, t, digits;
        int optimal = 0;
        if (!(in >> q >> t >> digits >> optimal)) continue;
        if (q == "-") q.clear();
        if (t == "-") t.clear();
        if (digits == "-") digits.clear();
        std::vector<AlignmentState> states;
        for (char c : digits) states.push_back(static_cast<AlignmentState>(c - '0'));
        AlignmentImpl a(q.c_str(), static_cast<int32_t>(q.size()), t.c_str(), static_cast<int32_t>(t.size()));
        a.set_alignment(states, optimal != 0);
        a.set_status(StatusType::success);
        a.set_alignment_type(AlignmentType::global_alignment);
        report("AlignmentImpl", a);

        // the same alignment as the banded aligner hands it out: run-length encoded, back to front, a view into a block
        for (int expand = 0; expand < 2; ++expand)
        {
            auto block           = std::make_shared<PackedAlignmentBlock>();
            block->expand_states = expand != 0;
            block->sequences_owned.assign(q.begin(), q.end());
            block->sequences_owned.insert(block->sequences_owned.end(), t.begin(), t.end());
            block->seq_starts_owned = {0, static_cast<int64_t>(q.size()), static_cast<int64_t>(q.size() + t.size())};
            block->sequences        = block->sequences_owned.data();
            block->seq_starts       = block->seq_starts_owned.data();
            std::vector<int8_t> ops;
            std::vector<int32_t> counts;
            for (size_t i = states.size(); i-- > 0;)
            {
                const int8_t s = static_cast<int8_t>(states[i]);
                if (!ops.empty() && ops.back() == s) counts.back()++;
                else { ops.push_back(s); counts.push_back(1); }
            }
            block->ops        = ops.data();
            block->counts     = counts.data();
            block->run_starts_owned = {0, static_cast<int32_t>(ops.size())};
            block->metadata_owned   = {optimal != 0 ? 0x80000000u : 0u};
            block->run_starts       = block->run_starts_owned.data();
            block->metadata         = block->metadata_owned.data();
            block->allocate_views(1);
            new (&block->alignments[0]) PackedAlignment(block.get(), 0);
            block->n_alignments = 1;
            std::shared_ptr<Alignment> view(block, &block->alignments[0]);
            report(expand ? "PackedAlignment(states)" : "PackedAlignment(runs)", *view);
            if (!expand)
            {
                std::string runs;
                for (size_t k = 0; k < view->get_actions().size(); ++k)
                    runs += std::to_string(view->get_runlengths()[k]) + "x" + std::to_string(static_cast<int>(view->get_actions()[k])) + ",";
                std::cout << "runs\t" << runs << '\n';
            }
        }
    }
    return 0;
}
