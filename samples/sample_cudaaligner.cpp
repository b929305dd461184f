// sample_cudaaligner.cpp -- banded global alignment of a few pairs through the C++ Aligner interface (counterpart of
// cudaaligner/samples/sample_cudaaligner.cpp). Build like samples/sample_cudapoa.cpp.
#include <claraparabricks/genomeworks/cudaaligner/aligner.hpp>
#include <claraparabricks/genomeworks/cudaaligner/alignment.hpp>

#include <iostream>
#include <memory>
#include <string>
#include <utility>
#include <vector>

using namespace claraparabricks::genomeworks;
using namespace claraparabricks::genomeworks::cudaaligner;

int main()
{
    const std::vector<std::pair<std::string, std::string>> pairs = {
        {"ACGTTGCAACGTACGTTAGC", "ACGTTGCATCGTACGTTAGC"}, {"TTGACCATTGACCA", "TTGACATTGACCAA"}, {"AAAA", "AAAT"}};
    std::unique_ptr<Aligner> aligner = create_aligner(AlignmentType::global_alignment, /*max_bandwidth*/ 64, nullptr, 0);
    for (const auto& p : pairs)
        if (aligner->add_alignment(p.first.c_str(), static_cast<int32_t>(p.first.size()), p.second.c_str(),
                                   static_cast<int32_t>(p.second.size())) != StatusType::success)
            std::cerr << "pair not added" << std::endl;
    aligner->align_all();
    aligner->sync_alignments();
    for (const auto& a : aligner->get_alignments())
    {
        const FormattedAlignment f = a->format_alignment();
        std::cout << f.query << "\n" << f.pairing << "\n" << f.target << "\ncigar " << a->convert_to_cigar() << "\n\n";
    }
    return 0;
}
