// align_overlaps_main.cpp -- `align_overlaps`: the `-a` (alignment) stage of cudamapper as a stand-alone consumer of
// cudaaligner: overlaps come in as PAF, the overlapped regions are aligned globally, PAF with `cg:Z:` CIGARs goes
// to stdout. Reference: cudamapper/src/main.cu:54-187 (align_overlaps / run_alignment_batch) and utils.cpp:41-124
// (print_paf); the option names -a / -m follow cudamapper's application_parameters.cpp.
#include <claraparabricks/genomeworks/cudamapper/overlap_alignment.hpp>
#include <claraparabricks/genomeworks/cudaaligner/cudaaligner.hpp>
#include <claraparabricks/genomeworks/utils/allocator.hpp>

#include <getopt.h>

#include <cstdlib>
#include <iostream>
#include <stdexcept>

using namespace claraparabricks::genomeworks;

namespace
{

[[noreturn]] void usage(int exit_code)
{
    std::cerr << "Usage: align_overlaps [options ...] <query.fasta> <target.fasta> <overlaps.paf>\n"
                 "     options:\n"
                 "        -a, --alignment-engines <int>\n"
                 "            number of alignment engines (host threads, each with its own stream and aligner) [1]\n"
                 "        -b, --batch-size <int>\n"
                 "            alignments per aligner batch [derived from the free device memory]\n"
                 "        -m, --max-cached-memory <int>\n"
                 "            device memory to preallocate, in GiB [2]\n"
                 "        -k, --kmer-size <int>\n"
                 "            factor applied to the residue-match column, as cudamapper prints it [1]\n"
                 "        -S\n"
                 "            print overlaps in SAM format (text; cudamapper's -S needs htslib, this writer does not)\n"
                 "        -h, --help\n";
    std::exit(exit_code);
}

} // namespace

int main(int argc, char** argv)
{
    int32_t engines = 1, batch_size = 0, kmer_size = 1;
    bool sam = false;
    int64_t cached_gib = 2;
    static const option long_options[] = {{"alignment-engines", required_argument, nullptr, 'a'},
                                          {"batch-size", required_argument, nullptr, 'b'},
                                          {"max-cached-memory", required_argument, nullptr, 'm'},
                                          {"kmer-size", required_argument, nullptr, 'k'},
                                          {"help", no_argument, nullptr, 'h'},
                                          {nullptr, 0, nullptr, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "a:b:m:k:Sh", long_options, nullptr)) != -1)
    {
        switch (c)
        {
        case 'a': engines = std::atoi(optarg); break;
        case 'b': batch_size = std::atoi(optarg); break;
        case 'm': cached_gib = std::atoll(optarg); break;
        case 'k': kmer_size = std::atoi(optarg); break;
        case 'S': sam = true; break;
        case 'h': usage(0);
        default: usage(1);
        }
    }
    if (argc - optind != 3 || engines < 1 || cached_gib < 1 || kmer_size < 1)
    {
        usage(1);
    }
    try
    {
        const std::vector<cudamapper::FastaSequence> queries = cudamapper::read_fasta(argv[optind]);
        const std::vector<cudamapper::FastaSequence> targets = cudamapper::read_fasta(argv[optind + 1]);
        const std::vector<cudamapper::Overlap> overlaps      = cudamapper::read_paf(argv[optind + 2], queries, targets);
        if (cudaaligner::Init() != cudaaligner::success) throw std::runtime_error("cudaaligner::Init failed");
        DefaultDeviceAllocator allocator = create_default_device_allocator(static_cast<std::size_t>(cached_gib) << 30);
        std::vector<std::string> cigars;
        cudamapper::align_overlaps(allocator, overlaps, queries, targets, engines, cigars, batch_size);
        if (sam)
        {
            std::string command_line;
            for (int k = 0; k < argc; ++k) command_line += (k ? " " : "") + std::string(argv[k]);
            cudamapper::print_sam(overlaps, cigars, queries, targets, "0.6.0-mi355x", command_line, stdout);
        }
        else
            cudamapper::print_paf(overlaps, cigars, queries, targets, kmer_size, stdout);
    }
    catch (const std::exception& e)
    {
        std::cerr << "align_overlaps: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
