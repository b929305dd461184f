// cudapoa_main.cpp -- the `cudapoa` command-line tool: consensus / MSA (and optionally the POA graphs) for every
// window of a cudapoa window file or of a set of FASTA files.
// Same options, outputs and messages as the reference tool (cudapoa/src/main.cpp:145-334,
// application_parameters.cpp:35-252): results go to stdout (one consensus per group, or the MSA rows),
// progress and errors to stderr.
#include <claraparabricks/genomeworks/cudapoa/batch.hpp>
#include <claraparabricks/genomeworks/cudapoa/utils.hpp>
#include <claraparabricks/genomeworks/utils/graph.hpp>
#include <claraparabricks/genomeworks/utils/signed_integer_utils.hpp>

#include <hip/hip_runtime_api.h>

#include <getopt.h>

#include <fstream>
#include <iostream>
#include <memory>
#include <stdexcept>

using namespace claraparabricks::genomeworks;
using namespace claraparabricks::genomeworks::cudapoa;

namespace
{

struct Options
{
    std::vector<std::string> input_paths;
    std::string graph_output_path;
    bool output_gfa           = false;
    bool all_fasta            = true;
    bool msa                  = false;
    BandMode band_mode        = BandMode::adaptive_band;
    int32_t band_width        = 256;
    int32_t max_groups        = -1;
    int32_t mismatch_score    = -6;
    int32_t gap_score         = -8;
    int32_t match_score       = 8;
    double gpu_mem_allocation = 0.9;
    float adaptive_storage    = 2.0f;
    float graph_length        = 3.0f;
    int32_t pred_distance     = 0;
};

[[noreturn]] void usage(int exit_code)
{
    std::cerr << "Usage: cudapoa [options ...]\n"
                 "     options:\n"
                 "        -i, --input <file>\n"
                 "            input in fasta/cudapoa format, can be used multiple times for multiple fasta files, but supports only one cudapoa file\n"
                 "        -a, --msa\n"
                 "            generates msa if this flag is passed [default: consensus]\n"
                 "        -b, --band-mode  <int>\n"
                 "            selects banding mode, 0: full-alignment, 1: static band, 2: adaptive band, 3: traceback static band, 4: traceback adaptive band [2]\n"
                 "        -w, --band-width <int>\n"
                 "            band-width for banded alignment (must be multiple of 128) [256]\n"
                 "        -s, --adaptive-storage  <float>\n"
                 "            factor to accommodate extra memory for adaptive score matrix. The factor represents ratio of adaptive-banded score matrix to static-banded score matrix [2.0]\n"
                 "        -l, --graph-length  <float>\n"
                 "            factor to determine maximum length of POA graph. The factor represents ratio of graph length to maximum sequence length in POA group [3.0]\n"
                 "        -D, --pred-distance <int>\n"
                 "            maximum distance of predecessor nodes that are considered in Needleman-Wunsch computations for static or adaptive-banded. If 0, it will be set equal to 2 x band-width [0]\n"
                 "        -d, --dot <file>\n"
                 "            output path for printing graph in DOT format [disabled]\n"
                 "        -G, --gfa <file>\n"
                 "            output path for printing graph in GFA format [disabled]\n"
                 "        -M, --max-groups  <int>\n"
                 "            maximum number of POA groups to create from file (-1 for all, > 0 for limited) [-1]\n"
                 "            repeats groups if less groups are present than specified\n"
                 "        -R, --gpu-mem-alloc <double>\n"
                 "            fraction of available GPU memory to be used for cudapoa [0.9]\n"
                 "        -m, --match  <int>\n"
                 "            score for matching bases (must be positive) [8]\n"
                 "        -n, --mismatch  <int>\n"
                 "            score for mismatching bases (must be non-positive) [-6]\n"
                 "        -g, --gap  <int>\n"
                 "            score for gaps (must be non-positive) [-8]\n"
                 "        -v, --version\n"
                 "            version information\n"
                 "        -h, --help\n"
                 "            prints usage"
              << std::endl;
    std::exit(exit_code);
}

Options parse_options(int argc, char* argv[])
{
    static const option long_options[] = {{"input", required_argument, nullptr, 'i'},
                                          {"msa", no_argument, nullptr, 'a'},
                                          {"band-mode", required_argument, nullptr, 'b'},
                                          {"band-width", required_argument, nullptr, 'w'},
                                          {"adaptive-storage", required_argument, nullptr, 's'},
                                          {"graph-length", required_argument, nullptr, 'l'},
                                          {"pred-distance", required_argument, nullptr, 'D'},
                                          {"dot", required_argument, nullptr, 'd'},
                                          {"gfa", required_argument, nullptr, 'G'},
                                          {"max-groups", required_argument, nullptr, 'M'},
                                          {"gpu-mem-alloc", required_argument, nullptr, 'R'},
                                          {"match", required_argument, nullptr, 'm'},
                                          {"mismatch", required_argument, nullptr, 'n'},
                                          {"gap", required_argument, nullptr, 'g'},
                                          {"version", no_argument, nullptr, 'v'},
                                          {"help", no_argument, nullptr, 'h'},
                                          {nullptr, 0, nullptr, 0}};
    Options o;
    int c;
    while ((c = getopt_long(argc, argv, "i:ab:w:s:l:D:d:G:M:R:m:n:g:vh", long_options, nullptr)) != -1)
    {
        switch (c)
        {
        case 'i': o.input_paths.emplace_back(optarg); break;
        case 'a': o.msa = true; break;
        case 'b':
        {
            const int mode = std::stoi(optarg);
            if (mode < 0 || mode > 4)
                throw std::runtime_error("band-mode must be either 0 for full bands, 1 for static bands, 2 for adaptive bands, 3 and 4 for static and adaptive bands with traceback");
            o.band_mode = static_cast<BandMode>(mode);
            break;
        }
        case 'w': o.band_width = std::stoi(optarg); break;
        case 's': o.adaptive_storage = std::stof(optarg); break;
        case 'l': o.graph_length = std::stof(optarg); break;
        case 'D':
            if (std::stoi(optarg) <= 0) throw std::runtime_error("pred-distance must be an integer greater than 0");
            o.pred_distance = std::stoi(optarg);
            break;
        case 'd': o.graph_output_path = optarg; break;
        case 'G':
            o.graph_output_path = optarg;
            o.output_gfa        = true;
            break;
        case 'M': o.max_groups = std::stoi(optarg); break;
        case 'R': o.gpu_mem_allocation = std::stod(optarg); break;
        case 'm': o.match_score = std::stoi(optarg); break;
        case 'n': o.mismatch_score = std::stoi(optarg); break;
        case 'g': o.gap_score = std::stoi(optarg); break;
        case 'v': std::cerr << "genomeworks_amd cudapoa (gfx950), GenomeWorks-compatible" << std::endl; std::exit(1);
        case 'h': usage(0);
        default: std::exit(1);
        }
    }
    if (o.gpu_mem_allocation <= 0 || o.gpu_mem_allocation > 1.0)
        throw std::runtime_error("gpu-mem-alloc must be greater than 0 and less than or equal to 1.0");
    if (o.band_mode != BandMode::adaptive_band && o.band_width < 1) throw std::runtime_error("band-width must be positive");
    if (o.match_score < 0) throw std::runtime_error("match score must be positive");
    if (o.max_groups == 0) throw std::runtime_error("max-groups cannot be 0");
    if (o.mismatch_score > 0) throw std::runtime_error("mismatch score must be non-positive");
    if (o.gap_score > 0) throw std::runtime_error("gap score must be non-positive");

    // every input is FASTA, or there is exactly one file (FASTA or cudapoa window format)
    for (const std::string& path : o.input_paths)
    {
        std::ifstream in(path);
        if (!in.good()) throw std::runtime_error("Invalid input file: " + path);
        std::string first;
        std::getline(in, first);
        if (first.empty() || first[0] != '>') o.all_fasta = false;
    }
    if (o.input_paths.empty() || (!o.all_fasta && o.input_paths.size() > 1))
    {
        std::cerr << "Invalid input. cudapoa needs input in either one cudapoa format file or in one/multiple fasta files." << std::endl;
        usage(1);
    }
    return o;
}

void report(StatusType status, const std::string& what)
{
    std::string message, hint;
    decode_error(status, message, hint);
    std::cerr << what << std::endl;
    std::cerr << message << std::endl << hint << std::endl;
}

// run the batch and print its results; group_ids[offset + g] is the input index of the g-th POA of the batch
void process_batch(Batch& batch, bool msa, const std::vector<int32_t>& group_ids, int32_t offset)
{
    batch.generate_poa();
    std::vector<StatusType> output_status;
    if (msa)
    {
        std::vector<std::vector<std::string>> rows;
        const StatusType status = batch.get_msa(rows, output_status);
        if (status != StatusType::success) report(status, "Could not generate MSA for batch : ");
        for (int32_t g = 0; g < get_size<int32_t>(rows); g++)
        {
            if (output_status[g] != StatusType::success)
                report(output_status[g], "Error generating  MSA for POA group " + std::to_string(group_ids[g + offset]));
            else
                for (const std::string& row : rows[g]) std::cout << row << std::endl;
        }
    }
    else
    {
        std::vector<std::string> consensus;
        std::vector<std::vector<uint16_t>> coverage;
        const StatusType status = batch.get_consensus(consensus, coverage, output_status);
        if (status != StatusType::success) report(status, "Could not generate consensus for batch : ");
        for (int32_t g = 0; g < get_size<int32_t>(consensus); g++)
        {
            if (output_status[g] != StatusType::success)
                report(output_status[g], "Error generating  consensus for POA group " + std::to_string(group_ids[g + offset]));
            else
                std::cout << consensus[g] << std::endl;
        }
    }
}

int run(int argc, char* argv[])
{
    const Options opt = parse_options(argc, argv);

    std::vector<std::vector<std::string>> windows;
    if (opt.all_fasta)
        parse_fasta_files(windows, opt.input_paths, opt.max_groups);
    else
        parse_cudapoa_file(windows, opt.input_paths[0], opt.max_groups);

    std::ofstream graph_output;
    if (!opt.graph_output_path.empty())
    {
        graph_output.open(opt.graph_output_path);
        if (!graph_output)
        {
            std::cerr << "Error opening " << opt.graph_output_path << " for graph output" << std::endl;
            return -1;
        }
    }

    std::vector<Group> poa_groups(windows.size());
    for (size_t i = 0; i < windows.size(); i++)
        for (const std::string& seq : windows[i]) poa_groups[i].push_back(Entry{seq.c_str(), nullptr, static_cast<int32_t>(seq.length())});

    // the fewest batch shapes that cover all groups
    std::vector<BatchConfig> batch_shapes;
    std::vector<std::vector<int32_t>> groups_per_batch;
    get_multi_batch_sizes(batch_shapes, groups_per_batch, poa_groups, opt.msa, opt.band_width, opt.band_mode, opt.adaptive_storage,
                          opt.graph_length, opt.pred_distance, nullptr, static_cast<float>(opt.gpu_mem_allocation),
                          opt.mismatch_score, opt.gap_score, opt.match_score);

    Init();
    int32_t groups_before = 0; // groups handled by earlier batch shapes (for the progress lines)
    for (int32_t b = 0; b < get_size<int32_t>(batch_shapes); b++)
    {
        const std::vector<int32_t>& ids = groups_per_batch[b];
        size_t free_bytes = 0, total_bytes = 0;
        if (hipSetDevice(0) != hipSuccess || hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess)
            throw std::runtime_error("no usable GPU");
        std::unique_ptr<Batch> batch =
            create_batch(0, nullptr, static_cast<int64_t>(opt.gpu_mem_allocation * static_cast<double>(free_bytes)),
                         opt.msa ? OutputType::msa : OutputType::consensus, batch_shapes[b], opt.gap_score, opt.mismatch_score,
                         opt.match_score);

        int32_t first_in_flight = 0; // index (into ids) of the first group of the batch being filled
        for (int32_t i = 0; i < get_size<int32_t>(ids);)
        {
            std::vector<StatusType> seq_status;
            const StatusType status = batch->add_poa_group(seq_status, poa_groups[ids[i]]);
            // run the batch when it is full, or when the last group has just been added
            if (status == StatusType::exceeded_maximum_poas || i == get_size<int32_t>(ids) - 1)
            {
                if (batch->get_total_poas() > 0)
                {
                    process_batch(*batch, opt.msa, ids, first_in_flight);
                    if (graph_output.is_open())
                    {
                        if (!graph_output.good()) throw std::runtime_error("Error writing dot file");
                        std::vector<DirectedGraph> graphs;
                        std::vector<StatusType> graph_status;
                        batch->get_graphs(graphs, graph_status);
                        for (DirectedGraph& g : graphs) graph_output << (opt.output_gfa ? g.serialize_to_gfa() : g.serialize_to_dot()) << std::endl;
                    }
                    batch->reset();
                    // a group that did not fit (exceeded_maximum_poas) is retried in the next round, so it is not in this range
                    const int32_t last = (status == StatusType::success) ? i : i - 1;
                    std::cerr << "Processed groups " << first_in_flight + groups_before << " - " << last + groups_before << " (batch " << b << ")" << std::endl;
                }
                else
                {
                    // even an empty batch cannot hold this group
                    std::cerr << "Could not add POA group " << ids[i] << " to batch " << b << std::endl;
                    i++;
                }
                first_in_flight = i;
            }
            if (status == StatusType::success)
            {
                int32_t dropped = 0;
                for (StatusType s : seq_status)
                    if (s == StatusType::exceeded_maximum_sequence_size) dropped++;
                if (dropped > 0)
                    std::cerr << "Dropping " << dropped << " sequence(s) in POA group " << ids[i] << " because it exceeded maximum size" << std::endl;
                i++;
            }
            else if (status != StatusType::exceeded_maximum_poas)
            {
                report(status, "Could not add POA group " + std::to_string(ids[i]) + " to batch " + std::to_string(b));
                i++;
            }
        }
        groups_before += get_size<int32_t>(ids);
    }
    return 0;
}

} // namespace

int main(int argc, char* argv[])
{
    try
    {
        return run(argc, argv);
    }
    catch (const std::exception& e)
    {
        std::cerr << "cudapoa: " << e.what() << std::endl;
        return 1;
    }
}
