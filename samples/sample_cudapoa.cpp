// sample_cudapoa.cpp -- consensus for a few POA groups through the C++ Batch interface (counterpart of
// cudapoa/samples/sample_cudapoa.cpp). Build:
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include samples/sample_cudapoa.cpp \
//       -L genomeworks_amd/lib -lgenomeworks_amd -lgwhip -L /opt/rocm/lib -lamdhip64 \
//       -Wl,-rpath,$PWD/genomeworks_amd/lib -Wl,-rpath,/opt/rocm/lib -o sample_cudapoa
#include <claraparabricks/genomeworks/cudapoa/batch.hpp>
#include <claraparabricks/genomeworks/cudapoa/utils.hpp>

#include <iostream>
#include <memory>

using namespace claraparabricks::genomeworks;
using namespace claraparabricks::genomeworks::cudapoa;

int main(int argc, char** argv)
{
    std::vector<std::vector<std::string>> windows;
    if (argc > 1)
        parse_cudapoa_file(windows, argv[1], -1);
    else
        windows = {{"ACGTTGCAACGTACGTTAGC", "ACGTTGCATCGTACGTTAGC", "ACGTTGCAACGTACGTAGC", "ACGTTGCAACGTACGTTAGC"},
                   {"TTGACCATTG", "TTGACATTG", "TTGACCATTG"}};
    Init();
    BatchConfig shape(1024, 32, 128, BandMode::static_band);
    std::unique_ptr<Batch> batch = create_batch(0, nullptr, int64_t(1) << 30, OutputType::consensus, shape, -8, -6, 8);
    for (const auto& w : windows)
    {
        Group group;
        for (const std::string& s : w) group.push_back(Entry{s.c_str(), nullptr, static_cast<int32_t>(s.size())});
        std::vector<StatusType> seq_status;
        if (batch->add_poa_group(seq_status, group) != StatusType::success) std::cerr << "group not added" << std::endl;
    }
    batch->generate_poa();
    std::vector<std::string> consensus;
    std::vector<std::vector<uint16_t>> coverage;
    std::vector<StatusType> status;
    batch->get_consensus(consensus, coverage, status);
    for (size_t i = 0; i < consensus.size(); i++)
        std::cout << (status[i] == StatusType::success ? consensus[i] : std::string("<error>")) << std::endl;
    return 0;
}
