// fill_probe.cpp -- where the time of the reference's multi-batch pattern goes on the host (2048 full-band
// windows take 250-284 ms whatever the number of batches while the kernels need ~100 ms). One Batch of BatchConfig(1024, 200)
// full band, W synthetic windows of 32 reads <= 1024 bp: the cost of creating the batch, of filling it (add_poa_group per
// window), of generate_poa() and of get_consensus(), each on the wall clock; then the same fill again (warm), and the fill's
// bytes written into ordinary heap memory for comparison.
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tools/fill_probe.cpp -L genomeworks_amd/lib \
//   usage: fill_probe [windows=2048] [GB=32] [window file]
//       -lgenomeworks_amd -lgwhip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/genomeworks_amd/lib -Wl,-rpath,/opt/rocm/lib -o tools/bin/fill_probe
#include <claraparabricks/genomeworks/cudapoa/batch.hpp>
#include <claraparabricks/genomeworks/cudapoa/utils.hpp>
#include <claraparabricks/genomeworks/cudapoa/multi_device.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <vector>

using namespace claraparabricks::genomeworks;
using namespace claraparabricks::genomeworks::cudapoa;

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
    const int W       = argc > 1 ? std::atoi(argv[1]) : 2048;
    const double gb   = argc > 2 ? std::atof(argv[2]) : 32.0;
    std::mt19937 rng(7);
    std::vector<std::vector<std::string>> windows(static_cast<size_t>(W));
    if (argc > 3) // a window file (cudapoa's text format); repeated cyclically up to W windows
    {
        std::vector<std::vector<std::string>> file_windows;
        parse_cudapoa_file(file_windows, argv[3], -1);
        for (size_t i = 0; i < windows.size(); ++i) windows[i] = file_windows[i % file_windows.size()];
    }
    else
    for (auto& w : windows)
    {
        std::string backbone(960, 'A');
        for (char& c : backbone) c = "ACGT"[rng() & 3];
        for (int r = 0; r < 32; ++r)
        {
            std::string read = backbone;
            for (int k = 0; k < 48; ++k) read[rng() % read.size()] = "ACGT"[rng() & 3];
            for (int k = 0; k < 12; ++k) read.erase(rng() % read.size(), 1);
            for (int k = 0; k < 12; ++k) read.insert(rng() % read.size(), 1, "ACGT"[rng() & 3]);
            w.push_back(read);
        }
    }
    Init();
    if (argc > 4) // fill_probe W GB file BATCHES: the multi-batch runner itself (cudapoa::process_windows_multi_device)
    {
        BatchConfig shape(1024, 200);
        std::printf("{\"windows\": %d, \"multi_device\": [", W);
        bool first = true;
        for (const char* p = argv[4]; *p; ++p)
        {
            const int nb = *p - '0';
            if (nb < 1 || nb > 9) continue;
            for (int rep = 0; rep < 2; ++rep)
            {
                MultiDeviceConfig mc;
                mc.devices            = {0};
                mc.batches_per_device = nb;
                mc.memory_per_device  = static_cast<int64_t>(gb * 1e9);
                MultiDeviceOutput out;
                const double a = now();
                process_windows_multi_device(out, windows, shape, mc);
                const double b = now();
                std::printf("%s{\"batches\": %d, \"call_ms\": %.1f, \"ms\": %.1f, \"ms_after_creation\": %.1f, \"launches\": %d}", first ? "" : ", ", nb,
                            (b - a) * 1e3, out.seconds * 1e3, out.seconds_after_creation * 1e3, out.launches);
                first = false;
            }
        }
        std::printf("]}\n");
        return 0;
    }
    double t0 = now();
    BatchConfig shape(1024, 200); // the reference benchmarks' shape: full band, 200 reads per POA
    std::unique_ptr<Batch> batch = create_batch(0, nullptr, static_cast<int64_t>(gb * 1e9), OutputType::consensus, shape, -8, -6, 8);
    const double t_create = now() - t0;
    std::printf("{\"windows\": %d, \"create_ms\": %.2f", W, t_create * 1e3);
    for (int round = 0; round < 3; ++round)
    {
        batch->reset();
        t0         = now();
        int added  = 0;
        double t_group = 0, t_add = 0;
        for (const auto& w : windows)
        {
            const double a = now();
            Group group;
            group.reserve(w.size());
            for (const std::string& s : w) group.push_back(Entry{s.c_str(), nullptr, static_cast<int32_t>(s.size())});
            std::vector<StatusType> seq_status;
            const double b = now();
            const StatusType st = batch->add_poa_group(seq_status, group);
            const double c = now();
            t_group += b - a;
            t_add += c - b;
            if (st != StatusType::success) break;
            ++added;
        }
        const double t_fill = now() - t0;
        t0 = now();
        batch->generate_poa();
        const double t_gen = now() - t0;
        t0 = now();
        std::vector<std::string> consensus;
        std::vector<std::vector<uint16_t>> coverage;
        std::vector<StatusType> status;
        batch->get_consensus(consensus, coverage, status);
        const double t_get = now() - t0;
        std::printf(", \"round%d\": {\"added\": %d, \"fill_ms\": %.2f, \"fill_us_per_window\": %.2f, \"group_build_ms\": %.2f, \"add_poa_group_ms\": %.2f, "
                    "\"generate_poa_call_ms\": %.2f, \"get_consensus_ms\": %.2f}",
                    round, added, t_fill * 1e3, t_fill * 1e6 / std::max(1, added), t_group * 1e3, t_add * 1e3, t_gen * 1e3, t_get * 1e3);
    }
    // the same bytes into ordinary heap memory (what a fill costs when the destination is cacheable, unpinned memory)
    {
        std::vector<uint8_t> seq(static_cast<size_t>(W) * 32 * 1024), wts(seq.size());
        std::memset(seq.data(), 1, seq.size());
        std::memset(wts.data(), 1, wts.size());
        t0       = now();
        size_t o = 0;
        for (const auto& w : windows)
            for (const std::string& s : w)
            {
                std::memcpy(seq.data() + o, s.data(), s.size());
                std::memset(wts.data() + o, 1, s.size());
                o += (s.size() + 3) & ~size_t(3);
            }
        std::printf(", \"heap_copy_ms\": %.2f", (now() - t0) * 1e3);
    }
    std::printf("}\n");
    return 0;
}
