// Driver for genomeworks_amd/host/base_packing.hpp: batches packed pair by pair as BandedAligner::add_alignment() does (queries
// and targets of odd and even lengths, so both halves of a byte start a sequence), expanded by a restatement of the device side
// (unpack_bases_kernel, gwhip_myers.hip: 16 bases per lane from 8 packed bytes, table 'A' 'C' 'T' 'G' 'N', clipped to the chunk's
// [first, last)), chunk by chunk at arbitrary cuts. Every expanded base must be indistinguishable from the caller's for the
// kernels: a query base equal to 'A' / 'C' / 'T' / 'G' exactly where the original is, a target base with the same pattern index
// (c >> 1) & 3. Prints "ok" or the first failure.
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "base_packing.hpp"

// what one lane of unpack_bases_kernel does for its group of 16 bases
static void unpack_group(const uint8_t* packed, char* out, int64_t i0, int64_t first, int64_t last)
{
    const uint64_t lut = 0x4e4e4e4e47544341ull;
    uint64_t bits;
    std::memcpy(&bits, packed + (i0 >> 1), 8);
    for (int k = 0; k < 16; ++k)
        if (i0 + k >= first && i0 + k < last) out[i0 + k] = static_cast<char>(lut >> (8 * ((bits >> (4 * k)) & 7u)));
}

static void unpack_range(const uint8_t* packed, char* out, int64_t first, int64_t last)
{
    for (int64_t i0 = first & ~int64_t(15); i0 < last; i0 += 16) unpack_group(packed, out, i0, first, last);
}

int main()
{
    std::mt19937 rng(11);
    const std::string alphabet = "ACGTACGTACGTNnacgtRYKMSWBDHV-*xU@~ ";
    for (int rep = 0; rep < 40; ++rep)
    {
        std::vector<char> seq;
        std::vector<uint8_t> packed;
        std::vector<int64_t> starts{0};
        const int pairs = 1 + static_cast<int>(rng() % 60);
        for (int p = 0; p < pairs; ++p)
        {
            const int32_t ql = static_cast<int32_t>(rng() % 70), tl = static_cast<int32_t>(rng() % 70);
            const int64_t at = static_cast<int64_t>(seq.size());
            for (int32_t i = 0; i < ql + tl; ++i) seq.push_back(rep % 4 == 3 ? static_cast<char>(rng() & 0xff) : alphabet[rng() % alphabet.size()]);
            // (the vector grows without initialising, as the pinned array does: old bytes are kept, new ones are garbage)
            const size_t old = packed.size();
            packed.resize((seq.size() + 1) / 2);
            for (size_t b = old; b < packed.size(); ++b) packed[b] = static_cast<uint8_t>(rng());
            gwhost::pack_bases(packed.data(), at, seq.data() + at, ql, gwhost::query_code);
            gwhost::pack_bases(packed.data(), at + ql, seq.data() + at + ql, tl, gwhost::target_code);
            starts.push_back(at + ql);
            starts.push_back(at + ql + tl);
        }
        const int64_t total = static_cast<int64_t>(seq.size());
        packed.resize(packed.size() + 16, 0xee); // the device staging buffer is padded (reads of whole groups)
        std::vector<char> out(static_cast<size_t>(total) + 16, '?');
        // chunks of consecutive pairs, as align_all() cuts them
        std::vector<int> cuts{0, pairs};
        for (int c = 0; c < 4; ++c) cuts.push_back(static_cast<int>(rng() % (pairs + 1)));
        std::sort(cuts.begin(), cuts.end());
        for (size_t c = 0; c + 1 < cuts.size(); ++c) unpack_range(packed.data(), out.data(), starts[2 * cuts[c]], starts[2 * cuts[c + 1]]);
        for (int64_t i = total; i < total + 16; ++i)
            if (out[static_cast<size_t>(i)] != '?')
            {
                std::printf("FAIL rep %d: wrote past the batch at %lld\n", rep, static_cast<long long>(i));
                return 1;
            }
        for (int p = 0; p < pairs; ++p)
            for (int64_t i = starts[2 * p]; i < starts[2 * p + 2]; ++i)
            {
                const char orig = seq[static_cast<size_t>(i)], got = out[static_cast<size_t>(i)];
                bool same;
                if (i < starts[2 * p + 1]) // query: equality with each of the four letters
                    same = (orig == 'A') == (got == 'A') && (orig == 'C') == (got == 'C') && (orig == 'T') == (got == 'T') && (orig == 'G') == (got == 'G');
                else
                    same = ((static_cast<unsigned char>(orig) >> 1) & 3) == ((static_cast<unsigned char>(got) >> 1) & 3);
                if (!same)
                {
                    std::printf("FAIL rep %d pair %d base %lld: %d -> %d\n", rep, p, static_cast<long long>(i), orig, got);
                    return 1;
                }
            }
    }
    std::printf("ok\n");
    return 0;
}
