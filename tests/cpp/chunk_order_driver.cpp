// Driver for genomeworks_amd/host/chunk_order.hpp: the processing order of every chunk equals a plain stable sort by descending
// pair length, and the chunks' workspace words equal gwhip_myers_banded_workspace_words over the whole chunk -- for batches cut
// into 1 .. 7 chunks, 1 .. 32 host threads, pieces from 64 pairs up (so small batches take the parallel path: histogram per piece,
// running sum, scatter), equal lengths, a few distinct lengths, wide ranges (the bucket table too large: one thread per chunk).
// Prints "ok" or the first difference.
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#include "../../include/gwhip.h"
#include "chunk_order.hpp"

int main()
{
    std::mt19937 rng(2026);
    int cases = 0, parallel_cases = 0;
    for (int rep = 0; rep < 160; ++rep)
    {
        const int32_t n        = rep < 8 ? rep + 1 : static_cast<int32_t>(1 + rng() % (rep % 5 == 0 ? 60000 : 3000));
        const int kind         = rep % 4; // 0 equal, 1 few lengths, 2 short-read like, 3 wide
        const int32_t n_chunks = std::max(1, std::min<int32_t>(1 + static_cast<int32_t>(rng() % 7), n));
        std::vector<int64_t> starts{0};
        std::vector<int32_t> bws;
        int64_t longest = 0;
        for (int32_t i = 0; i < n; ++i)
        {
            int32_t q, t;
            switch (kind)
            {
            case 0: q = 150, t = 150; break;
            case 1: q = 100 + 50 * static_cast<int32_t>(rng() % 3), t = q + static_cast<int32_t>(rng() % 2); break;
            case 2: q = 148 + static_cast<int32_t>(rng() % 5), t = 147 + static_cast<int32_t>(rng() % 7); break;
            default: q = 1 + static_cast<int32_t>(rng() % (rep % 8 == 3 ? 400000 : 5000)), t = 1 + static_cast<int32_t>(rng() % 5000); break;
            }
            starts.push_back(starts.back() + q);
            starts.push_back(starts.back() + t);
            bws.push_back(static_cast<int32_t>(16 + rng() % 1000));
            longest = std::max<int64_t>(longest, static_cast<int64_t>(q) + t);
        }
        std::vector<gwhost::PairRange> chunks;
        for (int32_t k = 0; k < n_chunks; ++k)
            chunks.push_back(gwhost::PairRange{static_cast<int32_t>(static_cast<int64_t>(n) * k / n_chunks), static_cast<int32_t>(static_cast<int64_t>(n) * (k + 1) / n_chunks)});
        const size_t threads      = 1 + rng() % 32;
        const int32_t pieces_from = rep % 3 == 0 ? 16384 : static_cast<int32_t>(64 + rng() % 900);
        std::vector<int32_t> order(static_cast<size_t>(n), -1);
        const std::vector<int64_t> words = gwhost::order_and_size_chunks(chunks, starts.data(), bws.data(), longest, threads, order.data(),
                                                                         &gwhip_myers_banded_workspace_words, pieces_from);
        for (int32_t k = 0; k < n_chunks; ++k)
        {
            const gwhost::PairRange& c = chunks[static_cast<size_t>(k)];
            const int32_t m            = c.hi - c.lo;
            std::vector<int32_t> ref(static_cast<size_t>(m));
            std::iota(ref.begin(), ref.end(), 0);
            auto len_of = [&](int32_t i) { return starts[2 * static_cast<size_t>(c.lo + i) + 2] - starts[2 * static_cast<size_t>(c.lo + i)]; };
            std::stable_sort(ref.begin(), ref.end(), [&](int32_t a, int32_t b) { return len_of(a) > len_of(b); });
            for (int32_t i = 0; i < m; ++i)
                if (order[static_cast<size_t>(c.lo + i)] != ref[static_cast<size_t>(i)])
                {
                    std::printf("FAIL rep %d (n %d, chunks %d, threads %zu, pieces from %d): chunk %d slot %d is %d, expected %d\n", rep, n, n_chunks, threads,
                                pieces_from, k, i, order[static_cast<size_t>(c.lo + i)], ref[static_cast<size_t>(i)]);
                    return 1;
                }
            const int64_t whole = gwhip_myers_banded_workspace_words(0, m, starts.data() + 2 * static_cast<size_t>(c.lo), bws.data() + c.lo, ref.data());
            if (whole != words[static_cast<size_t>(k)])
            {
                std::printf("FAIL rep %d chunk %d: %lld workspace words, expected %lld\n", rep, k, static_cast<long long>(words[static_cast<size_t>(k)]),
                            static_cast<long long>(whole));
                return 1;
            }
            if (threads / static_cast<size_t>(n_chunks) > 1 && m / pieces_from > 1) parallel_cases++;
        }
        cases++;
    }
    if (parallel_cases < 20)
    {
        std::printf("FAIL only %d of the chunks were large enough for pieces\n", parallel_cases);
        return 1;
    }
    std::printf("ok\n");
    return 0;
}
