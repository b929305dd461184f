// chunk_order.hpp -- processing order and workspace size of a banded-aligner batch that is cut into chunks of consecutive pairs
// (host/cudaaligner.cpp, BandedAligner::align_all). Within a chunk the longest pairs go first (aligner_global_myers_banded.cpp:
// 306-309: the lanes of one wavefront get similar work), ties in input order; indices are chunk-local. For a million short pairs
// this is 5 ms on one thread -- longer than the uploads and the kernels -- so large chunks are cut into pieces of whole waves
// (64 slots) for the host threads: a stable counting sort by descending pair length in three steps -- histogram per piece, first
// slot of every (length, piece) by a running sum, scatter per piece -- gives exactly the order of the one-thread sort; the
// workspace is then sized piece by piece (gwhip_myers_banded_workspace_words). Host only; tests/cpp/chunk_order_driver.cpp.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "host_common.hpp"

namespace gwhost
{

struct PairRange
{
    int32_t lo, hi; ///< pairs [lo, hi) of the batch
};

/// include/gwhip.h: gwhip_myers_banded_workspace_words
using WorkspaceWordsFn = int64_t (*)(int32_t first_slot, int32_t n_slots, const int64_t* sequence_starts_host, const int32_t* max_bandwidths_host,
                                     const int32_t* scheduling_index_host);

/// one thread, one chunk: stable, descending pair length
inline void order_chunk(const PairRange& c, const int64_t* seq_starts, int32_t* order)
{
    const int32_t m = c.hi - c.lo;
    int32_t* ord    = order + c.lo;
    auto len_of     = [&](int32_t i) { return seq_starts[2 * static_cast<size_t>(c.lo + i) + 2] - seq_starts[2 * static_cast<size_t>(c.lo + i)]; };
    int64_t longest = 0;
    for (int32_t i = 0; i < m; ++i) longest = std::max(longest, len_of(i));
    if (m >= 4096 && longest < (int64_t(1) << 22))
    {
        // a stable counting sort by descending pair length: linear in m
        std::vector<int32_t> first(static_cast<size_t>(longest) + 2, 0);
        for (int32_t i = 0; i < m; ++i) first[static_cast<size_t>(longest - len_of(i)) + 1]++;
        for (size_t k = 1; k < first.size(); ++k) first[k] += first[k - 1];
        for (int32_t i = 0; i < m; ++i) ord[static_cast<size_t>(first[static_cast<size_t>(longest - len_of(i))]++)] = i;
    }
    else
    {
        std::iota(ord, ord + m, 0);
        std::stable_sort(ord, ord + m, [&](int32_t a, int32_t b) { return len_of(a) > len_of(b); });
    }
}

/// Writes order[c.lo .. c.hi) of every chunk and returns the chunks' workspace words (the sum of words_of over their waves).
/// longest_pair = query + target length of the batch's longest pair; pieces_from = pairs a piece must at least have.
inline std::vector<int64_t> order_and_size_chunks(const std::vector<PairRange>& chunks, const int64_t* seq_starts, const int32_t* max_bandwidths,
                                                  int64_t longest_pair, size_t host_threads, int32_t* order, WorkspaceWordsFn words_of,
                                                  int32_t pieces_from = 16384)
{
    struct Piece
    {
        int32_t chunk, lo, hi; // chunk-local indices, and the same range of slots
        std::vector<int32_t> first;
        int64_t words;
    };
    const int32_t n_chunks = static_cast<int32_t>(chunks.size());
    const int64_t buckets  = longest_pair + 1;
    std::vector<Piece> pieces;
    for (int32_t k = 0; k < n_chunks; ++k)
    {
        const int32_t m       = chunks[static_cast<size_t>(k)].hi - chunks[static_cast<size_t>(k)].lo;
        const int32_t p_count = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(static_cast<int64_t>(host_threads) / std::max(1, n_chunks), m / pieces_from)));
        const int32_t share   = ((m + p_count - 1) / p_count + 63) & ~63;
        for (int32_t lo = 0; lo < m; lo += share) pieces.push_back(Piece{k, lo, std::min(m, lo + share), {}, 0});
    }
    const bool in_pieces = pieces.size() > static_cast<size_t>(n_chunks) && buckets * static_cast<int64_t>(pieces.size()) <= (int64_t(1) << 22);
    auto len_in_chunk    = [&](const PairRange& c, int32_t i) { return seq_starts[2 * static_cast<size_t>(c.lo + i) + 2] - seq_starts[2 * static_cast<size_t>(c.lo + i)]; };
    auto size_piece      = [&](Piece& pc) {
        const PairRange& c = chunks[static_cast<size_t>(pc.chunk)];
        pc.words           = words_of(pc.lo, pc.hi - pc.lo, seq_starts + 2 * static_cast<size_t>(c.lo), max_bandwidths + c.lo, order + c.lo);
    };
    if (in_pieces)
    {
        parallel_tasks(pieces.size(), pieces.size(), [&](size_t t) {
            Piece& pc          = pieces[t];
            const PairRange& c = chunks[static_cast<size_t>(pc.chunk)];
            pc.first.assign(static_cast<size_t>(buckets), 0);
            for (int32_t i = pc.lo; i < pc.hi; ++i) pc.first[static_cast<size_t>(longest_pair - len_in_chunk(c, i))]++;
        });
        for (size_t t0 = 0; t0 < pieces.size();) // the pieces of one chunk are consecutive
        {
            size_t t1 = t0;
            while (t1 < pieces.size() && pieces[t1].chunk == pieces[t0].chunk) ++t1;
            int32_t running = 0;
            for (int64_t b = 0; b < buckets; ++b)
                for (size_t t = t0; t < t1; ++t)
                {
                    const int32_t count                     = pieces[t].first[static_cast<size_t>(b)];
                    pieces[t].first[static_cast<size_t>(b)] = running;
                    running += count;
                }
            t0 = t1;
        }
        parallel_tasks(pieces.size(), pieces.size(), [&](size_t t) {
            Piece& pc          = pieces[t];
            const PairRange& c = chunks[static_cast<size_t>(pc.chunk)];
            int32_t* ord       = order + c.lo;
            for (int32_t i = pc.lo; i < pc.hi; ++i) ord[static_cast<size_t>(pc.first[static_cast<size_t>(longest_pair - len_in_chunk(c, i))]++)] = i;
        });
        parallel_tasks(pieces.size(), pieces.size(), [&](size_t t) { size_piece(pieces[t]); });
    }
    else
    {
        pieces.clear();
        for (int32_t k = 0; k < n_chunks; ++k) pieces.push_back(Piece{k, 0, chunks[static_cast<size_t>(k)].hi - chunks[static_cast<size_t>(k)].lo, {}, 0});
        parallel_tasks(pieces.size(), pieces.size(), [&](size_t t) {
            order_chunk(chunks[static_cast<size_t>(pieces[t].chunk)], seq_starts, order);
            size_piece(pieces[t]);
        });
    }
    std::vector<int64_t> words(chunks.size(), 0);
    for (const Piece& pc : pieces) words[static_cast<size_t>(pc.chunk)] += pc.words;
    return words;
}

} // namespace gwhost
