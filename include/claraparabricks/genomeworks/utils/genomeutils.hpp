// genomeutils.hpp -- random genome / read generators and reverse complement.
// Same names, arguments and RNG consumption order as the reference's
// common/base/include/claraparabricks/genomeworks/utils/genomeutils.hpp:32-167, so seeded inputs
// (std::minstd_rand + libstdc++ distributions) are reproducible across the two code bases.
#pragma once

#include <algorithm>
#include <cstdint>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <claraparabricks/genomeworks/utils/signed_integer_utils.hpp>

namespace claraparabricks
{
namespace genomeworks
{
namespace genomeutils
{

inline std::string generate_random_genome(const int32_t length, std::minstd_rand& rng)
{
    static const char alphabet[4] = {'A', 'C', 'G', 'T'};
    std::uniform_int_distribution<int32_t> random_index(0, 3);
    std::string genome;
    genome.reserve(length > 0 ? length : 0);
    for (int32_t i = 0; i < length; i++) genome += alphabet[random_index(rng)];
    return genome;
}

// Draw order per range: deletions (prob, then position), insertions (prob, position, base), substitutions
// (prob, position, base); each of the max_* trials fires when random_prob(rng) > 0.5.
inline std::string generate_random_sequence(const std::string& backbone, std::minstd_rand& rng, int max_mutations,
                                            int max_insertions, int max_deletions,
                                            std::vector<std::pair<int, int>>* ranges = nullptr)
{
    throw_on_negative(max_mutations, "max_mutations cannot be negative.");
    throw_on_negative(max_insertions, "max_insertions cannot be negative.");
    throw_on_negative(max_deletions, "max_deletions cannot be negative.");
    static const char alphabet[4] = {'A', 'C', 'G', 'T'};
    std::uniform_int_distribution<int> random_base(0, 3);
    std::string sequence = backbone;
    std::vector<std::pair<int, int>> full_range(1, std::make_pair(0, get_size<int>(backbone)));
    if (ranges == nullptr) ranges = &full_range;
    for (auto range : *ranges)
    {
        const int start_index = range.first;
        const int end_index   = range.second;
        throw_on_negative(start_index, "start_index of the range cannot be negative.");
        throw_on_negative(end_index - start_index, "end_index of the range cannot be smaller than start_index.");
        if (get_size<int>(backbone) < end_index)
            throw std::invalid_argument("end_index should be smaller than backbone's length.");
        const int range_length = end_index - start_index;
        std::string substring  = backbone.substr(start_index, range_length);
        std::uniform_real_distribution<double> random_prob(0, 1);
        for (int j = 0; j < std::min(max_deletions, range_length); j++)
        {
            if (random_prob(rng) > 0.5)
            {
                const int length = static_cast<int>(substring.length());
                std::uniform_int_distribution<int> random_del_pos(0, length - 1);
                substring.erase(random_del_pos(rng), 1);
            }
        }
        for (int j = 0; j < std::min(max_insertions, range_length); j++)
        {
            if (random_prob(rng) > 0.5)
            {
                const int length = static_cast<int>(substring.length());
                std::uniform_int_distribution<int> random_ins_pos(0, length);
                const int ins_pos  = random_ins_pos(rng);
                const int ins_base = random_base(rng);
                substring.insert(ins_pos, 1, alphabet[ins_base]);
            }
        }
        const int length = static_cast<int>(substring.length());
        if (length > 0)
        {
            std::uniform_int_distribution<int> random_mut_pos(0, length - 1);
            for (int j = 0; j < std::min(max_mutations, range_length); j++)
            {
                if (random_prob(rng) > 0.5)
                {
                    const int mut_pos   = random_mut_pos(rng);
                    const int swap_base = random_base(rng);
                    substring[mut_pos]  = alphabet[swap_base];
                }
            }
        }
        if (start_index < static_cast<int>(sequence.length())) sequence.replace(start_index, range_length, substring);
    }
    return sequence;
}

inline std::vector<std::string> generate_random_sequences(std::string const& backbone, int n, std::minstd_rand& rng,
                                                          int max_mutations = 1, int max_insertion = 1,
                                                          int max_deletions = 1)
{
    throw_on_negative(n, "n cannot be negative!");
    std::vector<std::string> sequences;
    sequences.reserve(n);
    sequences.push_back(backbone);
    for (int i = 1; i < n; i++)
        sequences.push_back(generate_random_sequence(backbone, rng, max_mutations, max_insertion, max_deletions));
    return sequences;
}

// A -> T, C -> G, T -> A, G -> C via (c >> 1) & 3 (genomeutils.hpp:144-160)
inline void reverse_complement(const char* src, const int32_t length, char* dest)
{
    constexpr char lookup[] = {'T', 'G', 'A', 'C'};
    for (int32_t pos = 0; pos < length; pos++)
    {
        const unsigned char nucleotide = static_cast<unsigned char>(src[length - 1 - pos]);
        dest[pos]                      = lookup[(nucleotide >> 1) & 0b11];
    }
}

/// Copies src to dest, or stores its reverse complement there (genomeutils.hpp:162-180).
inline void copy_sequence(const char* const src, const int32_t length, char* const dest, const bool do_reverse_complement)
{
    if (do_reverse_complement)
        reverse_complement(src, length, dest);
    else
        std::copy_n(src, length, dest);
}

} // namespace genomeutils
} // namespace genomeworks
} // namespace claraparabricks
