// genomeutils.hpp -- random genome / read generators and reverse complement.
// Public names and arguments are those of the reference's utils/genomeutils.hpp (it is part of its public include tree,
// common/base/include/claraparabricks/genomeworks/utils/genomeutils.hpp:32-167); the implementation is this project's
// own. What has to agree with the reference is the sequence of values taken from the std::minstd_rand engine, so
// that a seed names the same synthetic reads in both code bases -- see detail::EditDice and the known-answer test.
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

namespace detail
{
/// The random decisions of the sequence generators, each drawn from the engine through the standard distribution
/// (and parameters) that fixes how many engine values it consumes: seeded inputs are a contract between this code
/// base, its CPU oracle and the reference (same std::minstd_rand seed => same reads), pinned by the known-answer
/// test tests/test_genomeutils_known_answers.py against pairs printed by the reference's own generator.
class EditDice
{
public:
    explicit EditDice(std::minstd_rand& engine)
        : engine_(engine)
    {
    }
    /// one of the max_* trials: the edit happens when a uniform real in [0, 1) exceeds one half
    bool fires() { return unit_(engine_) > 0.5; }
    /// uniform integer in [0, last]
    int upto(int last) { return std::uniform_int_distribution<int>(0, last)(engine_); }
    /// a base, uniform over A, C, G, T
    char base() { return "ACGT"[base_(engine_)]; }

private:
    std::minstd_rand& engine_;
    std::uniform_real_distribution<double> unit_{0, 1};
    std::uniform_int_distribution<int> base_{0, 3};
};
} // namespace detail

/// `length` bases, uniform over A, C, G, T.
inline std::string generate_random_genome(const int32_t length, std::minstd_rand& rng)
{
    std::uniform_int_distribution<int32_t> pick(0, 3);
    std::string genome(static_cast<size_t>(length > 0 ? length : 0), 'A');
    for (char& c : genome) c = "ACGT"[pick(rng)];
    return genome;
}

/// A noisy copy of `backbone`. Inside every range [first, second) (default: the whole backbone) up to max_deletions
/// single-base deletions, then up to max_insertions single-base insertions, then up to max_mutations substitutions are
/// tried (each try capped by the range length and taken with probability one half); a substitution may redraw the base
/// it replaces. Edited ranges are written back over the same positions of the copy.
inline std::string generate_random_sequence(const std::string& backbone, std::minstd_rand& rng, int max_mutations,
                                            int max_insertions, int max_deletions,
                                            std::vector<std::pair<int, int>>* ranges = nullptr)
{
    throw_on_negative(max_mutations, "max_mutations cannot be negative.");
    throw_on_negative(max_insertions, "max_insertions cannot be negative.");
    throw_on_negative(max_deletions, "max_deletions cannot be negative.");
    const int backbone_length = get_size<int>(backbone);
    std::vector<std::pair<int, int>> whole{{0, backbone_length}};
    const std::vector<std::pair<int, int>>& todo = ranges != nullptr ? *ranges : whole;

    detail::EditDice dice(rng);
    std::string noisy = backbone;
    for (const std::pair<int, int>& range : todo)
    {
        const int first = range.first, span = range.second - range.first;
        throw_on_negative(first, "start_index of the range cannot be negative.");
        throw_on_negative(span, "end_index of the range cannot be smaller than start_index.");
        if (range.second > backbone_length) throw std::invalid_argument("end_index should be smaller than backbone's length.");

        std::string piece = backbone.substr(static_cast<size_t>(first), static_cast<size_t>(span));
        for (int tries = std::min(max_deletions, span); tries > 0; --tries)
            if (dice.fires()) piece.erase(static_cast<size_t>(dice.upto(static_cast<int>(piece.size()) - 1)), 1);
        for (int tries = std::min(max_insertions, span); tries > 0; --tries)
            if (dice.fires())
            {
                const int where = dice.upto(static_cast<int>(piece.size())); // position first, base second
                piece.insert(static_cast<size_t>(where), 1, dice.base());
            }
        if (!piece.empty())
        {
            const int last = static_cast<int>(piece.size()) - 1; // fixed before the first substitution
            for (int tries = std::min(max_mutations, span); tries > 0; --tries)
                if (dice.fires())
                {
                    const int where = dice.upto(last);
                    piece[static_cast<size_t>(where)] = dice.base();
                }
        }
        if (first < static_cast<int>(noisy.size())) noisy.replace(static_cast<size_t>(first), static_cast<size_t>(span), piece);
    }
    return noisy;
}

/// The backbone itself followed by n - 1 noisy copies of it.
inline std::vector<std::string> generate_random_sequences(std::string const& backbone, int n, std::minstd_rand& rng,
                                                          int max_mutations = 1, int max_insertion = 1,
                                                          int max_deletions = 1)
{
    throw_on_negative(n, "n cannot be negative!");
    std::vector<std::string> reads{backbone};
    while (static_cast<int>(reads.size()) < n)
        reads.push_back(generate_random_sequence(backbone, rng, max_mutations, max_insertion, max_deletions));
    return reads;
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
