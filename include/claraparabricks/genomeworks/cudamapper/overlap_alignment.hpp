// overlap_alignment.hpp -- the alignment stage of cudamapper as a consumer of cudaaligner.
// What it mirrors: `struct Overlap` / `RelativeStrand` (cudamapper/include/.../cudamapper/types.hpp:30-95),
// `align_overlaps()` + `run_alignment_batch()` (cudamapper/src/main.cu:54-187: several alignment engines on host
// threads share one allocator and one device, each pulling ranges of overlaps, `create_aligner(max_query, max_target,
// batch_size, global_alignment, allocator, stream, device)` -> add_alignment / align_all / sync_alignments /
// convert_to_cigar / reset) and `print_paf()` (cudamapper/src/utils.cpp:41-124). The overlap *finder* (index, matcher,
// chainer) is out of scope: overlaps arrive as PAF records.
#pragma once
#include <claraparabricks/genomeworks/utils/allocator.hpp>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudamapper
{

using read_id_t          = std::uint32_t;
using position_in_read_t = std::uint32_t;

enum class RelativeStrand : unsigned char
{
    Forward = '+',
    Reverse = '-',
};

/// One overlap between two substrings (field names as in the reference).
struct Overlap
{
    read_id_t query_read_id_;
    read_id_t target_read_id_;
    position_in_read_t query_start_position_in_read_;
    position_in_read_t target_start_position_in_read_;
    position_in_read_t query_end_position_in_read_;
    position_in_read_t target_end_position_in_read_;
    RelativeStrand relative_strand = RelativeStrand::Forward;
    std::uint32_t num_residues_    = 0;
    bool overlap_complete          = false;
};

/// A named sequence (io::FastaSequence of the reference: name + seq).
struct FastaSequence
{
    std::string name;
    std::string seq;
};

/// Reads every record of a (multi-line) FASTA file; names are cut at the first whitespace. Throws on I/O errors.
std::vector<FastaSequence> read_fasta(const std::string& path);

/// Parses the first nine PAF columns of every line; read names are resolved against the two sequence sets.
/// Throws std::runtime_error on malformed lines, unknown names or coordinates outside the reads.
std::vector<Overlap> read_paf(const std::string& path, const std::vector<FastaSequence>& queries,
                              const std::vector<FastaSequence>& targets);

/// Global alignment of the overlapped regions. `cigars` is resized to overlaps.size(); entry i belongs to overlap i
/// whatever the number of engines. `num_alignment_engines` host threads each own a stream and an Aligner and share
/// `allocator`. `max_alignments_per_batch` <= 0: derived from the allocator's free memory like the reference does
/// (85 % of it, 0.03 B per base pair of the largest overlap).
void align_overlaps(DefaultDeviceAllocator allocator, const std::vector<Overlap>& overlaps,
                    const std::vector<FastaSequence>& queries, const std::vector<FastaSequence>& targets,
                    std::int32_t num_alignment_engines, std::vector<std::string>& cigars,
                    std::int32_t max_alignments_per_batch = 0);

/// PAF lines in the reference's format (12 columns, `cg:Z:` tag when cigars is not empty).
void print_paf(const std::vector<Overlap>& overlaps, const std::vector<std::string>& cigars,
               const std::vector<FastaSequence>& queries, const std::vector<FastaSequence>& targets,
               std::int32_t kmer_size, std::FILE* out);

/// SAM text records for the overlaps (cudamapper -S: print_sam, cudamapper/src/utils.cpp:190-318, which the reference only
/// builds with htslib; this writer needs no library and emits text SAM only -- no BAM). Header: one @SQ line per distinct
/// target read in order of first appearance (the reference adds one per overlap and htslib refuses the duplicates) and the
/// @PG line of the tool; one record per overlap with the fields the reference fills: QNAME, FLAG 0 (16 for overlaps on the
/// reverse strand), RNAME / POS of the TARGET read and start (the reference leaves the target index at 0 and writes the
/// query start; a SAM consumer needs the position on RNAME), MAPQ 255 as in print_paf, the CIGAR when cigars is not empty
/// (else *), RNEXT * / PNEXT 0 / TLEN 0, the whole query sequence, QUAL *.
void print_sam(const std::vector<Overlap>& overlaps, const std::vector<std::string>& cigars,
               const std::vector<FastaSequence>& queries, const std::vector<FastaSequence>& targets,
               const std::string& program_version, const std::string& command_line, std::FILE* out);

} // namespace cudamapper
} // namespace genomeworks
} // namespace claraparabricks
