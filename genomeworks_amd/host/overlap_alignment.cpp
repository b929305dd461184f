// The alignment stage of cudamapper over our cudaaligner (see the header for the reference locations).
#include <claraparabricks/genomeworks/cudamapper/overlap_alignment.hpp>

#include <claraparabricks/genomeworks/cudaaligner/aligner.hpp>
#include <claraparabricks/genomeworks/cudaaligner/alignment.hpp>
#include <claraparabricks/genomeworks/utils/cudautils.hpp>
#include <claraparabricks/genomeworks/utils/signed_integer_utils.hpp>

#include <algorithm>
#include <cinttypes>
#include <cstdlib>
#include <fstream>
#include <future>
#include <iostream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudamapper
{

std::vector<FastaSequence> read_fasta(const std::string& path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open FASTA file " + path);
    std::vector<FastaSequence> records;
    std::string line;
    while (std::getline(in, line))
    {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>')
        {
            const size_t end = line.find_first_of(" \t", 1);
            records.push_back({line.substr(1, end == std::string::npos ? std::string::npos : end - 1), std::string()});
        }
        else
        {
            if (records.empty()) throw std::runtime_error("FASTA file " + path + " does not start with a header line");
            records.back().seq += line;
        }
    }
    return records;
}

std::vector<Overlap> read_paf(const std::string& path, const std::vector<FastaSequence>& queries,
                              const std::vector<FastaSequence>& targets)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open PAF file " + path);
    std::unordered_map<std::string, read_id_t> qid, tid;
    for (size_t i = 0; i < queries.size(); i++) qid.emplace(queries[i].name, static_cast<read_id_t>(i));
    for (size_t i = 0; i < targets.size(); i++) tid.emplace(targets[i].name, static_cast<read_id_t>(i));
    std::vector<Overlap> overlaps;
    std::string line;
    int64_t line_no = 0;
    while (std::getline(in, line))
    {
        line_no++;
        if (line.empty()) continue;
        std::istringstream ls(line);
        std::string qname, tname, strand;
        int64_t qlen, qs, qe, tlen, ts, te;
        if (!(ls >> qname >> qlen >> qs >> qe >> strand >> tname >> tlen >> ts >> te) || (strand != "+" && strand != "-"))
            throw std::runtime_error("malformed PAF line " + std::to_string(line_no) + " in " + path);
        const auto q = qid.find(qname);
        const auto t = tid.find(tname);
        if (q == qid.end() || t == tid.end())
            throw std::runtime_error("PAF line " + std::to_string(line_no) + ": unknown read name");
        const int64_t ql = get_size<int64_t>(queries[q->second].seq), tl = get_size<int64_t>(targets[t->second].seq);
        if (qs < 0 || qe < qs || qe > ql || ts < 0 || te < ts || te > tl)
            throw std::runtime_error("PAF line " + std::to_string(line_no) + ": coordinates outside the read");
        Overlap o{};
        o.query_read_id_                 = q->second;
        o.target_read_id_                = t->second;
        o.query_start_position_in_read_  = static_cast<position_in_read_t>(qs);
        o.query_end_position_in_read_    = static_cast<position_in_read_t>(qe);
        o.target_start_position_in_read_ = static_cast<position_in_read_t>(ts);
        o.target_end_position_in_read_   = static_cast<position_in_read_t>(te);
        o.relative_strand                = strand == "-" ? RelativeStrand::Reverse : RelativeStrand::Forward;
        int64_t residues                 = 0;
        if (ls >> residues) o.num_residues_ = static_cast<std::uint32_t>(std::max<int64_t>(residues, 0));
        overlaps.push_back(o);
    }
    return overlaps;
}

namespace
{

// one alignment engine (cudamapper/src/main.cu:54-124): pulls ranges of overlaps until none is left
void run_alignment_batch(DefaultDeviceAllocator allocator, std::mutex& overlap_idx_mtx, const std::vector<Overlap>& overlaps,
                         const std::vector<FastaSequence>& queries, const std::vector<FastaSequence>& targets,
                         int32_t& overlap_idx, const int32_t max_query_size, const int32_t max_target_size,
                         std::vector<std::string>& cigars, const int32_t batch_size, const int32_t device_id)
{
    scoped_device_switch dev(device_id);
    CudaStream stream = make_cuda_stream();
    std::unique_ptr<cudaaligner::Aligner> batch = cudaaligner::create_aligner(
        max_query_size, max_target_size, batch_size, cudaaligner::AlignmentType::global_alignment, allocator, stream.get(), device_id);
    while (true)
    {
        int32_t idx_start = 0, idx_end = 0;
        {
            std::lock_guard<std::mutex> lck(overlap_idx_mtx);
            if (overlap_idx == get_size<int32_t>(overlaps)) break;
            idx_start   = overlap_idx;
            idx_end     = std::min(idx_start + batch_size, get_size<int32_t>(overlaps));
            overlap_idx = idx_end;
        }
        for (int32_t idx = idx_start; idx < idx_end; idx++)
        {
            const Overlap& o            = overlaps[idx];
            const std::string& query    = queries[o.query_read_id_].seq;
            const std::string& target   = targets[o.target_read_id_].seq;
            const int32_t query_length  = static_cast<int32_t>(o.query_end_position_in_read_ - o.query_start_position_in_read_);
            const int32_t target_length = static_cast<int32_t>(o.target_end_position_in_read_ - o.target_start_position_in_read_);
            const cudaaligner::StatusType status =
                batch->add_alignment(query.data() + o.query_start_position_in_read_, query_length,
                                     target.data() + o.target_start_position_in_read_, target_length, false,
                                     o.relative_strand == RelativeStrand::Reverse);
            if (status != cudaaligner::success) throw std::runtime_error("Experienced error type " + std::to_string(status));
        }
        batch->align_all(); // asynchronous on the engine's stream
        batch->sync_alignments();
        const std::vector<std::shared_ptr<cudaaligner::Alignment>>& alignments = batch->get_alignments();
        for (int32_t i = 0; i < get_size<int32_t>(alignments); i++) cigars[idx_start + i] = alignments[i]->convert_to_cigar();
        batch->reset();
    }
}

} // namespace

void align_overlaps(DefaultDeviceAllocator allocator, const std::vector<Overlap>& overlaps,
                    const std::vector<FastaSequence>& queries, const std::vector<FastaSequence>& targets,
                    int32_t num_alignment_engines, std::vector<std::string>& cigars, int32_t max_alignments_per_batch)
{
    cigars.assign(overlaps.size(), std::string());
    if (overlaps.empty()) return;
    if (num_alignment_engines < 1) throw std::invalid_argument("num_alignment_engines must be at least 1");
    int32_t max_query_size = 0, max_target_size = 0;
    for (const Overlap& o : overlaps)
    {
        if (o.query_read_id_ >= queries.size() || o.target_read_id_ >= targets.size() ||
            o.query_end_position_in_read_ < o.query_start_position_in_read_ ||
            o.target_end_position_in_read_ < o.target_start_position_in_read_ ||
            o.query_end_position_in_read_ > queries[o.query_read_id_].seq.size() ||
            o.target_end_position_in_read_ > targets[o.target_read_id_].seq.size())
            throw std::invalid_argument("overlap refers to a read or a position that does not exist");
        max_query_size  = std::max(max_query_size, static_cast<int32_t>(o.query_end_position_in_read_ - o.query_start_position_in_read_));
        max_target_size = std::max(max_target_size, static_cast<int32_t>(o.target_end_position_in_read_ - o.target_start_position_in_read_));
    }
    int32_t batch_size = max_alignments_per_batch;
    if (batch_size <= 0)
    {
        // the reference's heuristic (main.cu:148-155): 85 % of the free memory, 0.03 B per base pair of the largest overlap
        const float memory_per_alignment = std::max(1.0f, 0.03f * static_cast<float>(max_query_size) * static_cast<float>(max_target_size));
        const double free_bytes          = static_cast<double>(allocator.get_size_of_largest_free_memory_block());
        const double max_alignments      = free_bytes * 85 / 100 / memory_per_alignment;
        batch_size = static_cast<int32_t>(std::min<double>(static_cast<double>(overlaps.size()), max_alignments)) / num_alignment_engines;
    }
    batch_size = std::max(batch_size, 1);
    std::cerr << "Aligning " << overlaps.size() << " overlaps (" << max_query_size << "x" << max_target_size << ") with batch size "
              << batch_size << std::endl;

    int32_t device_id = 0;
    GW_CU_CHECK_ERR(hipGetDevice(&device_id));
    int32_t overlap_idx = 0;
    std::mutex overlap_idx_mtx;
    // several engines on host threads: one engine's copies overlap another's kernels
    std::vector<std::future<void>> align_futures;
    for (int32_t t = 0; t < num_alignment_engines; t++)
        align_futures.push_back(std::async(std::launch::async, &run_alignment_batch, allocator, std::ref(overlap_idx_mtx),
                                           std::cref(overlaps), std::cref(queries), std::cref(targets), std::ref(overlap_idx),
                                           max_query_size, max_target_size, std::ref(cigars), batch_size, device_id));
    std::exception_ptr first_error;
    for (auto& f : align_futures)
    {
        try
        {
            f.get();
        }
        catch (...)
        {
            if (!first_error) first_error = std::current_exception();
        }
    }
    if (first_error) std::rethrow_exception(first_error);
}

void print_paf(const std::vector<Overlap>& overlaps, const std::vector<std::string>& cigars,
               const std::vector<FastaSequence>& queries, const std::vector<FastaSequence>& targets, int32_t kmer_size,
               std::FILE* out)
{
    if (!cigars.empty() && cigars.size() != overlaps.size()) throw std::invalid_argument("one CIGAR per overlap (or none)");
    std::string buffer;
    buffer.reserve(150 * overlaps.size());
    char num[160];
    for (size_t i = 0; i < overlaps.size(); ++i)
    {
        const Overlap& o          = overlaps[i];
        const FastaSequence& q    = queries.at(o.query_read_id_);
        const FastaSequence& t    = targets.at(o.target_read_id_);
        const int64_t approx_len  = std::max(std::llabs(static_cast<int64_t>(o.target_start_position_in_read_) - static_cast<int64_t>(o.target_end_position_in_read_)),
                                            std::llabs(static_cast<int64_t>(o.query_start_position_in_read_) - static_cast<int64_t>(o.query_end_position_in_read_)));
        buffer += q.name;
        std::snprintf(num, sizeof(num), "\t%zu\t%u\t%u\t%c\t", q.seq.length(), o.query_start_position_in_read_,
                      o.query_end_position_in_read_, static_cast<unsigned char>(o.relative_strand));
        buffer += num;
        buffer += t.name;
        std::snprintf(num, sizeof(num), "\t%zu\t%u\t%u\t%u\t%" PRId64 "\t%i", t.seq.length(), o.target_start_position_in_read_,
                      o.target_end_position_in_read_, o.num_residues_ * static_cast<std::uint32_t>(kmer_size), approx_len, 255);
        buffer += num;
        if (!cigars.empty())
        {
            buffer += "\tcg:Z:";
            buffer += cigars[i];
        }
        buffer += '\n';
    }
    std::fwrite(buffer.data(), 1, buffer.size(), out);
}

void print_sam(const std::vector<Overlap>& overlaps, const std::vector<std::string>& cigars,
               const std::vector<FastaSequence>& queries, const std::vector<FastaSequence>& targets,
               const std::string& program_version, const std::string& command_line, std::FILE* out)
{
    if (!cigars.empty() && cigars.size() != overlaps.size()) throw std::invalid_argument("one CIGAR per overlap (or none)");
    std::string buffer;
    char num[96];
    // header: @SQ of every target read an overlap names, in order of first appearance; @PG
    std::vector<char> listed(targets.size(), 0);
    for (const Overlap& o : overlaps)
    {
        const FastaSequence& t = targets.at(o.target_read_id_);
        if (listed[o.target_read_id_]) continue;
        listed[o.target_read_id_] = 1;
        buffer += "@SQ\tSN:";
        buffer += t.name;
        std::snprintf(num, sizeof(num), "\tLN:%zu\n", t.seq.length());
        buffer += num;
    }
    buffer += "@PG\tID:cudamapper\tPN:cudamapper\tVN:" + program_version;
    if (!command_line.empty()) buffer += "\tCL:" + command_line;
    buffer += '\n';
    for (size_t i = 0; i < overlaps.size(); ++i)
    {
        const Overlap& o       = overlaps[i];
        const FastaSequence& q = queries.at(o.query_read_id_);
        const FastaSequence& t = targets.at(o.target_read_id_);
        buffer += q.name;
        std::snprintf(num, sizeof(num), "\t%d\t", o.relative_strand == RelativeStrand::Reverse ? 16 : 0);
        buffer += num;
        buffer += t.name;
        std::snprintf(num, sizeof(num), "\t%u\t255\t", o.target_start_position_in_read_ + 1u); // SAM positions are 1-based
        buffer += num;
        // SEQ is the whole read on the strand it aligns in (flag 16: its reverse complement), and the CIGAR covers only
        // query_start..query_end of it: the unaligned ends become soft clips, so that the CIGAR's query length equals
        // len(SEQ) as the SAM specification asks (the reference hands htslib the forward read and the bare CIGAR,
        // cudamapper/src/utils.cpp:253-300; samtools rejects such a record for partial overlaps)
        const bool reverse  = o.relative_strand == RelativeStrand::Reverse;
        const bool has_cg   = !cigars.empty() && !cigars[i].empty();
        if (has_cg)
        {
            const int64_t qlen  = static_cast<int64_t>(q.seq.length());
            const int64_t head  = std::max<int64_t>(0, std::min<int64_t>(o.query_start_position_in_read_, qlen));
            const int64_t tail  = std::max<int64_t>(0, qlen - std::min<int64_t>(o.query_end_position_in_read_, qlen));
            const int64_t left  = reverse ? tail : head;
            const int64_t right = reverse ? head : tail;
            if (left > 0)
            {
                std::snprintf(num, sizeof(num), "%" PRId64 "S", left);
                buffer += num;
            }
            // cudaaligner's 'I' is a base present in the TARGET only and its 'D' one present in the query only
            // (cudaaligner.hpp:47-53), the opposite of SAM's operators, which are named from the reference sequence's side
            // (RNAME = the target read): swapped here, in the SAM text only (the PAF cg:Z: tag stays cudaaligner's)
            {
                const size_t at = buffer.size();
                buffer += cigars[i];
                for (size_t k = at; k < buffer.size(); ++k)
                    buffer[k] = buffer[k] == 'I' ? 'D' : (buffer[k] == 'D' ? 'I' : buffer[k]);
            }
            if (right > 0)
            {
                std::snprintf(num, sizeof(num), "%" PRId64 "S", right);
                buffer += num;
            }
        }
        else
            buffer += '*';
        buffer += "\t*\t0\t0\t";
        if (q.seq.empty())
            buffer += '*';
        else if (!reverse)
            buffer += q.seq;
        else
        {
            const size_t at = buffer.size();
            buffer.append(q.seq.rbegin(), q.seq.rend());
            for (size_t k = at; k < buffer.size(); ++k)
            {
                switch (buffer[k])
                {
                case 'A': buffer[k] = 'T'; break;
                case 'C': buffer[k] = 'G'; break;
                case 'G': buffer[k] = 'C'; break;
                case 'T': buffer[k] = 'A'; break;
                case 'a': buffer[k] = 't'; break;
                case 'c': buffer[k] = 'g'; break;
                case 'g': buffer[k] = 'c'; break;
                case 't': buffer[k] = 'a'; break;
                default: break; // N and the IUPAC codes that are their own complement stay
                }
            }
        }
        buffer += "\t*\n";
    }
    std::fwrite(buffer.data(), 1, buffer.size(), out);
}

} // namespace cudamapper
} // namespace genomeworks
} // namespace claraparabricks
