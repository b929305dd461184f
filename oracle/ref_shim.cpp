// ref_shim.cpp -- extern "C" doors onto the REFERENCE's own CPU aligner functions, compiled together with the
// reference sources (oracle/Makefile.ref) into oracle/_ref/libref_aligner.so. TEST INFRASTRUCTURE ONLY.
// Nothing here restates an algorithm: it only adapts std::string / std::vector to plain pointers.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <claraparabricks/genomeworks/utils/mathutils.hpp> // myers_cpu.hpp uses ceiling_divide without including it

#include "needleman_wunsch_cpu.hpp"
#include "ukkonen_cpu.hpp"
#include "myers_cpu.hpp"

using namespace claraparabricks::genomeworks::cudaaligner;

extern "C" {

// needleman_wunsch_cpu(text=target, query): per-cell AlignmentState path (cudaaligner/src/needleman_wunsch_cpu.cpp:139-191)
int32_t ref_needleman_wunsch_cpu(const char* target, int32_t tlen, const char* query, int32_t qlen, int8_t* out, int32_t cap)
{
    std::vector<int8_t> r = needleman_wunsch_cpu(std::string(target, tlen), std::string(query, qlen));
    if ((int32_t)r.size() > cap) return -1;
    std::memcpy(out, r.data(), r.size());
    return (int32_t)r.size();
}

// exact edit distance by the reference's CPU Myers (cudaaligner/src/myers_cpu.hpp:79-131)
int32_t ref_myers_edit_distance(const char* target, int32_t tlen, const char* query, int32_t qlen)
{
    return myers_compute_edit_distance(std::string(target, tlen), std::string(query, qlen));
}

// bottom-right entry of the naive NW matrix (needleman_wunsch_cpu.cpp:117-137)
int32_t ref_nw_edit_distance(const char* target, int32_t tlen, const char* query, int32_t qlen)
{
    auto m = needleman_wunsch_build_score_matrix_naive(std::string(target, tlen), std::string(query, qlen));
    return m(m.num_rows() - 1, m.num_cols() - 1);
}

int32_t ref_ukkonen_cpu(const char* target, int32_t tlen, const char* query, int32_t qlen, int32_t p, int8_t* out, int32_t cap)
{
    std::vector<int8_t> r = ukkonen_cpu(std::string(target, tlen), std::string(query, qlen), p);
    if ((int32_t)r.size() > cap) return -1;
    std::memcpy(out, r.data(), r.size());
    return (int32_t)r.size();
}
}
