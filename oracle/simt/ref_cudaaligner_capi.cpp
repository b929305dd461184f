// ref_cudaaligner_capi.cpp -- TEST INFRASTRUCTURE (oracle/simt): a flat C interface over the REFERENCE's cudaaligner library -- its
// own aligner*.cpp, myers_gpu.cu, hirschberg_myers_gpu.cu, ukkonen_gpu.cu ... compiled by g++ from /root/reference where they lie
// and run on the CPU by the SIMT emulator of simt.hpp (oracle/Makefile.ref, target ref_cudaaligner_simt ->
// oracle/_ref/libref_cudaaligner_simt.so). This file only calls the reference's API: the create_aligner factories of
// cudaaligner/include/.../aligner.hpp (default = Hirschberg + Myers; fixed band = banded Myers) and the constructors of the two
// classes its factory does not hand out (AlignerGlobalUkkonen, AlignerGlobalMyers: cudaaligner/src/*.hpp, as its tests do).
// Used by tests/ref_cudaaligner.py to check the aligner oracles and to write tests/golden/reference_simt_alignments.json.gz.
#include <claraparabricks/genomeworks/cudaaligner/aligner.hpp>
#include <claraparabricks/genomeworks/cudaaligner/alignment.hpp>
#include <claraparabricks/genomeworks/cudaaligner/cudaaligner.hpp>
#include <claraparabricks/genomeworks/utils/allocator.hpp>

#include "aligner_global_hirschberg_myers.hpp"
#include "aligner_global_myers.hpp"
#include "aligner_global_ukkonen.hpp"

#include <cstring>
#include <memory>
#include <vector>

using namespace claraparabricks::genomeworks;
using namespace claraparabricks::genomeworks::cudaaligner;

struct RefAligner
{
    std::unique_ptr<Aligner> aligner;
    std::vector<std::vector<int8_t>> states;
    std::vector<int> optimal, status;
};

#pragma GCC visibility push(default)
extern "C" {

// kind: 0 = create_aligner(max_query, max_target, max_alignments, global_alignment, ...) (the default: Hirschberg + Myers),
//       1 = AlignerGlobalUkkonen, 2 = AlignerGlobalMyers, 3 = AlignerGlobalHirschbergMyers (constructed directly)
void* ref_aligner_create(int kind, int max_query, int max_target, int max_alignments)
{
    try
    {
        auto* h = new RefAligner;
        if (kind == 0)
            h->aligner = create_aligner(max_query, max_target, max_alignments, AlignmentType::global_alignment, nullptr, 0, int64_t(256) << 20);
        else
        {
            DefaultDeviceAllocator allocator = create_default_device_allocator(int64_t(256) << 20);
            if (kind == 1) h->aligner = std::make_unique<AlignerGlobalUkkonen>(max_query, max_target, max_alignments, allocator, nullptr, 0);
            if (kind == 2) h->aligner = std::make_unique<AlignerGlobalMyers>(max_query, max_target, max_alignments, allocator, nullptr, 0);
            if (kind == 3) h->aligner = std::make_unique<AlignerGlobalHirschbergMyers>(max_query, max_target, max_alignments, allocator, nullptr, 0);
        }
        return h;
    }
    catch (...)
    {
        return nullptr;
    }
}

// create_aligner(global_alignment, max_bandwidth, stream, device, max_device_memory): banded Myers
void* ref_aligner_create_banded(int max_bandwidth, long long max_device_memory)
{
    try
    {
        auto* h    = new RefAligner;
        h->aligner = create_aligner(AlignmentType::global_alignment, max_bandwidth, nullptr, 0, max_device_memory);
        return h;
    }
    catch (...)
    {
        return nullptr;
    }
}

void ref_aligner_destroy(void* handle) { delete static_cast<RefAligner*>(handle); }

int ref_aligner_add(void* handle, const char* query, int query_length, const char* target, int target_length)
{
    try
    {
        return static_cast<int>(static_cast<RefAligner*>(handle)->aligner->add_alignment(query, query_length, target, target_length));
    }
    catch (...)
    {
        return -1;
    }
}

// align_all() + sync_alignments(); -> number of alignments, or -1 - status
int ref_aligner_run(void* handle)
{
    RefAligner* h = static_cast<RefAligner*>(handle);
    try
    {
        StatusType s = h->aligner->align_all();
        if (s != StatusType::success) return -1 - static_cast<int>(s);
        s = h->aligner->sync_alignments();
        if (s != StatusType::success) return -1 - static_cast<int>(s);
        h->states.clear(), h->optimal.clear(), h->status.clear();
        for (const std::shared_ptr<Alignment>& a : h->aligner->get_alignments())
        {
            std::vector<int8_t> st;
            for (AlignmentState x : a->get_alignment()) st.push_back(static_cast<int8_t>(x));
            // the banded aligner hands out run-length results (get_actions() / get_runlengths(), alignment_impl.hpp:104-127)
            const std::vector<int8_t>& actions  = a->get_actions();
            const std::vector<int32_t>& lengths = a->get_runlengths();
            if (st.empty())
                for (size_t r = 0; r < actions.size(); ++r) st.insert(st.end(), static_cast<size_t>(lengths[r]), actions[r]);
            h->states.push_back(st);
            h->optimal.push_back(a->is_optimal() ? 1 : 0);
            h->status.push_back(static_cast<int>(a->get_status()));
        }
        return static_cast<int>(h->states.size());
    }
    catch (...)
    {
        return -1000;
    }
}

int ref_aligner_states_length(void* handle, int i) { return static_cast<int>(static_cast<RefAligner*>(handle)->states[static_cast<size_t>(i)].size()); }
void ref_aligner_states(void* handle, int i, signed char* out)
{
    const std::vector<int8_t>& s = static_cast<RefAligner*>(handle)->states[static_cast<size_t>(i)];
    std::memcpy(out, s.data(), s.size());
}
int ref_aligner_is_optimal(void* handle, int i) { return static_cast<RefAligner*>(handle)->optimal[static_cast<size_t>(i)]; }
int ref_aligner_alignment_status(void* handle, int i) { return static_cast<RefAligner*>(handle)->status[static_cast<size_t>(i)]; }
void ref_aligner_reset(void* handle) { static_cast<RefAligner*>(handle)->aligner->reset(); }
}
#pragma GCC visibility pop
