// gwhip_poa_hooks.hip -- unit-test entry points that expose each POA device function on caller-provided
// arrays, mirroring the reference's test wrappers (runNW cudapoa_nw.cuh:499, runNWbanded
// cudapoa_nw_banded.cuh:608, runNWbandedTB cudapoa_nw_tb_banded.cuh:727, runTopSort cudapoa_topsort.cuh:220,
// addAlignment cudapoa_add_alignment.cuh:335, generateConsensusTestHost cudapoa_generate_consensus.cuh:395)
// so the reference's inline known-answer vectors apply 1:1. Arrays use the reference layout with SizeT=int32.
#include <hip/hip_runtime.h>

#include <string>

#include "poa_device.h"
#include "poa_full_device.h"
#include "poa_graph_device.h"
#include "poa_tb_device.h"

namespace gwhip
{
extern thread_local std::string g_last_error;

struct NwHookArgs
{
    gwhip_poa_config cfg;
    gwhip_poa_test_graph g;
    const uint8_t* read;
    int32_t read_length;
    uint8_t* scratch;
    size_t rowinfo_off, scores_off, trace_off;
    size_t scores_elems, trace_elems;
    int32_t scores_width;
    int32_t* alignment_graph;
    int32_t* alignment_read;
    int32_t* aligned_nodes;
};

static GraphView<int32_t> __device__ view_of(const gwhip_poa_test_graph& t)
{
    GraphView<int32_t> g{};
    g.nodes               = const_cast<uint8_t*>(t.nodes);
    g.sorted_poa          = const_cast<int32_t*>(t.graph);
    g.node_id_to_pos      = const_cast<int32_t*>(t.node_id_to_pos);
    g.incoming_edge_count = const_cast<uint16_t*>(t.incoming_edge_count);
    g.incoming_edges      = const_cast<int32_t*>(t.incoming_edges);
    g.outgoing_edge_count = const_cast<uint16_t*>(t.outgoing_edge_count);
    g.outgoing_edges      = const_cast<int32_t*>(t.outgoing_edges);
    return g;
}

template <typename TraceT>
__global__ __launch_bounds__(kWave) void nw_hook_kernel(NwHookArgs a)
{
    __shared__ __attribute__((aligned(16))) int16_t ring[4224];
    GraphView<int32_t> g      = view_of(a.g);
    RowInfo<false>* rowinfo = reinterpret_cast<RowInfo<false>*>(a.scratch + a.rowinfo_off);
    int16_t* scores           = reinterpret_cast<int16_t*>(a.scratch + a.scores_off);
    TraceT* trace             = reinterpret_cast<TraceT*>(a.scratch + a.trace_off);
    const gwhip_poa_config& c = a.cfg;
    const int lane            = threadIdx.x;
    build_rowinfo<int32_t, RowInfo<false>>(g, a.g.graph_count, rowinfo, lane);
    wave_sync();
    uint64_t cells  = 0;
    PhaseClock pc{nullptr, 0};
    const float buf = __fmul_rn((float)c.max_nodes_per_graph, (float)c.matrix_sequence_dimension);
    int32_t n;
    switch (c.band_mode)
    {
    case GWHIP_FULL_BAND:
        n = nw_full<int16_t, int32_t, RowInfo<false>>(g, rowinfo, a.g.graph_count, a.read, a.read_length, scores, a.scores_width, ring,
                                      (int32_t)sizeof(ring), a.alignment_graph, a.alignment_read, c.gap_score,
                                      c.mismatch_score, c.match_score, cells);
        break;
    case GWHIP_STATIC_BAND:
        n = nw_banded<int16_t, int32_t, RowInfo<false>, false, false>(g, rowinfo, a.g.graph_count, a.read, nullptr, a.read_length, scores, ring,
                                               (int32_t)sizeof(ring), buf, a.alignment_graph, a.alignment_read,
                                               c.alignment_band_width, c.gap_score, c.mismatch_score, c.match_score, 0, cells, pc);
        break;
    case GWHIP_ADAPTIVE_BAND:
        n = nw_banded<int16_t, int32_t, RowInfo<false>, true, false>(g, rowinfo, a.g.graph_count, a.read, nullptr, a.read_length, scores, ring,
                                              (int32_t)sizeof(ring), buf, a.alignment_graph, a.alignment_read,
                                              c.alignment_band_width, c.gap_score, c.mismatch_score, c.match_score, 0, cells, pc);
        break;
    case GWHIP_STATIC_BAND_TRACEBACK:
        n = nw_banded_tb<int16_t, int32_t, RowInfo<false>, TraceT, false>(g, rowinfo, a.g.graph_count, a.read, a.read_length, scores,
                                                          a.scores_elems, trace, a.trace_elems, buf, a.alignment_graph,
                                                          a.alignment_read, c.alignment_band_width,
                                                          c.max_banded_pred_distance, c.gap_score, c.mismatch_score,
                                                          c.match_score, 0, cells);
        break;
    default:
        n = nw_banded_tb<int16_t, int32_t, RowInfo<false>, TraceT, true>(g, rowinfo, a.g.graph_count, a.read, a.read_length, scores,
                                                         a.scores_elems, trace, a.trace_elems, buf, a.alignment_graph,
                                                         a.alignment_read, c.alignment_band_width,
                                                         c.max_banded_pred_distance, c.gap_score, c.mismatch_score,
                                                         c.match_score, 0, cells);
        break;
    }
    if (lane == 0) *a.aligned_nodes = n;
}

__global__ void topsort_hook_kernel(int32_t* sorted_poa, int32_t* node_id_to_pos, int32_t node_count,
                                    const uint16_t* incoming_edge_count, const int32_t* outgoing_edges,
                                    const uint16_t* outgoing_edge_count, uint16_t* local_cnt)
{
    topsort_kahn<int32_t>(sorted_poa, node_id_to_pos, node_count, incoming_edge_count, outgoing_edges,
                          outgoing_edge_count, local_cnt);
}

__global__ void add_alignment_hook_kernel(GraphView<int32_t> g, int32_t* node_count, int32_t alignment_length,
                                          const int32_t* alignment_graph, const uint8_t* read,
                                          const int32_t* alignment_read, const int8_t* base_weights,
                                          int32_t max_nodes_per_graph, int32_t* status)
{
    int32_t new_count = *node_count;
    uint8_t st = add_alignment_to_graph<int32_t, false>(new_count, g, *node_count, alignment_length, alignment_graph, read,
                                                        alignment_read, base_weights, nullptr, 0, 0,
                                                        (uint32_t)max_nodes_per_graph);
    *status = st;
    if (st == 0) *node_count = new_count;
}

__global__ void consensus_hook_kernel(GraphView<int32_t> g, int32_t node_count, int32_t* predecessors, int32_t* scores,
                                      uint8_t* consensus, uint16_t* coverage, int32_t max_consensus_size)
{
    generate_consensus<int32_t>(g, node_count, predecessors, scores, consensus, coverage, max_consensus_size);
}

static int hook_fail(hipError_t e, const char* what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return (int)e;
}

static void nw_hook_plan(const gwhip_poa_config& c, NwHookArgs& a, size_t& total)
{
    const size_t mn = (size_t)c.max_nodes_per_graph;
    size_t off      = 0;
    auto take       = [&](size_t b) { size_t o = off; off = gw_align_up(off + b, 256); return o; };
    a.rowinfo_off   = take((mn + 2) * sizeof(RowInfo<false>));
    const bool tb   = c.band_mode == GWHIP_STATIC_BAND_TRACEBACK || c.band_mode == GWHIP_ADAPTIVE_BAND_TRACEBACK;
    size_t width    = (size_t)c.matrix_sequence_dimension;
    if (c.band_mode == GWHIP_FULL_BAND)
    {
        size_t w2 = (size_t)((c.max_sequence_size + 1 + kCellsPerLane + 3) & ~3);
        width     = width > w2 ? width : w2;
    }
    a.scores_width = (int32_t)width;
    if (tb)
    {
        a.scores_elems = (size_t)c.max_banded_pred_distance * width;
        a.trace_elems  = mn * width;
    }
    else
    {
        a.scores_elems = mn * width;
        a.trace_elems  = 0;
    }
    a.scores_off = take(a.scores_elems * 2 + 64);
    a.trace_off  = take(a.trace_elems * 2 + 64);
    total        = off;
}

} // namespace gwhip

using namespace gwhip;

extern "C" {

size_t gwhip_poa_test_nw_scratch_bytes(const gwhip_poa_config* cfg)
{
    NwHookArgs a{};
    size_t total = 0;
    nw_hook_plan(*cfg, a, total);
    return total;
}

int gwhip_poa_test_nw(const gwhip_poa_config* cfg, const gwhip_poa_test_graph* g, const uint8_t* read,
                      int32_t read_length, void* scratch, int32_t* alignment_graph, int32_t* alignment_read,
                      int32_t* aligned_nodes, gwhip_stream_t stream)
{
    NwHookArgs a{};
    size_t total = 0;
    nw_hook_plan(*cfg, a, total);
    a.cfg             = *cfg;
    a.g               = *g;
    a.read            = read;
    a.read_length     = read_length;
    a.scratch         = (uint8_t*)scratch;
    a.alignment_graph = alignment_graph;
    a.alignment_read  = alignment_read;
    a.aligned_nodes   = aligned_nodes;
    if (cfg->trace16)
        hipLaunchKernelGGL(nw_hook_kernel<int16_t>, dim3(1), dim3(kWave), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(nw_hook_kernel<int8_t>, dim3(1), dim3(kWave), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hook_fail(e, "nw_hook_kernel launch");
}

int gwhip_poa_test_topsort(int32_t* sorted_poa, int32_t* node_id_to_pos, int32_t node_count,
                           const uint16_t* incoming_edge_count, const int32_t* outgoing_edges,
                           const uint16_t* outgoing_edge_count, uint16_t* local_incoming_edge_count,
                           gwhip_stream_t stream)
{
    hipLaunchKernelGGL(topsort_hook_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sorted_poa, node_id_to_pos,
                       node_count, incoming_edge_count, outgoing_edges, outgoing_edge_count, local_incoming_edge_count);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hook_fail(e, "topsort_hook_kernel launch");
}

int gwhip_poa_test_add_alignment(uint8_t* nodes, int32_t* node_count, int32_t* node_alignments,
                                 uint16_t* node_alignment_count, int32_t* incoming_edges,
                                 uint16_t* incoming_edge_count, int32_t* outgoing_edges,
                                 uint16_t* outgoing_edge_count, uint16_t* incoming_edge_w, int32_t alignment_length,
                                 const int32_t* alignment_graph, const uint8_t* read, const int32_t* alignment_read,
                                 uint16_t* node_coverage_counts, const int8_t* base_weights,
                                 int32_t max_nodes_per_graph, int32_t* status, gwhip_stream_t stream)
{
    GraphView<int32_t> g{};
    g.nodes                = nodes;
    g.node_alignments      = node_alignments;
    g.node_alignment_count = node_alignment_count;
    g.incoming_edges       = incoming_edges;
    g.incoming_edge_count  = incoming_edge_count;
    g.outgoing_edges       = outgoing_edges;
    g.outgoing_edge_count  = outgoing_edge_count;
    g.incoming_edge_w      = incoming_edge_w;
    g.coverage             = node_coverage_counts;
    hipLaunchKernelGGL(add_alignment_hook_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, g, node_count,
                       alignment_length, alignment_graph, read, alignment_read, base_weights, max_nodes_per_graph, status);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hook_fail(e, "add_alignment_hook_kernel launch");
}

int gwhip_poa_test_consensus(const uint8_t* nodes, int32_t node_count, const int32_t* graph,
                             const int32_t* node_id_to_pos, const int32_t* incoming_edges,
                             const uint16_t* incoming_edge_count, const int32_t* outgoing_edges,
                             const uint16_t* outgoing_edge_count, const uint16_t* incoming_edge_w,
                             int32_t* predecessors, int32_t* scores, uint8_t* consensus, uint16_t* coverage,
                             const uint16_t* node_coverage_counts, const int32_t* node_alignments,
                             const uint16_t* node_alignment_count, int32_t max_consensus_size, gwhip_stream_t stream)
{
    GraphView<int32_t> g{};
    g.nodes                = const_cast<uint8_t*>(nodes);
    g.sorted_poa           = const_cast<int32_t*>(graph);
    g.node_id_to_pos       = const_cast<int32_t*>(node_id_to_pos);
    g.incoming_edges       = const_cast<int32_t*>(incoming_edges);
    g.incoming_edge_count  = const_cast<uint16_t*>(incoming_edge_count);
    g.outgoing_edges       = const_cast<int32_t*>(outgoing_edges);
    g.outgoing_edge_count  = const_cast<uint16_t*>(outgoing_edge_count);
    g.incoming_edge_w      = const_cast<uint16_t*>(incoming_edge_w);
    g.coverage             = const_cast<uint16_t*>(node_coverage_counts);
    g.node_alignments      = const_cast<int32_t*>(node_alignments);
    g.node_alignment_count = const_cast<uint16_t*>(node_alignment_count);
    // `scores` needs node_count + 1 entries: element 0 is the guard in front of the per-node scores
    hipLaunchKernelGGL(consensus_hook_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, g, node_count, predecessors,
                       scores, consensus, coverage, max_consensus_size);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hook_fail(e, "consensus_hook_kernel launch");
}

} // extern "C"
