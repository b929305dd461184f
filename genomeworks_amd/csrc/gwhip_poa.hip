// gwhip_poa.hip -- kernels and C-ABI launchers of the POA hot path (gfx950 only).
// Replaces generatePOA() (cudapoa/src/cudapoa_kernels.cuh:544-1076): one wavefront per window builds the
// graph read by read; a second kernel extracts consensus (or MSA) -- see include/gwhip.h.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "poa_device.h"
#include "poa_full_device.h"
#include "poa_graph_device.h"
#include "poa_tb_device.h"

namespace gwhip
{

// This file is compiled nine times: as it stands (everything but the graph-build kernel's instantiations) and, through
// gwhip_poa_part{0..7}.hip, with GWHIP_POA_PART defined -- parts 0..3 once per (score type, id type) pair, parts 4 and 5 for
// the packed passes of band 128 and of bands 384 / 512 (launch_packed_variant below), part 6 for the traceback-buffer kernels
// with 16-bit scores, ids and traces, part 7 for the full band with 16-bit scores. Those translation units hold nothing but
// their poa_window_kernel instantiations and their launcher, so the heavy compilations run in parallel.
#ifndef GWHIP_POA_PART
thread_local std::string g_last_error;

static int fail(hipError_t e, const char* what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return (int)e;
}
static int fail_msg(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}
#endif

constexpr int kRingBytes   = 8448;  // LDS ring of recent score rows: 16 rows of a 256-band int16 row (264 x 2 B)
constexpr int kRowInfoLds  = 3074;  // rows of the LDS row table (covers max_nodes_per_graph <= 3072)
constexpr int kRowInfoBytes = kRowInfoLds * 8;
constexpr int kReadLds     = 2048;  // LDS copy of the current read (+ read-ahead slack)
constexpr int kCodeTileLds = 4096;  // LDS tile of trace codes for the traceback (64 rows x 64 columns)
static_assert(kReadLds + kCodeTileLds >= 3072 * 2, "the incremental topsort keeps the previous order (uint16 x 3072) in the read + code-tile regions");
// Graphs that do not fit the LDS tables (long reads: HBM row table, 32-bit cells, bands up to 1536 columns) spend their
// LDS on the forward pass instead: a ring of the most recent score rows wide enough for 5 rows of the widest band,
// the band starts of those rows, and a sliding window of the read. 4 blocks per CU still fit (4 x 35 KB).
constexpr int kBsRingBytes   = 2048; // band starts of the ring rows (64 x 4 B) + 64 staged rows of the HBM row table (64 x 24 B)
constexpr int kReadWinBytes  = 4096;
constexpr int kMwLds         = 768;  // arguments (128 B) and hand-over entries of the multi-wave forward pass (poa_device.h)
constexpr size_t kReservedCuLds = 128 * 1024; // LDS of a multi-wave block (160 KB per CU: nothing else of this kernel family, >= 35 KB per block, fits beside it)

struct KernelArgs
{
    gwhip_poa_config cfg;
    PoaLayout L;
    int32_t total_windows;
    const uint8_t* sequences;
    const int8_t* base_weights;
    int32_t* sequence_lengths;
    const gwhip_window_details* window_details;
    uint8_t* consensus;
    uint16_t* coverage;
    uint8_t* msa;
    uint8_t* workspace;
    uint8_t* full_scores; // full band: variable-width score regions after the slabs
    uint64_t* cells;
    uint64_t* phase_cycles; // optional [windows][kPhCount]
    int32_t debug_flags;    // profiling ablations (GWHIP_DEBUG env var); 0 in production
    int32_t cons_lds_nodes; // node capacity of the consensus kernel's LDS tables (poa_graph_device.h)
    int32_t wide_ring_bytes; // HBM-table layout: bytes of the LDS ring of score rows (the other regions follow it)
    uint32_t* work_counters; // persistent grid (fewer blocks than windows): [0] windows handed out beyond the first gridDim.x, [1] blocks done
};

template <typename IdT>
__device__ GraphView<IdT> carve_graph(uint8_t* slab, const PoaLayout& L)
{
    GraphView<IdT> g;
    g.nodes                = slab + L.nodes;
    g.incoming_edge_count  = (uint16_t*)(slab + L.in_cnt);
    g.outgoing_edge_count  = (uint16_t*)(slab + L.out_cnt);
    g.node_alignment_count = (uint16_t*)(slab + L.aln_cnt);
    g.coverage             = (uint16_t*)(slab + L.coverage);
    g.sorted_poa           = (IdT*)(slab + L.sorted);
    g.node_id_to_pos       = (IdT*)(slab + L.pos);
    g.local_cnt            = (uint16_t*)(slab + L.local_cnt);
    g.incoming_edges       = (IdT*)(slab + L.in_edges);
    g.incoming_edge_w      = (uint16_t*)(slab + L.in_w);
    g.outgoing_edges       = (IdT*)(slab + L.out_edges);
    g.node_alignments      = (IdT*)(slab + L.aligned);
    g.cons_scores          = (int32_t*)(slab + L.cons_scores);
    g.cons_pred            = (IdT*)(slab + L.cons_pred);
    g.marks                = slab + L.marks;
    g.check                = slab + L.check;
    g.to_visit             = (IdT*)(slab + L.to_visit);
    g.out_cov              = (uint16_t*)(slab + L.out_cov);
    g.out_cov_cnt          = (uint16_t*)(slab + L.out_cov_cnt);
    g.msa_pos              = (IdT*)(slab + L.msa_pos);
    g.seq_begin            = (IdT*)(slab + L.seq_begin);
    return g;
}

// ------------------------------------------------------------------------------------------------
// Graph-build kernel: grid = windows, block = one wavefront.
// ------------------------------------------------------------------------------------------------
// LDS_TABLES: the per-row table, the topsort working set and the current read live in LDS. It is a template
// flag (not a runtime select) so every pointer has one address space and the row loop issues ds_* ops only:
// a flat/global load in that loop would wait on the previous row's score store (shared in-order vmcnt).
// NW: wavefronts per window. 1 everywhere but for graphs beyond the LDS tables in the adaptive band mode (long reads),
// where the wide bands' forward pass is a pipeline of wavefronts over 256-column blocks (generic_forward_skew); every other
// phase is wave 0's, the helper wavefronts wait at a barrier in between.
// DBG: the instantiation that honours GWHIP_DEBUG selectors and the per-phase cycle accounting. Production launches
// (no selector, no phase buffer) run DBG = false, in which every selector test and profiling hook folds away.
// VARIANT: 0 = production, alignment_band_width 256 (and every width without a packed pass); 1 = debug (below); 2 = production
// for alignment_band_width 128 (the packed pass with the band in lanes 0..31 next to the 256-column one an adaptive band may
// widen to: a separate instantiation, so that the metric configuration's kernel carries none of its code -- with both in one
// kernel the headline lost 1 %); 3 = production for alignment_band_width 384 / 512 (the two-pass packed pass,
// poa_forward_moves_wide.h); 4 = debug for those widths; 5 / 6 = production / debug of the FULL band with int16 scores (the
// packed pass of poa_forward_moves_full.h in front of the generic nw_full; gwhip_poa_part7.hip).
template <typename ScoreT, typename IdT, typename TraceT, int BM, bool MSA, bool LDS_TABLES, int NW, int VARIANT>
__global__ __launch_bounds__(kWave * NW) void poa_window_kernel(KernelArgs a)
{
    constexpr bool DBG = VARIANT == 1 || VARIANT == 4 || VARIANT == 6;
    constexpr int PV   = VARIANT == 0 ? 0 : (VARIANT <= 2 ? 1 : 2); // packed passes of nw_banded
    constexpr bool kFullPacked = (VARIANT == 5 || VARIANT == 6) && BM == GWHIP_FULL_BAND && LDS_TABLES && std::is_same<ScoreT, int16_t>::value;
    const int32_t debug_flags = DBG ? a.debug_flags : 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane       = threadIdx.x & (kWave - 1);
    const int wave       = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // One block per window -- or, when the launcher put fewer blocks than windows on the device (a persistent grid of one block
    // per SIMD, single-wave kernels only), window after window from a shared counter: block b starts with window b.
    for (int32_t w = blockIdx.x; w < a.total_windows;)
    {
    const gwhip_poa_config& c = a.cfg;
    uint8_t* slab        = a.workspace + (size_t)w * a.L.per_window;
    GraphView<IdT> g     = carve_graph<IdT>(slab, a.L);
    const gwhip_window_details wd = a.window_details[w];
    int32_t* seq_lens    = a.sequence_lengths + wd.seq_len_buffer_offset;
    const uint8_t* sequence    = a.sequences + wd.seq_starts;
    const int8_t* base_weights = a.base_weights + wd.seq_starts;
    uint8_t* consensus   = a.consensus + (size_t)w * c.max_consensus_size;
    int32_t* alignment_graph = (int32_t*)(slab + a.L.align_graph);
    int32_t* alignment_read  = (int32_t*)(slab + a.L.align_read);

    // LDS carve: [ring | rowinfo (aliased by the topsort working set) | read]
    ScoreT* ring = reinterpret_cast<ScoreT*>(smem);
    uint8_t* lds_rowinfo_region = smem + kRingBytes;
    uint8_t* lds_read_buf       = smem + kRingBytes + kRowInfoBytes;
    uint8_t* lds_code_tile      = LDS_TABLES ? smem + kRingBytes + kRowInfoBytes + kReadLds : nullptr;
    const int32_t ring_bytes    = LDS_TABLES ? kRingBytes : a.wide_ring_bytes;
    int32_t* lds_bs_ring        = LDS_TABLES ? nullptr : reinterpret_cast<int32_t*>(smem + ring_bytes);
    uint8_t* lds_read_window    = LDS_TABLES ? nullptr : smem + ring_bytes + kBsRingBytes;
    uint8_t* codes              = a.L.codes ? slab + a.L.codes : nullptr;
    constexpr bool graph_fits_lds = LDS_TABLES;
    using RowT = RowInfo<LDS_TABLES>;
    RowT* rowinfo;
    if constexpr (LDS_TABLES)
        rowinfo = reinterpret_cast<RowT*>(smem + kRingBytes);
    else
        rowinfo = reinterpret_cast<RowT*>(slab + a.L.rowinfo);

    constexpr bool TB = (BM == GWHIP_STATIC_BAND_TRACEBACK || BM == GWHIP_ADAPTIVE_BAND_TRACEBACK);
    ScoreT* scores;
    uint8_t* full_moves = nullptr; // full band, int16 scores: one move byte per cell behind the window's score matrix
    if (BM == GWHIP_FULL_BAND)
    {
        // the window's region: scores_width x max_nodes score cells (+ as many move bytes with int16 scores, full_score_bytes)
        const size_t cell_bytes = sizeof(ScoreT) + (sizeof(ScoreT) == 2 ? 1 : 0);
        uint8_t* region         = a.full_scores + (size_t)wd.scores_offset * (size_t)c.max_nodes_per_graph * cell_bytes;
        scores                  = reinterpret_cast<ScoreT*>(region);
        if (sizeof(ScoreT) == 2) full_moves = region + (size_t)wd.scores_width * (size_t)c.max_nodes_per_graph * sizeof(ScoreT);
    }
    else
        scores = reinterpret_cast<ScoreT*>(slab + a.L.scores);
    TraceT* traceback = TB ? reinterpret_cast<TraceT*>(slab + a.L.trace) : nullptr;
    const float banded_buffer_size = __fmul_rn((float)c.max_nodes_per_graph, (float)c.matrix_sequence_dimension);
    MwArgs<ScoreT>* mw_args = nullptr;
    MwShared* mw_shared     = nullptr;
    if constexpr (NW > 1)
    {
        // Issue priority by the window's weight (its bases: 2^16 .. 2^19 and more -> 0 .. 3). A launch -- or several
        // concurrent launches of different size classes -- lasts as long as its heaviest window, one chain of dependent
        // steps; when lighter windows share its SIMDs, the arbiter should serve the heavy one first.
        {
            int32_t bases = lane < (int32_t)wd.num_seqs ? seq_lens[lane] : 0;
            for (int off = 32; off > 0; off >>= 1) bases += __shfl_xor(bases, off);
            bases = __builtin_amdgcn_readfirstlane(bases);
            const int32_t level = bases >= (1 << 19) ? 3 : (bases >= (1 << 18) ? 2 : (bases >= (1 << 17) ? 1 : 0));
            if (level == 3) __builtin_amdgcn_s_setprio(3);
            else if (level == 2) __builtin_amdgcn_s_setprio(2);
            else if (level == 1) __builtin_amdgcn_s_setprio(1);
        }
        static_assert(!LDS_TABLES, "the multi-wave forward pass belongs to the HBM-table layout");
        mw_args   = reinterpret_cast<MwArgs<ScoreT>*>(smem + ring_bytes + kBsRingBytes + kReadWinBytes);
        mw_shared = reinterpret_cast<MwShared*>(smem + ring_bytes + kBsRingBytes + kReadWinBytes + 128);
        if (wave != 0)
        {
            // helper wavefronts: parked at the barrier until wave 0 reaches a wide-band forward pass (or the end)
            for (;;)
            {
                block_barrier();
                const MwArgs<ScoreT> A = *mw_args;
                if (A.op != 1) return;
                generic_forward_skew<ScoreT, IdT, RowT, DBG>(A, g, rowinfo, ring, mw_shared, wave, lane);
            }
        }
    }

    // ---- backbone from read 0 (cudapoa_kernels.cuh:200-238), lanes in parallel ----
    const int32_t len0 = seq_lens[0];
    for (int32_t i = lane; i < len0; i += kWave)
    {
        g.nodes[i]                = sequence[i];
        g.sorted_poa[i]           = (IdT)i;
        g.node_id_to_pos[i]       = (IdT)i;
        g.node_alignment_count[i] = 0;
        g.coverage[i]             = 1;
        g.outgoing_edge_count[i]  = (i == len0 - 1) ? 0 : 1;
        // per-node record of the incremental topsort: queue length 1 | out-degree << 4 | in-degree << 10
        g.local_cnt[i] = (uint16_t)(1 | ((i == len0 - 1 ? 0 : 1) << 4) | ((i == 0 ? 0 : 1) << 10));
        if (i < len0 - 1)
        {
            g.outgoing_edges[(int64_t)i * kEdges] = (IdT)(i + 1);
            if (MSA)
            {
                g.out_cov[(int64_t)i * kEdges * c.max_sequences_per_poa] = 0;
                g.out_cov_cnt[(int64_t)i * kEdges]                       = 1;
            }
        }
        if (i == 0)
        {
            g.incoming_edge_count[0] = 0;
            g.incoming_edge_w[0]     = (uint16_t)base_weights[0];
            if (MSA) g.seq_begin[0] = 0;
        }
        else
        {
            g.incoming_edges[(int64_t)i * kEdges]  = (IdT)(i - 1);
            g.incoming_edge_w[(int64_t)i * kEdges] = (uint16_t)(base_weights[i - 1] + base_weights[i]);
            g.incoming_edge_count[i]               = 1;
        }
    }
    if (lane == 0) consensus[0] = 0;
    uint64_t cells = 0;
    int32_t node_count = len0;
    uint64_t phase_acc[kPhCount] = {0, 0, 0, 0, 0, 0};
    PhaseClock pc{(DBG && a.phase_cycles) ? phase_acc : nullptr, 0};
    wave_sync();
    pc.start();

    for (int32_t s = 1; s < (int32_t)wd.num_seqs; s++)
    {
        const int32_t seq_len = seq_lens[s];
        const int32_t adv     = ((s == 1 ? len0 : seq_lens[s - 1]) + 3) & ~3; // :248-249
        sequence += adv;
        base_weights += adv;

        if (node_count >= c.max_nodes_per_graph) // :253-265
        {
            if (lane == 0) { consensus[0] = kKernelError; consensus[1] = kNodeCountExceeded; }
            break;
        }
        pc.tick(kPhOther);
        // rows with 4..6 predecessors keep predecessors 3..5 in an LDS side table (the trace-code tile region is free
        // until the traceback)
        build_rowinfo<IdT, RowT>(g, node_count, rowinfo, lane, LDS_TABLES ? reinterpret_cast<uint64_t*>(lds_code_tile) : nullptr);
        // stage the read (plus the never-consumed read-ahead) in LDS when it fits
        constexpr bool LDS_READ = LDS_TABLES && BM != GWHIP_FULL_BAND; // (the traceback-buffer modes: for their packed pass)
        const uint8_t* lds_read = lds_read_buf;
        if constexpr (LDS_READ)
        {
            // adaptive bands may widen to 1536 columns: the staged span covers max(read, band) + read-ahead
            const int32_t stage_bytes = min(kReadLds, ((max(seq_len, kMaxAdaptiveBand) + 8) + 3) & ~3);
            for (int32_t i = lane * 4; i < stage_bytes; i += kWave * 4)
                *reinterpret_cast<uint32_t*>(lds_read_buf + i) = *reinterpret_cast<const uint32_t*>(sequence + i);
        }
        wave_sync();
        pc.tick(kPhRowInfo);

        int32_t alen = 0;
        // traceback-buffer modes with int16 scores, an int16 trace region and the tables in LDS: the packed pass first
        // (poa_forward_moves_tb.h); it declines what it cannot reproduce and the memory-faithful routine takes that read
        constexpr bool kTbPacked = TB && LDS_TABLES && std::is_same<ScoreT, int16_t>::value && std::is_same<TraceT, int16_t>::value;
        bool tb_handled = false;
        if constexpr (kTbPacked)
        {
            constexpr bool kAd = BM == GWHIP_ADAPTIVE_BAND_TRACEBACK;
            const bool adaptive_now = kAd && c.alignment_band_width < kMaxAdaptiveBand; // else the static routine, as below
            auto packed = [&](auto ad_tag, int32_t rerun) -> int32_t {
                return nw_banded_tb_packed<IdT, decltype(ad_tag)::value>(g, rowinfo, node_count, lds_read, seq_len, reinterpret_cast<int16_t*>(scores),
                                                                         a.L.scores_elems, reinterpret_cast<int16_t*>(traceback), a.L.trace_elems,
                                                                         banded_buffer_size, alignment_graph, alignment_read, c.alignment_band_width,
                                                                         c.max_banded_pred_distance, c.gap_score, c.mismatch_score, c.match_score, rerun,
                                                                         cells, smem, kRingBytes, reinterpret_cast<const uint64_t*>(lds_code_tile),
                                                                         debug_flags, tb_handled);
            };
            if (adaptive_now)
                alen = packed(std::integral_constant<bool, kAd>{}, 0);
            else
                alen = packed(std::false_type{}, 0);
            // a band-edge rerun doubles the band beyond the packed widths: the routine below handles it (and nothing else of
            // this read) -- tb_handled stays true, alen carries the shift
        }
        if (tb_handled && !(alen == kShiftLeft || alen == kShiftRight)) {}
        else if (BM == GWHIP_ADAPTIVE_BAND_TRACEBACK && c.alignment_band_width < kMaxAdaptiveBand)
        {
            if (tb_handled)
                alen = nw_banded_tb<ScoreT, IdT, RowT, TraceT, true>(g, rowinfo, node_count, sequence, seq_len, scores, a.L.scores_elems,
                                                               traceback, a.L.trace_elems, banded_buffer_size, alignment_graph,
                                                               alignment_read, c.alignment_band_width, c.max_banded_pred_distance,
                                                               c.gap_score, c.mismatch_score, c.match_score, alen, cells);
            else
            alen = nw_banded_tb<ScoreT, IdT, RowT, TraceT, true>(g, rowinfo, node_count, sequence, seq_len, scores, a.L.scores_elems,
                                                           traceback, a.L.trace_elems, banded_buffer_size, alignment_graph,
                                                           alignment_read, c.alignment_band_width, c.max_banded_pred_distance,
                                                           c.gap_score, c.mismatch_score, c.match_score, 0, cells);
            if (alen == kShiftLeft || alen == kShiftRight)
                alen = nw_banded_tb<ScoreT, IdT, RowT, TraceT, true>(g, rowinfo, node_count, sequence, seq_len, scores, a.L.scores_elems,
                                                               traceback, a.L.trace_elems, banded_buffer_size, alignment_graph,
                                                               alignment_read, c.alignment_band_width, c.max_banded_pred_distance,
                                                               c.gap_score, c.mismatch_score, c.match_score, alen, cells);
        }
        else if (TB)
        {
            alen = nw_banded_tb<ScoreT, IdT, RowT, TraceT, false>(g, rowinfo, node_count, sequence, seq_len, scores, a.L.scores_elems,
                                                            traceback, a.L.trace_elems, banded_buffer_size, alignment_graph,
                                                            alignment_read, c.alignment_band_width, c.max_banded_pred_distance,
                                                            c.gap_score, c.mismatch_score, c.match_score, 0, cells);
        }
        else if (BM == GWHIP_ADAPTIVE_BAND && c.alignment_band_width < kMaxAdaptiveBand)
        {
            alen = nw_banded<ScoreT, IdT, RowT, true, LDS_READ, PV>(g, rowinfo, node_count, sequence, lds_read, seq_len, scores, ring, ring_bytes,
                                                banded_buffer_size, alignment_graph, alignment_read, c.alignment_band_width,
                                                c.gap_score, c.mismatch_score, c.match_score, 0, cells, pc, debug_flags, codes, lds_code_tile, lds_read_window, lds_bs_ring,
                                                mw_args, mw_shared);
            if (alen == kShiftLeft || alen == kShiftRight)
                alen = nw_banded<ScoreT, IdT, RowT, true, LDS_READ, PV>(g, rowinfo, node_count, sequence, lds_read, seq_len, scores, ring, ring_bytes,
                                                    banded_buffer_size, alignment_graph, alignment_read,
                                                    c.alignment_band_width, c.gap_score, c.mismatch_score, c.match_score,
                                                    alen, cells, pc, debug_flags, codes, lds_code_tile, lds_read_window, lds_bs_ring,
                                                    mw_args, mw_shared);
        }
        else if (BM == GWHIP_STATIC_BAND || BM == GWHIP_ADAPTIVE_BAND)
        {
            alen = nw_banded<ScoreT, IdT, RowT, false, LDS_READ, PV>(g, rowinfo, node_count, sequence, lds_read, seq_len, scores, ring, ring_bytes,
                                                 banded_buffer_size, alignment_graph, alignment_read, c.alignment_band_width,
                                                 c.gap_score, c.mismatch_score, c.match_score, 0, cells, pc, debug_flags, codes, lds_code_tile, lds_read_window, lds_bs_ring);
        }
        else
        {
            bool full_handled = false;
            if constexpr (kFullPacked)
                alen = nw_full_packed<IdT>(g, rowinfo, node_count, sequence, seq_len, reinterpret_cast<int16_t*>(scores), wd.scores_width, full_moves,
                                           smem, kRingBytes, reinterpret_cast<const uint64_t*>(lds_code_tile), alignment_graph, alignment_read,
                                           c.gap_score, c.mismatch_score, c.match_score, cells, debug_flags, full_handled, pc);
            if (!full_handled)
                alen = nw_full<ScoreT, IdT, RowT>(g, rowinfo, node_count, sequence, seq_len, scores, wd.scores_width, ring, kRingBytes,
                                            alignment_graph, alignment_read, c.gap_score, c.mismatch_score, c.match_score, cells);
        }
        // SizeT alignment_length in the reference: the value is narrowed to SizeT (cudapoa_kernels.cuh:268)
        alen = (int32_t)(IdT)alen;

        uint8_t err = 0;
        if (alen == kNwLoopFailed) err = kLoopCountExceeded;
        else if (alen == kNwAdaptiveStorageFailed) err = kExceededAdaptiveBandedMatrixSize;
        else if (TB && alen == kNwTracebackBufferFailed) err = kExceededMaximumPredecessorDistance;
        else if (alen == kNwPipelineFailed || alen == kNwScoreWrapped) err = kGenericError;
        if (err)
        {
            if (lane == 0) { consensus[0] = kKernelError; consensus[1] = err; }
            break;
        }

        pc.tick(kPhTraceback);
        int32_t status_and_count = 0;
        int32_t par_rc           = -1; // -1: use the serial merge
        if constexpr (!LDS_TABLES && !TB && BM != GWHIP_FULL_BAND)
        {
            // graphs beyond the LDS tables (long reads): same lane-parallel merge, its scratch in the score matrix
            // (dead until the next read's forward pass); 2 x read length + node bitset always fit in it
            int32_t new_count   = 0;
            int32_t* scratch    = reinterpret_cast<int32_t*>(scores);
            const int32_t lpad  = (seq_len + 63) & ~63;
            par_rc = add_alignment_parallel<IdT, MSA, int32_t>(new_count, g, node_count, alen, alignment_graph, alignment_read,
                                                               sequence, base_weights, seq_len, MSA ? g.seq_begin + s : nullptr,
                                                               (uint16_t)s, (uint32_t)c.max_sequences_per_poa,
                                                               c.max_nodes_per_graph, scratch, scratch + lpad,
                                                               reinterpret_cast<uint32_t*>(scratch + 2 * lpad), lane,
                                                               debug_flags, pc.acc ? &pc.acc[kPhOther] : nullptr);
            if (par_rc == 0)
            {
                if (lane == 0) seq_lens[0] = new_count; // :506
                status_and_count = new_count;
            }
            else if (par_rc > 0)
            {
                if (lane == 0) { consensus[0] = kKernelError; consensus[1] = (uint8_t)par_rc; }
                status_and_count = -1;
            }
        }
        if constexpr (LDS_TABLES)
        {
            int32_t new_count = 0;
            par_rc = add_alignment_parallel<IdT, MSA>(new_count, g, node_count, alen, alignment_graph, alignment_read,
                                                      sequence, base_weights, seq_len, MSA ? g.seq_begin + s : nullptr,
                                                      (uint16_t)s, (uint32_t)c.max_sequences_per_poa,
                                                      c.max_nodes_per_graph, reinterpret_cast<int16_t*>(smem),
                                                      reinterpret_cast<int16_t*>(smem) + 2048,
                                                      reinterpret_cast<uint32_t*>(lds_rowinfo_region), lane, debug_flags,
                                                      pc.acc ? &pc.acc[kPhOther] : nullptr);
            if (par_rc == 0)
            {
                if (lane == 0) seq_lens[0] = new_count; // :506
                status_and_count = new_count;
            }
            else if (par_rc > 0)
            {
                if (lane == 0) { consensus[0] = kKernelError; consensus[1] = (uint8_t)par_rc; }
                status_and_count = -1;
            }
        }
        if (lane == 0 && par_rc < 0)
        {
            int32_t new_count = 0;
            uint8_t e = add_alignment_to_graph<IdT, MSA>(new_count, g, node_count, alen, alignment_graph, sequence,
                                                         alignment_read, base_weights, MSA ? g.seq_begin + s : nullptr,
                                                         (uint16_t)s, (uint32_t)c.max_sequences_per_poa,
                                                         (uint32_t)c.max_nodes_per_graph);
            if (e != 0)
            {
                consensus[0]     = kKernelError;
                consensus[1]     = e;
                status_and_count = -1;
            }
            else
            {
                seq_lens[0]      = new_count; // :506
                status_and_count = new_count;
            }
        }
        pc.tick(kPhAddAlignment);
        status_and_count = wave_first(status_and_count);
        // graphs beyond the LDS tables in the banded score-matrix modes: Kahn order with an LDS cache of node records
        // filled along the previous order (the ring and the score matrix are idle here)
        constexpr bool kCachedSort = !LDS_TABLES && !TB && BM != GWHIP_FULL_BAND;
        bool sorted_here = false;
        if constexpr (kCachedSort)
        {
            // incremental order (replay of the previous run in blocks) when the ids fit its 17-bit fields and the score
            // matrix holds its working set (24 bytes per node); GWHIP_DEBUG bit 17 selects the cached full re-sort instead
            // (A/B: the choice must be the same for every read of a window, the two keep different things in local_cnt)
            const bool incr_ok = c.max_nodes_per_graph <= 131071 && !(debug_flags & (1 << 17)) &&
                                 (int64_t)a.L.scores_elems * (int64_t)sizeof(ScoreT) >= (int64_t)c.max_nodes_per_graph * 24 + 2048;
            if (status_and_count >= 0 && !c.spoa_accurate && !(debug_flags & (1 << 21)) && incr_ok)
            {
                wave_sync();
                // hot state in LDS (the idle score ring) when the graph's counters fit beside the window of the previous
                // order; GWHIP_DEBUG bit 16: the HBM routine only (A/B)
                bool done = false;
                if (!(debug_flags & (1 << 16)) && topsort_incr_cnt8_lds_bytes(status_and_count) <= ring_bytes)
                    done = topsort_kahn_incr_cnt8<IdT>(g, node_count, status_and_count, smem, reinterpret_cast<int32_t*>(scores), lane);
                if (!done) topsort_kahn_incr_hbm<IdT>(g, node_count, status_and_count, reinterpret_cast<int32_t*>(scores), lane);
                sorted_here = true;
            }
            else if (status_and_count >= 0 && !c.spoa_accurate && !(debug_flags & (1 << 21)) &&
                     (int64_t)a.L.scores_elems * (int64_t)sizeof(ScoreT) >= (int64_t)(2 * ((node_count + 63) & ~63)) * 4)
            {
                wave_sync();
                topsort_kahn_cached<IdT>(g, node_count, status_and_count, smem, reinterpret_cast<int32_t*>(scores), lane);
                sorted_here = true;
            }
        }
        if (lane == 0 && status_and_count >= 0 && !sorted_here)
        {
            const int32_t new_count = status_and_count;
            if (c.spoa_accurate)
                topsort_racon<IdT>(g, new_count, (int32_t)(uint16_t)c.max_nodes_per_graph); // (uint16_t) cast :519
            else if (!graph_fits_lds)
                topsort_kahn<IdT>(g.sorted_poa, g.node_id_to_pos, new_count, g.incoming_edge_count,
                                  g.outgoing_edges, g.outgoing_edge_count, g.local_cnt);
        }
        wave_sync();
        if (status_and_count >= 0 && !c.spoa_accurate && graph_fits_lds)
        {
            if constexpr (LDS_TABLES)
            {
                if (!(debug_flags & (1 << 21))) // GWHIP_DEBUG bit 21: full re-sort after every read (A/B switch)
                    topsort_kahn_incr_lds<IdT>(g, node_count, status_and_count, lds_rowinfo_region, smem, lds_read_buf, lane,
                                               debug_flags, pc.acc ? &pc.acc[kPhOther] : nullptr);
                else
                    topsort_kahn_lds<IdT>(g, status_and_count, lds_rowinfo_region, smem, lane, debug_flags, pc.acc ? &pc.acc[kPhOther] : nullptr);
            }
        }
        pc.tick(kPhTopsort);
        if (status_and_count < 0) break;
        node_count = status_and_count;
    }
    if (lane == 0 && a.cells) a.cells[w] = cells;
    if (DBG && lane == 0 && a.phase_cycles)
        for (int k = 0; k < kPhCount; k++) a.phase_cycles[(size_t)w * kPhCount + k] = phase_acc[k];
    if constexpr (NW > 1)
    {
        if (lane == 0) mw_args->op = 2; // the helper wavefronts leave
        block_barrier();
        break;
    }
    else
    {
        if ((int32_t)gridDim.x >= a.total_windows || a.work_counters == nullptr) break; // one block per window
        wave_sync();
        int32_t next = 0;
        if (lane == 0) next = (int32_t)gridDim.x + (int32_t)atomicAdd(a.work_counters, 1u);
        w = wave_first(next);
    }
    }
    if constexpr (NW == 1)
        if ((int32_t)gridDim.x < a.total_windows && a.work_counters != nullptr && threadIdx.x == 0)
        {
            // the last block to leave puts the counters back to zero for the next launch (every other block has made its last
            // request by then)
            __threadfence();
            if (atomicAdd(a.work_counters + 1, 1u) == gridDim.x - 1)
            {
                a.work_counters[0] = 0;
                a.work_counters[1] = 0;
                __threadfence();
            }
        }
}

#ifndef GWHIP_POA_PART
// ------------------------------------------------------------------------------------------------
// Consensus kernel: one wavefront per window (the reference runs one THREAD per window,
// cudapoa_generate_consensus.cuh:286-354, 512 per block => 2 blocks for 1024 windows).
// ------------------------------------------------------------------------------------------------
template <typename IdT>
__global__ __launch_bounds__(kWave) void poa_consensus_kernel(KernelArgs a)
{
    const int32_t w    = blockIdx.x;
    const gwhip_poa_config& c = a.cfg;
    uint8_t* consensus = a.consensus + (size_t)w * c.max_consensus_size;
    if (consensus[0] == kKernelError) return;
    uint8_t* slab      = a.workspace + (size_t)w * a.L.per_window;
    GraphView<IdT> g   = carve_graph<IdT>(slab, a.L);
    const int32_t n    = a.sequence_lengths[a.window_details[w].seq_len_buffer_offset];
    if (threadIdx.x == 0)
        generate_consensus<IdT>(g, n, g.cons_pred, g.cons_scores, consensus, a.coverage + (size_t)w * c.max_consensus_size,
                                c.max_consensus_size);
}

// LDS flavour (16-bit ids, <= 3072 nodes): generate_consensus_lds
__global__ __launch_bounds__(kWave) void poa_consensus_lds_kernel(KernelArgs a)
{
    extern __shared__ __align__(16) uint8_t cons_smem[];
    const int32_t w    = blockIdx.x;
    const gwhip_poa_config& c = a.cfg;
    uint8_t* consensus = a.consensus + (size_t)w * c.max_consensus_size;
    if (consensus[0] == kKernelError) return;
    uint8_t* slab          = a.workspace + (size_t)w * a.L.per_window;
    GraphView<int16_t> g   = carve_graph<int16_t>(slab, a.L);
    const int32_t n        = a.sequence_lengths[a.window_details[w].seq_len_buffer_offset];
    if (n <= a.cons_lds_nodes)
        generate_consensus_lds<int16_t>(g, n, cons_smem, a.cons_lds_nodes, consensus, a.coverage + (size_t)w * c.max_consensus_size,
                                        c.max_consensus_size, threadIdx.x & (kWave - 1),
                                        (a.debug_flags & 16) != 0); // GWHIP_DEBUG bit 4: the first bundle pass node by node (A/B)
    else if (threadIdx.x == 0) // a graph beyond the LDS tables of this launch: the serial HBM routine
        generate_consensus<int16_t>(g, n, g.cons_pred, g.cons_scores, consensus, a.coverage + (size_t)w * c.max_consensus_size,
                                    c.max_consensus_size);
}

// MSA kernel: lane 0 does the racon topsort + column assignment, then one lane per sequence
// (loops when num_seqs > 64; the reference launches max_sequences_per_poa threads, cudapoa_kernels.cuh:1025-1026).
constexpr int kMsaStackEntries = 4096; // LDS stack of the wave-wide racon order (its depth is a few dozen on real graphs)

template <typename IdT>
__global__ __launch_bounds__(kWave) void poa_msa_kernel(KernelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t msa_smem[]; // [stack | one state byte per node], or nothing
    __shared__ int32_t msa_length;
    const int32_t w    = blockIdx.x;
    const gwhip_poa_config& c = a.cfg;
    uint8_t* consensus = a.consensus + (size_t)w * c.max_consensus_size;
    if (consensus[0] == kKernelError) return;
    uint8_t* slab      = a.workspace + (size_t)w * a.L.per_window;
    GraphView<IdT> g   = carve_graph<IdT>(slab, a.L);
    const gwhip_window_details wd = a.window_details[w];
    const int32_t n    = a.sequence_lengths[wd.seq_len_buffer_offset];
    const int lane     = threadIdx.x & (kWave - 1);
    bool done          = false;
    if (a.cons_lds_nodes >= n && n > 0) // the launch gave this kernel LDS for graphs of up to cons_lds_nodes nodes
    {
        uint32_t* stack = reinterpret_cast<uint32_t*>(msa_smem);
        uint8_t* state  = msa_smem + (size_t)kMsaStackEntries * 4;
        if (topsort_racon_wave<IdT>(g, n, state, stack, kMsaStackEntries, lane))
        {
            const int32_t len = node_id_to_msa_pos_wave<IdT>(g, n, state, lane);
            if (lane == 0) msa_length = len;
            done = true;
        }
    }
    if (!done && threadIdx.x == 0)
    {
        // static_cast<SizeT>(max_nodes_per_graph), cudapoa_generate_msa.cuh:196
        topsort_racon<IdT>(g, n, (int32_t)(IdT)c.max_nodes_per_graph);
        msa_length = node_id_to_msa_pos<IdT>(g, n);
    }
    wave_sync();
    if (threadIdx.x == 0 && (uint32_t)msa_length >= (uint32_t)c.max_consensus_size)
    {
        consensus[0] = kKernelError;
        consensus[1] = kExceededMaximumSequenceSize;
    }
    wave_sync();
    if (consensus[0] == kKernelError) return;
    uint8_t* msa = a.msa + (size_t)w * c.max_sequences_per_poa * c.max_consensus_size;
    if (done) // wave mode: one lane per node scatters into the rows
        generate_msa_rows_wave<IdT>(g, n, (int32_t)wd.num_seqs, msa, msa_length, (uint32_t)c.max_sequences_per_poa,
                                    (uint32_t)c.max_consensus_size, lane);
    else
        for (int32_t s = threadIdx.x; s < (int32_t)wd.num_seqs; s += kWave)
            generate_msa_row<IdT>(g, (uint16_t)s, msa, msa_length, (uint32_t)c.max_sequences_per_poa, (uint32_t)c.max_consensus_size);
}

template <typename IdT>
__global__ void poa_export_graph_kernel(KernelArgs a, int32_t first_window, uint8_t* nodes, int32_t* in_edges, uint16_t* in_w,
                                        uint16_t* in_cnt, int32_t* out_edges, uint16_t* out_cnt)
{
    const int32_t w  = first_window + blockIdx.x; // window in the batch
    const size_t o   = blockIdx.x;                // its slot in the output arrays
    const int32_t mn = a.cfg.max_nodes_per_graph;
    uint8_t* slab    = a.workspace + (size_t)w * a.L.per_window;
    GraphView<IdT> g = carve_graph<IdT>(slab, a.L);
    const int32_t n  = a.sequence_lengths[a.window_details[w].seq_len_buffer_offset];
    if (a.consensus[(size_t)w * a.cfg.max_consensus_size] == kKernelError) return;
    for (int32_t i = threadIdx.x; i < n && i < mn; i += blockDim.x)
    {
        nodes[o * mn + i]  = g.nodes[i];
        uint16_t ic        = g.incoming_edge_count[i];
        in_cnt[o * mn + i] = ic;
        for (int32_t e = 0; e < ic && e < kEdges; e++)
        {
            in_edges[(o * mn + i) * kEdges + e] = g.incoming_edges[(int64_t)i * kEdges + e];
            in_w[(o * mn + i) * kEdges + e]     = g.incoming_edge_w[(int64_t)i * kEdges + e];
        }
        if (out_edges && out_cnt)
        {
            uint16_t oc         = g.outgoing_edge_count[i];
            out_cnt[o * mn + i] = oc;
            for (int32_t e = 0; e < oc && e < kEdges; e++)
                out_edges[(o * mn + i) * kEdges + e] = g.outgoing_edges[(int64_t)i * kEdges + e];
        }
    }
}

#endif // GWHIP_POA_PART

// ------------------------------------------------------------------------------------------------
// host side of the C-ABI
// ------------------------------------------------------------------------------------------------
#ifndef GWHIP_DEVICE_ONLY // tools/isa_dump.sh compiles a single kernel instantiation without the host dispatch
#ifndef GWHIP_POA_PART
static size_t full_score_bytes(const gwhip_poa_config& c, int32_t windows, uint64_t sum_scores_width)
{
    if (c.band_mode != GWHIP_FULL_BAND) return 0;
    uint64_t per_window_width = (uint64_t)((c.max_sequence_size + 1 + kCellsPerLane + 3) & ~3);
    uint64_t width_sum        = sum_scores_width ? sum_scores_width : per_window_width * (uint64_t)windows;
    // int16 scores: + one move byte per cell (poa_forward_moves_full.h); the kernel carves a window's region the same way
    return (size_t)(width_sum * (uint64_t)c.max_nodes_per_graph * (c.score32 ? 4u : 3u)) + 256;
}

static bool validate(const gwhip_poa_args* args)
{
    const gwhip_poa_config& c = args->cfg;
    if (args->total_windows < 0 || c.max_nodes_per_graph <= 0 || c.max_consensus_size < 2) return false;
    if (c.band_mode < 0 || c.band_mode > GWHIP_ADAPTIVE_BAND_TRACEBACK) return false;
    if (c.band_mode != GWHIP_FULL_BAND && (c.alignment_band_width % 128 != 0 || c.alignment_band_width <= 0)) return false;
    if (!c.size32 && c.max_nodes_per_graph > 32767) return false;
    return true;
}
#endif // GWHIP_POA_PART

// The kernels of the packed int16 passes other than the metric configuration's -- VARIANT 2 (band 128), 3 and 4 (bands 384 /
// 512, production and debug) of poa_window_kernel<int16, int16, int8, static / adaptive band, MSA, LDS tables> -- are compiled
// in translation units of their own (gwhip_poa_part4.hip, gwhip_poa_part5.hip) behind this launcher.
template <int BM, bool MSA, int VARIANT>
hipError_t launch_packed_variant(const KernelArgs& ka, dim3 grid, size_t lds, hipStream_t stream);
#if defined(GWHIP_POA_PART) && GWHIP_POA_PART >= 4
template <int BM, bool MSA, int VARIANT>
hipError_t launch_packed_variant(const KernelArgs& ka, dim3 grid, size_t lds, hipStream_t stream)
{
    hipLaunchKernelGGL((poa_window_kernel<int16_t, int16_t, int8_t, BM, MSA, true, 1, VARIANT>), grid, dim3(kWave), lds, stream, ka);
    return hipGetLastError();
}
#if GWHIP_POA_PART == 7
// full band, int16 scores: the packed pass of poa_forward_moves_full.h (production and debug)
template hipError_t launch_packed_variant<GWHIP_FULL_BAND, false, 5>(const KernelArgs&, dim3, size_t, hipStream_t);
template hipError_t launch_packed_variant<GWHIP_FULL_BAND, true, 5>(const KernelArgs&, dim3, size_t, hipStream_t);
template hipError_t launch_packed_variant<GWHIP_FULL_BAND, false, 6>(const KernelArgs&, dim3, size_t, hipStream_t);
template hipError_t launch_packed_variant<GWHIP_FULL_BAND, true, 6>(const KernelArgs&, dim3, size_t, hipStream_t);
#elif GWHIP_POA_PART == 4 || GWHIP_POA_PART == 5
#if GWHIP_POA_PART == 4
#define GW_PACKED_VARIANTS(BM, MSA) template hipError_t launch_packed_variant<BM, MSA, 2>(const KernelArgs&, dim3, size_t, hipStream_t);
#else
#define GW_PACKED_VARIANTS(BM, MSA)                                                                       \
    template hipError_t launch_packed_variant<BM, MSA, 3>(const KernelArgs&, dim3, size_t, hipStream_t); \
    template hipError_t launch_packed_variant<BM, MSA, 4>(const KernelArgs&, dim3, size_t, hipStream_t);
#endif
GW_PACKED_VARIANTS(GWHIP_STATIC_BAND, false)
GW_PACKED_VARIANTS(GWHIP_STATIC_BAND, true)
GW_PACKED_VARIANTS(GWHIP_ADAPTIVE_BAND, false)
GW_PACKED_VARIANTS(GWHIP_ADAPTIVE_BAND, true)
#undef GW_PACKED_VARIANTS
#endif
#endif

template <typename ScoreT, typename IdT, typename TraceT, bool MSA, bool LDS_TABLES>
static hipError_t launch_window_kernel(const KernelArgs& ka_in, hipStream_t stream)
{
    KernelArgs ka = ka_in;
    constexpr size_t wide_rest = (size_t)kBsRingBytes + kReadWinBytes + kMwLds;
    // HBM-table layout, single-wave kernels: a ring of five rows of the widest band, four blocks per CU
    ka.wide_ring_bytes = 5 * (kMaxAdaptiveBand + kRightPad) * 4;
    const size_t lds   = LDS_TABLES ? (size_t)kRingBytes + kRowInfoBytes + kReadLds + kCodeTileLds : (size_t)ka.wide_ring_bytes + wide_rest;
    dim3 grid(ka.total_windows);
    // more windows than one wavefront per SIMD (LDS-table kernels: 39 KB of LDS = four one-wave blocks per CU): a persistent grid
    if (LDS_TABLES && ka.work_counters != nullptr)
    {
        static const int simds = [] {
            int cus = 0, dev = 0;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            return 4 * cus;
        }();
        if (ka.total_windows > simds) grid = dim3(simds);
    }
    if (grid.x >= (unsigned)ka.total_windows) ka.work_counters = nullptr;
#define GW_LAUNCH(BM)                                                                                              \
    {                                                                                                              \
        constexpr int NW = (!LDS_TABLES && BM == GWHIP_ADAPTIVE_BAND) ? kSkWaves : 1;                              \
        size_t lds_req = lds;                                                                                      \
        if (NW > 1) /* multi-wave blocks (long reads): the register count allows one 8-wave block per CU anyway; it   \
                       owns the CU's LDS and puts it to use as a ring of 20 rows of the widest band */             \
        {                                                                                                          \
            lds_req            = kReservedCuLds;                                                                   \
            ka.wide_ring_bytes = (int32_t)(kReservedCuLds - wide_rest);                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&poa_window_kernel<ScoreT, IdT, TraceT, BM, MSA, LDS_TABLES, NW, 0>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kReservedCuLds);             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&poa_window_kernel<ScoreT, IdT, TraceT, BM, MSA, LDS_TABLES, NW, 1>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kReservedCuLds);             \
        }                                                                                                          \
        constexpr bool kHasPacked = LDS_TABLES && std::is_same<ScoreT, int16_t>::value && std::is_same<TraceT, int8_t>::value && \
                                    (BM == GWHIP_STATIC_BAND || BM == GWHIP_ADAPTIVE_BAND);                        \
        const bool debug = ka.debug_flags != 0 || ka.phase_cycles != nullptr;                                      \
        const int32_t bw = ka.cfg.alignment_band_width;                                                            \
        bool launched    = false;                                                                                  \
        if constexpr (kHasPacked)                                                                                  \
        {                                                                                                          \
            constexpr int PBM = kHasPacked ? BM : GWHIP_STATIC_BAND;                                               \
            if (bw == 384 || bw == 512)                                                                            \
            {                                                                                                      \
                hipError_t pe = debug ? launch_packed_variant<PBM, MSA, 4>(ka, grid, lds_req, stream)             \
                                      : launch_packed_variant<PBM, MSA, 3>(ka, grid, lds_req, stream);            \
                if (pe != hipSuccess) return pe;                                                                   \
                launched = true;                                                                                   \
            }                                                                                                      \
            else if (bw == 128 && !debug)                                                                          \
            {                                                                                                      \
                hipError_t pe = launch_packed_variant<PBM, MSA, 2>(ka, grid, lds_req, stream);                     \
                if (pe != hipSuccess) return pe;                                                                   \
                launched = true;                                                                                   \
            }                                                                                                      \
        }                                                                                                          \
        if constexpr (BM == GWHIP_FULL_BAND && LDS_TABLES && std::is_same<ScoreT, int16_t>::value && std::is_same<IdT, int16_t>::value && \
                      std::is_same<TraceT, int8_t>::value)                                                         \
        {                                                                                                          \
            hipError_t pe = debug ? launch_packed_variant<GWHIP_FULL_BAND, MSA, 6>(ka, grid, lds_req, stream)      \
                                  : launch_packed_variant<GWHIP_FULL_BAND, MSA, 5>(ka, grid, lds_req, stream);     \
            if (pe != hipSuccess) return pe;                                                                       \
            launched = true;                                                                                       \
        }                                                                                                          \
        if (launched) {}                                                                                           \
        else if (debug)                                                                                            \
            hipLaunchKernelGGL((poa_window_kernel<ScoreT, IdT, TraceT, BM, MSA, LDS_TABLES, NW, 1>), grid, dim3(kWave * NW), lds_req, stream, ka); \
        else                                                                                                       \
            hipLaunchKernelGGL((poa_window_kernel<ScoreT, IdT, TraceT, BM, MSA, LDS_TABLES, NW, 0>), grid, dim3(kWave * NW), lds_req, stream, ka); \
    }                                                                                                              \
    break;
    switch (ka.cfg.band_mode)
    {
    case GWHIP_FULL_BAND: GW_LAUNCH(GWHIP_FULL_BAND)
    case GWHIP_STATIC_BAND: GW_LAUNCH(GWHIP_STATIC_BAND)
    case GWHIP_ADAPTIVE_BAND: GW_LAUNCH(GWHIP_ADAPTIVE_BAND)
    case GWHIP_STATIC_BAND_TRACEBACK: GW_LAUNCH(GWHIP_STATIC_BAND_TRACEBACK)
    default: GW_LAUNCH(GWHIP_ADAPTIVE_BAND_TRACEBACK)
    }
#undef GW_LAUNCH
    return hipGetLastError();
}

template <typename ScoreT, typename IdT, typename TraceT>
hipError_t launch_msa_split(const KernelArgs& ka, hipStream_t stream)
{
    // LDS tables need 16-bit ids, <= 3072 graph rows and a read (+ read-ahead) that fits the LDS copy
    constexpr bool can_lds = sizeof(IdT) == 2;
    const bool lds = can_lds && ka.cfg.max_nodes_per_graph + 2 <= kRowInfoLds && ka.cfg.max_sequence_size + 16 <= kReadLds;
    if (ka.cfg.output_mask & 2)
    {
        if constexpr (can_lds)
            if (lds) return launch_window_kernel<ScoreT, IdT, TraceT, true, true>(ka, stream);
        return launch_window_kernel<ScoreT, IdT, TraceT, true, false>(ka, stream);
    }
    if constexpr (can_lds)
        if (lds) return launch_window_kernel<ScoreT, IdT, TraceT, false, true>(ka, stream);
    return launch_window_kernel<ScoreT, IdT, TraceT, false, false>(ka, stream);
}

// the traceback-buffer kernels with int16 scores, ids and traces (the packed pass of poa_forward_moves_tb.h lives there)
// are compiled in a translation unit of their own (gwhip_poa_part6.hip)
#if defined(GWHIP_POA_PART) && GWHIP_POA_PART == 6
template hipError_t launch_msa_split<int16_t, int16_t, int16_t>(const KernelArgs&, hipStream_t);
#else
extern template hipError_t launch_msa_split<int16_t, int16_t, int16_t>(const KernelArgs&, hipStream_t);
#endif

template <typename ScoreT, typename IdT>
hipError_t launch_trace_split(const KernelArgs& ka, hipStream_t stream)
{
    // TraceT only matters in the traceback modes; the other modes share the int8_t instantiation
    const bool tb = ka.cfg.band_mode == GWHIP_STATIC_BAND_TRACEBACK || ka.cfg.band_mode == GWHIP_ADAPTIVE_BAND_TRACEBACK;
    if (tb && ka.cfg.trace16) return launch_msa_split<ScoreT, IdT, int16_t>(ka, stream);
    return launch_msa_split<ScoreT, IdT, int8_t>(ka, stream);
}
#ifdef GWHIP_POA_PART
#if GWHIP_POA_PART == 0
template hipError_t launch_trace_split<int32_t, int32_t>(const KernelArgs&, hipStream_t);
#elif GWHIP_POA_PART == 1
template hipError_t launch_trace_split<int32_t, int16_t>(const KernelArgs&, hipStream_t);
#elif GWHIP_POA_PART == 2
template hipError_t launch_trace_split<int16_t, int32_t>(const KernelArgs&, hipStream_t);
#elif GWHIP_POA_PART == 3
template hipError_t launch_trace_split<int16_t, int16_t>(const KernelArgs&, hipStream_t);
#endif
#else
extern template hipError_t launch_trace_split<int32_t, int32_t>(const KernelArgs&, hipStream_t);
extern template hipError_t launch_trace_split<int32_t, int16_t>(const KernelArgs&, hipStream_t);
extern template hipError_t launch_trace_split<int16_t, int32_t>(const KernelArgs&, hipStream_t);
extern template hipError_t launch_trace_split<int16_t, int16_t>(const KernelArgs&, hipStream_t);

static KernelArgs make_kernel_args(const gwhip_poa_args* args)
{
    KernelArgs ka{};
    ka.cfg             = args->cfg;
    ka.L               = make_poa_layout(args->cfg);
    ka.total_windows   = args->total_windows;
    ka.sequences       = args->sequences;
    ka.base_weights    = args->base_weights;
    ka.sequence_lengths = args->sequence_lengths;
    ka.window_details  = args->window_details;
    ka.consensus       = args->consensus;
    ka.coverage        = args->coverage;
    ka.msa             = args->msa;
    ka.workspace       = (uint8_t*)args->workspace;
    ka.full_scores     = ka.workspace + (size_t)args->total_windows * ka.L.per_window;
    ka.cells           = args->cells;
    ka.phase_cycles    = args->phase_cycles;
    ka.work_counters   = args->work_counters;
    if (const char* pg = std::getenv("GWHIP_POA_PERSISTENT")) // A/B: 0 = always one block per window
        if (pg[0] == '0') ka.work_counters = nullptr;
    {
        const char* dbg = std::getenv("GWHIP_DEBUG");
        ka.debug_flags  = dbg ? std::atoi(dbg) : 0;
    }
    return ka;
}
#endif // GWHIP_POA_PART

#endif // GWHIP_DEVICE_ONLY
} // namespace gwhip

#if !defined(GWHIP_DEVICE_ONLY) && !defined(GWHIP_POA_PART)
using namespace gwhip;

extern "C" {

size_t gwhip_poa_workspace_bytes(const gwhip_poa_config* cfg, int32_t windows, uint64_t sum_scores_width)
{
    PoaLayout L = make_poa_layout(*cfg);
    return (size_t)windows * L.per_window + full_score_bytes(*cfg, windows, sum_scores_width);
}

void gwhip_poa_bytes_per_window(const gwhip_poa_config* cfg, int64_t* per_poa, int64_t* per_matrix)
{
    PoaLayout L      = make_poa_layout(*cfg);
    const bool tb    = cfg->band_mode == GWHIP_STATIC_BAND_TRACEBACK || cfg->band_mode == GWHIP_ADAPTIVE_BAND_TRACEBACK;
    int64_t matrix   = (int64_t)cfg->matrix_sequence_dimension * (int64_t)cfg->max_nodes_per_graph *
                     (tb ? L.trace_bytes : L.score_bytes);
    if (cfg->band_mode == GWHIP_FULL_BAND)
    {
        // worst-case per-window score row: align4(max_sequence_size + 1 + 4) (cudapoa_batch.cuh:502)
        *per_poa    = (int64_t)L.per_window;
        *per_matrix = (int64_t)((cfg->max_sequence_size + 1 + kCellsPerLane + 3) & ~3) * (int64_t)cfg->max_nodes_per_graph *
                      (cfg->score32 ? 4 : 3); // int16 scores: + one move byte per cell (full_score_bytes)
    }
    else
    {
        *per_poa    = (int64_t)L.per_window - matrix;
        *per_matrix = matrix;
    }
    // outputs + inputs the host classes allocate next to the slab
    *per_poa += (int64_t)cfg->max_consensus_size * 3;
    if (cfg->output_mask & 2) *per_poa += (int64_t)cfg->max_consensus_size * cfg->max_sequences_per_poa;
    *per_poa += (int64_t)cfg->max_sequences_per_poa * cfg->max_sequence_size * 2 + (int64_t)cfg->max_sequences_per_poa * 4 + 32;
}

int32_t gwhip_poa_resident_windows(const gwhip_poa_config* cfg)
{
    if (cfg == nullptr) return 0;
    // the condition of launch_msa_split() for the LDS-table kernels (one 64-lane block per window, 39 KB of LDS: four per CU)
    const bool lds = !cfg->size32 && cfg->max_nodes_per_graph + 2 <= kRowInfoLds && cfg->max_sequence_size + 16 <= kReadLds;
    if (!lds) return 0;
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    {
        (void)hipGetLastError();
        cus = 256;
    }
    return 4 * cus;
}

int gwhip_poa_generate(const gwhip_poa_args* args, gwhip_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || !validate(args)) return fail_msg((int)hipErrorInvalidValue, "gwhip_poa_generate: invalid arguments");
    if (args->total_windows == 0) return 0;
    KernelArgs ka = make_kernel_args(args);
    if (!args->workspace || ((uintptr_t)args->workspace & 255) != 0)
        return fail_msg((int)hipErrorInvalidValue, "gwhip_poa_generate: workspace must be 256-byte aligned");
    if (args->workspace_bytes < gwhip_poa_workspace_bytes(&args->cfg, args->total_windows, 0))
        return fail_msg((int)hipErrorInvalidValue, "gwhip_poa_generate: workspace_bytes is smaller than gwhip_poa_workspace_bytes(cfg, total_windows, 0)");
    hipError_t e;
    const bool msa = (args->cfg.output_mask & 2) != 0;
    if (args->cfg.score32)
        e = args->cfg.size32 ? launch_trace_split<int32_t, int32_t>(ka, stream) : launch_trace_split<int32_t, int16_t>(ka, stream);
    else
        e = args->cfg.size32 ? launch_trace_split<int16_t, int32_t>(ka, stream) : launch_trace_split<int16_t, int16_t>(ka, stream);
    if (e != hipSuccess) return fail(e, "poa_window_kernel launch");
    if (args->event_after_graph_build)
    {
        e = hipEventRecord((hipEvent_t)args->event_after_graph_build, stream);
        if (e != hipSuccess) return fail(e, "hipEventRecord");
    }
    dim3 grid(args->total_windows), block(kWave);
    if (msa)
    {
        // wave-wide racon order with its marks in LDS when a byte per node fits (long-read graphs: one block per CU)
        const size_t msa_lds = (size_t)kMsaStackEntries * 4 + (((size_t)args->cfg.max_nodes_per_graph + 15) & ~size_t(15));
        const char* msa_dbg  = std::getenv("GWHIP_MSA_SERIAL"); // debugging: the serial HBM routine
        const bool msa_wave  = msa_lds <= 150 * 1024 && !(msa_dbg && msa_dbg[0] == '1');
        ka.cons_lds_nodes    = msa_wave ? args->cfg.max_nodes_per_graph : 0;
        if (args->cfg.size32)
        {
            if (msa_wave) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&poa_msa_kernel<int32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)msa_lds);
            hipLaunchKernelGGL(poa_msa_kernel<int32_t>, grid, block, msa_wave ? msa_lds : 0, stream, ka);
        }
        else
        {
            if (msa_wave) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&poa_msa_kernel<int16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)msa_lds);
            hipLaunchKernelGGL(poa_msa_kernel<int16_t>, grid, block, msa_wave ? msa_lds : 0, stream, ka);
        }
    }
    else
    {
        const char* cons_dbg = std::getenv("GWHIP_CONSENSUS_SERIAL"); // debugging: force the HBM-only kernel
        if (args->cfg.size32) hipLaunchKernelGGL(poa_consensus_kernel<int32_t>, grid, block, 0, stream, ka);
        else if (args->cfg.max_nodes_per_graph <= kConsLdsNodes && !(cons_dbg && cons_dbg[0] == '1'))
        {
            // 55 KB of LDS for the largest graph lets two blocks share a CU; with more windows than that holds at once
            // the tables are sized for 2176 nodes (39 KB, four blocks per CU) and larger graphs take the HBM routine
            int cus = 0, dev = 0;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            const int32_t cap = (args->cfg.max_nodes_per_graph > kConsLdsNodesSmall && (args->total_windows > 2 * cus || args->shared_device != 0) &&
                                 !(cons_dbg && cons_dbg[0] == '2'))
                                    ? kConsLdsNodesSmall
                                    : std::min<int32_t>(args->cfg.max_nodes_per_graph, kConsLdsNodes);
            ka.cons_lds_nodes = cap;
            hipLaunchKernelGGL(poa_consensus_lds_kernel, grid, block, cons_lds_bytes(cap), stream, ka);
        }
        else hipLaunchKernelGGL(poa_consensus_kernel<int16_t>, grid, block, 0, stream, ka);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return fail(e, "poa output kernel launch");
    return 0;
}

int gwhip_poa_export_graphs_range(const gwhip_poa_args* args, int32_t first_window, int32_t n_windows, uint8_t* nodes,
                                  int32_t* incoming_edges, uint16_t* incoming_edge_weights, uint16_t* incoming_edge_count,
                                  int32_t* outgoing_edges, uint16_t* outgoing_edge_count, gwhip_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || !validate(args) || first_window < 0 || n_windows < 0 || first_window + n_windows > args->total_windows)
        return fail_msg((int)hipErrorInvalidValue, "gwhip_poa_export_graphs: invalid arguments");
    if (n_windows == 0) return 0;
    if (args->workspace_bytes < gwhip_poa_workspace_bytes(&args->cfg, args->total_windows, 0))
        return fail_msg((int)hipErrorInvalidValue, "gwhip_poa_export_graphs: workspace_bytes is smaller than gwhip_poa_workspace_bytes(cfg, total_windows, 0)");
    KernelArgs ka = make_kernel_args(args);
    dim3 grid(n_windows), block(256);
    if (args->cfg.size32)
        hipLaunchKernelGGL(poa_export_graph_kernel<int32_t>, grid, block, 0, stream, ka, first_window, nodes, incoming_edges,
                           incoming_edge_weights, incoming_edge_count, outgoing_edges, outgoing_edge_count);
    else
        hipLaunchKernelGGL(poa_export_graph_kernel<int16_t>, grid, block, 0, stream, ka, first_window, nodes, incoming_edges,
                           incoming_edge_weights, incoming_edge_count, outgoing_edges, outgoing_edge_count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(e, "poa_export_graph_kernel launch");
    return 0;
}

int gwhip_poa_export_graphs(const gwhip_poa_args* args, uint8_t* nodes, int32_t* incoming_edges,
                            uint16_t* incoming_edge_weights, uint16_t* incoming_edge_count, int32_t* outgoing_edges,
                            uint16_t* outgoing_edge_count, gwhip_stream_t stream)
{
    if (!args) return fail_msg((int)hipErrorInvalidValue, "gwhip_poa_export_graphs: invalid arguments");
    return gwhip_poa_export_graphs_range(args, 0, args->total_windows, nodes, incoming_edges, incoming_edge_weights,
                                         incoming_edge_count, outgoing_edges, outgoing_edge_count, stream);
}

int gwhip_last_error_string(char* buf, size_t len)
{
    if (buf && len)
    {
        std::strncpy(buf, g_last_error.c_str(), len - 1);
        buf[len - 1] = 0;
    }
    return (int)g_last_error.size();
}

const char* gwhip_build_arch(void) { return "gfx950"; }
int gwhip_abi_version(void) { return 1; }

} // extern "C"
#endif // GWHIP_DEVICE_ONLY
