// poa_device.h -- gfx950 device code for the POA hot path (one 64-lane wavefront per window).
//
// What the reference computes is restated in oracle/ and SURVEY.md Appendix A; this file is how it is
// computed on CDNA4:
//   * the band row (256 cells = 64 lanes x 4 cells) is one wave-wide step; the previous row stays in
//     registers and is re-aligned to the new band start with DPP lane shifts, so the common predecessor
//     (distance 1) costs no memory round trip;
//   * nearby predecessors come from an LDS ring of recent rows, far ones from the HBM score matrix;
//   * the horizontal max-plus recurrence is a wave-level prefix-max (fixpoint of the reference's
//     shfl/any relaxation loop, identical unless an int16 store wraps -- see DESIGN.md);
//   * score rows stream to HBM as one aligned 8/16-byte store per lane (512 B / 1 KiB per wave-row);
//   * per-row predecessor info (base, predecessor rows, sink flag) is gathered once per read by all
//     lanes into an LDS table so neither the forward pass nor the traceback chases graph pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "poa_layout.h"

namespace gwhip
{

constexpr int kWave = 64;

template <typename T> struct Limits;
template <> struct Limits<int16_t> { static constexpr int32_t min = -32768; };
template <> struct Limits<int32_t> { static constexpr int32_t min = INT32_MIN; };

template <typename ScoreT> struct alignas(sizeof(ScoreT) * 4) Quad { ScoreT v[4]; };

// Narrowing of a score to the matrix type with a sticky "did not fit" flag. The kernels agree with the reference only while
// no int16 store wraps (with a wrap the reference's result depends on its relaxation order): the
// general forward passes raise the flag and the window then ends with StatusType::generic_error instead of a silent
// divergence; the packed pass of the 256-column band is only taken when bounds on the scores exclude a wrap (nw_banded).
template <typename ScoreT> __device__ __forceinline__ int32_t narrow_chk(int32_t v, bool& wrapped)
{
    const int32_t n = (int32_t)(ScoreT)v;
    if constexpr (sizeof(ScoreT) == 2) wrapped |= n != v;
    return n;
}

// Per-row table entry: node base, predecessor count, sink flag, the row's band start and the score-matrix rows
// (node_id_to_pos + 1) of the first 3 predecessor slots. The LDS-resident flavour (PACKED) is bit-packed into one 64-bit
// word (one ds_read_b64, decoded on the scalar unit): rows <= 4095, band starts <= 2044.
template <bool PACKED> struct RowInfo;
template <> struct alignas(8) RowInfo<true>
{
    uint64_t w; // [0:8) base  [8:14) pred count  [14] sink  [15:24) band_start/4  [24:36) [36:48) [48:60) pred rows
    __device__ __forceinline__ int32_t base() const { return (int32_t)(w & 0xff); }
    __device__ __forceinline__ int32_t cnt() const { return (int32_t)((w >> 8) & 0x3f); }
    __device__ __forceinline__ bool sink() const { return ((w >> 14) & 1) != 0; }
    __device__ __forceinline__ int32_t bs() const { return (int32_t)((w >> 15) & 0x1ff) << 2; }
    __device__ __forceinline__ int32_t pred(int k) const { return (int32_t)((w >> (24 + 12 * k)) & 0xfff); }
    __device__ __forceinline__ void set(int32_t base, int32_t cnt, bool sink, int32_t p0, int32_t p1, int32_t p2)
    {
        w = (uint64_t)(base & 0xff) | ((uint64_t)(cnt & 0x3f) << 8) | ((uint64_t)(sink ? 1 : 0) << 14) |
            ((uint64_t)(p0 & 0xfff) << 24) | ((uint64_t)(p1 & 0xfff) << 36) | ((uint64_t)(p2 & 0xfff) << 48);
    }
    __device__ __forceinline__ void set_bs(int32_t bs) { w = (w & ~(0x1ffull << 15)) | ((uint64_t)((bs >> 2) & 0x1ff) << 15); }
};
template <> struct alignas(8) RowInfo<false>
{
    uint8_t base_;
    uint8_t cnt_sink_;
    uint16_t pad_;
    int32_t bs_;
    int32_t pred_[3];
    int32_t pad2_;
    __device__ __forceinline__ int32_t base() const { return base_; }
    __device__ __forceinline__ int32_t cnt() const { return cnt_sink_ & 0x7f; }
    __device__ __forceinline__ bool sink() const { return (cnt_sink_ & 0x80) != 0; }
    __device__ __forceinline__ int32_t bs() const { return bs_; }
    // selects, not pred_[k]: a runtime index would put the struct in scratch memory, and a scratch load in the row
    // loop waits for every outstanding score store
    __device__ __forceinline__ int32_t pred(int k) const { return k == 0 ? pred_[0] : (k == 1 ? pred_[1] : pred_[2]); }
    __device__ __forceinline__ void set(int32_t base, int32_t cnt, bool sink, int32_t p0, int32_t p1, int32_t p2)
    {
        base_ = (uint8_t)base; cnt_sink_ = (uint8_t)((cnt & 0x7f) | (sink ? 0x80 : 0)); pad_ = 0; bs_ = 0;
        pred_[0] = p0; pred_[1] = p1; pred_[2] = p2; pad2_ = 0;
    }
    __device__ __forceinline__ void set_bs(int32_t bs) { bs_ = bs; }
};

template <typename IdT> struct GraphView
{
    uint8_t* nodes;
    IdT* node_alignments;
    uint16_t* node_alignment_count;
    IdT* incoming_edges;
    uint16_t* incoming_edge_count;
    IdT* outgoing_edges;
    uint16_t* outgoing_edge_count;
    uint16_t* incoming_edge_w;
    IdT* sorted_poa;
    IdT* node_id_to_pos;
    uint16_t* local_cnt;
    uint16_t* coverage;
    int32_t* cons_scores;
    IdT* cons_pred;
    uint8_t* marks;
    uint8_t* check;
    IdT* to_visit;
    uint16_t* out_cov;
    uint16_t* out_cov_cnt;
    IdT* msa_pos;
    IdT* seq_begin;
};

// ---- band placement: IEEE fp32 multiply + truncation (cudapoa_nw_banded.cuh:67-78) ----
__device__ __forceinline__ int32_t band_start_for_row(int32_t row, float gradient, int32_t band_width,
                                                      int32_t band_shift, int32_t max_column)
{
    int32_t diagonal_index = (int32_t)(__fmul_rn((float)row, gradient));
    int32_t start_pos      = max(0, diagonal_index - band_shift);
    if (max_column < start_pos + band_width) start_pos = max(0, max_column - band_width + kCellsPerLane);
    start_pos = start_pos - (start_pos % kCellsPerLane);
    return start_pos;
}

// ---- optional per-phase cycle accounting (s_memtime), enabled by passing a non-null accumulator ----
enum Phase { kPhRowInfo = 0, kPhForward, kPhTraceback, kPhAddAlignment, kPhTopsort, kPhOther, kPhCount };
struct PhaseClock
{
    uint64_t* acc; // [kPhCount] or nullptr
    uint64_t last;
    __device__ void start() { if (acc) last = clock64(); }
    __device__ void tick(int phase)
    {
        if (acc)
        {
            uint64_t now = clock64();
            acc[phase] += now - last;
            last = now;
        }
    }
};

// ---- wave-level helpers ----
// Ordering point of ONE wavefront with itself: everything the wave wrote (LDS, global) is complete and visible to
// every lane of the same wave afterwards. A window is worked on by one wavefront in all phases but the multi-wave
// forward pass (generic_forward_skew below), so this is what the single-wave code needs where a block-wide kernel
// would write __syncthreads() -- and it contains no s_barrier, which the helper wavefronts of a multi-wave block
// (parked at their own barrier) must not be released by.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
// barrier of all wavefronts of the block (multi-wave forward pass only)
__device__ __forceinline__ void block_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ int32_t wave_bcast(int32_t v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int32_t wave_first(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ uint64_t wave_first64(uint64_t v)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// wave-uniform copy of a row-table entry (moves the packed word into SGPRs so its decode runs on the scalar unit)
template <typename RowT> __device__ __forceinline__ RowT uniform_row(RowT r) { return r; }
template <> __device__ __forceinline__ RowInfo<true> uniform_row(RowInfo<true> r)
{
    r.w = wave_first64(r.w);
    return r;
}
// Load through an explicit LDS pointer. Where an LDS branch and an HBM branch do the same loads the optimiser would
// otherwise merge them behind one flat pointer (flat loads wait on both memory counters).
template <typename T> __device__ __forceinline__ T lds_ld(const T* p)
{
    return *(const __attribute__((address_space(3))) T*)p;
}
template <typename ScoreT> __device__ __forceinline__ Quad<ScoreT> lds_ld_quad(const ScoreT* p)
{
    const __attribute__((address_space(3))) ScoreT* q = (const __attribute__((address_space(3))) ScoreT*)p;
    Quad<ScoreT> r;
    r.v[0] = q[0]; r.v[1] = q[1]; r.v[2] = q[2]; r.v[3] = q[3];
    return r;
}

// four cells as one 8 / 16-byte store through an explicit LDS or global pointer (a struct cannot be assigned through
// an address-space pointer, a vector can)
template <typename ScoreT> struct QuadVec;
template <> struct QuadVec<int16_t> { typedef short type __attribute__((ext_vector_type(4))); };
template <> struct QuadVec<int32_t> { typedef int type __attribute__((ext_vector_type(4))); };
template <typename ScoreT> __device__ __forceinline__ void lds_st_quad(ScoreT* p, ScoreT a, ScoreT b, ScoreT c, ScoreT d)
{
    typedef typename QuadVec<ScoreT>::type V;
    V v = {a, b, c, d};
    *(__attribute__((address_space(3))) V*)p = v;
}
template <typename ScoreT>
__device__ __forceinline__ void global_st_quad(__attribute__((address_space(1))) ScoreT* p, ScoreT a, ScoreT b, ScoreT c, ScoreT d)
{
    typedef typename QuadVec<ScoreT>::type V;
    V v = {a, b, c, d};
    *(__attribute__((address_space(1))) V*)p = v;
}

// loads through an LDS byte address
template <typename ScoreT> __device__ __forceinline__ typename QuadVec<ScoreT>::type lds_ld_qv(uint32_t addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) typename QuadVec<ScoreT>::type*>(addr);
}
template <typename ScoreT> __device__ __forceinline__ ScoreT lds_ld_at(uint32_t addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) ScoreT*>(addr);
}

// a row-table record from LDS, dword by dword through an explicit LDS pointer
template <typename RowT> __device__ __forceinline__ RowT lds_ld_row(const RowT* p)
{
    static_assert(sizeof(RowT) % 4 == 0, "whole dwords");
    uint32_t w[sizeof(RowT) / 4];
    const __attribute__((address_space(3))) uint32_t* q = (const __attribute__((address_space(3))) uint32_t*)p;
#pragma unroll
    for (int k = 0; k < (int)(sizeof(RowT) / 4); k++) w[k] = q[k];
    RowT r;
    __builtin_memcpy(&r, w, sizeof(r));
    return r;
}

// every lane loaded the same record: moving it to scalar registers lets the row loop branch and index on the scalar unit
template <> __device__ __forceinline__ RowInfo<false> uniform_row(RowInfo<false> r)
{
    static_assert(sizeof(RowInfo<false>) == 24, "six dwords");
    int32_t w[6];
    __builtin_memcpy(w, &r, sizeof(r));
#pragma unroll
    for (int k = 0; k < 6; k++) w[k] = wave_first(w[k]);
    __builtin_memcpy(&r, w, sizeof(r));
    return r;
}

// Single-lane LDS stores without a branch: the compiler turns `if (lane == 0) *p = v;` into an exec-mask save,
// a skip branch and a restore; inside wave-uniform code (all 64 lanes active) narrowing exec around the store is
// cheaper. `p` must point to LDS.
__device__ __forceinline__ void lane0_store_u16(void* p, uint32_t v)
{
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)p;
    asm volatile("s_mov_b64 exec, 1\n\tds_write_b16 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lane0_store_u8(void* p, uint32_t v)
{
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)p;
    asm volatile("s_mov_b64 exec, 1\n\tds_write_b8 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lane0_store_u32(void* p, uint32_t v)
{
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)p;
    asm volatile("s_mov_b64 exec, 1\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(v) : "memory");
}

// Inclusive prefix-max across the 64 lanes with the gfx9 DPP row-shift / row-broadcast sequence.
__device__ __forceinline__ int32_t wave_inclusive_max(int32_t v)
{
    constexpr int32_t ident = INT32_MIN;
#define GW_DPP_MAX(ctrl, rmask) v = max(v, __builtin_amdgcn_update_dpp(ident, v, ctrl, rmask, 0xf, false))
    GW_DPP_MAX(0x111, 0xf); // row_shr:1
    GW_DPP_MAX(0x112, 0xf); // row_shr:2
    GW_DPP_MAX(0x114, 0xf); // row_shr:4
    GW_DPP_MAX(0x118, 0xf); // row_shr:8
    GW_DPP_MAX(0x142, 0xa); // row_bcast:15 into rows 1,3
    GW_DPP_MAX(0x143, 0xc); // row_bcast:31 into rows 2,3
#undef GW_DPP_MAX
    return v;
}
// value of lane-1 (lane 0 receives `first`)
__device__ __forceinline__ int32_t wave_shr1(int32_t v, int32_t first)
{
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false); // wave_shr:1
}
// value of lane+1 (lane 63 receives `last`)
__device__ __forceinline__ int32_t wave_shl1(int32_t v, int32_t last)
{
    return __builtin_amdgcn_update_dpp(last, v, 0x130, 0xf, 0xf, false); // wave_shl:1
}

// ------------------------------------------------------------------------------------------------
// Banded score-matrix storage (our layout): row r occupies stride = band_width + 8 elements; the value
// the reference keeps at relative index `rel` (= column - band_start(r), 0 = left-boundary slot) is
// stored at element rel + 3, so each lane's 4 cells (rel 1+4l .. 4+4l) form one naturally aligned Quad.
// ------------------------------------------------------------------------------------------------
constexpr int kRelShift = 3;

template <typename ScoreT> struct BandedCtx
{
    ScoreT* scores;    // HBM score matrix of this window
    ScoreT* ring;      // LDS ring of recent rows (same row layout), ring_rows rows
    int32_t ring_rows; // 0 = no ring
    int32_t stride;
    int32_t band_width, band_shift, max_column;
    float gradient;
    int32_t min_score;
};

// scalar read with the reference's get_score() predicate (cudapoa_nw_banded.cuh:80-102), HBM copy
template <typename ScoreT>
__device__ __forceinline__ int32_t get_score(const BandedCtx<ScoreT>& b, int32_t row, int32_t column)
{
    int32_t bs   = band_start_for_row(row, b.gradient, b.band_width, b.band_shift, b.max_column);
    int32_t bend = min(bs + b.band_width, b.max_column);
    if ((column > bend || column < bs) && column != -1) return b.min_score;
    int32_t rel = column == -1 ? 0 : column - bs;
    // the relative-0 slot of a row whose band starts past column 0 is min_score by construction (and is not stored)
    if (rel == 0 && bs > 0 && row > 0) return b.min_score;
    return b.scores[(int64_t)row * b.stride + rel + kRelShift];
}

// ------------------------------------------------------------------------------------------------
// Lane-0 traceback by recomputation, exact restatement of cudapoa_nw_banded.cuh:428-549, reading the HBM
// score matrix and the LDS row table. Returns alignment length or an error / rerun code.
// ------------------------------------------------------------------------------------------------
template <typename ScoreT, typename IdT, typename RowT, bool ADAPTIVE>
__device__ __forceinline__ int32_t traceback_banded(const BandedCtx<ScoreT>& b, const GraphView<IdT>& g, const RowT* rowinfo,
                                    int32_t graph_count, const uint8_t* read, int32_t read_length, int32_t start_i,
                                    int32_t* alignment_graph, int32_t* alignment_read, int32_t gap_score,
                                    int32_t mismatch_score, int32_t match_score, int32_t rerun)
{
    int32_t aligned_nodes = 0;
    int32_t i = start_i, j = read_length;
    int32_t prev_i = 0, prev_j = 0;
    int32_t loop_count = 0;
    const int32_t bound = read_length + graph_count + 2;
    while (!(i == 0 && j == 0) && loop_count < bound)
    {
        loop_count++;
        int32_t scores_ij = get_score(b, i, j);
        bool pred_found   = false;
        RowT ri{};
        int32_t pred_count = 0, node_id = 0;
        if (i != 0)
        {
            ri         = rowinfo[i];
            pred_count = ri.cnt();
        }
        auto pred_row = [&](int32_t p) -> int32_t {
            if (pred_count == 0) return 0;
            if (p < 3) return ri.pred(p);
            return (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1;
        };
        if (i != 0 && pred_count > 3) node_id = g.sorted_poa[i - 1];
        if (i != 0 && j != 0)
        {
            if (ADAPTIVE)
            {
                if (rerun == 0 && b.band_width < kMaxAdaptiveBand)
                {
                    int32_t threshold = max(1, b.max_column / 1024);
                    if (j > threshold && j < b.max_column - threshold)
                    {
                        int32_t bs = band_start_for_row(i, b.gradient, b.band_width, b.band_shift, b.max_column);
                        if (j <= bs + threshold) { aligned_nodes = kShiftLeft; break; }
                        if (j >= (bs + b.band_width - threshold)) { aligned_nodes = kShiftRight; break; }
                    }
                }
            }
            int32_t match_cost = (ri.base() == read[j - 1] ? match_score : mismatch_score);
            int32_t np         = max(pred_count, 1);
            for (int32_t p = 0; p < np; p++)
            {
                int32_t pi = pred_row(p);
                if (scores_ij == get_score(b, pi, j - 1) + match_cost)
                {
                    prev_i = pi; prev_j = j - 1; pred_found = true;
                    break;
                }
            }
        }
        if (!pred_found && i != 0)
        {
            int32_t np = max(pred_count, 1);
            for (int32_t p = 0; p < np; p++)
            {
                int32_t pi = pred_row(p);
                if (scores_ij == get_score(b, pi, j) + gap_score)
                {
                    prev_i = pi; prev_j = j; pred_found = true;
                    break;
                }
            }
        }
        if (!pred_found && scores_ij == get_score(b, i, j - 1) + gap_score)
        {
            prev_i = i; prev_j = j - 1; pred_found = true;
        }
        alignment_graph[aligned_nodes] = (i == prev_i ? -1 : (int32_t)g.sorted_poa[i - 1]);
        alignment_read[aligned_nodes]  = (j == prev_j ? -1 : j - 1);
        aligned_nodes++;
        i = prev_i;
        j = prev_j;
    }
    if (loop_count >= bound) aligned_nodes = kNwLoopFailed;
    return aligned_nodes;
}

// ------------------------------------------------------------------------------------------------
// Tiled traceback for graphs whose row table lives in HBM (long reads). traceback_banded_tiled keeps the score matrix
// around the path in LDS, but with the row table, the read and the output arrays in HBM every step still paid global
// round trips behind its own stores (about 1 450 cycles per step). Here everything a step touches is staged with the
// tile, by all lanes, in the same round trip: the row-table records of the 64 tile rows, 256 characters of the read
// around the anchor column, and the walk's output (cells, 64 per flush; graph positions are translated to node ids by
// all lanes after the walk). Same decision sequence as traceback_banded (cudapoa_nw_banded.cuh:428-549).
// LDS (the forward pass's ring, dead by now): tile[64][64] | meta[64] | rows[64] | read[256] | stage[64].
// ------------------------------------------------------------------------------------------------
template <typename ScoreT, typename IdT, typename RowT, bool ADAPTIVE>
__device__ __forceinline__ int32_t traceback_banded_tiled_staged(const BandedCtx<ScoreT>& b, const GraphView<IdT>& g,
                                                                 const RowT* rowinfo, int32_t graph_count, const uint8_t* read,
                                                                 int32_t read_length, int32_t start_i, int32_t* alignment_graph,
                                                                 int32_t* alignment_read, int32_t gap_score, int32_t mismatch_score,
                                                                 int32_t match_score, int32_t rerun, uint8_t* lds)
{
    constexpr int kTileRows = 64, kTileCols = 64, kReanchor = 44, kReadWin = 256, kStage = 64;
    ScoreT* tile       = reinterpret_cast<ScoreT*>(lds);
    int32_t* tile_meta = reinterpret_cast<int32_t*>(lds + kTileRows * kTileCols * sizeof(ScoreT));
    RowT* ri_tile      = reinterpret_cast<RowT*>(reinterpret_cast<uint8_t*>(tile_meta) + kTileRows * sizeof(int32_t));
    uint8_t* rd_tile   = reinterpret_cast<uint8_t*>(ri_tile) + kTileRows * sizeof(RowT);
    uint64_t* stage    = reinterpret_cast<uint64_t*>(rd_tile + kReadWin);
    const int lane      = threadIdx.x & (kWave - 1);
    const int32_t bound = read_length + graph_count + 2;
    int32_t aligned_nodes = 0;
    int32_t i = start_i, j = read_length, prev_i = 0, prev_j = 0;
    int32_t tile_top = -1, rd_lo = 0;

    auto load_tile = [&](int32_t top, int32_t col) {
        wave_sync();
        const int32_t row = top - lane;
        int32_t bs = 0, e0 = 0;
        if (row >= 0)
        {
            bs = band_start_for_row(row, b.gradient, b.band_width, b.band_shift, b.max_column);
            e0 = (col - lane - 40) - bs + kRelShift;
            e0 = min(max(e0 & ~3, 0), b.stride - kTileCols);
            const ScoreT* src = b.scores + (int64_t)row * b.stride + e0;
            ScoreT* dst       = tile + lane * kTileCols;
#pragma unroll
            for (int k = 0; k < kTileCols; k += 4)
                *reinterpret_cast<Quad<ScoreT>*>(dst + k) = *reinterpret_cast<const Quad<ScoreT>*>(src + k);
            if (row >= 1) ri_tile[lane] = rowinfo[row];
        }
        tile_meta[lane] = (row >= 0) ? ((bs & 0xffff) | ((e0 - kRelShift + bs) << 16)) : 0x7fff0000;
        // read characters of columns [rd_lo + 1, rd_lo + 256]: the walk only moves left, at most one tile width plus the
        // window's own drift before the next re-anchor (the input buffer keeps zero slack behind the last read)
        rd_lo = max(col - 192, 0) & ~3;
        *reinterpret_cast<uint32_t*>(rd_tile + 4 * lane) = *reinterpret_cast<const uint32_t*>(read + rd_lo + 4 * lane);
        wave_sync();
    };
    auto score_at = [&](int32_t row, int32_t column) -> int32_t {
        const int32_t rr = tile_top - row;
        if (tile_top >= 0 && rr >= 0 && rr < kTileRows)
        {
            const int32_t meta = tile_meta[rr];
            const int32_t bs   = meta & 0xffff;
            const int32_t lo   = meta >> 16;
            const int32_t bend = min(bs + b.band_width, b.max_column);
            if ((column > bend || column < bs) && column != -1) return b.min_score;
            const int32_t col = column == -1 ? bs : column;
            if (col == bs && bs > 0 && row > 0) return b.min_score;
            const int32_t off = col - lo;
            if (off >= 0 && off < kTileCols) return tile[rr * kTileCols + off];
        }
        return get_score(b, row, column);
    };
    // stage[k] = the cell step k left; the reference's entry for a step follows from two consecutive cells
    auto flush_stage = [&](int32_t first, int32_t count) {
        if (lane < count)
        {
            const uint64_t cur = stage[lane];
            const uint64_t nxt = lane + 1 < count ? stage[lane + 1] : ((uint64_t)(uint32_t)i | ((uint64_t)(uint32_t)j << 32));
            const int32_t ci = (int32_t)(uint32_t)cur, cj = (int32_t)(cur >> 32);
            const int32_t ni = (int32_t)(uint32_t)nxt, nj = (int32_t)(nxt >> 32);
            alignment_graph[first + lane] = ci == ni ? -1 : ci - 1; // sorted position; node ids are filled in below
            alignment_read[first + lane]  = cj == nj ? -1 : cj - 1;
        }
    };

    bool rerun_break = false;
    while (!(i == 0 && j == 0) && aligned_nodes < bound) // every step appends one entry: the reference's loop counter
    {
        {
            const int32_t rr = tile_top - i;
            bool reload      = tile_top < 0 || rr < 0 || rr >= kReanchor;
            if (!reload)
            {
                const int32_t meta = tile_meta[rr];
                const int32_t off  = j - (meta >> 16);
                reload             = (off < 2 || off >= kTileCols);
            }
            reload = reload || (j - 1 - rd_lo) < 0; // (cannot happen between two re-anchors; kept as a guard)
            if (reload && i > 0)
            {
                tile_top = i;
                load_tile(i, j);
            }
        }
        const int32_t scores_ij = score_at(i, j);
        bool pred_found         = false;
        RowT ri{};
        int32_t pred_count = 0, node_id = 0;
        if (i != 0)
        {
            ri         = uniform_row(ri_tile[tile_top - i]);
            pred_count = ri.cnt();
        }
        auto pred_row = [&](int32_t p) -> int32_t {
            if (pred_count == 0) return 0;
            if (p < 3) return ri.pred(p);
            return (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1;
        };
        if (i != 0 && pred_count > 3) node_id = g.sorted_poa[i - 1];
        if (i != 0 && j != 0)
        {
            if (ADAPTIVE)
            {
                if (rerun == 0 && b.band_width < kMaxAdaptiveBand)
                {
                    int32_t threshold = max(1, b.max_column / 1024);
                    if (j > threshold && j < b.max_column - threshold)
                    {
                        int32_t bs = band_start_for_row(i, b.gradient, b.band_width, b.band_shift, b.max_column);
                        if (j <= bs + threshold) { aligned_nodes = kShiftLeft; rerun_break = true; }
                        else if (j >= (bs + b.band_width - threshold)) { aligned_nodes = kShiftRight; rerun_break = true; }
                    }
                }
            }
            if (!rerun_break)
            {
                const int32_t rc   = j - 1 - rd_lo;
                const uint32_t ch  = (uint32_t)rc < (uint32_t)kReadWin ? rd_tile[rc] : read[j - 1];
                int32_t match_cost = ((uint32_t)ri.base() == ch ? match_score : mismatch_score);
                int32_t np         = max(pred_count, 1);
                for (int32_t p = 0; p < np; p++)
                {
                    int32_t pi = pred_row(p);
                    if (scores_ij == score_at(pi, j - 1) + match_cost)
                    {
                        prev_i = pi; prev_j = j - 1; pred_found = true;
                        break;
                    }
                }
            }
        }
        if (rerun_break) break;
        if (!pred_found && i != 0)
        {
            int32_t np = max(pred_count, 1);
            for (int32_t p = 0; p < np; p++)
            {
                int32_t pi = pred_row(p);
                if (scores_ij == score_at(pi, j) + gap_score)
                {
                    prev_i = pi; prev_j = j; pred_found = true;
                    break;
                }
            }
        }
        if (!pred_found && scores_ij == score_at(i, j - 1) + gap_score)
        {
            prev_i = i; prev_j = j - 1; pred_found = true;
        }
        if (lane == 0) stage[aligned_nodes & (kStage - 1)] = (uint64_t)(uint32_t)i | ((uint64_t)(uint32_t)j << 32);
        aligned_nodes++;
        i = prev_i;
        j = prev_j;
        if ((aligned_nodes & (kStage - 1)) == 0)
        {
            wave_sync();
            flush_stage(aligned_nodes - kStage, kStage);
        }
    }
    if (rerun_break) return aligned_nodes;
    wave_sync();
    if (aligned_nodes > 0 && (aligned_nodes & (kStage - 1)) != 0)
        flush_stage(aligned_nodes & ~(kStage - 1), aligned_nodes & (kStage - 1));
    if (aligned_nodes >= bound) aligned_nodes = kNwLoopFailed;
    wave_sync();
    for (int32_t k0 = lane; k0 < aligned_nodes; k0 += 4 * kWave) // positions -> node ids, four load chains per lane
    {
        int32_t pos[4], node[4];
#pragma unroll
        for (int u = 0; u < 4; u++) pos[u] = (k0 + u * kWave < aligned_nodes) ? alignment_graph[k0 + u * kWave] : -1;
#pragma unroll
        for (int u = 0; u < 4; u++) node[u] = (int32_t)g.sorted_poa[max(pos[u], 0)];
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (pos[u] >= 0) alignment_graph[k0 + u * kWave] = node[u];
    }
    wave_sync();
    return aligned_nodes;
}

// ------------------------------------------------------------------------------------------------
// Traceback by recomputation, candidate-per-lane. Same decision sequence as cudapoa_nw_banded.cuh:428-549
// (diagonal move through predecessor 0..n-1, then vertical through predecessor 0..n-1, then horizontal; first
// equality wins), but one step evaluates every candidate at once: lanes 0..30 test the diagonal moves, lanes
// 31..61 the vertical moves and lane 62 the horizontal move; a ballot + find-first-set reproduces the
// reference's priority order. Rows with more than 31 predecessors take a wave-uniform sequential loop.
//
// The walk is latency-bound on a lone wavefront, so a step is organised as ONE LDS round trip:
//   * the score matrix around the path is cached in an LDS tile of 60 rows x 64 ABSOLUTE columns; the window of
//     tile row t starts at column ((anchor_col - 40 - t) & ~3) + 1 -- arithmetic, independent of the band start
//     (band starts are multiples of 4, so the HBM source is Quad-aligned) -- and the loader applies
//     get_score()'s band predicate (cudapoa_nw_banded.cuh:80-102) once per cached cell, so a candidate lane
//     needs no band math at all;
//   * every candidate lane also prefetches the row-table word and the read character its move would need
//     next; the winner's copies are broadcast with v_readlane, so the next step starts without a dependent
//     LDS read; the winner's score becomes the next step's H(i, j);
//   * (position, read index) pairs are staged in LDS and flushed 64 at a time: an HBM store inside the walk
//     would make each step wait for the previous write acknowledgement (vmcnt is shared by loads and stores).
// Cells outside the tile (far predecessors) use the HBM copy through get_score(). alignment_graph receives
// sorted positions during the walk and is translated to node ids by all lanes afterwards.
// ------------------------------------------------------------------------------------------------
template <typename ScoreT, typename IdT, bool ADAPTIVE>
__device__ __forceinline__ int32_t traceback_banded_lanes(const BandedCtx<ScoreT>& b, const GraphView<IdT>& g,
                                                          const RowInfo<true>* rowinfo, int32_t graph_count,
                                                          const uint8_t* read, int32_t read_length, int32_t start_i,
                                                          int32_t* alignment_graph, int32_t* alignment_read,
                                                          int32_t gap_score, int32_t mismatch_score,
                                                          int32_t match_score, int32_t rerun, ScoreT* tile, int32_t dbg = 0,
                                                          uint64_t* prof_acc = nullptr)
{
    constexpr int kTileRows = 60, kTileCols = 64, kTileStride = 68, kReanchor = 44, kLead = 40, kHalf = 31;
    constexpr int kStage = 64;
    const int lane      = threadIdx.x & (kWave - 1);
    const int32_t bound = read_length + graph_count + 2;
    int32_t aligned_nodes = 0, loop_count = 0;
    int32_t i = start_i, j = read_length, prev_i = 0, prev_j = 0;
    int32_t tile_top = -1, tile_col = 0; // matrix row in tile row 0 and the column the windows are anchored on
    uint32_t* stage  = reinterpret_cast<uint32_t*>(tile + kTileRows * kTileStride);
    // profiling (GWHIP_DEBUG bits 22-24): 4 cycles in the steps (incl. their tile loads), 5 their number (x 1000), 6 cycles of
    // the post-pass; arrives in the "other" accumulator
    const int32_t psel = prof_acc ? (dbg >> 22) & 7 : 0;
    uint64_t pacc      = 0;

    auto window_lo = [&](int32_t t) -> int32_t { return ((tile_col - kLead - t) & ~3) + 1; };
    auto load_tile = [&](int32_t top, int32_t col) {
        wave_sync();
        tile_top = top;
        tile_col = col;
        const int32_t row = top - lane;
        const bool ok_row = row >= 0 && lane < kTileRows;
        int32_t e0 = 0, klo = 0, khi = -1;
        if (ok_row)
        {
            const int32_t bs   = band_start_for_row(row, b.gradient, b.band_width, b.band_shift, b.max_column);
            const int32_t bend = min(bs + b.band_width, b.max_column);
            e0                 = window_lo(lane) - bs + kRelShift; // multiple of 4
            // stored elements get_score() would return: columns bs (or bs + 1 when the relative-0 slot is synthetic) .. bend
            klo = ((bs > 0 && row > 0) ? kRelShift + 1 : kRelShift) - e0;
            khi = bend - bs + kRelShift - e0;
        }
        const ScoreT* src = b.scores + (int64_t)row * b.stride + e0;
        ScoreT* dst       = tile + lane * kTileStride;
        const bool whole  = klo <= 0 && khi >= kTileCols - 1;
        if (__ballot(ok_row && !whole) == 0)
        {
            if (ok_row)
            {
#pragma unroll
                for (int k = 0; k < kTileCols; k += 4)
                    *reinterpret_cast<Quad<ScoreT>*>(dst + k) = *reinterpret_cast<const Quad<ScoreT>*>(src + k);
            }
        }
        else if (ok_row) // a window reaching past a band edge: per-cell predicate
        {
#pragma nounroll
            for (int k = 0; k < kTileCols; k++) dst[k] = (k >= klo && k <= khi) ? src[k] : (ScoreT)b.min_score;
        }
        wave_sync();
    };
    // stage[k] = cell (row | column << 16) step k LEFT; the entry the reference records for a step -- graph position or
    // -1 when the row did not change, read position or -1 when the column did not change -- follows from two
    // consecutive cells, so the walk stores one word per step and the lanes format 64 steps at a time. The cell
    // after the last staged step is the walk's current cell (i, j).
    auto flush_stage = [&](int32_t first, int32_t count) {
        if (lane < count)
        {
            const uint32_t cur = stage[lane];
            const uint32_t nxt = lane + 1 < count ? stage[lane + 1] : ((uint32_t)i | ((uint32_t)j << 16));
            const int32_t ci = (int32_t)(cur & 0xffff), cj = (int32_t)(cur >> 16);
            const int32_t ni = (int32_t)(nxt & 0xffff), nj = (int32_t)(nxt >> 16);
            alignment_graph[first + lane] = ci == ni ? -1 : ci - 1; // sorted position; node ids are filled in below
            alignment_read[first + lane]  = cj == nj ? -1 : cj - 1;
        }
    };

    // lane roles (VGPR constants)
    const int kind        = lane < kHalf ? 0 : (lane < 2 * kHalf ? 1 : (lane == 2 * kHalf ? 2 : 3));
    const int p           = kind == 0 ? lane : lane - kHalf;
    const int psh         = 24 + 12 * min(p, 2);
    const bool is_diag    = kind == 0, is_vert = kind == 1, is_horiz = kind == 2, is_self = kind == 3;
    const int32_t col_dec = (is_vert | is_self) ? 0 : 1; // candidate column = j - col_dec
    // (the 256-column band of int16 scores has its own walk over move bytes, poa_traceback_moves.h; this one serves the
    // other band widths of graphs that fit the LDS tables)
    constexpr bool use_tile = true;
    // Wave-uniform walk state, kept in SGPRs (every update goes through readfirstlane / readlane so that the
    // loop control stays scalar): position, H(i, j), the row-table word of row i and the read character j - 1.
    int32_t scores_ij = 0;
    uint32_t ri_lo = 0, ri_hi = 0, rch = 0;
    bool have = false;
    // (every step appends one entry, so the reference's loop counter is aligned_nodes)
    while (!(i == 0 && j == 0) && aligned_nodes < bound)
    {
        const uint64_t t_rc = psel == 4 ? clock64() : 0;
        if (psel == 5) pacc += 1000;
        // keep the current cell and its near predecessors inside the tile
        if (use_tile)
        {
            const int32_t t   = tile_top - i;
            const int32_t off = j - window_lo(t);
            const bool reload = (tile_top < 0) | (t < 0) | (t >= kReanchor) | (off < 2) | (off >= kTileCols);
            if (reload && i > 0) load_tile(i, j);
        }
        const bool need_self = !have && !use_tile; // H(i, j) arrives with the candidates
        if (!have)
        {
            if (use_tile) scores_ij = wave_first(get_score(b, i, j));
            const uint64_t w = i != 0 ? wave_first64(rowinfo[i].w) : 0;
            ri_lo = (uint32_t)w; ri_hi = (uint32_t)(w >> 32);
            rch   = j > 0 ? (uint32_t)wave_first((int32_t)read[j - 1]) : 0u;
        }
        const int32_t pred_count = (int32_t)((ri_lo >> 8) & 0x3f);
        const int32_t np         = max(pred_count, 1);
        if (ADAPTIVE)
        {
            if (i != 0 && j != 0 && rerun == 0 && b.band_width < kMaxAdaptiveBand)
            {
                int32_t threshold = max(1, b.max_column / 1024);
                if (j > threshold && j < b.max_column - threshold)
                {
                    int32_t bs = band_start_for_row(i, b.gradient, b.band_width, b.band_shift, b.max_column);
                    if (j <= bs + threshold) { aligned_nodes = kShiftLeft; break; }
                    if (j >= (bs + b.band_width - threshold)) { aligned_nodes = kShiftRight; break; }
                }
            }
        }
        const int32_t match_cost = ((ri_lo & 0xff) == rch ? match_score : mismatch_score);
        bool found         = false;
        int32_t next_i = prev_i, next_j = prev_j, next_score = 0;
        uint32_t next_lo = 0, next_hi = 0, next_rch = 0;
        if (np <= kHalf)
        {
            // candidate of this lane: predecessor row (diag / vert) or the row itself (horiz), and its column
            const bool en = (is_diag & (i != 0) & (j != 0) & (p < np)) | (is_vert & (i != 0) & (p < np)) | is_horiz;
            const uint64_t riw = (uint64_t)ri_lo | ((uint64_t)ri_hi << 32);
            int32_t crow       = pred_count != 0 ? (int32_t)((riw >> psh) & 0xfff) : 0;
            crow               = (is_horiz | is_self) ? i : crow;
            if (pred_count > 3) // predecessor slots beyond the three packed ones live in the HBM edge list
            {
                if (en && !is_horiz && p >= 3)
                {
                    const int32_t node_id = g.sorted_poa[i - 1];
                    crow = (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1;
                }
            }
            const int32_t ccol = j - col_dec;
            const int32_t cost = is_diag ? match_cost : gap_score;
            const int32_t t    = tile_top - crow;
            const int32_t off  = ccol - window_lo(t);
            const bool in_tile = use_tile & ((uint32_t)t < (uint32_t)kTileRows) & ((uint32_t)off < (uint32_t)kTileCols);
            // three independent LDS reads, one round trip: the candidate cell, and what the next step needs if this
            // candidate wins (its row-table word -- row 0 holds 0 -- and the read character left of its column)
            int32_t val        = tile[in_tile ? t * kTileStride + off : 0];
            const uint64_t cw  = rowinfo[crow].w;
            const uint32_t cch = read[max(ccol - 1, 0)];
            const bool from_hbm = (en | (is_self & need_self)) & !in_tile;
            if (__ballot(from_hbm) != 0) // no tile / far predecessor / band edge: the HBM copy
            {
                if (from_hbm) val = get_score(b, crow, ccol);
            }
            if (need_self) scores_ij = __builtin_amdgcn_readlane(val, kWave - 1);
            const bool hit   = en & (scores_ij == val + cost);
            const uint64_t m = __ballot(hit);
            found            = m != 0;
            const int sel    = found ? __ffsll((unsigned long long)m) - 1 : 0;
            const int32_t si = __builtin_amdgcn_readlane(crow, sel);
            const int32_t sj = __builtin_amdgcn_readlane(ccol, sel);
            next_i           = found ? si : prev_i;
            next_j           = found ? sj : prev_j;
            next_score       = __builtin_amdgcn_readlane(val, sel);
            next_lo          = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)cw, sel);
            next_hi          = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)(cw >> 32), sel);
            next_rch         = (uint32_t)__builtin_amdgcn_readlane((int32_t)cch, sel);
        }
        else // more predecessors than candidate lanes: the reference's sequential order, wave-uniform
        {
            if (need_self) scores_ij = wave_first(get_score(b, i, j));
            const int32_t node_id = g.sorted_poa[i - 1];
            RowInfo<true> ri;
            ri.w = (uint64_t)ri_lo | ((uint64_t)ri_hi << 32);
            auto pred_row = [&](int32_t q) -> int32_t {
                if (q < 3) return ri.pred(q);
                return wave_first((int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + q]] + 1);
            };
            bool f = false;
            if (j != 0)
                for (int32_t q = 0; q < np && !f; q++)
                {
                    const int32_t pi = pred_row(q);
                    if (scores_ij == wave_first(get_score(b, pi, j - 1)) + match_cost) { next_i = pi; next_j = j - 1; f = true; }
                }
            for (int32_t q = 0; q < np && !f; q++)
            {
                const int32_t pi = pred_row(q);
                if (scores_ij == wave_first(get_score(b, pi, j)) + gap_score) { next_i = pi; next_j = j; f = true; }
            }
            if (!f && scores_ij == wave_first(get_score(b, i, j - 1)) + gap_score) { next_i = i; next_j = j - 1; f = true; }
            found = false; // walk state is re-read at the top of the next step
        }
        prev_i = next_i;
        prev_j = next_j;
        lane0_store_u32(stage + (aligned_nodes & (kStage - 1)), (uint32_t)i | ((uint32_t)j << 16));
        aligned_nodes++;
        i         = prev_i;
        j         = prev_j;
        if ((aligned_nodes & (kStage - 1)) == 0) flush_stage(aligned_nodes - kStage, kStage);
        scores_ij = next_score;
        ri_lo     = next_lo;
        ri_hi     = next_hi;
        rch       = next_rch;
        have      = found;
        if (psel == 4) pacc += clock64() - t_rc;
    }
    const uint64_t t_pp = psel == 6 ? clock64() : 0;
    if (aligned_nodes > 0 && (aligned_nodes & (kStage - 1)) != 0)
        flush_stage(aligned_nodes & ~(kStage - 1), aligned_nodes & (kStage - 1));
    if ((dbg & 2) && prof_acc && lane == 0) *prof_acc += (uint64_t)max(aligned_nodes, 0); // profiling: all steps
    if (aligned_nodes >= bound) aligned_nodes = kNwLoopFailed;
    wave_sync();
    for (int32_t k0 = lane; k0 < aligned_nodes; k0 += 4 * kWave) // 4 independent load chains per lane in flight
    {
        int32_t pos[4], node[4];
#pragma unroll
        for (int u = 0; u < 4; u++) pos[u] = (k0 + u * kWave < aligned_nodes) ? alignment_graph[k0 + u * kWave] : -1;
#pragma unroll
        for (int u = 0; u < 4; u++) node[u] = (int32_t)g.sorted_poa[max(pos[u], 0)];
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (pos[u] >= 0) alignment_graph[k0 + u * kWave] = node[u];
    }
    wave_sync();
    if (psel == 6) pacc += clock64() - t_pp;
    if (psel && lane == 0) *prof_acc += pacc;
    return aligned_nodes;
}

// ------------------------------------------------------------------------------------------------
// Per-read row table: all lanes gather (base, predecessor rows, sink flag) for rows 1..N.
// ------------------------------------------------------------------------------------------------
template <typename IdT, typename RowT>
__device__ __forceinline__ void build_rowinfo(const GraphView<IdT>& g, int32_t graph_count, RowT* rowinfo, int lane,
                                              uint64_t* xpred = nullptr)
{
    // xpred (optional, LDS, 256 slots keyed by row & 255): predecessor rows 3..5 of the rows that have more than the
    // three the row table holds: {row [0:12), valid [12], count [13:19), pred 3 [20:32), pred 4 [32:44), pred 5 [44:56)}.
    // The last writer of a slot wins; a reader checks the row id (poa_forward_packed.h: classify_rows).
    if (xpred != nullptr)
    {
#pragma unroll
        for (int q = 0; q < 4; q++) xpred[q * kWave + lane] = 0;
        wave_sync();
    }
    if (lane == 0) rowinfo[0].set(0, 0, false, 0, 0, 0); // row 0 (virtual source): no predecessors
    // Four 64-row chunks per iteration: a chunk is three dependent HBM round trips (row -> node -> its first three
    // in-edges -> their rows), and the chunks of one iteration share them (18 round trips per read instead of 66).
    // Edge slots past the in-degree may hold stale or uninitialised ids; they are range-checked and masked.
    constexpr int kU = 4;
    for (int32_t r0 = 1 + lane; r0 <= graph_count; r0 += kU * kWave)
    {
        int32_t node[kU], cnt[kU], oc[kU], bas[kU], e0[kU], e1[kU], e2[kU], q0[kU], q1[kU], q2[kU];
        int32_t e3[kU], e4[kU], e5[kU], q3[kU] = {}, q4[kU] = {}, q5[kU] = {};
#pragma unroll
        for (int u = 0; u < kU; u++) node[u] = g.sorted_poa[min(r0 + u * kWave, graph_count) - 1];
#pragma unroll
        for (int u = 0; u < kU; u++)
        {
            cnt[u] = g.incoming_edge_count[node[u]];
            oc[u]  = g.outgoing_edge_count[node[u]];
            bas[u] = g.nodes[node[u]];
            e0[u]  = g.incoming_edges[(int64_t)node[u] * kEdges + 0];
            e1[u]  = g.incoming_edges[(int64_t)node[u] * kEdges + 1];
            e2[u]  = g.incoming_edges[(int64_t)node[u] * kEdges + 2];
        }
#pragma unroll
        for (int u = 0; u < kU; u++)
        {
            q0[u] = g.node_id_to_pos[(uint32_t)e0[u] < (uint32_t)graph_count ? e0[u] : 0];
            q1[u] = g.node_id_to_pos[(uint32_t)e1[u] < (uint32_t)graph_count ? e1[u] : 0];
            q2[u] = g.node_id_to_pos[(uint32_t)e2[u] < (uint32_t)graph_count ? e2[u] : 0];
            // in-edges 3..5 of the few rows that have them ride along with this round trip (only those lanes load)
            e3[u] = e4[u] = e5[u] = 0;
            if (xpred != nullptr && cnt[u] > 3)
            {
                e3[u] = g.incoming_edges[(int64_t)node[u] * kEdges + 3];
                e4[u] = g.incoming_edges[(int64_t)node[u] * kEdges + 4];
                e5[u] = g.incoming_edges[(int64_t)node[u] * kEdges + 5];
            }
        }
        if (xpred != nullptr)
        {
            bool any = false;
#pragma unroll
            for (int u = 0; u < kU; u++) any |= cnt[u] > 3;
            if (__ballot(any) != 0)
            {
#pragma unroll
                for (int u = 0; u < kU; u++)
                {
                    q3[u] = q4[u] = q5[u] = 0;
                    if (cnt[u] > 3)
                    {
                        q3[u] = g.node_id_to_pos[(uint32_t)e3[u] < (uint32_t)graph_count ? e3[u] : 0];
                        q4[u] = g.node_id_to_pos[(uint32_t)e4[u] < (uint32_t)graph_count ? e4[u] : 0];
                        q5[u] = g.node_id_to_pos[(uint32_t)e5[u] < (uint32_t)graph_count ? e5[u] : 0];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kU; u++)
        {
            const int32_t r = r0 + u * kWave;
            if (r <= graph_count)
            {
                RowT ri{};
                ri.set(bas[u], cnt[u], oc[u] == 0, cnt[u] > 0 ? q0[u] + 1 : 0, cnt[u] > 1 ? q1[u] + 1 : 0, cnt[u] > 2 ? q2[u] + 1 : 0);
                rowinfo[r] = ri;
                if (xpred != nullptr && cnt[u] > 3)
                    xpred[r & 255] = (uint64_t)((uint32_t)r | (1u << 12) | ((uint32_t)(cnt[u] & 63) << 13) | ((uint32_t)((q3[u] + 1) & 0xfff) << 20)) |
                                     ((uint64_t)((q4[u] + 1) & 0xfff) << 32) | ((uint64_t)((q5[u] + 1) & 0xfff) << 44);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Lean forward pass for bands <= 256 columns (one 64-lane pass per row), PACKED row table in LDS, read in LDS.
// Everything wave-uniform (row table word, band starts, ring slots, carry, boundary values) lives in SGPRs;
// per row the vector unit only does the 4-cell recurrence, the DPP prefix-max and two stores.
//
// Rows whose only predecessor is the previous row and whose band moved by 0 or 4 columns ("simple" rows, the
// bulk) run in a tight inner loop with no branches besides the loop itself: the previous row is in registers
// and is re-aligned with DPP lane shifts. All other rows take the general body (LDS ring, far rows from HBM).
// The relative-0 slot of a row is only materialised when its band starts at column 0; for every other row it is
// min_score by construction (cudapoa_nw_banded.cuh:158-175), so readers synthesise it.
// Under the no-int16-wrap precondition values are kept in 32-bit registers and narrowed on store.
// ------------------------------------------------------------------------------------------------
template <typename ScoreT, typename IdT>
__device__ __forceinline__ void banded_forward_1pass(const GraphView<IdT>& g, const RowInfo<true>* rowinfo,
                                                     int32_t graph_count, const uint8_t* lds_read, ScoreT* scores,
                                                     ScoreT* ring, int32_t ring_rows, int32_t band_width,
                                                     int32_t max_column, int32_t gap_score, int32_t mismatch_score,
                                                     int32_t match_score, int32_t dbg, uint64_t* general_row_acc, bool& wrapped)
{
    const int lane          = threadIdx.x & (kWave - 1);
    const int32_t min_score = Limits<ScoreT>::min / 2;
    const int32_t stride    = band_width + kRightPad;
    const int32_t lane4     = lane * 4;
    uint64_t general_acc    = 0; // profiling (GWHIP_DEBUG bit 2): cycles (or, with bit 3, count) of general rows
    const bool full_wave    = band_width == 256; // every lane owns 4 in-band cells
    const bool active       = lane4 < band_width;
    const int32_t K0 = (lane4 + 0) * gap_score, K1 = (lane4 + 1) * gap_score, K2 = (lane4 + 2) * gap_score,
                  K3 = (lane4 + 3) * gap_score;
    // row 0 in registers: cells rel 1+4l .. 4+4l = (rel) * gap
    int32_t P0 = K0 + gap_score, P1 = K1 + gap_score, P2 = K2 + gap_score, P3 = K3 + gap_score;
    int32_t prev_bs = 0, prev_rel0 = 0;
    int32_t ring_slot = 0; // slot of row r-1 (row 0 sits in slot 0)
    bool hbm_dirty    = false;
    ScoreT* row_out   = scores; // advanced by stride per row
    ScoreT* ring_out  = ring;   // ring row of slot `ring_slot`

    // shared row tail: horizontal max-plus scan (prefix max of u[t] = v[t] - t*gap, carry-in as element -1),
    // then the two row stores
    auto finish_row = [&](int32_t s0, int32_t s1, int32_t s2, int32_t s3, int32_t fe, int32_t rel0_val, int32_t bs) {
        const int32_t u0 = s0 - K0, u1 = s1 - K1, u2 = s2 - K2, u3 = s3 - K3;
        const int32_t m1 = max(u0, u1), m2 = max(m1, u2), m3 = max(m2, u3);
        const int32_t incl = wave_inclusive_max(m3);
        const int32_t excl = max(wave_shr1(incl, INT32_MIN), fe + gap_score);
        P0 = max(u0, excl) + K0;
        P1 = max(m1, excl) + K1;
        P2 = max(m2, excl) + K2;
        P3 = max(m3, excl) + K3;
        row_out += stride;
        ring_slot = (ring_slot + 1 == ring_rows) ? 0 : ring_slot + 1;
        ring_out  = (ring_slot == 0) ? ring : ring_out + stride;
        Quad<ScoreT> out;
        out.v[0] = (ScoreT)narrow_chk<ScoreT>(P0, wrapped); out.v[1] = (ScoreT)narrow_chk<ScoreT>(P1, wrapped);
        out.v[2] = (ScoreT)narrow_chk<ScoreT>(P2, wrapped); out.v[3] = (ScoreT)narrow_chk<ScoreT>(P3, wrapped);
        if (full_wave)
        {
            if (!(dbg & 1)) *reinterpret_cast<Quad<ScoreT>*>(row_out + lane4 + 1 + kRelShift) = out;
            *reinterpret_cast<Quad<ScoreT>*>(ring_out + lane4 + 1 + kRelShift) = out;
        }
        else if (active)
        {
            if (!(dbg & 1)) *reinterpret_cast<Quad<ScoreT>*>(row_out + lane4 + 1 + kRelShift) = out;
            *reinterpret_cast<Quad<ScoreT>*>(ring_out + lane4 + 1 + kRelShift) = out;
        }
        if (bs == 0) // only rows whose band starts at column 0 have a real left-boundary value
        {
            if (lane == 0)
            {
                row_out[kRelShift]  = (ScoreT)rel0_val;
                ring_out[kRelShift] = (ScoreT)rel0_val;
            }
        }
        hbm_dirty = true;
        prev_bs   = bs;
        prev_rel0 = rel0_val;
    };

    // the table entry of row r+1 is loaded (raw, VGPR pair) while row r is computed and only moved to SGPRs at
    // the top of the next iteration, so its LDS latency is off the critical path
    RowInfo<true> raw_next = rowinfo[1];
    int32_t r              = 1;
    RowInfo<true> ri       = uniform_row(raw_next);
    raw_next               = rowinfo[min(2, graph_count)];
    while (r <= graph_count)
    {
        // ================= tight loop: consecutive simple rows =================
        for (;;)
        {
            const int32_t pred_count = ri.cnt();
            const int32_t bs         = ri.bs();
            const int32_t q          = (bs - prev_bs) >> 2;
            const bool simple        = (pred_count == 1) & (ri.pred(0) == r - 1) & (q <= 1);
            if (!simple) break;
            const uint32_t base = (uint32_t)ri.base();
            const int32_t c     = bs + lane4;
            const uint32_t rd4  = *reinterpret_cast<const uint32_t*>(lds_read + c);
            const int32_t cp0   = ((rd4 & 0xff) == base) ? match_score : mismatch_score;
            const int32_t cp1   = (((rd4 >> 8) & 0xff) == base) ? match_score : mismatch_score;
            const int32_t cp2   = (((rd4 >> 16) & 0xff) == base) ? match_score : mismatch_score;
            const int32_t cp3   = ((rd4 >> 24) == base) ? match_score : mismatch_score;
            const int32_t fe    = (bs > kCellsPerLane) ? min_score + gap_score : max(min_score, prev_rel0) + gap_score;
            const int32_t rel0_val = bs == 0 ? fe : min_score;
            const int32_t pend  = min(prev_bs + band_width - kCellsPerLane, max_column);
            const bool q0       = (q == 0);
            const int32_t a0    = wave_shr1(P3, prev_rel0);
            const int32_t b0 = wave_shl1(P0, 0), b1 = wave_shl1(P1, 0), b2 = wave_shl1(P2, 0), b3 = wave_shl1(P3, 0);
            const int32_t S0 = q0 ? a0 : P3, S1 = q0 ? P0 : b0, S2 = q0 ? P1 : b1, S3 = q0 ? P2 : b2, S4 = q0 ? P3 : b3;
            const bool valid = c <= pend;
            const int32_t s0 = valid ? max(S0 + cp0, S1 + gap_score) : min_score;
            const int32_t s1 = valid ? max(S1 + cp1, S2 + gap_score) : min_score;
            const int32_t s2 = valid ? max(S2 + cp2, S3 + gap_score) : min_score;
            const int32_t s3 = valid ? max(S3 + cp3, S4 + gap_score) : min_score;
            finish_row(s0, s1, s2, s3, fe, rel0_val, bs);
            r++;
            if (r > graph_count) break;
            ri       = uniform_row(raw_next);
            raw_next = rowinfo[min(r + 1, graph_count)];
        }
        if (r > graph_count) break;

        // ================= general row =================
        {
            const uint64_t t_general = (dbg & 4) ? clock64() : 0;
            const int32_t pred_count = ri.cnt();
            const int32_t bs         = ri.bs();
            const uint32_t base      = (uint32_t)ri.base();
            const int32_t c          = bs + lane4;
            const int32_t my_slot    = (ring_slot + 1 == ring_rows) ? 0 : ring_slot + 1; // slot this row will occupy
            const uint32_t rd4 = *reinterpret_cast<const uint32_t*>(lds_read + c);
            const int32_t cp0  = ((rd4 & 0xff) == base) ? match_score : mismatch_score;
            const int32_t cp1  = (((rd4 >> 8) & 0xff) == base) ? match_score : mismatch_score;
            const int32_t cp2  = (((rd4 >> 16) & 0xff) == base) ? match_score : mismatch_score;
            const int32_t cp3  = ((rd4 >> 24) == base) ? match_score : mismatch_score;

            auto from_regs = [&](int32_t& t0, int32_t& t1, int32_t& t2, int32_t& t3) {
                const int32_t q    = (bs - prev_bs) >> 2;
                const int32_t pend = min(prev_bs + band_width - kCellsPerLane, max_column);
                int32_t S0, S1, S2, S3, S4;
                if (q == 0)
                {
                    S0 = wave_shr1(P3, prev_rel0);
                    S1 = P0; S2 = P1; S3 = P2; S4 = P3;
                }
                else if (q == 1)
                {
                    S0 = P3;
                    S1 = wave_shl1(P0, 0); S2 = wave_shl1(P1, 0); S3 = wave_shl1(P2, 0); S4 = wave_shl1(P3, 0);
                }
                else
                {
                    const int src = lane + q;
                    S0 = __shfl(P3, src - 1);
                    S1 = __shfl(P0, src); S2 = __shfl(P1, src); S3 = __shfl(P2, src); S4 = __shfl(P3, src);
                }
                const bool valid = c <= pend; // c >= prev_bs always (band starts never decrease)
                t0 = valid ? max(S0 + cp0, S1 + gap_score) : min_score;
                t1 = valid ? max(S1 + cp1, S2 + gap_score) : min_score;
                t2 = valid ? max(S2 + cp2, S3 + gap_score) : min_score;
                t3 = valid ? max(S3 + cp3, S4 + gap_score) : min_score;
            };
            auto from_memory = [&](int32_t prow, int32_t& t0, int32_t& t1, int32_t& t2, int32_t& t3) {
                const int32_t pbs  = prow == 0 ? 0 : uniform_row(rowinfo[prow]).bs();
                const int32_t pend = min(pbs + band_width - kCellsPerLane, max_column);
                const bool valid   = !(c > pend || c < pbs);
                const int32_t dist = r - prow;
                const bool in_ring = dist < ring_rows;
                int32_t S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0;
                if (in_ring)
                {
                    int32_t slot = my_slot - dist;
                    if (slot < 0) slot += ring_rows;
                    if (valid)
                    {
                        const ScoreT* rowp = ring + slot * stride + (c - pbs) + kRelShift;
                        S0 = rowp[0];
                        const Quad<ScoreT> qd = *reinterpret_cast<const Quad<ScoreT>*>(rowp + 1);
                        S1 = qd.v[0]; S2 = qd.v[1]; S3 = qd.v[2]; S4 = qd.v[3];
                    }
                }
                else
                {
                    if (hbm_dirty) { wave_sync(); hbm_dirty = false; }
                    if (valid)
                    {
                        const ScoreT* rowp = scores + (int64_t)prow * stride + (c - pbs) + kRelShift;
                        S0 = rowp[0];
                        const Quad<ScoreT> qd = *reinterpret_cast<const Quad<ScoreT>*>(rowp + 1);
                        S1 = qd.v[0]; S2 = qd.v[1]; S3 = qd.v[2]; S4 = qd.v[3];
                    }
                }
                if (pbs > 0 && c == pbs) S0 = min_score; // relative-0 slot of a row whose band starts past column 0
                t0 = valid ? max(S0 + cp0, S1 + gap_score) : min_score;
                t1 = valid ? max(S1 + cp1, S2 + gap_score) : min_score;
                t2 = valid ? max(S2 + cp2, S3 + gap_score) : min_score;
                t3 = valid ? max(S3 + cp3, S4 + gap_score) : min_score;
            };
            // relative-0 slot of an older row
            auto rel0_of = [&](int32_t prow) -> int32_t {
                if (prow == r - 1) return prev_rel0;
                const int32_t pbs = prow == 0 ? 0 : uniform_row(rowinfo[prow]).bs();
                if (pbs > 0) return min_score;
                const int32_t dist = r - prow;
                if (dist < ring_rows)
                {
                    int32_t slot = my_slot - dist;
                    if (slot < 0) slot += ring_rows;
                    return wave_first((int32_t)ring[slot * stride + kRelShift]);
                }
                if (hbm_dirty) { wave_sync(); hbm_dirty = false; }
                return wave_first((int32_t)scores[(int64_t)prow * stride + kRelShift]);
            };

            const int32_t p0row   = pred_count == 0 ? 0 : ri.pred(0);
            const int32_t node_id = (pred_count > 3) ? (int32_t)g.sorted_poa[r - 1] : 0;
            auto pred_row = [&](int32_t p) -> int32_t {
                if (p == 0) return p0row;
                if (p == 1) return ri.pred(1);
                if (p == 2) return ri.pred(2);
                return (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1;
            };
            int32_t fe = 0, rel0_val = min_score;
            if (pred_count == 0)
            {
                if (bs == 0) rel0_val = gap_score; // carry-in stays 0 (reference quirk)
            }
            else
            {
                if (bs > kCellsPerLane && pred_count == 1)
                    fe = min_score + gap_score;
                else
                {
                    int32_t penalty = min_score;
                    for (int32_t p = 0; p < pred_count; p++) penalty = max(penalty, rel0_of(pred_row(p)));
                    fe = penalty + gap_score;
                }
                if (bs == 0) rel0_val = fe;
            }
            int32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            const int32_t np = max(pred_count, 1);
            for (int32_t p = 0; p < np; p++)
            {
                const int32_t prow = pred_row(p);
                int32_t t0, t1, t2, t3;
                if (prow == r - 1) from_regs(t0, t1, t2, t3);
                else from_memory(prow, t0, t1, t2, t3);
                if (p == 0) { s0 = t0; s1 = t1; s2 = t2; s3 = t3; }
                else { s0 = max(s0, t0); s1 = max(s1, t1); s2 = max(s2, t2); s3 = max(s3, t3); }
            }
            finish_row(s0, s1, s2, s3, fe, rel0_val, bs);
            r++;
            if (r <= graph_count)
            {
                ri       = uniform_row(raw_next);
                raw_next = rowinfo[min(r + 1, graph_count)];
            }
            if (dbg & 4) general_acc += (dbg & 8) ? 1 : clock64() - t_general;
        }
    }
    if ((dbg & 4) && general_row_acc) *general_row_acc += general_acc;
}

} // namespace gwhip
#include "poa_forward_packed.h"
#include "poa_forward_moves.h"
#include "poa_forward_moves_wide.h"
#include "poa_forward_moves_tb.h"
#include "poa_traceback_moves.h"
namespace gwhip
{

// ------------------------------------------------------------------------------------------------
// Multi-wave forward pass for wide bands (long reads: adaptive bands of 512 .. 1536 columns, HBM row table).
//
// A row of a 1536-column band is six 256-column passes; on one wavefront they run back to back (about 4 000 cycles per
// row). Round 2 first gave every pass of a row its own wavefront with two block barriers per row (carries exchanged in
// "u space"): about 2 300 cycles per row, because record decode, the LDS round trips and the two barriers are per row.
// This version has no per-row barrier at all: it is a pipeline of wavefronts over ABSOLUTE column blocks.
//
//   * Block b is the chunk anchors c in [256 b, 256 b + 256), i.e. the cells of columns 256 b + 1 .. 256 b + 256; it
//     belongs to wave b mod kSkWaves for every row. Band starts only move right, so a wave keeps its block until the band
//     has passed it and then moves on to block b + kSkWaves; a band of at most 1536 columns touches at most 7 blocks, so
//     with 8 waves no wave ever owns two blocks of a row. A lane's columns are the same in every row: its four read
//     characters are loaded once per block, not once per row.
//   * Everything a block of row r needs from other wavefronts comes from the block to its LEFT: the horizontal carry
//     (the cell of column 256 b of row r) and, for the lane at the block's first chunk, the cell of column 256 b of each
//     predecessor row. Wave w therefore runs one row (or more) behind wave w - 1 and never waits for a wave on its
//     right -- except for ring space, see below. All of it goes through the LDS ring of recent rows; the only
//     synchronisation is one progress word per wave (done[w] = last row the wave has finished or skipped), written
//     after the row's ring stores (one wavefront's LDS operations execute in order) and polled by the neighbours.
//     Candidates and the in-block prefix maximum do not depend on the carry, so the poll for the left neighbour's row r
//     comes after them and normally finds it there.
//   * Ring space: slot r mod R is rewritten by row r, so a wave may not run more than kSkLead rows ahead of its right
//     neighbour, and a predecessor counts as "in the ring" only up to R - kSkLead - 1 rows back. Rows with a predecessor
//     farther back (or with more than three predecessors) read it from the HBM matrix, which other wavefronts have
//     written: all waves take a block barrier in front of such a row (the decision is the same in all of them).
//   * The row table is read 64 rows at a time into registers (lane l holds row base + l, a row's record is six
//     v_readlane), prefetched one batch ahead, so the row loop issues no load that would wait for the score stores.
//
// The pass also writes the move bytes of poa_traceback_moves.h (one byte per cell: the move the reference's traceback
// takes from that cell -- rows up << 1 | columns left -- 0 = undecided), which turn the traceback of wide bands into a
// walk over a sheared LDS tile with run skipping. A move is written only where the forward pass saw exactly the operands
// the traceback's get_score() would see: not in chunks outside some predecessor's band, not in the band's first cell
// (its horizontal operand is the carry-in), not where the predecessor that attains the maximum is more than 63 rows up.
//
// All waves of the block call generic_forward_skew with the same arguments (wave 0 hands them over through MwArgs in
// LDS). Wave 0 alone runs every other phase.
// ------------------------------------------------------------------------------------------------
constexpr int kSkWaves = 8; // > kMaxAdaptiveBand / 256 + 1 blocks of a row
constexpr int kSkLead  = 4; // rows a wave may run ahead of its right neighbour

template <typename ScoreT> struct MwArgs
{
    int32_t op; // 1 = run a forward pass, 2 = leave the kernel
    int32_t graph_count, read_length, band_width, band_shift, max_column;
    float gradient;
    int32_t gap_score, mismatch_score, match_score;
    int32_t ring_rows;
    const uint8_t* read;
    ScoreT* scores;
    uint8_t* codes;
    int32_t dbg; // GWHIP_DEBUG (profiling counters, bits 12-15)
};
struct MwShared // in LDS, behind the regions of the single-wave layout
{
    // per wave and row & 7: {the last cell of the wave's block in row r, r}, written with ONE 8-byte store when the wave has
    // finished (or skipped) row r -- behind the row's ring stores, one wavefront's LDS operations execute in order. A wave is
    // never more than kSkLead + 1 rows ahead of its reader, so entry r & 7 says r until the reader is past row r; "wave w has
    // finished row r" is hand[w][r & 7][1] >= r. (Round 4: a progress word per wave next to the carries cost every row a
    // second LDS store and every reader a second load.)
    int32_t hand[kSkWaves][8][2];
    // per wave: the last row it SKIPPED (the band does not touch its block there). Skipped rows leave the entries alone: a wave
    // that works row r and then runs through skipped rows r + 1 .. r + 8 -- nothing holds it back there -- would otherwise
    // overwrite row r's entry while its right neighbour may still be rows behind (found by the protocol model in
    // tests/test_long_read_handover_model.py, never seen on the GPU). "Wave w has finished row r" is therefore
    // max(hand[w][r & 7][1], skipped[w]) >= r; a carry is only ever asked of a row its wave worked on.
    int32_t skipped[kSkWaves];
    unsigned long long prof; // profiling: the selected counter, summed over the waves
    int32_t fail;            // a wavefront gave up on a bounded wait (protocol error): the window reports a failure status
};
static_assert(sizeof(MwShared) <= 640, "kMwLds (gwhip_poa.hip) reserves 128 bytes for MwArgs and 640 for MwShared");
// Entry r & 7 is rewritten by row r + 8, which its wave finishes only after the reader has finished row r + 8 - kSkLead - 1
// (the ring-space rule): that must not be before row r (modelled in tests/test_long_read_handover_model.py).
static_assert(kSkLead + 1 <= 8, "a hand-over entry must survive until its reader is past the row");

__device__ __forceinline__ int32_t lds_poll(const int32_t* p)
{
    const int32_t v = *(const volatile __attribute__((address_space(3))) int32_t*)p;
    return __builtin_amdgcn_readfirstlane(v);
}

template <typename ScoreT, typename IdT, typename RowT, bool PROF = true>
__device__ __forceinline__ void generic_forward_skew(const MwArgs<ScoreT>& A, const GraphView<IdT>& g, const RowT* rowinfo, ScoreT* ring,
                                                     MwShared* shared, int wave, int lane, uint64_t* prof_out = nullptr)
{
    // profiling (GWHIP_DEBUG bits 12-15, sum over the waves, arrives in the "other" phase): 1 rows worked on, 2 rows
    // skipped, 3 block barriers (x 1000), 4 cycles in the pass, 5 cycles waiting for the left neighbour, 6 cycles waiting
    // for ring space, 7 rows on the general path, 8 cycles in the row bodies (waits included), 9 polls, 10 cycles waiting
    // for the left neighbour at the start of a row, 11 cycles in the bodies of first-block rows, 12 their number, 13 cycles
    // in the row-table batches, 14 cycles in the block barriers of far rows
    // (PROF = false: the helper wavefronts of a production kernel, whose copy of the arguments comes out of LDS and could
    // not be folded: no counter code in their row loop. Wave 0 passes a compile-time 0 in production kernels.)
    const int32_t sksel = PROF ? (A.dbg >> 12) & 15 : 0;
    uint64_t skacc      = 0;
    const uint64_t t_pass = sksel == 4 ? clock64() : 0;
    static_assert(sizeof(RowT) == 24, "the register copy of the row table holds six dwords per row");
    const int32_t graph_count = A.graph_count, read_length = A.read_length, band_width = A.band_width;
    const int32_t max_column = A.max_column, gap_score = A.gap_score;
    const int32_t min_score  = Limits<ScoreT>::min / 2;
    int32_t stride           = band_width + kRightPad;
    asm volatile("" : "+s"(stride)); // one scalar value: the row pointers advance by it with one add each, not by band_width and the pad
    const int32_t ring_rows  = A.ring_rows;
    const int32_t near_rows  = ring_rows - kSkLead - 1; // a predecessor closer than this is read from the ring
    // explicit global pointers: the arguments come out of an LDS struct, and behind a pointer of unknown address space
    // every access would be a flat instruction, which the row loop must not contain (it waits on both memory counters)
    typedef __attribute__((address_space(1))) ScoreT GScore;
    typedef __attribute__((address_space(1))) uint8_t GByte;
    // (the helper wavefronts' copy of A comes out of LDS, i.e. out of vector registers: made scalar here, once)
    GScore* scores           = (GScore*)wave_first64((uint64_t)A.scores);
    GByte* codes             = (GByte*)wave_first64((uint64_t)A.codes);
    const GByte* read        = (const GByte*)wave_first64((uint64_t)A.read);
    // match / mismatch in vector registers for the whole pass: the selects of the row loop take them as they are
    const int32_t match_v = (int32_t)pin_vgpr((uint32_t)A.match_score), mismatch_v = (int32_t)pin_vgpr((uint32_t)A.mismatch_score);
    const int left = (wave + kSkWaves - 1) % kSkWaves, right = (wave + 1) % kSkWaves;

    // ---- this wave's block: per-lane values that only change when the wave moves on to its next block ----
    int32_t blk = wave;
    uint32_t rd4;                           // read characters of columns c .. c+3 (the characters of cells c+1 .. c+4)
    int32_t cvec;                           // chunk anchor column c = 256 blk + 4 lane
    int32_t cg0, cg1, cg2, cg3;             // (c + k) * gap: the cells' offsets in "u space"
    bool c_in_read;                         // c <= max_column
    int32_t blk_carry_gap;                  // u-space offset of the left neighbour's last cell (column 256 blk)
    auto enter_block = [&](int32_t block) {
        cvec = block * 256 + 4 * lane;
        uint32_t v = cvec < read_length + 8 ? *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(read + cvec) : 0u;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory"); // here, once per block, not at its first use in the row loop
        rd4 = v;
        cg0 = cvec * gap_score; cg1 = cg0 + gap_score; cg2 = cg1 + gap_score; cg3 = cg2 + gap_score;
        c_in_read = cvec <= max_column;
        blk_carry_gap = (256 * block - 1) * gap_score;
    };
    enter_block(blk);

    // ---- row table in registers, 64 rows at a time: lane l holds row batch_base + l ----
    // raw dwords of the next batch (prefetched), and of the current one what the row loop reads with v_readlane:
    //   m0 = base [0:8) | predecessor count [8:15) | general path [15] | distance to predecessor 0 [16:24) | 1 [24:32)
    //   m1 = distance to predecessor 2 [0:8) | (band start - predecessor k's band start) / 4 for k = 0, 1, 2 [8:16) [16:24) [24:32)
    // A row takes the general path when it has more than three predecessors, or one that is not safely in the ring, or
    // one whose band start is too far left to encode.
    int32_t nxt[6];
    int32_t m0v = 0, m1v = 0, bsv = 0, p0v = 0, p1v = 0, p2v = 0, prev_bsv = 0;
    //   kb = band start [0:16) | what THIS wave does with the row [16:19) | the wave's block in that row [19:27)
    //        0..2 straight-line row with 1 / 2 / 3 predecessors, 3..5 the same in the first block of the band, 6 general row,
    //        7 the band does not reach the wave's block. Band starts only move right, so the wave's block of a row is a
    //        function of the row alone: the first block b >= band start / 256 with b = wave (mod kSkWaves).
    int32_t kbv = 0;
    auto load_batch = [&](int32_t base, int32_t (&dst)[6]) {
        const int32_t row = min(base + lane, graph_count);
        const __attribute__((address_space(1))) int32_t* src = (const __attribute__((address_space(1))) int32_t*)(rowinfo + row);
#pragma unroll
        for (int k = 0; k < 6; k++) dst[k] = src[k];
        // wait for the batch here, once per 64 rows: left to the compiler the wait lands in front of every row's first
        // use of the registers (it cannot prove across the loop that the load has landed), i.e. behind every row's stores
        asm volatile("" : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]));
    };
    auto adopt_batch = [&](int32_t base) { // nxt holds rows base .. base + 63
        prev_bsv          = bsv;
        const int32_t row = base + lane;
        const int32_t w0  = nxt[0];
        const int32_t cnt = (w0 >> 8) & 0x7f;
        bsv = nxt[1];
        p0v = cnt > 0 ? nxt[2] : 0;
        p1v = nxt[3];
        p2v = nxt[4];
        auto band_start_of = [&](int32_t p) -> int32_t { // band start of row p <= row, from this batch or the previous one
            const int32_t idx = p - base;
            const int32_t cur = __builtin_amdgcn_ds_bpermute((idx & 63) << 2, bsv);
            const int32_t prv = __builtin_amdgcn_ds_bpermute(((idx + 64) & 63) << 2, prev_bsv);
            return p == 0 ? 0 : (idx >= 0 ? cur : prv);
        };
        const int32_t d0 = row - p0v, d1 = cnt > 1 ? row - p1v : 0, d2 = cnt > 2 ? row - p2v : 0;
        // (the three permutes run with every lane active: a lane that is switched off supplies no value)
        const int32_t q0 = band_start_of(p0v), q1 = band_start_of(p1v), q2 = band_start_of(p2v);
        const int32_t e0 = bsv - q0, e1 = cnt > 1 ? bsv - q1 : 0, e2 = cnt > 2 ? bsv - q2 : 0;
        const int32_t dmax = max(d0, max(d1, d2)), emax = max(e0, max(e1, e2));
        const bool general = cnt > 3 || dmax >= near_rows || dmax > 63 || emax > 1020 || row > graph_count;
        m0v = (w0 & 0x7fff) | (general ? 0x8000 : 0) | ((d0 & 0xff) << 16) | ((d1 & 0xff) << 24);
        m1v = (d2 & 0xff) | (((e0 >> 2) & 0xff) << 8) | (((e1 >> 2) & 0xff) << 16) | (((e2 >> 2) & 0xff) << 24);
        const int32_t b_lo  = bsv >> 8;
        const int32_t blk_r = b_lo + ((wave - b_lo) & (kSkWaves - 1));
        const bool skip     = (blk_r << 8) >= bsv + band_width;
        const int32_t kind  = skip ? 7 : (general ? 6 : (blk_r == b_lo ? 3 : 0) + min(max(cnt, 1), 3) - 1);
        kbv = bsv | (kind << 16) | (blk_r << 19);
    };
    load_batch(1, nxt);
    adopt_batch(1);
    load_batch(65, nxt);

    // ---- the neighbours' progress (MwShared::hand): entry [w][r & 7] = {wave w's last cell of row r, r} ----
    const uint32_t left_hand  = lds_addr(&shared->hand[left][0][0]);
    const uint32_t right_hand = lds_addr(&shared->hand[right][0][0]);
    const uint32_t my_hand    = lds_addr(&shared->hand[wave][0][0]);
    auto hand_load = [&](uint32_t base, int32_t row) -> uint2 { // one 8-byte LDS load (volatile: polled)
        const uint32_t addr = base + 8u * (uint32_t)(row & 7);
        const uint64_t v    = *reinterpret_cast<const volatile __attribute__((address_space(3))) uint64_t*>(addr);
        return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    };
    auto hand_wait = [&]() {};
    auto publish = [&](int32_t row, int32_t carry) { // behind the row's ring stores: LDS runs one wave's operations in order
        const uint32_t ha = my_hand + 8u * (uint32_t)(row & 7);
        u32x2 pr;
        pr.x = (uint32_t)carry;
        pr.y = (uint32_t)row;
        asm volatile("s_mov_b64 exec, 1\n\tds_write_b64 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(ha), "v"(pr) : "memory");
    };
    int32_t left_done = 0, right_done = 0; // cached: the neighbour has finished at least this row
    // Waits are bounded: a protocol error must not end as a wavefront spinning forever. After the first timeout a wave
    // stops waiting altogether (both cached values jump to the largest row) and raises MwShared::fail; wave 0 turns that into
    // a failure status of the window (nw_banded returns kNwPipelineFailed -> StatusType::generic_error): never a silent result.
    auto give_up = [&]() {
        shared->fail = 1; // (every lane, the same word: no divergent branch next to the cached values, which must stay scalar)
        left_done  = INT32_MAX;
        right_done = INT32_MAX;
    };
    typedef __attribute__((address_space(3))) const volatile int32_t* LdsWord;
    const LdsWord left_skipped  = (LdsWord)&shared->skipped[left];
    const LdsWord right_skipped = (LdsWord)&shared->skipped[right];
    const uint32_t my_skipped   = lds_addr(&shared->skipped[wave]);
    auto publish_skip = [&](int32_t row) {
        asm volatile("s_mov_b64 exec, 1\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(my_skipped), "v"(row) : "memory");
    };
    auto wait_left = [&](int32_t row) {
        const uint64_t t_w = sksel == 5 ? clock64() : 0;
        int32_t spins = 0;
        while (left_done < row)
        {
            if (sksel == 9) skacc++;
            const uint2 e    = hand_load(left_hand, row);
            const int32_t sk = *left_skipped;
            left_done        = max(left_done, max(wave_first((int32_t)e.y), wave_first(sk)));
            if (left_done < row)
            {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) give_up();
            }
        }
        if (sksel == 5) skacc += clock64() - t_w;
    };
    auto wait_right = [&](int32_t row) {
        const uint64_t t_w = sksel == 6 ? clock64() : 0;
        int32_t spins = 0;
        while (right_done < row)
        {
            if (sksel == 9) skacc++;
            const uint2 e    = hand_load(right_hand, row);
            const int32_t sk = *right_skipped;
            right_done       = max(right_done, max(wave_first((int32_t)e.y), wave_first(sk)));
            if (right_done < row)
            {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) give_up();
            }
        }
        if (sksel == 6) skacc += clock64() - t_w;
    };
    auto left_carry = [&](int32_t row) -> int32_t { // of a row the left neighbour has finished
        const uint2 e = hand_load(left_hand, row);
        hand_wait();
        return wave_first((int32_t)e.x);
    };

    // per-row state kept incrementally: the ring slot of row r, its byte offset, the row's HBM score / code rows
    const uint32_t ring_lds = lds_addr(ring);
    const int32_t row_bytes = stride * (int32_t)sizeof(ScoreT);
    const int32_t ring_span = ring_rows * row_bytes;
    uint32_t ring_off_r = (uint32_t)((1 % ring_rows) * row_bytes);
    GScore* scores_row  = scores + stride;
    GByte* codes_row    = codes ? codes + stride : codes;
    auto next_row = [&]() {
        ring_off_r = ring_off_r + (uint32_t)row_bytes;
        ring_off_r = ring_off_r == (uint32_t)ring_span ? 0u : ring_off_r;
        scores_row += stride;
        codes_row += stride;
    };
    // (two loops, the outer one over the 64-row batches of the register row table: the row loop carries no batch test)
    for (int32_t r0 = 1; r0 <= graph_count; r0 += 64)
    {
    if (r0 > 1)
    {
        const uint64_t t_b = sksel == 13 ? clock64() : 0;
        adopt_batch(r0);
        load_batch(r0 + 64, nxt);
        if (sksel == 13) skacc += clock64() - t_b;
    }
    const int32_t r_last = min(r0 + 63, graph_count);
    for (int32_t r = r0; r <= r_last; r++, next_row())
    {
        const int32_t ridx = r - r0;
        const uint32_t m0        = (uint32_t)__builtin_amdgcn_readlane(m0v, ridx);
        const uint32_t m1        = (uint32_t)__builtin_amdgcn_readlane(m1v, ridx);
        const uint32_t kb        = (uint32_t)__builtin_amdgcn_readlane(kbv, ridx);
        const int32_t bs         = (int32_t)(kb & 0xffffu);
        const int32_t kind       = (int32_t)((kb >> 16) & 7u);
        const uint32_t base      = m0 & 0xffu;
        const int32_t pred_count = (int32_t)((m0 >> 8) & 0x7fu);
        const bool general       = (m0 & 0x8000u) != 0;
        auto slot_now  = [&]() -> int32_t { return (int32_t)(ring_off_r / (uint32_t)row_bytes); }; // general rows only
        auto slot_back = [&](int32_t d) -> int32_t { // slot of row r - d, d < ring_rows
            const int32_t sl = slot_now() - d;
            return sl < 0 ? sl + ring_rows : sl;
        };
        // A predecessor that is not safely in the ring comes from the HBM matrix, written by other wavefronts: every wave
        // finishes its earlier rows and its stores first (same decision in all waves).
        int32_t p0 = 0, p1 = 0, p2 = 0;
        bool far = false;
        if (general)
        {
            p0 = __builtin_amdgcn_readlane(p0v, ridx);
            p1 = __builtin_amdgcn_readlane(p1v, ridx);
            p2 = __builtin_amdgcn_readlane(p2v, ridx);
            if (pred_count <= 3)
                far = (r - p0 >= near_rows) || (pred_count > 1 && r - p1 >= near_rows) || (pred_count > 2 && r - p2 >= near_rows);
            else
            {
                const int32_t node_id = wave_first((int32_t)g.sorted_poa[r - 1]);
                for (int32_t p = 0; p < pred_count; p++)
                {
                    const int32_t prow = p == 0 ? p0 : (p == 1 ? p1 : (p == 2 ? p2 : wave_first((int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1)));
                    far                = far || (r - prow >= near_rows);
                }
            }
            const uint64_t t_bar = (far && sksel == 14) ? clock64() : 0;
            if (far) block_barrier();
            if (far && sksel == 3) skacc += 1000;
            if (far && sksel == 14) skacc += clock64() - t_bar;
        }

        if ((int32_t)(kb >> 19) != blk) // the band has passed this wave's block: on to the next one
        {
            blk = (int32_t)(kb >> 19);
            enter_block(blk);
        }
        if (kind == 7) // the band has not reached this block yet, or has passed it
        {
            publish_skip(r); // (not an entry of MwShared::hand: see there)
            if (sksel == 2) skacc++;
            continue;
        }
        const bool first_block = (bs >> 8) == blk;
        const int32_t tg       = cvec - bs; // index of the lane's first cell in the band
        // cells of column 256 blk of the predecessor rows are the left neighbour's: it must be past row r - 1
        const uint64_t t_ws = sksel == 10 ? clock64() : 0;
        if (left_done < r - 1) wait_left(r - 1);
        if (sksel == 10) skacc += clock64() - t_ws;
        if (sksel == 1) skacc++;
        if (sksel == 12 && first_block) skacc++;
        const uint64_t t_rb = (sksel == 8 || (sksel == 11 && first_block)) ? clock64() : 0;

        // what the two row bodies leave behind
        int32_t H0, H1, H2, H3;
        uint32_t code4;
        int32_t rel0_val = min_score;

        // match / mismatch costs of the row's base against the lane's four read characters
        const int32_t cp0 = ((rd4 & 0xff) == base) ? match_v : mismatch_v;
        const int32_t cp1 = (((rd4 >> 8) & 0xff) == base) ? match_v : mismatch_v;
        const int32_t cp2 = (((rd4 >> 16) & 0xff) == base) ? match_v : mismatch_v;
        const int32_t cp3 = ((rd4 >> 24) == base) ? match_v : mismatch_v;
        // the row's finish, shared by both bodies: prefix maximum in u space (u = v - (c + k) * gap, an offset common to the
        // whole row, so the carry is converted with the column it belongs to), the hand-over from the left, H and the stores
        auto finish_row = [&](int32_t s0, int32_t s1, int32_t s2, int32_t s3, int32_t fe_if_first) {
            int32_t u0 = s0 - cg0, u1 = s1 - cg1, u2 = s2 - cg2, u3 = s3 - cg3;
            if (first_block) // lanes left of the band start carry no cells (lanes right of its end feed nobody)
            {
                const bool dead = tg < 0;
                u0 = dead ? INT32_MIN : u0; u1 = dead ? INT32_MIN : u1; u2 = dead ? INT32_MIN : u2; u3 = dead ? INT32_MIN : u3;
            }
            const int32_t m0_ = u0, m1_ = max(m0_, u1), m2_ = max(m1_, u2), m3_ = max(m2_, u3);
            const int32_t incl = wave_inclusive_max(m3_);
            int32_t carry_u;
            if (first_block)
                carry_u = fe_if_first - (bs - 1) * gap_score; // the carry-in is the element of column bs - 1
            else
            {
                if (left_done < r) wait_left(r);
                const int32_t hl = left_carry(r);
                carry_u          = hl - (256 * blk - 1) * gap_score; // the left neighbour's cell of column 256 blk
            }
            const int32_t excl = max(wave_shr1(incl, INT32_MIN), carry_u);
            H0 = (ScoreT)(max(m0_, excl) + cg0);
            H1 = (ScoreT)(max(m1_, excl) + cg1);
            H2 = (ScoreT)(max(m2_, excl) + cg2);
            H3 = (ScoreT)(max(m3_, excl) + cg3);
        };

        if (!general)
        {
            // ========= at most three predecessors, all in the ring: no global load, no loop, its own tail =========
            // (one straight-line instantiation per predecessor count and per "first block of the row": a lone wavefront
            // pays about 7 cycles per issued instruction, so the row's length in instructions is the row's time)
            auto fast_row = [&](auto np_tag, auto first_tag) {
                constexpr int NP     = decltype(np_tag)::value;
                constexpr bool FIRST = decltype(first_tag)::value;
                constexpr uint32_t esz = sizeof(ScoreT);
                asm volatile("; FASTROW_BEGIN %0 %1" ::"n"(NP), "n"((int)FIRST));
                const int32_t wlim = band_width - kCellsPerLane;
                auto pred_base = [&](int32_t d, int32_t e) -> uint32_t { // LDS byte address of that row's element (tg = 0) + kRelShift
                    int32_t o = (int32_t)ring_off_r - d * row_bytes;
                    o         = o < 0 ? o + ring_span : o;
                    return ring_lds + (uint32_t)o + (uint32_t)(kRelShift + e) * esz;
                };
                const int32_t d0 = (int32_t)((m0 >> 16) & 0xff), e0 = (int32_t)((m1 >> 8) & 0xff) << 2;
                const int32_t d1 = (int32_t)(m0 >> 24), d2 = (int32_t)(m1 & 0xff); // rows up to predecessors 1 and 2
                const int32_t tg4 = tg * (int32_t)esz;
                // predecessor k: cells of columns c .. c+4 sit at ring element (c - pbs) + kRelShift of its row; a read outside
                // the LDS allocation (chunks that lie outside that row's band) returns zero and is masked below
                const uint32_t a0 = pred_base(d0, e0) + (uint32_t)tg4;
                const int32_t Sa  = lds_ld_at<ScoreT>(a0);
                const auto qa     = lds_ld_qv<ScoreT>(a0 + esz);
                const bool va     = (uint32_t)(tg + e0) <= (uint32_t)wlim && c_in_read;
                int32_t Sb = 0, Sc = 0;
                typename QuadVec<ScoreT>::type qb = {0, 0, 0, 0}, qc = {0, 0, 0, 0};
                bool vb = true, vc = true;
                if constexpr (NP > 1)
                {
                    const int32_t e1 = (int32_t)((m1 >> 16) & 0xff) << 2;
                    const uint32_t a1 = pred_base(d1, e1) + (uint32_t)tg4;
                    Sb = lds_ld_at<ScoreT>(a1);
                    qb = lds_ld_qv<ScoreT>(a1 + esz);
                    vb = (uint32_t)(tg + e1) <= (uint32_t)wlim && c_in_read;
                }
                if constexpr (NP > 2)
                {
                    const int32_t e2 = (int32_t)(m1 >> 24) << 2;
                    const uint32_t a2 = pred_base(d2, e2) + (uint32_t)tg4;
                    Sc = lds_ld_at<ScoreT>(a2);
                    qc = lds_ld_qv<ScoreT>(a2 + esz);
                    vc = (uint32_t)(tg + e2) <= (uint32_t)wlim && c_in_read;
                }
                // the left neighbour's entry of this row is requested now, looked at after the arithmetic
                uint2 probe = make_uint2(0u, 0u);
                if constexpr (!FIRST) probe = hand_load(left_hand, r);
                // match / mismatch costs of the row's base against the lane's four read characters
                const int32_t cp0 = ((rd4 & 0xff) == base) ? match_v : mismatch_v;
                const int32_t cp1 = (((rd4 >> 8) & 0xff) == base) ? match_v : mismatch_v;
                const int32_t cp2 = (((rd4 >> 16) & 0xff) == base) ? match_v : mismatch_v;
                const int32_t cp3 = ((rd4 >> 24) == base) ? match_v : mismatch_v;
                int32_t D[4], V[4], s[4];
                D[0] = Sa + cp0; D[1] = (int32_t)qa.x + cp1; D[2] = (int32_t)qa.y + cp2; D[3] = (int32_t)qa.z + cp3;
                V[0] = (int32_t)qa.x + gap_score; V[1] = (int32_t)qa.y + gap_score; V[2] = (int32_t)qa.z + gap_score; V[3] = (int32_t)qa.w + gap_score;
#pragma unroll
                for (int k = 0; k < 4; k++) s[k] = va ? (int32_t)(ScoreT)max(D[k], V[k]) : min_score;
                // the move a cell's maximum stands for (poa_traceback_moves.h): rows up << 1 | columns left
                const uint32_t mvV0 = (uint32_t)d0 << 1, mvD0 = mvV0 | 1u;
                uint32_t kD[4] = {mvD0, mvD0, mvD0, mvD0}, kV[4] = {mvV0, mvV0, mvV0, mvV0};
                bool undecided = !va;
                if constexpr (NP > 1)
                {
                    int32_t Db[4], Vb[4];
                    Db[0] = Sb + cp0; Db[1] = (int32_t)qb.x + cp1; Db[2] = (int32_t)qb.y + cp2; Db[3] = (int32_t)qb.z + cp3;
                    Vb[0] = (int32_t)qb.x + gap_score; Vb[1] = (int32_t)qb.y + gap_score; Vb[2] = (int32_t)qb.z + gap_score; Vb[3] = (int32_t)qb.w + gap_score;
                    undecided = undecided || !vb;
                    const uint32_t mvV1 = (uint32_t)d1 << 1, mvD1 = mvV1 | 1u;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        s[k]  = max(s[k], vb ? (int32_t)(ScoreT)max(Db[k], Vb[k]) : min_score);
                        kD[k] = Db[k] > D[k] ? mvD1 : kD[k]; D[k] = max(D[k], Db[k]);
                        kV[k] = Vb[k] > V[k] ? mvV1 : kV[k]; V[k] = max(V[k], Vb[k]);
                    }
                }
                if constexpr (NP > 2)
                {
                    int32_t Dc[4], Vc[4];
                    Dc[0] = Sc + cp0; Dc[1] = (int32_t)qc.x + cp1; Dc[2] = (int32_t)qc.y + cp2; Dc[3] = (int32_t)qc.z + cp3;
                    Vc[0] = (int32_t)qc.x + gap_score; Vc[1] = (int32_t)qc.y + gap_score; Vc[2] = (int32_t)qc.z + gap_score; Vc[3] = (int32_t)qc.w + gap_score;
                    undecided = undecided || !vc;
                    const uint32_t mvV2 = (uint32_t)d2 << 1, mvD2 = mvV2 | 1u;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        s[k]  = max(s[k], vc ? (int32_t)(ScoreT)max(Dc[k], Vc[k]) : min_score);
                        kD[k] = Dc[k] > D[k] ? mvD2 : kD[k]; D[k] = max(D[k], Dc[k]);
                        kV[k] = Vc[k] > V[k] ? mvV2 : kV[k]; V[k] = max(V[k], Vc[k]);
                    }
                }
                // prefix maximum in u space (u = v - (c + k) * gap: an offset common to the whole row, so a carry is converted
                // with the column it belongs to)
                int32_t u0 = s[0] - cg0, u1 = s[1] - cg1, u2 = s[2] - cg2, u3 = s[3] - cg3;
                if constexpr (FIRST) // lanes left of the band start carry no cells (lanes right of its end feed nobody)
                {
                    const bool dead = tg < 0;
                    u0 = dead ? INT32_MIN : u0; u1 = dead ? INT32_MIN : u1; u2 = dead ? INT32_MIN : u2; u3 = dead ? INT32_MIN : u3;
                }
                const int32_t m0_ = u0, m1_ = max(m0_, u1), m2_ = max(m1_, u2), m3_ = max(m2_, u3);
                const int32_t incl = wave_inclusive_max(m3_);
                int32_t carry_u, rel0 = min_score;
                if constexpr (FIRST)
                {
                    // left boundary (cudapoa_nw_banded.cuh:293-326); the relative-0 slot of a row whose band starts past column
                    // 0 holds min_score, so only the rows at the top of the matrix read their predecessors' slots
                    int32_t fe = 0;
                    if (pred_count == 0)
                    {
                        if (bs == 0) rel0 = (ScoreT)gap_score; // carry-in stays 0: reference quirk
                    }
                    else
                    {
                        if (bs > kCellsPerLane && NP == 1)
                            fe = min_score + gap_score;
                        else
                        {
                            auto rel0_of = [&](int32_t d, int32_t e) -> int32_t {
                                if (bs - e > 0) return min_score;
                                return wave_first((int32_t)lds_ld_at<ScoreT>(pred_base(d, -kRelShift) + kRelShift * esz));
                            };
                            int32_t penalty = max(min_score, rel0_of(d0, e0));
                            if constexpr (NP > 1) penalty = max(penalty, rel0_of((int32_t)(m0 >> 24), (int32_t)((m1 >> 16) & 0xff) << 2));
                            if constexpr (NP > 2) penalty = max(penalty, rel0_of((int32_t)(m1 & 0xff), (int32_t)(m1 >> 24) << 2));
                            fe = penalty + gap_score;
                        }
                        if (bs == 0) rel0 = (ScoreT)fe;
                    }
                    carry_u = fe - (bs - 1) * gap_score; // the carry-in is the element of column bs - 1
                }
                else
                {
                    hand_wait();
                    const int32_t probe_row = wave_first((int32_t)probe.y);
                    int32_t hl              = wave_first((int32_t)probe.x);
                    if (probe_row < r) // the probe came too early
                    {
                        if (left_done < r) wait_left(r);
                        hl = left_carry(r);
                    }
                    left_done = max(left_done, r);
                    carry_u   = hl - blk_carry_gap; // the left neighbour's cell of column 256 blk
                }
                const int32_t excl = max(wave_shr1(incl, INT32_MIN), carry_u);
                const int32_t Hk[4] = {(int32_t)(ScoreT)(max(m0_, excl) + cg0), (int32_t)(ScoreT)(max(m1_, excl) + cg1),
                                       (int32_t)(ScoreT)(max(m2_, excl) + cg2), (int32_t)(ScoreT)(max(m3_, excl) + cg3)};
                // move bytes: diagonal through the first slot attaining H, else vertical, else horizontal (0 rows up, 1 left)
                uint32_t code4 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const uint32_t ck = Hk[k] == D[k] ? kD[k] : (Hk[k] == V[k] ? kV[k] : 1u);
                    code4 |= ck << (8 * k);
                }
                if constexpr (FIRST) code4 = tg == 0 ? (code4 & 0xffffff00u) : code4; // the band's first cell
                code4 = undecided ? 0u : code4;
                // flow control: slot r mod R still holds row r - R, which the right neighbour may need until it has finished
                // row r - kSkLead - 1
                if (right_done < r - kSkLead - 1) wait_right(r - kSkLead - 1);
                if ((uint32_t)tg < (uint32_t)band_width)
                {
                    typedef typename QuadVec<ScoreT>::type V4;
                    const V4 out = {(ScoreT)Hk[0], (ScoreT)Hk[1], (ScoreT)Hk[2], (ScoreT)Hk[3]};
                    // explicit LDS store: the hand-over entry below is ordered behind it by the LDS queue, not by a wait
                    *reinterpret_cast<__attribute__((address_space(3))) V4*>(ring_lds + ring_off_r + (uint32_t)(kRelShift + 1) * esz + (uint32_t)tg4) = out;
                    // (streaming store: the matrix is read again only by far predecessors, the sink scan and recomputed steps,
                    // and 1 TB of it per long-read set should not push the graphs and the trace codes out of the L2).
                    // Row base in a scalar register pair + the lane's 32-bit offset (tg >= 0 here): no 64-bit vector address
                    // arithmetic per row.
                    if constexpr (sizeof(ScoreT) == 4)
                        asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt" ::"v"((uint32_t)tg4), "v"(out), "s"(scores_row),
                                     "n"((kRelShift + 1) * 4)
                                     : "memory");
                    else
                        __builtin_nontemporal_store(out, reinterpret_cast<__attribute__((address_space(1))) V4*>(scores_row + kRelShift + 1 + tg));
                    if (codes)
                        asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"((uint32_t)tg), "v"(code4), "s"(codes_row), "n"(kRelShift + 1)
                                     : "memory");
                }
                if constexpr (FIRST)
                {
                    if (lane == 0)
                    {
                        *reinterpret_cast<__attribute__((address_space(3))) ScoreT*>(ring_lds + ring_off_r + (uint32_t)kRelShift * esz) = (ScoreT)rel0;
                        scores_row[kRelShift] = (ScoreT)rel0;
                    }
                }
                publish(r, __builtin_amdgcn_readlane(Hk[3], kWave - 1));
                asm volatile("; FASTROW_END %0 %1" ::"n"(NP), "n"((int)FIRST));
            };
            if (kind == 0) fast_row(std::integral_constant<int, 1>{}, std::false_type{});
            else if (kind == 1) fast_row(std::integral_constant<int, 2>{}, std::false_type{});
            else if (kind == 3) fast_row(std::integral_constant<int, 1>{}, std::true_type{});
            else if (kind == 4) fast_row(std::integral_constant<int, 2>{}, std::true_type{});
            else if (kind == 2) fast_row(std::integral_constant<int, 3>{}, std::false_type{});
            else fast_row(std::integral_constant<int, 3>{}, std::true_type{});
            if (sksel == 8 || (sksel == 11 && first_block)) skacc += clock64() - t_rb;
            continue;
        }
        else
        {
            // ================= general row: any number of predecessors, far ones from the HBM matrix =================
            if (sksel == 7) skacc++;
            const bool many       = pred_count > 3;
            const int32_t node_id = many ? wave_first((int32_t)g.sorted_poa[r - 1]) : 0;
            auto pred_row = [&](int32_t p) -> int32_t {
                if (pred_count == 0) return 0;
                if (p < 3) return p == 0 ? p0 : (p == 1 ? p1 : p2);
                return wave_first((int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1);
            };
            auto bs_of = [&](int32_t row) -> int32_t {
                return row == 0 ? 0 : band_start_for_row(row, A.gradient, band_width, A.band_shift, max_column);
            };
            const int32_t np = max(pred_count, 1);
            int32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            int32_t bD0 = 0, bD1 = 0, bD2 = 0, bD3 = 0, bV0 = 0, bV1 = 0, bV2 = 0, bV3 = 0;
            int32_t kD0 = 0, kD1 = 0, kD2 = 0, kD3 = 0, kV0 = 0, kV1 = 0, kV2 = 0, kV3 = 0;
            bool undecided = false;
            for (int32_t p = 0; p < np; p++)
            {
                const int32_t prow = pred_row(p);
                const int32_t pbs  = bs_of(prow);
                const int32_t pend = min(pbs + band_width - kCellsPerLane, max_column);
                const bool valid   = !(cvec > pend || cvec < pbs);
                int32_t S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0; // predecessor row, columns c .. c+4
                if (r - prow < near_rows) // wave-uniform
                {
                    if (valid)
                    {
                        const ScoreT* rowp = ring + slot_back(r - prow) * stride + (cvec - pbs) + kRelShift;
                        S0 = lds_ld(rowp);
                        const Quad<ScoreT> qd = lds_ld_quad(rowp + 1);
                        S1 = qd.v[0]; S2 = qd.v[1]; S3 = qd.v[2]; S4 = qd.v[3];
                    }
                }
                else if (valid) // the HBM matrix (complete: block barrier above)
                {
                    const GScore* rowp = scores + (int64_t)prow * stride + (cvec - pbs) + kRelShift;
                    S0 = rowp[0]; S1 = rowp[1]; S2 = rowp[2]; S3 = rowp[3]; S4 = rowp[4];
                }
                int32_t D0, D1, D2, D3, V0, V1, V2, V3, t0, t1, t2, t3;
                if (valid)
                {
                    D0 = S0 + cp0; D1 = S1 + cp1; D2 = S2 + cp2; D3 = S3 + cp3;
                    V0 = S1 + gap_score; V1 = S2 + gap_score; V2 = S3 + gap_score; V3 = S4 + gap_score;
                    t0 = (ScoreT)max(D0, V0); t1 = (ScoreT)max(D1, V1); t2 = (ScoreT)max(D2, V2); t3 = (ScoreT)max(D3, V3);
                }
                else
                {
                    D0 = D1 = D2 = D3 = V0 = V1 = V2 = V3 = t0 = t1 = t2 = t3 = min_score;
                    undecided = true;
                }
                const int32_t up = r - prow <= MtGeometry<true>::kMaxUp ? r - prow : 0; // rows up, 0 = too far for a move byte
                if (p == 0)
                {
                    s0 = t0; s1 = t1; s2 = t2; s3 = t3;
                    bD0 = D0; bD1 = D1; bD2 = D2; bD3 = D3; bV0 = V0; bV1 = V1; bV2 = V2; bV3 = V3;
                    kD0 = kD1 = kD2 = kD3 = kV0 = kV1 = kV2 = kV3 = up;
                }
                else
                {
                    s0 = max(s0, t0); s1 = max(s1, t1); s2 = max(s2, t2); s3 = max(s3, t3);
                    if (D0 > bD0) { bD0 = D0; kD0 = up; }
                    if (D1 > bD1) { bD1 = D1; kD1 = up; }
                    if (D2 > bD2) { bD2 = D2; kD2 = up; }
                    if (D3 > bD3) { bD3 = D3; kD3 = up; }
                    if (V0 > bV0) { bV0 = V0; kV0 = up; }
                    if (V1 > bV1) { bV1 = V1; kV1 = up; }
                    if (V2 > bV2) { bV2 = V2; kV2 = up; }
                    if (V3 > bV3) { bV3 = V3; kV3 = up; }
                }
            }
            int32_t fe = 0;
            if (first_block) // cudapoa_nw_banded.cuh:293-326
            {
                auto rel0_of = [&](int32_t row) -> int32_t {
                    if (r - row < near_rows) return wave_first((int32_t)lds_ld(ring + slot_back(r - row) * stride + kRelShift));
                    return wave_first((int32_t)scores[(int64_t)row * stride + kRelShift]);
                };
                if (pred_count == 0)
                {
                    if (bs == 0) rel0_val = (ScoreT)gap_score; // carry-in stays 0: reference quirk
                }
                else
                {
                    if (bs > kCellsPerLane && pred_count == 1)
                        fe = min_score + gap_score;
                    else
                    {
                        int32_t penalty = min_score;
                        for (int32_t p = 0; p < pred_count; p++) penalty = max(penalty, rel0_of(pred_row(p)));
                        fe = penalty + gap_score;
                    }
                    if (bs == 0) rel0_val = (ScoreT)fe;
                }
            }
            // This path's global loads end here, explicitly: the compiler merges the two paths' common tail, and a join
            // with loads possibly in flight on one side gets a vmcnt(0) wait that the load-free path would pay as well --
            // behind its own score stores, one HBM acknowledgement per row.
            __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
            finish_row(s0, s1, s2, s3, fe);
            auto code_of = [&](int32_t H, int32_t bD, int32_t kD, int32_t bV, int32_t kV) -> uint32_t {
                const uint32_t cd = kD != 0 ? ((uint32_t)kD << 1) | 1u : 0u; // (kD, kV: rows up to the slot that attains the maximum)
                const uint32_t cv = kV != 0 ? (uint32_t)kV << 1 : 0u;
                return H == bD ? cd : (H == bV ? cv : 1u);
            };
            code4 = code_of(H0, bD0, kD0, bV0, kV0) | (code_of(H1, bD1, kD1, bV1, kV1) << 8) |
                    (code_of(H2, bD2, kD2, bV2, kV2) << 16) | (code_of(H3, bD3, kD3, bV3, kV3) << 24);
            if (tg == 0) code4 &= 0xffffff00u; // the band's first cell
            if (undecided) code4 = 0u;
        }

        // flow control: slot r mod R still holds row r - R, which the right neighbour may need until it has finished row
        // r - kSkLead - 1
        wait_right(r - kSkLead - 1);
        if ((uint32_t)tg < (uint32_t)band_width)
        {
            const int32_t rel = tg + 1;
            // explicit LDS stores: the progress word below is ordered behind them by the LDS queue, not by a wait
            const int32_t slot_r = slot_now();
            lds_st_quad<ScoreT>(ring + slot_r * stride + rel + kRelShift, (ScoreT)H0, (ScoreT)H1, (ScoreT)H2, (ScoreT)H3);
            global_st_quad<ScoreT>(scores + (int64_t)r * stride + rel + kRelShift, (ScoreT)H0, (ScoreT)H1, (ScoreT)H2, (ScoreT)H3);
            if (codes) *reinterpret_cast<__attribute__((address_space(1))) uint32_t*>(codes + (int64_t)r * stride + rel + kRelShift) = code4;
        }
        if (first_block && lane == 0)
        {
            *(__attribute__((address_space(3))) ScoreT*)(ring + slot_now() * stride + kRelShift) = (ScoreT)rel0_val;
            scores[(int64_t)r * stride + kRelShift]                                          = (ScoreT)rel0_val;
        }
        publish(r, __builtin_amdgcn_readlane(H3, kWave - 1));
        if (sksel == 8 || (sksel == 11 && first_block)) skacc += clock64() - t_rb;
    }
    }
    if (sksel == 4) skacc += clock64() - t_pass;
    if (sksel && lane == 0) __hip_atomic_fetch_add(&shared->prof, (unsigned long long)skacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    block_barrier(); // the score and code matrices are complete in HBM (wave 0's traceback reads them)
    if (sksel && prof_out && lane == 0) *prof_out += shared->prof;
}

// ------------------------------------------------------------------------------------------------
// Banded NW (score-matrix modes): forward pass wave-wide, then sink selection (wave reduction with the
// reference's first-maximum tie rule) and the lane-0 traceback.
// ------------------------------------------------------------------------------------------------
// PV: which packed int16 passes the instantiation carries -- 0: the 256-column band only; 1: + the 128-column band
// (poa_forward_moves.h, band in lanes 0..31); 2: the two-pass 384- and 512-column bands (poa_forward_moves_wide.h)
template <typename ScoreT, typename IdT, typename RowT, bool ADAPTIVE, bool LDS_READ, int PV = 0>
__device__ __forceinline__ int32_t nw_banded(const GraphView<IdT>& g, RowT* rowinfo, int32_t graph_count, const uint8_t* read,
                             const uint8_t* lds_read, int32_t read_length, ScoreT* scores, ScoreT* ring_base, int32_t ring_bytes,
                             float max_buffer_size, int32_t* alignment_graph, int32_t* alignment_read,
                             int32_t band_width, int32_t gap_score, int32_t mismatch_score, int32_t match_score,
                             int32_t rerun, uint64_t& cells, PhaseClock& pc, int32_t dbg = 0, uint8_t* codes = nullptr,
                             uint8_t* code_tile = nullptr, uint8_t* read_window = nullptr, int32_t* bs_ring = nullptr,
                             MwArgs<ScoreT>* mw_args = nullptr, MwShared* mw_shared = nullptr)
{
    const int lane              = threadIdx.x & (kWave - 1);
    const int32_t min_score     = Limits<ScoreT>::min / 2;
    const float gradient        = __fdiv_rn((float)(read_length + 1), (float)(graph_count + 1));
    const int32_t max_column    = read_length + 1;

    if (ADAPTIVE) // cudapoa_nw_banded.cuh:213-234
    {
        if ((double)gradient > 1.1)
            band_width = max(band_width, ((int32_t)((double)max_column * 0.08 * (double)gradient) + 127) & ~127);
        if ((double)gradient < 0.8)
            band_width = max(band_width, ((int32_t)((double)max_column * 0.1 / (double)gradient) + 127) & ~127);
        band_width = min(band_width, kMaxAdaptiveBand);
        if (band_width == kMaxAdaptiveBand && rerun != 0) return rerun;
    }
    int32_t band_shift = band_width / 2;
    if (ADAPTIVE) // :239-265
    {
        if (rerun == kShiftLeft && band_width <= kMaxAdaptiveBand / 2)
        {
            band_width *= 2;
            band_shift = (int32_t)((double)band_shift * 2.5);
        }
        if (rerun == kShiftRight && band_width <= kMaxAdaptiveBand / 2)
        {
            band_width *= 2;
            band_shift = (int32_t)((double)band_shift * 1.5);
        }
        float required = __fmul_rn((float)graph_count, (float)(band_width + kRightPad));
        if (required > max_buffer_size) return kNwAdaptiveStorageFailed;
    }
    cells += (uint64_t)graph_count * (uint64_t)band_width;

    BandedCtx<ScoreT> b;
    b.scores     = scores;
    b.stride     = band_width + kRightPad;
    b.band_width = band_width;
    b.band_shift = band_shift;
    b.max_column = max_column;
    b.gradient   = gradient;
    b.min_score  = min_score;
    b.ring       = ring_base;
    b.ring_rows  = ring_bytes / (int32_t)(b.stride * sizeof(ScoreT));
    if (b.ring_rows < 2) b.ring_rows = 0;
    if (bs_ring != nullptr) b.ring_rows = min(b.ring_rows, 64); // entries of bs_ring
    const int32_t stride = b.stride;
    const int32_t npass  = (band_width + 255) / 256;
    const bool reg_path  = (npass == 1); // previous row carried in registers

    // row 0: H[0][rel] = rel * gap for rel < stride (:269-272); only rel <= band_width is ever read.
    for (int32_t rel = lane; rel <= band_width; rel += kWave)
    {
        ScoreT v = (ScoreT)(rel * gap_score);
        scores[rel + kRelShift] = v;
        if (b.ring_rows) b.ring[rel + kRelShift] = v;
    }
    // registers: previous row's cells (columns prev_bs+1+4*lane .. +4), valid when prev_row == r-1
    int32_t P0 = 0, P1 = 0, P2 = 0, P3 = 0;
    int32_t prev_bs   = 0;         // band start of the row held in P*
    int32_t prev_rel0 = 0;         // its relative-0 slot value
    if (reg_path)
    {
        int32_t c = 4 * lane; // row 0, band start 0: cells rel 1+4l..4+4l
        P0 = (ScoreT)((c + 1) * gap_score);
        P1 = (ScoreT)((c + 2) * gap_score);
        P2 = (ScoreT)((c + 3) * gap_score);
        P3 = (ScoreT)((c + 4) * gap_score);
        prev_rel0 = 0; // row 0, rel 0 = 0 * gap
    }
    // band start of every row, once per read, into the row table: the row loop then does no fp math at all
    for (int32_t r = 1 + lane; r <= graph_count; r += kWave)
        rowinfo[r].set_bs(band_start_for_row(r, gradient, band_width, band_shift, max_column));
    bool hbm_dirty = true; // stores since the last workgroup sync (needed before reading the HBM matrix)
    wave_sync();
    hbm_dirty = false;
    // rows still in the LDS ring have their band start there too (bs_ring): with the row table in HBM a load of it
    // in the middle of a row would wait for the previous row's score stores (loads and stores return in order)
    auto bs_of = [&](int32_t row, int32_t r) -> int32_t {
        if (row == 0) return 0;
        if (bs_ring && b.ring_rows && r - row < b.ring_rows) return bs_ring[row % b.ring_rows];
        return uniform_row(rowinfo[row]).bs();
    };
    // sliding LDS window over the read (graphs whose tables live in HBM): columns [staged_end - 4096, staged_end)
    constexpr int32_t kWin = 4096, kWinStep = 1024;
    int32_t staged_end = 0;
    auto stage_read = [&](int32_t need_end) { // wave-uniform
        while (staged_end < need_end)
        {
            for (int32_t i = lane * 4; i < kWinStep; i += kWave * 4)
            {
                const int32_t col = staged_end + i;
                // positions past the read are never consumed (the input buffer keeps zero slack behind every read)
                const uint32_t v = col < read_length + 8 ? *reinterpret_cast<const uint32_t*>(read + col) : 0u;
                *reinterpret_cast<uint32_t*>(read_window + (col & (kWin - 1))) = v;
            }
            staged_end += kWinStep;
        }
        wave_sync();
    };

    constexpr bool kFastOk = std::is_same<RowT, RowInfo<true>>::value && LDS_READ;
    bool fast_done = false, codes_valid = false, moves_valid = false;
    bool packed_selective = false; // the packed pass of band 256 / 128 ran and kept the score rows nobody reads out of HBM
    bool wrapped = false; // a score did not fit the matrix type (narrow_chk)
    if constexpr (kFastOk && std::is_same<ScoreT, int16_t>::value)
    {
        // packed 16-bit pass for the 256-column band (preconditions: poa_forward_packed.h)
        const bool width_ok  = PV == 2 ? (band_width == 384 || band_width == 512) : (band_width == 256 || (PV == 1 && band_width == 128));
        const int32_t u_span = max(band_width, 256) * abs(gap_score); // u-space offset of the last band cell
        const bool packed_ok = width_ok && max_column >= band_width && ring_bytes >= kPkSlots * kPkSlotBytes &&
                               abs(gap_score) <= 30 && abs(match_score) <= 100 && abs(mismatch_score) <= 100 &&
                               codes != nullptr && !(dbg & 256) && ring_bytes >= kMtBytes &&
                               // no packed operation can leave int16: the largest score (all matches) plus the u-space offset of
                               // the last band cell, and the smallest (every step at the worst penalty; min_score-derived cells)
                               max(match_score, 0) * min(read_length, graph_count) + u_span + abs(match_score) <= 32767 &&
                               (graph_count + read_length) * min(min(gap_score, mismatch_score), 0) >= -32768 + 256 &&
                               min_score + 4 * min(min(gap_score, mismatch_score), 0) - u_span >= -32768;
        static_assert(kWdSlots * kWdSlotBytes <= kPkSlots * kPkSlotBytes, "the two-pass ring lives in the 256-column pass's ring region");
        if (packed_ok)
        {
            // move bytes, row kinds, descriptors in registers (poa_forward_moves.h, poa_forward_moves_wide.h)
            if constexpr (PV == 2)
            {
                if (band_width == 384)
                    banded_forward_moves_wide<IdT, 384>(g, rowinfo, graph_count, lds_read, scores, codes, reinterpret_cast<uint8_t*>(ring_base),
                                                        reinterpret_cast<const uint64_t*>(code_tile), max_column, gap_score, mismatch_score, match_score, dbg);
                else
                    banded_forward_moves_wide<IdT, 512>(g, rowinfo, graph_count, lds_read, scores, codes, reinterpret_cast<uint8_t*>(ring_base),
                                                        reinterpret_cast<const uint64_t*>(code_tile), max_column, gap_score, mismatch_score, match_score, dbg);
            }
            else if (band_width == 256)
            {
                banded_forward_moves<IdT, 256>(g, rowinfo, graph_count, lds_read, scores, codes, reinterpret_cast<uint8_t*>(ring_base),
                                               reinterpret_cast<const uint64_t*>(code_tile), max_column, gap_score, mismatch_score, match_score,
                                               dbg, pc.acc ? &pc.acc[kPhOther] : nullptr, (dbg & (1 << 25)) != 0);
                packed_selective = !(dbg & (1 << 25));
            }
            else if constexpr (PV == 1)
            {
                banded_forward_moves<IdT, 128>(g, rowinfo, graph_count, lds_read, scores, codes, reinterpret_cast<uint8_t*>(ring_base),
                                               reinterpret_cast<const uint64_t*>(code_tile), max_column, gap_score, mismatch_score, match_score,
                                               dbg, pc.acc ? &pc.acc[kPhOther] : nullptr, (dbg & (1 << 25)) != 0);
                packed_selective = !(dbg & (1 << 25));
            }
            fast_done   = true;
            moves_valid = true;
        }
    }
    if constexpr (kFastOk)
    {
        if (!fast_done && npass == 1 && b.ring_rows >= 2)
        {
            banded_forward_1pass<ScoreT, IdT>(g, rowinfo, graph_count, lds_read, scores, b.ring, b.ring_rows, band_width,
                                              max_column, gap_score, mismatch_score, match_score, dbg,
                                              pc.acc ? &pc.acc[kPhOther] : nullptr, wrapped);
            fast_done = true;
        }
    }
    // wide bands in a multi-wave block: a pipeline of wavefronts over absolute 256-column blocks (generic_forward_skew)
    if constexpr (!std::is_same<RowT, RowInfo<true>>::value)
    {
        // rows the ring may hold there: the band start must move less than one block between a row and the row that
        // reuses its slot (see the function's header)
        const int32_t sk_rows = min(b.ring_rows, (int32_t)(200.0f / (gradient + 1.0f)));
        if (!fast_done && mw_args != nullptr && npass >= 2 && sk_rows >= kSkLead + 3 && !(dbg & (1 << 18)))
        {
            MwArgs<ScoreT> A;
            A.op = 1;
            A.graph_count = graph_count; A.read_length = read_length; A.band_width = band_width; A.band_shift = band_shift;
            A.max_column = max_column; A.gradient = gradient;
            A.gap_score = gap_score; A.mismatch_score = mismatch_score; A.match_score = match_score;
            A.ring_rows = sk_rows; A.read = read; A.scores = scores; A.codes = codes; A.dbg = dbg;
            if (lane == 0) *mw_args = A;
            (&mw_shared->hand[0][0][0])[lane]      = 0; // kSkWaves x 8 entries of two words: row 0 is "finished" everywhere
            (&mw_shared->hand[0][0][0])[lane + 64] = 0;
            if (lane < kSkWaves) mw_shared->skipped[lane] = 0;
            if (lane == 0) { mw_shared->prof = 0; mw_shared->fail = 0; }
            block_barrier(); // the helper wavefronts wait here for their arguments
            generic_forward_skew<ScoreT, IdT, RowT>(A, g, rowinfo, b.ring, mw_shared, 0, lane, pc.acc ? &pc.acc[kPhOther] : nullptr);
            if (wave_first(mw_shared->fail) != 0) return kNwPipelineFailed; // a bounded hand-over wait ran out
            fast_done   = true;
            codes_valid = codes != nullptr;
        }
    }
    // Row table through LDS when it lives in HBM: 64 rows at a time, so the row loop itself issues no global load
    // (one would wait for the previous row's score stores: loads and stores return in order).
    RowT* ri_stage       = nullptr;
    int32_t ri_stage_end = 0;
    if constexpr (!std::is_same<RowT, RowInfo<true>>::value)
        if (bs_ring != nullptr) ri_stage = reinterpret_cast<RowT*>(bs_ring + 64);
    auto fetch_ri = [&](int32_t row) -> RowT {
        if (ri_stage == nullptr) return uniform_row(rowinfo[row]);
        if (row >= ri_stage_end) // wave-uniform
        {
            wave_sync();
            if (row + lane <= graph_count) ri_stage[(row + lane) & 63] = rowinfo[row + lane];
            ri_stage_end = row + 64;
            wave_sync();
        }
        return uniform_row(ri_stage[row & 63]);
    };
    // One row. FAST: every predecessor row is still in the LDS ring (and there are at most three), so this
    // instantiation contains LDS traffic and score stores only -- no load that would drain the store queue.
    auto row_body = [&](auto fast_tag, const int32_t r, const int32_t slot_r, const RowT& ri) {
        constexpr bool FAST      = decltype(fast_tag)::value;
        const int32_t pred_count = ri.cnt();
        const int32_t bs         = ri.bs();
        const int32_t node_id    = (!FAST && pred_count > 3) ? (int32_t)g.sorted_poa[r - 1] : 0;
        auto pred_row = [&](int32_t p) -> int32_t {
            if (pred_count == 0) return 0;
            if (FAST || p < 3) return ri.pred(p);
            return (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1;
        };
        // ring slot of an earlier row that is still in the ring
        auto slot_of = [&](int32_t row) -> int32_t {
            if (!FAST) return row % b.ring_rows;
            const int32_t sl = slot_r - (r - row);
            return sl < 0 ? sl + b.ring_rows : sl;
        };
        // relative-0 slot of an arbitrary earlier row (get_score(row, -1): reads rel 0 unconditionally)
        auto rel0_of = [&](int32_t row) -> int32_t {
            if (reg_path && row == r - 1) return prev_rel0;
            if (FAST || (b.ring_rows && r - row < b.ring_rows)) return lds_ld(b.ring + slot_of(row) * stride + kRelShift);
            if (hbm_dirty) { wave_sync(); hbm_dirty = false; }
            return scores[(int64_t)row * stride + kRelShift];
        };

        // ---- left boundary / carry-in (:293-326), wave-uniform ----
        int32_t fe       = 0;          // first_element_prev_score
        int32_t rel0_val = min_score;  // value kept in this row's relative-0 slot
        const int32_t pred_idx0 = pred_row(0);
        if (pred_count == 0)
        {
            if (bs == 0) rel0_val = (ScoreT)gap_score; // carry-in stays 0: reference quirk
        }
        else
        {
            if (bs > kCellsPerLane && pred_count == 1)
                fe = min_score + gap_score;
            else
            {
                int32_t penalty = max(min_score, rel0_of(pred_idx0));
                for (int32_t p = 0; p < pred_count; p++) penalty = max(penalty, rel0_of(pred_row(p)));
                fe = penalty + gap_score;
            }
            if (bs == 0) rel0_val = (ScoreT)fe;
        }

        if (read_window != nullptr && bs + npass * 256 + 4 > staged_end) stage_read(bs + npass * 256 + 4);
        int32_t carry = fe;
        int32_t N0 = 0, N1 = 0, N2 = 0, N3 = 0; // this row's cells of the (single) register pass
        for (int32_t pass = 0; pass < npass; pass++)
        {
            const int32_t c      = bs + pass * 256 + 4 * lane; // chunk anchor column (cells c+1..c+4)
            const bool active    = (pass * 256 + 4 * lane) < band_width;
            // read characters c .. c+3 (positions past the read are never consumed; buffer has zero slack).
            // They come from the LDS copy when the read was staged: a global load here would make every row
            // wait for the previous row's score store (loads and stores share the in-order vmcnt counter).
            const uint32_t rd4   = LDS_READ ? *reinterpret_cast<const uint32_t*>(lds_read + c)
                                   : read_window != nullptr ? *reinterpret_cast<const uint32_t*>(read_window + (c & (kWin - 1)))
                                                            : *reinterpret_cast<const uint32_t*>(read + c);
            const int32_t cp0    = ((rd4 & 0xff) == (uint32_t)ri.base()) ? match_score : mismatch_score;
            const int32_t cp1    = (((rd4 >> 8) & 0xff) == (uint32_t)ri.base()) ? match_score : mismatch_score;
            const int32_t cp2    = (((rd4 >> 16) & 0xff) == (uint32_t)ri.base()) ? match_score : mismatch_score;
            const int32_t cp3    = ((rd4 >> 24) == (uint32_t)ri.base()) ? match_score : mismatch_score;
            int32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            const int32_t np = max(pred_count, 1);
            for (int32_t p = 0; p < np; p++)
            {
                const int32_t prow = (p == 0) ? pred_idx0 : pred_row(p);
                const int32_t pbs  = (reg_path && prow == r - 1) ? prev_bs
                                     : FAST ? (prow == 0 ? 0 : bs_ring[slot_of(prow)]) : bs_of(prow, r);
                const int32_t pend = min(pbs + band_width - kCellsPerLane, max_column);
                const bool valid   = !(c > pend || c < pbs);
                int32_t S0, S1, S2, S3, S4; // predecessor row columns c .. c+4
                if (reg_path && prow == r - 1)
                {
                    const int32_t q = (bs - prev_bs) >> 2; // lane shift between the two bands
                    if (q == 0)
                    {
                        S0 = wave_shr1(P3, prev_rel0);
                        S1 = P0; S2 = P1; S3 = P2; S4 = P3;
                    }
                    else if (q == 1)
                    {
                        S0 = P3;
                        S1 = wave_shl1(P0, 0); S2 = wave_shl1(P1, 0); S3 = wave_shl1(P2, 0); S4 = wave_shl1(P3, 0);
                    }
                    else
                    {
                        int src = lane + q;
                        S0 = __shfl(P3, src - 1);
                        S1 = __shfl(P0, src); S2 = __shfl(P1, src); S3 = __shfl(P2, src); S4 = __shfl(P3, src);
                    }
                }
                else
                {
                    const bool in_ring = FAST || (b.ring_rows && r - prow < b.ring_rows);
                    if (!in_ring && hbm_dirty) { wave_sync(); hbm_dirty = false; }
                    S0 = S1 = S2 = S3 = S4 = 0;
                    if (valid)
                    {
                        const int32_t rel = c - pbs; // multiple of 4
                        if (in_ring) // LDS ring: ds_read ops only
                        {
                            const ScoreT* rowp = b.ring + slot_of(prow) * stride;
                            S0 = lds_ld(rowp + rel + kRelShift);
                            Quad<ScoreT> qd = lds_ld_quad(rowp + rel + kRelShift + 1);
                            S1 = qd.v[0]; S2 = qd.v[1]; S3 = qd.v[2]; S4 = qd.v[3];
                        }
                        else // HBM score matrix
                        {
                            const ScoreT* rowp = scores + (int64_t)prow * stride;
                            S0 = rowp[rel + kRelShift];
                            Quad<ScoreT> qd = *reinterpret_cast<const Quad<ScoreT>*>(rowp + rel + kRelShift + 1);
                            S1 = qd.v[0]; S2 = qd.v[1]; S3 = qd.v[2]; S4 = qd.v[3];
                        }
                    }
                }
                int32_t t0, t1, t2, t3;
                if (valid)
                {
                    t0 = narrow_chk<ScoreT>(max(S0 + cp0, S1 + gap_score), wrapped);
                    t1 = narrow_chk<ScoreT>(max(S1 + cp1, S2 + gap_score), wrapped);
                    t2 = narrow_chk<ScoreT>(max(S2 + cp2, S3 + gap_score), wrapped);
                    t3 = narrow_chk<ScoreT>(max(S3 + cp3, S4 + gap_score), wrapped);
                }
                else { t0 = t1 = t2 = t3 = min_score; }
                if (p == 0) { s0 = t0; s1 = t1; s2 = t2; s3 = t3; }
                else { s0 = max(s0, t0); s1 = max(s1, t1); s2 = max(s2, t2); s3 = max(s3, t3); }
            }
            // ---- horizontal max-plus scan: H[t] = max(v[t], H[t-1] + gap), H[-1] = carry ----
            // with u[t] = v[t] - t*gap this is a prefix maximum; t = 4*lane + k inside the pass.
            const int32_t tb = 4 * lane;
            int32_t u0 = s0 - (tb + 0) * gap_score;
            int32_t u1 = s1 - (tb + 1) * gap_score;
            int32_t u2 = s2 - (tb + 2) * gap_score;
            int32_t u3 = s3 - (tb + 3) * gap_score;
            if (!active) u0 = u1 = u2 = u3 = INT32_MIN;
            const int32_t m0 = u0, m1 = max(m0, u1), m2 = max(m1, u2), m3 = max(m2, u3);
            const int32_t incl  = wave_inclusive_max(m3);
            const int32_t cu    = carry + gap_score; // carry as element t = -1: carry - (-1)*gap
            const int32_t excl  = max(wave_shr1(incl, INT32_MIN), cu);
            N0 = narrow_chk<ScoreT>(max(m0, excl) + (tb + 0) * gap_score, wrapped);
            N1 = narrow_chk<ScoreT>(max(m1, excl) + (tb + 1) * gap_score, wrapped);
            N2 = narrow_chk<ScoreT>(max(m2, excl) + (tb + 2) * gap_score, wrapped);
            N3 = narrow_chk<ScoreT>(max(m3, excl) + (tb + 3) * gap_score, wrapped);
            // carry into the next pass = last cell of the last lane
            carry = wave_bcast(N3, kWave - 1);
            if (active)
            {
                Quad<ScoreT> out;
                out.v[0] = (ScoreT)N0; out.v[1] = (ScoreT)N1; out.v[2] = (ScoreT)N2; out.v[3] = (ScoreT)N3;
                const int32_t rel = pass * 256 + 4 * lane + 1;
                if (!(dbg & 1)) *reinterpret_cast<Quad<ScoreT>*>(scores + (int64_t)r * stride + rel + kRelShift) = out;
                if (b.ring_rows && !(dbg & 2))
                    *reinterpret_cast<Quad<ScoreT>*>(b.ring + slot_r * stride + rel + kRelShift) = out;
            }
        }
        if (lane == 0)
        {
            if (!(dbg & 1)) scores[(int64_t)r * stride + kRelShift] = (ScoreT)rel0_val;
            if (b.ring_rows && !(dbg & 2)) b.ring[slot_r * stride + kRelShift] = (ScoreT)rel0_val;
            if (bs_ring && b.ring_rows) bs_ring[slot_r] = bs;
        }
        hbm_dirty = true;
        if (reg_path)
        {
            P0 = N0; P1 = N1; P2 = N2; P3 = N3;
            prev_bs   = bs;
            prev_rel0 = rel0_val;
        }
    };
    if (!fast_done)
    {
        RowT ri_next   = fetch_ri(1);
        int32_t slot_r = b.ring_rows ? 1 % b.ring_rows : 0; // r % ring_rows, kept incrementally
        for (int32_t r = 1; r <= graph_count; r++)
        {
            const RowT ri = ri_next;
            if (r < graph_count) ri_next = fetch_ri(r + 1); // prefetch: hides the LDS latency of the table
            bool all_in_ring = ri_stage != nullptr && b.ring_rows >= 2 && ri.cnt() <= 3;
            if (all_in_ring)
            {
                const int32_t np = max(ri.cnt(), 1);
                for (int32_t p = 0; p < np; p++) all_in_ring = all_in_ring && (r - (ri.cnt() ? ri.pred(p) : 0) < b.ring_rows);
            }
            if ((dbg & (1 << 20)) && pc.acc) pc.acc[kPhOther] += all_in_ring ? 1 : (1 << 20); // profiling: rows per path
            if (wave_first((int32_t)all_in_ring))
                row_body(std::true_type{}, r, slot_r, ri);
            else
                row_body(std::false_type{}, r, slot_r, ri);
            slot_r = (b.ring_rows && slot_r + 1 == b.ring_rows) ? 0 : slot_r + 1;
        }
    }
    wave_sync(); // score matrix complete and visible to lane 0's traceback
    pc.tick(kPhForward);
    if (__ballot(wrapped) != 0) return kNwScoreWrapped; // enforced precondition, see narrow_chk

    // ---- sink selection (:410-426): first row with the strictly greatest H(row, L) among sink rows ----
    const uint64_t t_sink = (pc.acc && ((dbg >> 22) & 7) == 1) ? clock64() : 0;
    int32_t best = min_score, best_i = 0;
    if constexpr (std::is_same<RowT, RowInfo<false>>::value)
    {
        // row table in HBM: eight rows per lane in flight (one row per round trip was 470 dependent round trips for a 30 k
        // row graph, per read); per lane the rows are visited in ascending order, as below
        constexpr int kU = 8;
        for (int32_t base = 1 + lane; base <= graph_count; base += kU * kWave)
        {
            bool sk[kU];
#pragma unroll
            for (int u = 0; u < kU; u++) sk[u] = rowinfo[min(base + u * kWave, graph_count)].sink();
#pragma unroll
            for (int u = 0; u < kU; u++)
            {
                const int32_t idx = base + u * kWave;
                if (idx <= graph_count && sk[u])
                {
                    int32_t s = get_score(b, idx, read_length);
                    if (best < s) { best = s; best_i = idx; }
                }
            }
        }
    }
    else
    for (int32_t idx = 1 + lane; idx <= graph_count; idx += kWave)
    {
        if (rowinfo[idx].sink())
        {
            int32_t s = get_score(b, idx, read_length);
            if (best < s) { best = s; best_i = idx; }
        }
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        int32_t ob = __shfl_xor(best, off), oi = __shfl_xor(best_i, off);
        // strictly greater wins; on ties the lower row index wins; rows that never beat min_score stay 0
        if (ob > best || (ob == best && oi != 0 && (best_i == 0 || oi < best_i))) { best = ob; best_i = oi; }
    }

    if (pc.acc && ((dbg >> 22) & 7) == 1) pc.acc[kPhOther] += clock64() - t_sink;
    int32_t aligned_nodes = 0;
    const bool tile_fits = (int32_t)(64 * 64 * sizeof(ScoreT) + 64 * sizeof(int32_t)) <= ring_bytes; // also covers the 60 x 68 + 64-word layout
    constexpr bool kLanesOk = std::is_same<RowT, RowInfo<true>>::value && LDS_READ;
    bool tb_done = false;
    if constexpr (kLanesOk && std::is_same<ScoreT, int16_t>::value)
    {
        if (moves_valid)
        {
            aligned_nodes = traceback_moves<int16_t, IdT, RowInfo<true>, ADAPTIVE>(b, g, rowinfo, graph_count, lds_read, read_length, wave_first(best_i),
                                                           alignment_graph, alignment_read, gap_score, mismatch_score, match_score,
                                                           rerun, reinterpret_cast<uint8_t*>(ring_base), codes, nullptr,
                                                           packed_selective ? ((dbg & (1 << 23)) ? 2 : 0) : 1);
            if (aligned_nodes == kNwNeedScoreRows)
            {
                // the walk met a cell it has to step by recomputation in a row whose scores stayed out of HBM (the band's first
                // cell, a chunk outside a predecessor's band): the same pass again with every row stored, then the same walk
                // (GWHIP_DEBUG, debug instantiation: bit 24 counts these reruns into the "other" phase accumulator, bit 23 makes
                // every recomputed step below row 0 take this path, bit 25 stores every row in the first place)
                if (pc.acc && (dbg & (1 << 24))) pc.acc[kPhOther] += 1;
                if (band_width == 256)
                    banded_forward_moves<IdT, 256>(g, rowinfo, graph_count, lds_read, scores, codes, reinterpret_cast<uint8_t*>(ring_base),
                                                   reinterpret_cast<const uint64_t*>(code_tile), max_column, gap_score, mismatch_score, match_score,
                                                   dbg, nullptr, true);
                else if constexpr (PV == 1)
                    banded_forward_moves<IdT, 128>(g, rowinfo, graph_count, lds_read, scores, codes, reinterpret_cast<uint8_t*>(ring_base),
                                                   reinterpret_cast<const uint64_t*>(code_tile), max_column, gap_score, mismatch_score, match_score,
                                                   dbg, nullptr, true);
                wave_sync();
                aligned_nodes = traceback_moves<int16_t, IdT, RowInfo<true>, ADAPTIVE>(b, g, rowinfo, graph_count, lds_read, read_length, wave_first(best_i),
                                                               alignment_graph, alignment_read, gap_score, mismatch_score, match_score,
                                                               rerun, reinterpret_cast<uint8_t*>(ring_base), codes, nullptr, 1);
            }
            tb_done = true;
        }
    }
    if constexpr (kLanesOk)
    {
        if (!tb_done && tile_fits && b.stride >= 64 && !(dbg & 32))
        {
            aligned_nodes = traceback_banded_lanes<ScoreT, IdT, ADAPTIVE>(b, g, rowinfo, graph_count, lds_read, read_length,
                                                                          wave_first(best_i), alignment_graph, alignment_read, gap_score,
                                                                          mismatch_score, match_score, rerun, ring_base, dbg,
                                                                          pc.acc ? &pc.acc[kPhOther] : nullptr);
            tb_done = true;
        }
    }
    constexpr bool kStagedOk = std::is_same<RowT, RowInfo<false>>::value && !LDS_READ;
    const bool staged_fits   = (int32_t)(64 * 64 * sizeof(ScoreT) + 64 * 4 + 64 * sizeof(RowT) + 256 + 64 * 8) <= ring_bytes;
    if constexpr (kStagedOk)
    {
        if (codes_valid && codes != nullptr && ring_bytes >= kMtBytesWide && b.stride >= 80 && !(dbg & 64))
        {
            // wide bands, graphs beyond the LDS tables: the pipelined forward pass left move bytes
            aligned_nodes = traceback_moves<ScoreT, IdT, RowT, ADAPTIVE>(b, g, rowinfo, graph_count, read, read_length, wave_first(best_i),
                                                                         alignment_graph, alignment_read, gap_score, mismatch_score,
                                                                         match_score, rerun, reinterpret_cast<uint8_t*>(ring_base), codes);
            tb_done = true;
        }
    }
    if (tb_done) {}
    else if (kStagedOk && staged_fits && b.stride >= 64 && !(dbg & 128))
    {
        // graphs beyond the LDS tables: score tile, row records, read window and output all staged in the dead ring
        aligned_nodes = traceback_banded_tiled_staged<ScoreT, IdT, RowT, ADAPTIVE>(b, g, rowinfo, graph_count, read, read_length,
                                                                                  wave_first(best_i), alignment_graph, alignment_read,
                                                                                  gap_score, mismatch_score, match_score, rerun,
                                                                                  reinterpret_cast<uint8_t*>(ring_base));
    }
    else if (lane == 0)
        aligned_nodes = traceback_banded<ScoreT, IdT, RowT, ADAPTIVE>(b, g, rowinfo, graph_count, read, read_length, best_i,
                                                                alignment_graph, alignment_read, gap_score,
                                                                mismatch_score, match_score, rerun);
    aligned_nodes = wave_first(aligned_nodes);
    pc.tick(kPhTraceback);
    return aligned_nodes;
}

} // namespace gwhip
