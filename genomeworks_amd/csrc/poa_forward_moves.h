// poa_forward_moves.h -- banded NW forward pass for the 256-column band with int16 scores, third generation
// (round 3). What it computes is cudapoa_nw_banded.cuh:269-408 (restated in oracle/poa_nw.inc); the arithmetic of a
// row is that of poa_forward_packed.h (two score cells per register, v_pk_* arithmetic, prefix maximum of
// u[t] = v[t] - t * gap, LDS ring of 8 rows at absolute column slots with sentinel cells behind the band end).
// What changed is everything around the arithmetic, because a lone wavefront pays ~4.5 cycles per issued
// instruction of any kind and ~25 per taken branch:
//
//   * MOVE BYTES instead of move codes. The byte stored per cell no longer names a predecessor SLOT (which the
//     traceback had to resolve through the row table) but the move itself: rows up (0..7) in bits 1..3 and columns
//     left (0/1) in bit 0 -- horizontal = 1, diagonal through a predecessor d rows up = 2 d + 1, vertical = 2 d,
//     0 = undecided here (the traceback recomputes that step from the score matrix). The forward pass knows d as a
//     per-row scalar, so the bytes cost the same instructions as the codes did, and the traceback needs neither
//     the row table nor any decode (poa_traceback_moves.h).
//   * ROW KINDS, decided once per read by all lanes (classify_kinds): 0 previous row, band not moved (registers +
//     one DPP shift); 1 previous row, band moved by one quad (registers + DPP lane shift: no LDS round trip; the
//     read characters of the next quad are prefetched one band move ahead); 2 one predecessor 1..7 rows up (LDS
//     ring); 3 two to six predecessors (LDS ring); 4 everything else (general 32-bit routine, HBM matrix).
//   * ROW DESCRIPTORS in registers: 64 rows at a time every lane decodes its row's table word into a ready-made
//     32-bit descriptor (kind, band start, ring slots and distances of the predecessors) and the base replicated
//     into four bytes; the row loop fetches them with v_readlane (one issue slot, no LDS latency, no scalar decode
//     for the common kinds).
//   * TWO PHASES: rows whose band starts at column 0 (the first ~band/2 rows of a read: their left boundary is a
//     real cell) and the rest (left boundary = min_score by construction) run in separate instantiations, so the
//     bulk of the rows carries no boundary code at all.
//   * Stores go through per-lane running pointers (one 64-bit add per row and stream).
//
// Preconditions are those of poa_forward_packed.h (checked by nw_banded).
#pragma once

namespace gwhip
{

constexpr int kKindShift = 60; // row kind in bits 60..62 of the packed row-table word
constexpr uint32_t kMoveHoriz = 1;

__device__ __forceinline__ uint32_t pk_mad_u16_vsv(uint32_t a, uint32_t s, uint32_t c)
{
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(s), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t pk_mad_u16_vvs(uint32_t a, uint32_t b, uint32_t s)
{
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(s));
    return d;
}
// streaming 8-byte store (round 5): a score row is read again by general rows, the sink scan and recomputed traceback steps only,
// and a launch writes 34 GB of score and move rows -- kept out of the L2's way the kernel runs 1-6 % faster in wall time
__device__ __forceinline__ void gstore_nt_u64(void* p, uint32_t lo, uint32_t hi)
{
    u32x2 v;
    v.x = lo; v.y = hi;
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
}
// lane 0 only: the 2 bytes just below the lane's own pointer (the left-boundary slot of a score row). The address is a
// per-lane VGPR pair: an inline-asm VMEM instruction must not take a scalar base the compiler may just have reloaded with
// v_readlane (the VALU-writes-SGPR -> VMEM hazard is not tracked across inline asm).
__device__ __forceinline__ void gstore_u16_lane0_below(const void* vptr, uint32_t v)
{
    asm volatile("s_mov_b64 exec, 1\n\tglobal_store_short %0, %1, off offset:-2\n\ts_mov_b64 exec, -1" ::"v"(vptr), "v"(v) : "memory");
}

// ------------------------------------------------------------------------------------------------
// Row kinds, all lanes in parallel. Needs the band starts in the table already. Returns the first row whose band
// starts past column 0 (graph_count + 1 if there is none); band starts never decrease from row to row.
// ------------------------------------------------------------------------------------------------
// MAXD: rows up a ring predecessor may be (ring rows - 1, or ring rows: poa_forward_moves_wide.h). TBRULES: the
// traceback-buffer modes (poa_forward_moves_tb.h) keep gap_score in the boundary slot of a row without predecessors whatever
// its band start, so a row that has such a row (band start past column 0) among its predecessors takes the general routine.
template <int MAXD = kPkMaxDist, bool TBRULES = false>
__device__ __forceinline__ int32_t classify_kinds(RowInfo<true>* rowinfo, int32_t graph_count, int lane, const uint64_t* xpred, int32_t dbg = 0)
{
    int32_t first_moved = graph_count + 1;
    for (int32_t r = 1 + lane; r <= graph_count; r += kWave)
    {
        RowInfo<true> ri  = rowinfo[r];
        const int32_t cnt = ri.cnt(), bs = ri.bs();
        uint64_t kind     = 4;
        // 4..6 predecessors: rows 3..5 from the side table build_rowinfo left (a row that lost its slot stays kind 4)
        const uint64_t xe  = (cnt > 3 && cnt <= 6 && xpred != nullptr) ? xpred[r & 255] : 0ull;
        const bool many_ok = cnt > 3 && cnt <= 6 && xpred_hit(xe, r, cnt) && !(dbg & (1 << 30));
        if (cnt >= 1 && (cnt <= 3 || many_ok))
        {
            bool ok    = true;
            int32_t d0 = 0, pbs0 = 0;
            for (int32_t k = 0; k < cnt; k++)
            {
                const int32_t p   = k < 3 ? ri.pred(k) : xpred_row(xe, k);
                const int32_t d   = r - p;
                const RowInfo<true> pri = rowinfo[p];
                const int32_t pbs = pri.bs(); // row 0 holds band start 0
                ok                = ok && d >= 1 && d <= MAXD && (bs - pbs) <= kPkGuardCols && (bs == 0 || pbs > 0);
                if constexpr (TBRULES) ok = ok && !(pri.cnt() == 0 && pbs > 0);
                if (k == 0) { d0 = d; pbs0 = pbs; }
            }
            if (ok) kind = cnt > 1 ? 3 : ((d0 == 1 && pbs0 == bs) ? 0 : ((d0 == 1 && bs - pbs0 == kCellsPerLane) ? 1 : 2));
        }
        // A/B selectors (GWHIP_DEBUG, debug instantiation): demote kinds so that the routines can be checked against each
        // other -- bit 10: registers -> ring (kinds 0, 1 -> 2), bit 15: kind 1 -> 2, bit 9: ring -> general (2, 3 -> 4),
        // bit 11: registers -> general (0, 1 -> 4), bit 30 (above): rows with 4..6 predecessors -> general
        if ((dbg & 1024) && kind <= 1) kind = 2;
        if ((dbg & 32768) && kind == 1) kind = 2;
        if ((dbg & 512) && (kind == 2 || kind == 3)) kind = 4;
        if ((dbg & 2048) && kind <= 1) kind = 4;
        ri.w       = (ri.w & ~(7ull << kKindShift)) | (kind << kKindShift);
        rowinfo[r] = ri;
        if (bs > 0) first_moved = min(first_moved, r);
    }
    for (int off = 32; off > 0; off >>= 1) first_moved = min(first_moved, __shfl_xor(first_moved, off));
    return __builtin_amdgcn_readfirstlane(first_moved);
}

// ------------------------------------------------------------------------------------------------
// Which rows' SCORES anybody will read back from HBM (round 5; bit 63 of the row word): the predecessors of general rows (they
// come from the HBM matrix), every row whose cells the walk may have to step by recomputation -- general rows and rows with more
// than three predecessors -- together with its predecessors, and the sink rows (sink selection). Every other row keeps its
// score row out of HBM: move bytes, ring and registers carry it (a launch of the metric batch used to write 22.6 GB of score
// rows of which well under 1 GB is ever read). Needs the row kinds in the table. all_rows: mark everything (A/B, reruns).
// ------------------------------------------------------------------------------------------------
constexpr uint64_t kRowScoresInHbm = 1ull << 63;
template <typename IdT>
__device__ __forceinline__ void mark_score_rows(const GraphView<IdT>& g, RowInfo<true>* rowinfo, int32_t graph_count, int lane, bool all_rows)
{
    auto mark = [&](int32_t row) {
        __hip_atomic_fetch_or(reinterpret_cast<uint32_t*>(&rowinfo[row].w) + 1, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (int32_t r = 1 + lane; r <= graph_count; r += kWave)
    {
        const uint64_t w    = rowinfo[r].w;
        const uint32_t kind = (uint32_t)(w >> kKindShift) & 7u;
        const int32_t cnt   = (int32_t)((w >> 8) & 0x3fu);
        const bool reads_or_recomputes = kind >= 4u || cnt > 3;
        if (reads_or_recomputes || ((w >> 14) & 1u) != 0 || all_rows) mark(r);
        if (reads_or_recomputes)
        {
            const int32_t node_id = cnt > 3 ? (int32_t)g.sorted_poa[r - 1] : 0;
            for (int32_t k = 0; k < cnt; k++)
                mark(k < 3 ? (int32_t)((w >> (24 + 12 * k)) & 0xfffu) : (int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + k]] + 1);
        }
    }
    wave_sync();
}
// the walk may step a cell of this row by recomputation: the row and its predecessors have their scores in HBM -- marked
// above, or a row whose band starts at column 0 (those store their scores anyway, and so do their predecessors, whose bands
// cannot start further right; every walk ends there, in the band's first column)
__device__ __forceinline__ bool row_recompute_safe(uint64_t w)
{
    return ((uint32_t)(w >> kKindShift) & 7u) >= 4u || ((w >> 8) & 0x3fu) > 3u || ((w >> 15) & 0x1ffu) == 0u;
}

// ------------------------------------------------------------------------------------------------
// The forward pass. `ring` is kPkSlots * kPkSlotBytes of LDS at LDS address 0 .. (the launcher's carve puts the ring
// first); `scores` the HBM score matrix and `moves` the HBM move-byte matrix (row stride 264 elements / bytes);
// lds_read the LDS copy of the read.
// ------------------------------------------------------------------------------------------------
// BW = 256: every lane owns four band cells. BW = 128: the same instruction stream with the band in lanes 0..31 -- lanes
// 32..63 compute cells right of the band that nobody reads (a prefix scan runs left to right, so they cannot reach into the
// band) and their stores are masked; a row then costs what a 256-column row costs, which is still 2-3 x less than the
// general pass.
template <typename IdT, int BW>
__device__ __forceinline__ void banded_forward_moves(const GraphView<IdT>& g, RowInfo<true>* rowinfo, int32_t graph_count,
                                                     const uint8_t* lds_read, int16_t* scores, uint8_t* moves, uint8_t* ring,
                                                     const uint64_t* xpred, int32_t max_column, int32_t gap_score,
                                                     int32_t mismatch_score, int32_t match_score, int32_t dbg,
                                                     uint64_t* prof_acc, bool store_all = true)
{
    // timing ablations (GWHIP_DEBUG, debug instantiation only; results are only meaningful on a relaunch over the
    // buffers of an unablated launch of the same batch): bit 26 no score-row stores, bit 27 no move-row stores
    // (tools/microbench_rows.hip only, results are garbage: bit 20 no ring write, 19 no guard write, 18 no cross-lane scan,
    // 17 no move bytes, 16 no rows at all)
    // (all of them only with bit 14 set: the same bits are the merge / topsort profiling selectors of the window kernel, 16-18 and
    // 25-27 -- tools/profile_subphases.sh's topsort selectors 2, 3 and 6 switched the row stores off until round 6)
    const bool abl       = (dbg & (1 << 14)) != 0;
    const bool st_scores = !(abl && (dbg & (1 << 26))), st_moves = !(abl && (dbg & (1 << 27)));
    const bool ab_ring = !(abl && (dbg & (1 << 20))), ab_guard = !(abl && (dbg & (1 << 19))), ab_scan = !(abl && (dbg & (1 << 18))),
               ab_moves = !(abl && (dbg & (1 << 17))), ab_rows = !(abl && (dbg & (1 << 16)));
    // profiling (GWHIP_DEBUG bits 28-30 = row kind + 1): cycles spent in rows of that kind, or with bit 12 their number;
    // arrives in the "other" phase accumulator
    const int32_t ksel = prof_acc ? ((dbg >> 28) & 7) - 1 : -1;
    const bool kcount  = (dbg & (1 << 12)) != 0;
    uint64_t kacc      = 0;
    static_assert(BW == 128 || BW == 256, "band widths of the packed pass");
    constexpr int32_t band_width = BW;
    constexpr int32_t stride     = band_width + kRightPad;
    constexpr int kBandLanes     = BW / kCellsPerLane; // lanes that own band cells
    const int lane               = threadIdx.x & (kWave - 1);
    const bool band_lane         = lane < kBandLanes;
    const int32_t lane4 = lane * 4, lane8 = lane * 8;
    const int32_t min_score = Limits<int16_t>::min / 2;

    const int32_t first_moved = classify_kinds(rowinfo, graph_count, lane, xpred, dbg);
    wave_sync();
    // rows past the first band move write their score row to HBM only when it will be read back (mark_score_rows); the rows
    // whose band starts at column 0 (a tenth of them) keep storing theirs together with the real left-boundary value
    mark_score_rows<IdT>(g, rowinfo, graph_count, lane, store_all);

    const uint32_t MIN2   = pin_vgpr(pk_dup(min_score));
    const uint32_t SENT2  = pin_vgpr(pk_dup(kPkSentinel));
    const uint32_t GAP2   = pin_vgpr(pk_dup(gap_score));
    const uint32_t MAT2   = pin_vgpr(pk_dup(match_score));
    const uint32_t DIF2   = pin_vgpr(pk_dup(mismatch_score - match_score));
    const uint32_t ONE2   = pin_vgpr(0x00010001u);
    const uint32_t THREE2 = pin_vgpr(0x00030003u);
    const uint32_t NEG1   = pin_vgpr(0xffffffffu);
    // t * gap for the lane's cells t = 4*lane + k
    const uint32_t K01 = pk_make((lane4 + 0) * gap_score, (lane4 + 1) * gap_score);
    const uint32_t K23 = pk_make((lane4 + 2) * gap_score, (lane4 + 3) * gap_score);
    const uint32_t ring_base = lds_addr(ring);
    const uint32_t read_base = lds_addr(lds_read);
    // guard store: lanes 0..15 write sentinel cells for columns band_end + 1 .. + 64, lane 16 the quad that ends in
    // the left-boundary slot (column band_start); byte offsets relative to the lane's own cell offset
    // (lanes 17..63 take part in the same store -- masking them off costs two writes of EXEC per row, ~45 cycles for a lone
    // wavefront -- and write quads of columns band_start - 7 and below, lane l at byte 2 band_start - 16 - 8 (l - 17): no reader
    // of this row looks left of column band_start - 3, and the lowest of them, 191 columns down, still lies clear of the
    // sentinel cells when the slot wraps)
    const uint32_t guard_off  = lane < 16 ? (uint32_t)(2 * BW) : (lane == 16 ? (uint32_t)-136 : (uint32_t)(120 - 16 * lane));
    const bool is_lane16      = lane == 16;
    const bool is_lane63      = lane == kBandLanes - 1; // the band's last lane
    const uint32_t move_keep  = lane == 0 ? 0xffffff00u : 0xffffffffu; // the band's first cell stays undecided
    // the guard quad's second dword in rows whose left boundary is min_score by construction
    const uint32_t GUARD_HI_MIN = pin_vgpr(is_lane16 ? (((uint32_t)kPkSentinel & 0xffffu) | ((uint32_t)min_score << 16)) : pk_dup(kPkSentinel));
    const uint32_t S0X_MIN      = (uint32_t)min_score << 16; // "cell of column band_start" of such a row, in the high half

    // state carried from row to row
    uint32_t P01 = pk_make((lane4 + 1) * gap_score, (lane4 + 2) * gap_score); // row 0: H[0][x] = x * gap
    uint32_t P23 = pk_make((lane4 + 3) * gap_score, (lane4 + 4) * gap_score);
    int32_t prev_rel0 = 0; // left-boundary value of the row in P (tracked while band starts are 0, and by general rows)
    // per-lane values that only change when the band moves: read characters of columns c+1..c+4 (rd4) and of the next
    // quad (rd4n), ring byte offset of the lane's quad (a1) and of its guard quad (ga)
    uint32_t rd4  = lds_load_u32(read_base + lane4);
    uint32_t rd4n = lds_load_u32(read_base + lane4 + 4);
    uint32_t a1   = (uint32_t)lane8;
    uint32_t ga   = (a1 + guard_off) & (kPkSlotBytes - 1);
    // per-lane pointers to the lane's quad in the HBM score row / to its four bytes in the move row of the CURRENT row
    uint8_t* const score_base = reinterpret_cast<uint8_t*>(scores) + lane8 + 2 * (1 + kRelShift); // the lane's quad in row 0
    uint8_t* move_ptr  = moves + lane4 + (1 + kRelShift);

    // row 0 into ring slot 0
    if (BW == 256 || band_lane) lds_store_u64(ring_base + a1, P01, P23);
    lds_store_guard(ring_base + ga, SENT2, is_lane16 ? (((uint32_t)kPkSentinel & 0xffffu) | (0u << 16)) : SENT2);

    // horizontal max-plus scan of the row's candidates; cu = carry-in as element t = -1 of u; leaves the row in P01/P23
    auto scan_row = [&](uint32_t s01, uint32_t s23, int32_t cu) {
        const uint32_t u01 = pk_sub(s01, K01), u23 = pk_sub(s23, K23);
        const uint32_t pm01 = pk_max(u01, (u01 << 16) | 0x8000u);
        const uint32_t pm23 = pk_max(u23, (u23 << 16) | 0x8000u);
        const int32_t m3    = (int32_t)pk_max(pm01, pm23) >> 16; // max(u0..u3)
        const int32_t incl  = ab_scan ? wave_inclusive_max(m3) : m3;
        const int32_t excl  = max(wave_shr1(incl, cu), cu); // lane 0: the carry-in alone
        const uint32_t ex2  = __builtin_amdgcn_perm((uint32_t)excl, (uint32_t)excl, 0x01000100u);
        const uint32_t m1b  = __builtin_amdgcn_perm(pm01, pm01, 0x03020302u); // max(u0,u1) in both halves
        P01 = pk_add(pk_max(pm01, ex2), K01);
        P23 = pk_add(pk_max(pk_max(pm23, m1b), ex2), K23);
    };
    // 0 where the halves are equal, 1 where they differ
    auto nz = [&](uint32_t a, uint32_t b) -> uint32_t { return pk_min_u16(pk_sub(a, b), ONE2); };
    // match / mismatch cost pairs of this row's base (replicated into four bytes) against the lane's read characters
    auto costs = [&](uint32_t base4, uint32_t& c01, uint32_t& c23) {
        const uint32_t x   = rd4 ^ base4;
        const uint32_t x01 = __builtin_amdgcn_perm(0u, x, 0x0c010c00u); // (byte0, byte1) zero-extended to halves
        const uint32_t x23 = __builtin_amdgcn_perm(0u, x, 0x0c030c02u);
        c01 = pk_mad_u16(pk_min_u16(x01, ONE2), DIF2, MAT2);
        c23 = pk_mad_u16(pk_min_u16(x23, ONE2), DIF2, MAT2);
    };
    // diagonal / vertical candidates of the four cells from one predecessor row: q01/q23 = its cells of columns
    // c+1..c+4, s0x = its cell of column c in the HIGH half
    auto from_pred = [&](uint32_t s0x, uint32_t q01, uint32_t q23, uint32_t c01, uint32_t c23, uint32_t& D01, uint32_t& D23,
                         uint32_t& V01, uint32_t& V23) {
        D01 = pk_add(__builtin_amdgcn_alignbit(q01, s0x, 16), c01);
        D23 = pk_add(__builtin_amdgcn_alignbit(q23, q01, 16), c23);
        V01 = pk_add(q01, GAP2);
        V23 = pk_add(q23, GAP2);
    };
    // the finished row (P01/P23) of row r: HBM score row, ring slot r & 7 with its guard quad, and its move bytes
    auto store_row = [&](auto bs0_tag, int32_t r, int32_t rel0_val, uint32_t mv4, bool scores_to_hbm) {
        constexpr bool BS0 = decltype(bs0_tag)::value;
        move_ptr += stride;
        const uint32_t sbase = ring_base + (((uint32_t)r & (kPkSlots - 1)) * kPkSlotBytes);
        if (BS0 || __builtin_expect(scores_to_hbm, 0)) // (wave-uniform)
        {
            uint8_t* score_ptr = score_base + (uint32_t)r * (uint32_t)(stride * 2);
            if (st_scores && (BW == 256 || band_lane)) gstore_nt_u64(score_ptr, P01, P23);
            if constexpr (BS0) gstore_u16_lane0_below(score_ptr, (uint32_t)rel0_val); // a real left-boundary value
        }
        if (ab_ring && (BW == 256 || band_lane)) lds_store_u64(sbase + a1, P01, P23);
        if constexpr (BS0)
        {
            const uint32_t rel0pk = ((uint32_t)kPkSentinel & 0xffffu) | ((uint32_t)rel0_val << 16);
            if (ab_guard) lds_store_guard(sbase + ga, SENT2, is_lane16 ? rel0pk : SENT2);
            prev_rel0 = rel0_val;
        }
        else if (ab_guard)
            lds_store_guard(sbase + ga, SENT2, GUARD_HI_MIN);
        if (st_moves && (BW == 256 || band_lane)) *reinterpret_cast<uint32_t*>(move_ptr) = mv4;
    };
    // four move bytes from two registers of 16-bit moves
    auto pack_moves = [&](uint32_t m01, uint32_t m23) -> uint32_t {
        return __builtin_amdgcn_perm(m23, m01, 0x06040200u) & move_keep; // low byte of each half
    };

    // ---------------- general row (kind 4): 32-bit arithmetic, previous row from registers, others from HBM ----------------
    auto general_row = [&](int32_t r, int32_t& prev_rel0_io) {
        const RowInfo<true> ri = uniform_row(rowinfo[r]);
        const int32_t bs       = ri.bs();
        const uint32_t base    = (uint32_t)ri.base();
        const int32_t prev_bs  = r > 1 ? uniform_row(rowinfo[r - 1]).bs() : 0;
        a1   = (uint32_t)(2 * bs + lane8) & (kPkSlotBytes - 1);
        ga   = (a1 + guard_off) & (kPkSlotBytes - 1);
        rd4  = lds_load_u32(read_base + bs + lane4);
        rd4n = lds_load_u32(read_base + bs + lane4 + 4);
        const int32_t pred_count = ri.cnt();
        const int32_t c          = bs + lane4;
        const int32_t cp0 = ((rd4 & 0xff) == base) ? match_score : mismatch_score;
        const int32_t cp1 = (((rd4 >> 8) & 0xff) == base) ? match_score : mismatch_score;
        const int32_t cp2 = (((rd4 >> 16) & 0xff) == base) ? match_score : mismatch_score;
        const int32_t cp3 = ((rd4 >> 24) == base) ? match_score : mismatch_score;
        const int32_t R0 = pk_lo(P01), R1 = pk_hi(P01), R2 = pk_lo(P23), R3 = pk_hi(P23);
        bool synced = false;
        auto from_regs = [&](int32_t& t0, int32_t& t1, int32_t& t2, int32_t& t3) {
            const int32_t q    = (bs - prev_bs) >> 2;
            const int32_t pend = min(prev_bs + band_width - kCellsPerLane, max_column);
            const int src      = lane + q;
            // the shuffle must run with every lane active (a lane that is masked off does not supply its value)
            const int32_t from_left = __shfl(R3, src - 1);
            const int32_t S0        = (q == 0 && lane == 0) ? prev_rel0_io : from_left;
            const int32_t S1 = __shfl(R0, src), S2 = __shfl(R1, src), S3 = __shfl(R2, src), S4 = __shfl(R3, src);
            const bool valid = c <= pend;
            t0 = valid ? max(S0 + cp0, S1 + gap_score) : min_score;
            t1 = valid ? max(S1 + cp1, S2 + gap_score) : min_score;
            t2 = valid ? max(S2 + cp2, S3 + gap_score) : min_score;
            t3 = valid ? max(S3 + cp3, S4 + gap_score) : min_score;
        };
        auto from_hbm = [&](int32_t prow, int32_t& t0, int32_t& t1, int32_t& t2, int32_t& t3) {
            const int32_t pbs  = prow == 0 ? 0 : uniform_row(rowinfo[prow]).bs();
            const int32_t pend = min(pbs + band_width - kCellsPerLane, max_column);
            const bool valid   = !(c > pend || c < pbs);
            if (!synced) { wave_sync(); synced = true; }
            // no divergent branch in this routine (the row loop stays scalar control flow): lanes outside the predecessor's
            // band load from the row's first quad and discard it
            const int16_t* rowp = scores + (int64_t)prow * stride + (valid ? (c - pbs) : 0) + kRelShift;
            int32_t S0 = rowp[0];
            const Quad<int16_t> qd = *reinterpret_cast<const Quad<int16_t>*>(rowp + 1);
            const int32_t S1 = qd.v[0], S2 = qd.v[1], S3 = qd.v[2], S4 = qd.v[3];
            if (pbs > 0 && c == pbs) S0 = min_score; // relative-0 slot of a row whose band starts past column 0
            t0 = valid ? max(S0 + cp0, S1 + gap_score) : min_score;
            t1 = valid ? max(S1 + cp1, S2 + gap_score) : min_score;
            t2 = valid ? max(S2 + cp2, S3 + gap_score) : min_score;
            t3 = valid ? max(S3 + cp3, S4 + gap_score) : min_score;
        };
        auto rel0_of = [&](int32_t prow) -> int32_t {
            if (prow == r - 1) return prev_rel0_io;
            const int32_t pbs = prow == 0 ? 0 : uniform_row(rowinfo[prow]).bs();
            if (pbs > 0) return min_score;
            if (!synced) { wave_sync(); synced = true; }
            return wave_first((int32_t)scores[(int64_t)prow * stride + kRelShift]);
        };
        const int32_t node_id = (pred_count > 3) ? (int32_t)g.sorted_poa[r - 1] : 0;
        auto pred_row = [&](int32_t p) -> int32_t {
            if (pred_count == 0) return 0;
            if (p < 3) return ri.pred(p);
            return wave_first((int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1);
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
            else from_hbm(prow, t0, t1, t2, t3);
            if (p == 0) { s0 = t0; s1 = t1; s2 = t2; s3 = t3; }
            else { s0 = max(s0, t0); s1 = max(s1, t1); s2 = max(s2, t2); s3 = max(s3, t3); }
        }
        scan_row(pk_make(s0, s1), pk_make(s2, s3), fe + gap_score);
        // stores (either flavour of left boundary)
        uint8_t* score_ptr = score_base + (uint32_t)r * (uint32_t)(stride * 2);
        move_ptr += stride;
        const uint32_t sbase  = ring_base + (((uint32_t)r & (kPkSlots - 1)) * kPkSlotBytes);
        const uint32_t rel0pk = ((uint32_t)kPkSentinel & 0xffffu) | ((uint32_t)rel0_val << 16);
        if (BW == 256 || band_lane)
        {
            *reinterpret_cast<uint2*>(score_ptr) = make_uint2(P01, P23);
            lds_store_u64(sbase + a1, P01, P23);
        }
        lds_store_guard(sbase + ga, SENT2, is_lane16 ? rel0pk : SENT2);
        if (bs == 0) gstore_u16_lane0_below(score_ptr, (uint32_t)rel0_val);
        if (BW == 256 || band_lane) *reinterpret_cast<uint32_t*>(move_ptr) = 0u;
        prev_rel0_io = rel0_val;
    };

    // ---------------- kinds 0 / 1: one predecessor, the previous row, in registers ----------------
    auto reg_row = [&](auto bs0_tag, auto moved_tag, int32_t r, uint32_t d0, uint32_t base4, bool scores_to_hbm) {
        constexpr bool BS0   = decltype(bs0_tag)::value;
        constexpr bool MOVED = decltype(moved_tag)::value;
        uint32_t s0x, q01, q23;
        if constexpr (!MOVED)
        {
            s0x = (uint32_t)wave_shr1((int32_t)P23, BS0 ? (int32_t)((uint32_t)prev_rel0 << 16) : (int32_t)S0X_MIN);
            q01 = P01; q23 = P23;
        }
        else
        {
            // band moved by one quad: the lane's columns are those of the next lane of the previous row
            s0x  = P23;
            q01  = (uint32_t)wave_shl1((int32_t)P01, (int32_t)pk_dup(kPkSentinel));
            q23  = (uint32_t)wave_shl1((int32_t)P23, (int32_t)pk_dup(kPkSentinel));
            a1   = (a1 + 8) & (kPkSlotBytes - 1);
            ga   = (ga + 8) & (kPkSlotBytes - 1);
            rd4  = rd4n;
            rd4n = lds_load_u32(read_base + (((d0 >> 3) & 0x1ffu) << 2) + lane4 + 4);
        }
        int32_t cu = min_score + 2 * gap_score, rel0_val = min_score;
        if constexpr (BS0)
        {
            rel0_val = max(min_score, prev_rel0) + gap_score;
            cu       = rel0_val + gap_score;
        }
        uint32_t c01, c23, D01, D23, V01, V23;
        costs(base4, c01, c23);
        from_pred(s0x, q01, q23, c01, c23, D01, D23, V01, V23);
        uint32_t s01 = pk_max(D01, V01), s23 = pk_max(D23, V23);
        if constexpr (MOVED) // lane 63: the chunk beyond the previous row's band
        {
            s01 = is_lane63 ? MIN2 : s01;
            s23 = is_lane63 ? MIN2 : s23;
        }
        scan_row(s01, s23, cu);
        // move = H == D ? 3 : H == V ? 2 : 1   ==  3 + [H != D] * (-1 - [H != V])
        const uint32_t m01 = pk_mad_u16(nz(P01, D01), pk_mad_u16(nz(P01, V01), NEG1, NEG1), THREE2);
        const uint32_t m23 = pk_mad_u16(nz(P23, D23), pk_mad_u16(nz(P23, V23), NEG1, NEG1), THREE2);
        uint32_t mv4 = ab_moves ? pack_moves(m01, m23) : 0u;
        if constexpr (MOVED) mv4 = is_lane63 ? 0u : mv4;
        store_row(bs0_tag, r, rel0_val, mv4, scores_to_hbm);
    };

    // ---------------- the rows of one phase ----------------
    auto run_rows = [&](auto bs0_tag, int32_t r_from, int32_t r_to) {
        constexpr bool BS0 = decltype(bs0_tag)::value;
        int32_t r = r_from;
        while (r <= r_to)
        {
            // descriptors of rows r .. r + 63, one per lane
            const int32_t r0   = r;
            uint32_t D0v, D1v;
            {
                const int32_t rr    = min(r0 + lane, graph_count);
                const uint64_t w    = rowinfo[rr].w;
                const uint32_t kind = (uint32_t)(w >> kKindShift) & 7u;
                const uint32_t cnt  = (uint32_t)(w >> 8) & 0x3fu;
                const uint32_t bs4  = (uint32_t)(w >> 15) & 0x1ffu;
                const uint32_t p0 = (uint32_t)(w >> 24) & 0xfffu, p1 = (uint32_t)(w >> 36) & 0xfffu, p2 = (uint32_t)(w >> 48) & 0xfffu;
                const uint32_t slots = (p0 & 7u) | ((p1 & 7u) << 3) | ((p2 & 7u) << 6);
                const uint32_t dists = (((uint32_t)rr - p0) & 7u) | ((((uint32_t)rr - p1) & 7u) << 3) | ((((uint32_t)rr - p2) & 7u) << 6);
                D0v = kind | (bs4 << 3) | (slots << 12) | (dists << 21) | ((cnt <= 3 ? cnt : 0u) << 30);
                D1v = ((uint32_t)w & 0xffu) * 0x01010101u;
            }
            // rows past the end of the phase read as kind 7 = "end of block"
            D0v = (r0 + lane <= r_to) ? D0v : 7u;
            // bit j: row r0 + j writes its score row to HBM (mark_score_rows)
            const uint64_t need64 = BS0 ? ~0ull : __ballot((rowinfo[min(r0 + lane, graph_count)].w & kRowScoresInHbm) != 0);
            int32_t k      = 0;
            uint32_t d0    = (uint32_t)__builtin_amdgcn_readlane((int32_t)D0v, 0);
            uint32_t base4 = (uint32_t)__builtin_amdgcn_readlane((int32_t)D1v, 0);
            uint32_t kind  = d0 & 7u;
            auto advance = [&]() {
                r++;
                k++;
                const uint32_t nd = (uint32_t)__builtin_amdgcn_readlane((int32_t)D0v, k & (kWave - 1));
                base4 = (uint32_t)__builtin_amdgcn_readlane((int32_t)D1v, k & (kWave - 1));
                d0    = k == kWave ? 7u : nd; // a select, not a branch
                kind  = d0 & 7u;
            };
            for (;;)
            {
                // streak of rows whose predecessor is the previous row and whose band did not move
                const uint64_t t_k0 = ksel == 0 ? clock64() : 0;
                const int32_t r_k0  = r;
                while (kind == 0)
                {
                    reg_row(bs0_tag, std::false_type{}, r, d0, base4, ((need64 >> k) & 1ull) != 0);
                    advance();
                }
                if (ksel == 0) kacc += kcount ? (uint64_t)(r - r_k0) : clock64() - t_k0;
                if (kind == 7u) break;
                const int32_t kind_now = (int32_t)min(kind, 4u);
                const uint64_t t_kx    = ksel == kind_now ? clock64() : 0;
                const bool scores_to_hbm = ((need64 >> k) & 1ull) != 0;
                if (kind == 1)
                    reg_row(bs0_tag, std::true_type{}, r, d0, base4, scores_to_hbm);
                else if (kind <= 3)
                {
                    // ===== predecessors from the LDS ring =====
                    const uint32_t bs = ((d0 >> 3) & 0x1ffu) << 2;
                    a1 = (2u * bs + (uint32_t)lane8) & (kPkSlotBytes - 1);
                    ga = (a1 + guard_off) & (kPkSlotBytes - 1);
                    const uint32_t a0 = (a1 - 4) & (kPkSlotBytes - 1); // dword whose high half is the cell of column c
                    const uint32_t b0 = ring_base + (((d0 >> 12) & 7u) * kPkSlotBytes);
                    const uint32_t sent16 = (uint32_t)kPkSentinel & 0xffffu;
                    const uint32_t dd0 = (d0 >> 21) & 7u;
                    if (kind == 2)
                    {
                        // all loads first (one LDS round trip), then the arithmetic
                        const uint32_t x0 = lds_load_u32(b0 + a0);
                        const uint2 q0    = lds_load_u64(b0 + a1);
                        rd4  = lds_load_u32(read_base + bs + lane4);
                        rd4n = lds_load_u32(read_base + bs + lane4 + 4);
                        int32_t cu = min_score + 2 * gap_score, rel0_val = min_score;
                        if constexpr (BS0)
                        {
                            rel0_val = max(min_score, (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(b0 + kPkSlotBytes - 4) >> 16))) + gap_score;
                            cu       = rel0_val + gap_score;
                        }
                        uint32_t c01, c23, D01, D23, V01, V23;
                        costs(base4, c01, c23);
                        from_pred(x0, q0.x, q0.y, c01, c23, D01, D23, V01, V23);
                        const bool outside = (q0.x & 0xffffu) == sent16; // chunk beyond the predecessor's band
                        const uint32_t s01 = pk_max(D01, V01), s23 = pk_max(D23, V23);
                        scan_row(outside ? MIN2 : s01, outside ? MIN2 : s23, cu);
                        // move = H == D ? 2 d + 1 : H == V ? 2 d : 1  ==  (2 d + 1) + [H != D] * (-1 + [H != V] * (1 - 2 d))
                        const uint32_t cD = pk_dup((int32_t)(2u * dd0 + 1u)), cV = pk_dup(1 - (int32_t)(2u * dd0));
                        const uint32_t m01 = pk_mad_u16_vvs(nz(P01, D01), pk_mad_u16_vsv(nz(P01, V01), cV, NEG1), cD);
                        const uint32_t m23 = pk_mad_u16_vvs(nz(P23, D23), pk_mad_u16_vsv(nz(P23, V23), cV, NEG1), cD);
                        const uint32_t mv4 = outside ? 0u : pack_moves(m01, m23);
                        store_row(bs0_tag, r, rel0_val, mv4, scores_to_hbm);
                    }
                    else
                    {
                        const uint32_t cnt3    = d0 >> 30;              // 2, 3, or 0 = more than three
                        const int32_t cnt      = cnt3 == 2 ? 2 : 3;     // predecessors in the descriptor
                        const uint32_t b1 = ring_base + (((d0 >> 15) & 7u) * kPkSlotBytes);
                        const uint32_t b2 = cnt > 2 ? ring_base + (((d0 >> 18) & 7u) * kPkSlotBytes) : b0;
                        const uint32_t dd1 = (d0 >> 24) & 7u, dd2 = (d0 >> 27) & 7u;
                        const uint32_t x0 = lds_load_u32(b0 + a0);
                        const uint2 q0    = lds_load_u64(b0 + a1);
                        const uint32_t x1 = lds_load_u32(b1 + a0);
                        const uint2 q1    = lds_load_u64(b1 + a1);
                        uint32_t x2 = 0;
                        uint2 q2 = make_uint2(0, 0);
                        if (cnt > 2)
                        {
                            x2 = lds_load_u32(b2 + a0);
                            q2 = lds_load_u64(b2 + a1);
                        }
                        rd4  = lds_load_u32(read_base + bs + lane4);
                        rd4n = lds_load_u32(read_base + bs + lane4 + 4);
                        int32_t fe = min_score + gap_score;
                        if constexpr (BS0) // left boundary in band: carry-in from the predecessors' column-0 values (:293-326)
                        {
                            int32_t pen = max(min_score, (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(b0 + kPkSlotBytes - 4) >> 16)));
                            pen = max(pen, (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(b1 + kPkSlotBytes - 4) >> 16)));
                            if (cnt > 2) pen = max(pen, (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(b2 + kPkSlotBytes - 4) >> 16)));
                            fe = pen + gap_score;
                        }
                        uint32_t c01, c23;
                        costs(base4, c01, c23);
                        // best diagonal / vertical candidate over the predecessors and the move of the first slot that attains it
                        uint32_t D0a, D0b, V0a, V0b, D1a, D1b, V1a, V1b;
                        from_pred(x0, q0.x, q0.y, c01, c23, D0a, D0b, V0a, V0b);
                        from_pred(x1, q1.x, q1.y, c01, c23, D1a, D1b, V1a, V1b);
                        const bool out0 = (q0.x & 0xffffu) == sent16, out1 = (q1.x & 0xffffu) == sent16;
                        bool undecided  = out0 | out1;
                        D0a = out0 ? MIN2 : D0a; D0b = out0 ? MIN2 : D0b; V0a = out0 ? MIN2 : V0a; V0b = out0 ? MIN2 : V0b;
                        D1a = out1 ? MIN2 : D1a; D1b = out1 ? MIN2 : D1b; V1a = out1 ? MIN2 : V1a; V1b = out1 ? MIN2 : V1b;
                        uint32_t bD01 = pk_max(D0a, D1a), bD23 = pk_max(D0b, D1b), bV01 = pk_max(V0a, V1a), bV23 = pk_max(V0b, V1b);
                        // diagonal move through slot k = 2 d_k + 1, vertical = 2 d_k: first attaining slot
                        //   A = mD0 + n0 * (E1 + n1 * E2),  E1 = 2 (d1 - d0), E2 = 2 (d2 - d1),  n_k = [slot k misses the maximum]
                        const uint32_t mD0 = pk_dup((int32_t)(2u * dd0 + 1u)), mV0 = pk_dup((int32_t)(2u * dd0));
                        const uint32_t E1  = pk_dup(2 * ((int32_t)dd1 - (int32_t)dd0));
                        uint32_t A01, A23, B01, B23;
                        if (cnt > 2)
                        {
                            uint32_t D2a, D2b, V2a, V2b;
                            from_pred(x2, q2.x, q2.y, c01, c23, D2a, D2b, V2a, V2b);
                            const bool out2 = (q2.x & 0xffffu) == sent16;
                            undecided       = undecided | out2;
                            D2a = out2 ? MIN2 : D2a; D2b = out2 ? MIN2 : D2b; V2a = out2 ? MIN2 : V2a; V2b = out2 ? MIN2 : V2b;
                            bD01 = pk_max(bD01, D2a); bD23 = pk_max(bD23, D2b); bV01 = pk_max(bV01, V2a); bV23 = pk_max(bV23, V2b);
                            const uint32_t E2  = pin_vgpr(pk_dup(2 * ((int32_t)dd2 - (int32_t)dd1)));
                            const uint32_t E1v = pin_vgpr(E1);
                            A01 = pk_mad_u16_vvs(nz(bD01, D0a), pk_mad_u16(nz(bD01, D1a), E2, E1v), mD0);
                            A23 = pk_mad_u16_vvs(nz(bD23, D0b), pk_mad_u16(nz(bD23, D1b), E2, E1v), mD0);
                            B01 = pk_mad_u16_vvs(nz(bV01, V0a), pk_mad_u16(nz(bV01, V1a), E2, E1v), mV0);
                            B23 = pk_mad_u16_vvs(nz(bV23, V0b), pk_mad_u16(nz(bV23, V1b), E2, E1v), mV0);
                        }
                        else
                        {
                            const uint32_t E1v = pin_vgpr(E1);
                            A01 = pk_mad_u16_vvs(nz(bD01, D0a), E1v, mD0); A23 = pk_mad_u16_vvs(nz(bD23, D0b), E1v, mD0);
                            B01 = pk_mad_u16_vvs(nz(bV01, V0a), E1v, mV0); B23 = pk_mad_u16_vvs(nz(bV23, V0b), E1v, mV0);
                        }
                        if (cnt3 == 0)
                        {
                            // predecessors 3..5 (rows from the side table, cells from the ring): they raise the maxima; where only
                            // they attain a maximum the first attaining slot is >= 3, whose distance the pass does not track -> move 0
                            const uint64_t xe     = wave_first64(xpred[r & 255]);
                            const int32_t cnt_all = (int32_t)((xe >> 13) & 63u);
                            uint32_t xD01 = MIN2, xD23 = MIN2, xV01 = MIN2, xV23 = MIN2;
                            int32_t pen_x = min_score;
                            for (int32_t kk = 3; kk < cnt_all; kk++)
                            {
                                const uint32_t bk = ring_base + (((uint32_t)xpred_row(xe, kk) & 7u) * kPkSlotBytes);
                                const uint32_t xk = lds_load_u32(bk + a0);
                                const uint2 qk    = lds_load_u64(bk + a1);
                                uint32_t Da, Db, Va, Vb;
                                from_pred(xk, qk.x, qk.y, c01, c23, Da, Db, Va, Vb);
                                const bool outk = (qk.x & 0xffffu) == sent16;
                                undecided       = undecided | outk;
                                xD01 = pk_max(xD01, outk ? MIN2 : Da); xD23 = pk_max(xD23, outk ? MIN2 : Db);
                                xV01 = pk_max(xV01, outk ? MIN2 : Va); xV23 = pk_max(xV23, outk ? MIN2 : Vb);
                                if constexpr (BS0) pen_x = max(pen_x, (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(bk + kPkSlotBytes - 4) >> 16)));
                            }
                            if constexpr (BS0) fe = max(fe - gap_score, pen_x) + gap_score;
                            const uint32_t fD01 = pk_max(bD01, xD01), fD23 = pk_max(bD23, xD23), fV01 = pk_max(bV01, xV01), fV23 = pk_max(bV23, xV23);
                            // A *= [max of the first three == overall max]
                            A01 = pk_mad_u16(nz(bD01, fD01), pk_sub(0u, A01), A01); A23 = pk_mad_u16(nz(bD23, fD23), pk_sub(0u, A23), A23);
                            B01 = pk_mad_u16(nz(bV01, fV01), pk_sub(0u, B01), B01); B23 = pk_mad_u16(nz(bV23, fV23), pk_sub(0u, B23), B23);
                            bD01 = fD01; bD23 = fD23; bV01 = fV01; bV23 = fV23;
                        }
                        const int32_t rel0_val = BS0 ? fe : min_score;
                        scan_row(pk_max(bD01, bV01), pk_max(bD23, bV23), fe + gap_score);
                        // move = H == bestD ? A : H == bestV ? B : 1
                        auto move_of = [&](uint32_t H, uint32_t bD, uint32_t bV, uint32_t A, uint32_t B) -> uint32_t {
                            const uint32_t t1 = pk_mad_u16(nz(H, bV), pk_sub(ONE2, B), B);
                            return pk_mad_u16(nz(H, bD), pk_sub(t1, A), A);
                        };
                        const uint32_t m01 = move_of(P01, bD01, bV01, A01, B01);
                        const uint32_t m23 = move_of(P23, bD23, bV23, A23, B23);
                        const uint32_t mv4 = undecided ? 0u : pack_moves(m01, m23);
                        store_row(bs0_tag, r, rel0_val, mv4, scores_to_hbm);
                    }
                }
                else
                {
                    general_row(r, prev_rel0);
                    if constexpr (!BS0) prev_rel0 = min_score;
                }
                if (ksel == kind_now) kacc += kcount ? 1 : clock64() - t_kx;
                advance();
            }
        }
    };

    const int32_t bs0_end = min(first_moved - 1, graph_count); // last row whose band starts at column 0
    if (!ab_rows) return;
    run_rows(std::true_type{}, 1, bs0_end);
    // from here on every row's left boundary is min_score by construction; the first such row sees the previous row's
    // real boundary through the general routine (band-start transition rows are kind 4)
    run_rows(std::false_type{}, bs0_end + 1, graph_count);
    if (ksel >= 0 && lane == 0) *prof_acc += kacc;
}

} // namespace gwhip
