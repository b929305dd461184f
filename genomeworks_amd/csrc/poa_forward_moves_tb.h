// poa_forward_moves_tb.h -- the packed forward pass of poa_forward_moves.h / poa_forward_moves_wide.h for the
// traceback-buffer modes (round 4): cudapoa_nw_tb_banded.cuh:264-531 (restated in oracle/poa_nw_tb.inc) for bands of 128,
// 256 (one register pass per row, LDS ring of 8 rows), 384 and 512 columns (two passes, ring of 4 rows) with int16 scores.
//
// What differs from the score-matrix modes, and how it is kept:
//   * THE CELL RULE (:140-262, :476-513). Per predecessor, in edge order: candidate = max(diagonal, vertical), diagonal
//     on a tie; a candidate replaces the running score only if STRICTLY greater, starting from min_score (so a cell no
//     candidate beats stays min_score with trace 0); the horizontal move wins only if strictly greater and then zeroes
//     the trace. Per cell that is: s = max(min_score, max_k M_k), M_k = max(D_k, V_k); winner = first k with M_k == s;
//     trace = +d_k if M_k == D_k else -d_k; trace = 0 if s == min_score or H != s. Evaluated on packed 16-bit pairs.
//   * THE SCORE RING in HBM keeps the reference's own element indices (row % H, relative index, stride band + 8, no
//     alignment shift): the general rows below and the sink selection read it exactly as the memory-faithful routine of
//     poa_tb_device.h does, including the stray set_score_tb(column = -1) store (:47-69) where it can be observed (band
//     starts beyond the row's own slot). A lane's four cells sit at relative index 4 l + 1 .. 4 l + 4, so every lane
//     stores the ALIGNED quad 4 l .. 4 l + 3 (its left neighbour's last cell -- the boundary slot for lane 0 -- and its
//     own first three) and the band's last lane adds the last cell.
//   * THE TRACE MATRIX is internal to one alignment (written by the forward pass, read by its own walk), so the packed
//     pass keeps it as two byte planes inside the reference's int16 region (trace16 configurations only): plane 0 =
//     move bytes in the format and layout of the score-matrix modes' move matrix (rows up << 1 | columns left, element
//     rel + 3), which the sheared-tile walk of poa_traceback_moves.h reads unchanged -- here every cell of a packed row
//     is decided, there is no recomputation; plane 1 = the exact int8 trace of the GENERAL rows (pred_count 0, band-start
//     transitions, predecessors beyond the LDS ring), whose plane-0 bytes are 0 so that the walk takes them one step
//     at a time.
//   * Anything the byte planes cannot hold (a predecessor 127 or more rows up) or that depends on ring-slot aliasing (a
//     slot-0 predecessor H or more rows up, :456-457) makes the pass return false; the caller then runs the
//     memory-faithful routine for that read.
#pragma once

namespace gwhip
{

// plane-1 bytes: the exact trace of a general row's cell, -126 .. 126; a row without predecessors names row 0 (the
// reference's trace is -r / +r there, which a byte cannot hold)
constexpr int kTbMaxDelta   = 126;
constexpr int kTbVertToRow0 = -128;
constexpr int kTbDiagToRow0 = 127;

struct TbPlanes
{
    int16_t* ring;      // HBM score ring, reference layout: H rows of (band + 8) elements
    size_t ring_elems;
    int32_t H;          // max_banded_pred_distance
    uint8_t* plane0;    // move bytes, row stride band + 8, element rel + kRelShift
    uint8_t* plane1;    // exact int8 traces of the general rows (kTb* codes above), same indexing
};

// lane 0 only: one byte just below the lane's own pointer (the boundary cell of a move / trace row)
__device__ __forceinline__ void gstore_u8_lane0_below(const void* vptr, uint32_t v)
{
    asm volatile("s_mov_b64 exec, 1\n\tglobal_store_byte %0, %1, off offset:-1\n\ts_mov_b64 exec, -1" ::"v"(vptr), "v"(v) : "memory");
}
// one lane only (mask): the high half of v as a 2-byte store 8 bytes above the lane's own pointer
__device__ __forceinline__ void gstore_hi16_lane_above8(uint64_t lane_mask, const void* vptr, uint32_t v)
{
    asm volatile("s_mov_b64 exec, %0\n\tglobal_store_short_d16_hi %1, %2, off offset:8\n\ts_mov_b64 exec, -1" ::"s"(lane_mask), "v"(vptr), "v"(v) : "memory");
}

template <typename IdT, int BW>
__device__ __forceinline__ bool banded_forward_tb(const GraphView<IdT>& g, RowInfo<true>* rowinfo, int32_t graph_count,
                                                  const uint8_t* lds_read, uint8_t* ring, const uint64_t* xpred, int32_t max_column,
                                                  int32_t gap_score, int32_t mismatch_score, int32_t match_score, const TbPlanes& tb,
                                                  int32_t dbg)
{
    static_assert(BW == 128 || BW == 256 || BW == 384 || BW == 512, "band widths of the packed passes");
    constexpr int32_t band_width = BW;
    constexpr int32_t stride     = band_width + kRightPad;
    constexpr int NP             = BW > 256 ? 2 : 1;                      // register passes of 256 columns per row
    constexpr int kLastLanes     = (BW - 256 * (NP - 1)) / kCellsPerLane; // lanes that own band cells of the last pass
    constexpr int kSlots         = NP == 2 ? kWdSlots : kPkSlots;         // LDS ring rows
    constexpr int kSlotBytes     = NP == 2 ? kWdSlotBytes : kPkSlotBytes;
    constexpr int kMaxD          = NP == 2 ? kWdMaxDist : kPkMaxDist;     // rows up a ring predecessor may be
    constexpr uint32_t kMask     = kSlotBytes - 1;
    const int lane               = threadIdx.x & (kWave - 1);
    const bool last_pass_lane    = lane < kLastLanes; // this lane owns band cells of the last pass (every lane owns cells of the others)
    const int32_t lane4 = lane * 4, lane8 = lane * 8;
    const int32_t min_score = Limits<int16_t>::min / 2;
    const int32_t H         = tb.H;

    const int32_t first_moved = classify_kinds<kMaxD, true>(rowinfo, graph_count, lane, xpred, dbg);
    // rows this pass cannot reproduce: a slot-0 predecessor H or more rows up (it is read through an aliased ring slot), or too
    // far for a plane-1 byte
    {
        bool far0 = false;
        for (int32_t r = 1 + lane; r <= graph_count; r += kWave)
        {
            const RowInfo<true> ri = rowinfo[r];
            far0 = far0 || (ri.cnt() > 0 && r - ri.pred(0) >= min(H, kTbMaxDelta + 1));
        }
        if (__ballot(far0) != 0) return false;
    }
    wave_sync();

    const uint32_t MIN2   = pin_vgpr(pk_dup(min_score));
    const uint32_t SENT2  = pin_vgpr(pk_dup(kPkSentinel));
    const uint32_t GAP2   = pin_vgpr(pk_dup(gap_score));
    const uint32_t MAT2   = pin_vgpr(pk_dup(match_score));
    const uint32_t DIF2   = pin_vgpr(pk_dup(mismatch_score - match_score));
    const uint32_t ONE2   = pin_vgpr(0x00010001u);
    const uint32_t NEG1   = pin_vgpr(0xffffffffu);
    // t * gap for the lane's cells: pass p, t = 256 p + 4 lane + k
    uint32_t K01[NP], K23[NP];
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        K01[p] = pk_make((256 * p + lane4 + 0) * gap_score, (256 * p + lane4 + 1) * gap_score);
        K23[p] = pk_make((256 * p + lane4 + 2) * gap_score, (256 * p + lane4 + 3) * gap_score);
    }
    const uint32_t ring_base = lds_addr(ring);
    const uint32_t read_base = lds_addr(lds_read);
    // guard store: lanes 0..15 write sentinel cells for columns band_end + 1 .. + 64, lane 16 the quad that ends in the
    // left-boundary slot (column band_start); byte offsets relative to the lane's own pass-0 offset
    // (lanes 17..63 take part in the same store -- masking them off costs two writes of EXEC per row, ~45 cycles for a lone
    // wavefront -- and write quads of columns band_start - 7 and below, lane l at byte 2 band_start - 16 - 8 (l - 17): no reader
    // of this row looks left of column band_start - 3, and the lowest of them, 191 columns down, still lies clear of the
    // sentinel cells when the slot wraps)
    const uint32_t guard_off  = lane < 16 ? (uint32_t)(2 * BW) : (lane == 16 ? (uint32_t)-136 : (uint32_t)(120 - 16 * lane));
    const bool is_lane16      = lane == 16;
    const bool is_last        = lane == kLastLanes - 1; // the band's last lane (of the last pass)
    const uint64_t last_mask  = 1ull << (kLastLanes - 1);
    const uint32_t S0X_MIN    = (uint32_t)min_score << 16;

    // state carried from row to row: the previous row's cells of every pass; row 0: H[0][x] = x * gap
    uint32_t P01[NP], P23[NP], rd4[NP], rd4n[NP], a1[NP];
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        P01[p]  = pk_make((256 * p + lane4 + 1) * gap_score, (256 * p + lane4 + 2) * gap_score);
        P23[p]  = pk_make((256 * p + lane4 + 3) * gap_score, (256 * p + lane4 + 4) * gap_score);
        rd4[p]  = lds_load_u32(read_base + 256 * p + lane4);
        rd4n[p] = lds_load_u32(read_base + 256 * p + lane4 + 4);
        a1[p]   = (uint32_t)(lane8 + 512 * p);
    }
    int32_t prev_rel0 = 0; // content of the boundary slot (relative index 0) of the row in P
    uint32_t ga       = (a1[0] + guard_off) & kMask;
    // per-lane pointers of the CURRENT row: its aligned quad in the HBM ring (slot row % H), its four move bytes in plane 0
    // (pass p: 512 / 256 bytes further)
    uint8_t* const ring_lane0 = reinterpret_cast<uint8_t*>(tb.ring) + lane8;
    uint8_t* ring_ptr         = ring_lane0;
    int32_t slot              = 0;
    uint8_t* move_ptr         = tb.plane0 + lane4 + (1 + kRelShift);
    bool unsupported          = false;

    auto owns = [&](int p) -> bool { return p < NP - 1 || kLastLanes == kWave || last_pass_lane; };
#pragma unroll
    for (int p = 0; p < NP; p++)
        if (owns(p)) lds_store_u64(ring_base + a1[p], P01[p], P23[p]);
    lds_store_guard(ring_base + ga, SENT2, is_lane16 ? (((uint32_t)kPkSentinel & 0xffffu) | (0u << 16)) : SENT2);

    // horizontal max-plus scan of the row's candidates, pass after pass; cu = carry-in as element t = -1 of u
    auto scan_row = [&](const uint32_t (&s01)[NP], const uint32_t (&s23)[NP], int32_t cu) {
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            const uint32_t u01 = pk_sub(s01[p], K01[p]), u23 = pk_sub(s23[p], K23[p]);
            const uint32_t pm01 = pk_max(u01, (u01 << 16) | 0x8000u);
            const uint32_t pm23 = pk_max(u23, (u23 << 16) | 0x8000u);
            const int32_t m3    = (int32_t)pk_max(pm01, pm23) >> 16;
            const int32_t incl  = wave_inclusive_max(m3);
            const int32_t excl  = max(wave_shr1(incl, cu), cu);
            const uint32_t ex2  = __builtin_amdgcn_perm((uint32_t)excl, (uint32_t)excl, 0x01000100u);
            const uint32_t m1b  = __builtin_amdgcn_perm(pm01, pm01, 0x03020302u);
            P01[p] = pk_add(pk_max(pm01, ex2), K01[p]);
            P23[p] = pk_add(pk_max(pk_max(pm23, m1b), ex2), K23[p]);
            // the next pass continues behind this one's last cell
            if (p + 1 < NP) cu = max(__builtin_amdgcn_readlane(incl, kWave - 1), cu);
        }
    };
    auto nz = [&](uint32_t a, uint32_t b) -> uint32_t { return pk_min_u16(pk_sub(a, b), ONE2); };
    // 0xffff where the halves are equal, 0 where they differ
    auto eqm = [&](uint32_t a, uint32_t b) -> uint32_t { return pk_sub(nz(a, b), ONE2); };
    auto bfi = [&](uint32_t m, uint32_t a, uint32_t b) -> uint32_t { return (m & a) | (~m & b); };
    auto costs = [&](uint32_t rd, uint32_t base4, uint32_t& c01, uint32_t& c23) {
        const uint32_t x   = rd ^ base4;
        const uint32_t x01 = __builtin_amdgcn_perm(0u, x, 0x0c010c00u);
        const uint32_t x23 = __builtin_amdgcn_perm(0u, x, 0x0c030c02u);
        c01 = pk_mad_u16(pk_min_u16(x01, ONE2), DIF2, MAT2);
        c23 = pk_mad_u16(pk_min_u16(x23, ONE2), DIF2, MAT2);
    };
    // one predecessor's candidate M = max(D, V) of the four cells and the move byte it stands for (2 d + 1 diagonal, 2 d
    // vertical; diagonal on a tie): q01/q23 = its cells of columns c+1..c+4, s0x = its cell of column c in the HIGH half
    auto from_pred = [&](uint32_t s0x, uint32_t q01, uint32_t q23, uint32_t c01, uint32_t c23, uint32_t d, uint32_t& M01, uint32_t& M23,
                         uint32_t& mv01, uint32_t& mv23) {
        const uint32_t D01 = pk_add(__builtin_amdgcn_alignbit(q01, s0x, 16), c01);
        const uint32_t D23 = pk_add(__builtin_amdgcn_alignbit(q23, q01, 16), c23);
        const uint32_t V01 = pk_add(q01, GAP2);
        const uint32_t V23 = pk_add(q23, GAP2);
        M01 = pk_max(D01, V01);
        M23 = pk_max(D23, V23);
        const uint32_t diag = pk_dup((int32_t)(2u * d + 1u));
        mv01 = pk_mad_u16_vvs(nz(M01, D01), NEG1, diag); // 2 d + 1 - [M != D]
        mv23 = pk_mad_u16_vvs(nz(M23, D23), NEG1, diag);
    };
    // the stored moves of a pass's four cells: the winner's move where the winner stands (H == s and s above min_score),
    // else horizontal
    auto final_moves = [&](int p, uint32_t s01, uint32_t s23, uint32_t w01, uint32_t w23) -> uint32_t {
        const uint32_t k01 = eqm(P01[p], s01) & pk_sub(0u, nz(s01, MIN2));
        const uint32_t k23 = eqm(P23[p], s23) & pk_sub(0u, nz(s23, MIN2));
        return __builtin_amdgcn_perm(bfi(k23, w23, ONE2), bfi(k01, w01, ONE2), 0x06040200u); // low byte of each half
    };
    // the finished row: LDS ring slot with its guard quad, HBM ring slot r % H, plane-0 bytes (mv4 per pass) and the
    // boundary cell's move byte (mv_boundary; rel0_val = content of the boundary slot)
    auto store_row = [&](int32_t r, int32_t rel0_val, const uint32_t (&mv4)[NP], uint32_t mv_boundary) {
        ring_ptr += stride * 2;
        slot++;
        if (slot == H)
        {
            slot     = 0;
            ring_ptr = ring_lane0;
        }
        move_ptr += stride;
        const uint32_t sbase = ring_base + (((uint32_t)r & (kSlots - 1)) * kSlotBytes);
        {
            // the boundary slot's real content behind the guard cells (a row without predecessors keeps gap_score there)
            const uint32_t rel0pk = ((uint32_t)kPkSentinel & 0xffffu) | ((uint32_t)rel0_val << 16);
            lds_store_guard(sbase + ga, SENT2, is_lane16 ? rel0pk : SENT2);
        }
        prev_rel0 = rel0_val;
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            // HBM ring, reference indices: the aligned quad rel 256 p + 4 l .. + 3 (the left neighbour's last cell first; lane 0:
            // the boundary slot, or the previous pass's last cell)
            const int32_t left0 = p == 0 ? (int32_t)((uint32_t)rel0_val << 16) : __builtin_amdgcn_readlane((int32_t)P23[p > 0 ? p - 1 : 0], kWave - 1);
            const uint32_t X    = (uint32_t)wave_shr1((int32_t)P23[p], left0);
            const uint32_t G01  = __builtin_amdgcn_alignbit(P01[p], X, 16);
            const uint32_t G23  = __builtin_amdgcn_alignbit(P23[p], P01[p], 16);
            if (owns(p))
            {
                lds_store_u64(sbase + a1[p], P01[p], P23[p]);
                *reinterpret_cast<uint2*>(ring_ptr + 512 * p)     = make_uint2(G01, G23);
                *reinterpret_cast<uint32_t*>(move_ptr + 256 * p) = mv4[p];
            }
        }
        gstore_hi16_lane_above8(last_mask, ring_ptr + 512 * (NP - 1), P23[NP - 1]); // the band's last cell
        gstore_u8_lane0_below(move_ptr, mv_boundary);
    };

    // ---------------- general row (kind 4): the reference's arithmetic in 32 bits, predecessors from registers / the HBM ring ----------------
    auto general_row = [&](int32_t r) {
        const RowInfo<true> ri = uniform_row(rowinfo[r]);
        const int32_t bs       = ri.bs();
        const uint32_t base    = (uint32_t)ri.base();
        const int32_t prev_bs  = r > 1 ? uniform_row(rowinfo[r - 1]).bs() : 0;
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            a1[p]   = (uint32_t)(2 * bs + lane8 + 512 * p) & kMask;
            rd4[p]  = lds_load_u32(read_base + bs + 256 * p + lane4);
            rd4n[p] = lds_load_u32(read_base + bs + 256 * p + lane4 + 4);
        }
        ga = (a1[0] + guard_off) & kMask;
        const int32_t pred_count = ri.cnt();
        int32_t R0[NP], R1[NP], R2[NP], R3[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) { R0[p] = pk_lo(P01[p]); R1[p] = pk_hi(P01[p]); R2[p] = pk_lo(P23[p]); R3[p] = pk_hi(P23[p]); }
        bool synced = false;
        const int32_t node_id = (pred_count > 3) ? (int32_t)g.sorted_poa[r - 1] : 0;
        auto pred_row = [&](int32_t p) -> int32_t {
            if (pred_count == 0) return 0;
            if (p < 3) return ri.pred(p);
            return wave_first((int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1);
        };
        // get_score_tb(row, -1) :119-138: the boundary slot of that row's ring slot
        auto rel0_of = [&](int32_t prow) -> int32_t {
            if (prow == r - 1) return prev_rel0;
            if (!synced) { wave_sync(); synced = true; }
            return wave_first((int32_t)tb.ring[(int64_t)(prow % H) * stride]);
        };
        // boundary :362-434
        int32_t fe = 0, rel0_val = min_score, trace0 = 0;
        bool stored_fe = false;
        if (pred_count == 0)
        {
            rel0_val = gap_score;     // scores[index] = gap_score, whatever the band start
            trace0   = kTbVertToRow0; // -r
        }
        else
        {
            const int32_t p0 = pred_row(0);
            trace0           = -(r - p0); // <= kTbMaxDelta: checked before the pass
            if (bs > kCellsPerLane && pred_count == 1)
                fe = min_score + gap_score;
            else
            {
                int32_t penalty = max(min_score, rel0_of(p0));
                for (int32_t p = 1; p < pred_count; p++)
                {
                    const int32_t pit = pred_row(p);
                    if ((r - pit) < H)
                    {
                        const int32_t st = rel0_of(pit);
                        if (penalty < st)
                        {
                            penalty = st;
                            trace0  = -(r - pit);
                            if (r - pit > kTbMaxDelta) unsupported = true;
                        }
                    }
                }
                fe        = penalty + gap_score;
                stored_fe = true;
                if (bs == 0) rel0_val = fe;
            }
        }
        // the stray store of set_score_tb(column = -1) :47-69 lands at relative index band_start of the ring row: inside the row
        // itself it is overwritten by the row's own cells (or is the boundary slot), beyond it it hits another ring row
        if (stored_fe && bs >= stride && lane == 0)
        {
            const int64_t idx = (int64_t)bs + (int64_t)(r % H) * stride;
            if ((size_t)idx < tb.ring_elems) tb.ring[idx] = (int16_t)fe;
        }
        uint32_t s01[NP], s23[NP];
        int32_t sv[NP][4], tv[NP][4];
#pragma unroll
        for (int ps = 0; ps < NP; ps++)
        {
            const int32_t c   = bs + 256 * ps + lane4;
            const uint32_t rd = rd4[ps];
            const int32_t cp[4] = {((rd & 0xff) == base) ? match_score : mismatch_score, (((rd >> 8) & 0xff) == base) ? match_score : mismatch_score,
                                   (((rd >> 16) & 0xff) == base) ? match_score : mismatch_score, ((rd >> 24) == base) ? match_score : mismatch_score};
#pragma unroll
            for (int k = 0; k < 4; k++) { sv[ps][k] = min_score; tv[ps][k] = 0; }
            const int32_t np = max(pred_count, 1);
            for (int32_t p = 0; p < np; p++)
            {
                const int32_t prow = pred_row(p);
                if (p > 0 && !((r - prow) < H)) continue; // :463
                const int32_t delta = r - prow;
                if (pred_count != 0 && delta > kTbMaxDelta) unsupported = true;
                const int32_t t_diag = pred_count == 0 ? kTbDiagToRow0 : delta, t_vert = pred_count == 0 ? kTbVertToRow0 : -delta;
                int32_t S[5];
                bool valid;
                if (prow == r - 1)
                {
                    // the previous row is in registers: virtual lane v of its quads (pass 0: 0..63, pass 1: 64..127); every lane
                    // takes part in the shuffles
                    const int32_t pend = min(prev_bs + band_width - kCellsPerLane, max_column);
                    valid              = c <= pend;
                    const int32_t v    = (c - prev_bs) >> 2;
                    auto vget = [&](const int32_t (&R)[NP], int32_t vl) -> int32_t {
                        int32_t x = __shfl(R[0], vl & 63);
                        if constexpr (NP == 2)
                        {
                            const int32_t y = __shfl(R[1], vl & 63);
                            x               = (vl & 64) ? y : x;
                        }
                        return x;
                    };
                    const int32_t left = vget(R3, v - 1);
                    S[0] = v == 0 ? prev_rel0 : left;
                    S[1] = vget(R0, v); S[2] = vget(R1, v); S[3] = vget(R2, v); S[4] = vget(R3, v);
                }
                else
                {
                    const int32_t pbs  = prow == 0 ? 0 : uniform_row(rowinfo[prow]).bs();
                    const int32_t pend = min(pbs + band_width - kCellsPerLane, max_column);
                    valid              = !(c > pend || c < pbs);
                    if (!synced) { wave_sync(); synced = true; }
                    const int16_t* pp = tb.ring + (int64_t)(prow % H) * stride + (valid ? (c - pbs) : 0);
#pragma unroll
                    for (int k = 0; k < 5; k++) S[k] = pp[k];
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const int32_t d = S[k] + cp[k], vv = S[k + 1] + gap_score;
                    const bool take_d = valid && d >= vv && d > sv[ps][k], take_v = valid && d < vv && vv > sv[ps][k];
                    tv[ps][k] = take_d ? t_diag : (take_v ? t_vert : tv[ps][k]);
                    sv[ps][k] = take_d ? d : (take_v ? vv : sv[ps][k]);
                }
            }
            s01[ps] = pk_make(sv[ps][0], sv[ps][1]);
            s23[ps] = pk_make(sv[ps][2], sv[ps][3]);
        }
        scan_row(s01, s23, fe + gap_score);
        // stores: plane 0 all zero (the walk takes this row step by step), plane 1 exact
        const uint32_t zero4[NP] = {};
        store_row(r, rel0_val, zero4, 0u);
        uint8_t* p1 = tb.plane1 + (move_ptr - tb.plane0);
#pragma unroll
        for (int ps = 0; ps < NP; ps++)
        {
            // horizontal strictly greater: trace 0 (:476-513)
            const int32_t t0 = pk_lo(P01[ps]) != sv[ps][0] ? 0 : tv[ps][0], t1 = pk_hi(P01[ps]) != sv[ps][1] ? 0 : tv[ps][1];
            const int32_t t2 = pk_lo(P23[ps]) != sv[ps][2] ? 0 : tv[ps][2], t3 = pk_hi(P23[ps]) != sv[ps][3] ? 0 : tv[ps][3];
            const uint32_t t4 = ((uint32_t)t0 & 0xffu) | (((uint32_t)t1 & 0xffu) << 8) | (((uint32_t)t2 & 0xffu) << 16) | ((uint32_t)t3 << 24);
            if (owns(ps)) *reinterpret_cast<uint32_t*>(p1 + 256 * ps) = t4;
        }
        gstore_u8_lane0_below(p1, (uint32_t)trace0 & 0xffu);
    };

    // ---------------- kinds 0 / 1: one predecessor, the previous row, in registers ----------------
    auto reg_row = [&](auto bs0_tag, auto moved_tag, int32_t r, uint32_t d0, uint32_t base4) {
        constexpr bool BS0   = decltype(bs0_tag)::value;
        constexpr bool MOVED = decltype(moved_tag)::value;
        uint32_t s0x[NP], q01[NP], q23[NP];
        if constexpr (!MOVED)
        {
#pragma unroll
            for (int p = 0; p < NP; p++)
            {
                const int32_t left0 = p == 0 ? (BS0 ? (int32_t)((uint32_t)prev_rel0 << 16) : (int32_t)S0X_MIN)
                                             : __builtin_amdgcn_readlane((int32_t)P23[p > 0 ? p - 1 : 0], kWave - 1);
                s0x[p] = (uint32_t)wave_shr1((int32_t)P23[p], left0);
                q01[p] = P01[p];
                q23[p] = P23[p];
            }
        }
        else
        {
            // band moved by one quad: the lane's columns are those of the next lane of the previous row; a pass's last lane
            // takes the next pass's first quad, the band's last lane the sentinel
#pragma unroll
            for (int p = 0; p < NP; p++)
            {
                s0x[p] = P23[p];
                const int32_t n01 = p + 1 < NP ? __builtin_amdgcn_readlane((int32_t)P01[p + 1 < NP ? p + 1 : p], 0) : (int32_t)pk_dup(kPkSentinel);
                const int32_t n23 = p + 1 < NP ? __builtin_amdgcn_readlane((int32_t)P23[p + 1 < NP ? p + 1 : p], 0) : (int32_t)pk_dup(kPkSentinel);
                q01[p] = (uint32_t)wave_shl1((int32_t)P01[p], n01);
                q23[p] = (uint32_t)wave_shl1((int32_t)P23[p], n23);
                a1[p]  = (a1[p] + 8) & kMask;
                rd4[p] = rd4n[p];
                rd4n[p] = lds_load_u32(read_base + (((d0 >> 3) & 0x1ffu) << 2) + 256 * p + lane4 + 4);
            }
            ga = (ga + 8) & kMask;
        }
        int32_t cu = min_score + 2 * gap_score, rel0_val = min_score;
        if constexpr (BS0)
        {
            rel0_val = max(min_score, prev_rel0) + gap_score;
            cu       = rel0_val + gap_score;
        }
        uint32_t s01[NP], s23[NP], w01[NP], w23[NP];
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            uint32_t c01, c23, M01, M23;
            costs(rd4[p], base4, c01, c23);
            from_pred(s0x[p], q01[p], q23[p], c01, c23, 1u, M01, M23, w01[p], w23[p]);
            if (MOVED && p == NP - 1) // the band's last lane: the chunk beyond the previous row's band
            {
                M01 = is_last ? MIN2 : M01;
                M23 = is_last ? MIN2 : M23;
            }
            s01[p] = pk_max(M01, MIN2);
            s23[p] = pk_max(M23, MIN2);
        }
        scan_row(s01, s23, cu);
        uint32_t mv4[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) mv4[p] = final_moves(p, s01[p], s23[p], w01[p], w23[p]);
        store_row(r, rel0_val, mv4, 2u /* vertical, one row up */);
    };

    // ---------------- the rows of one phase ----------------
    auto run_rows = [&](auto bs0_tag, int32_t r_from, int32_t r_to) {
        constexpr bool BS0 = decltype(bs0_tag)::value;
        int32_t r = r_from;
        while (r <= r_to)
        {
            const int32_t r0 = r;
            uint32_t D0v, D1v;
            {
                const int32_t rr    = min(r0 + lane, graph_count);
                const uint64_t w    = rowinfo[rr].w;
                const uint32_t kind = (uint32_t)(w >> kKindShift) & 7u;
                const uint32_t cnt  = (uint32_t)(w >> 8) & 0x3fu;
                const uint32_t bs4  = (uint32_t)(w >> 15) & 0x1ffu;
                const uint32_t p0 = (uint32_t)(w >> 24) & 0xfffu, p1 = (uint32_t)(w >> 36) & 0xfffu, p2 = (uint32_t)(w >> 48) & 0xfffu;
                constexpr uint32_t sm = kSlots - 1;
                const uint32_t slots = (p0 & sm) | ((p1 & sm) << 3) | ((p2 & sm) << 6);
                const uint32_t dists = (((uint32_t)rr - p0) & 7u) | ((((uint32_t)rr - p1) & 7u) << 3) | ((((uint32_t)rr - p2) & 7u) << 6);
                D0v = kind | (bs4 << 3) | (slots << 12) | (dists << 21) | ((cnt <= 3 ? cnt : 0u) << 30);
                D1v = ((uint32_t)w & 0xffu) * 0x01010101u;
            }
            D0v = (r0 + lane <= r_to) ? D0v : 7u;
            int32_t k      = 0;
            uint32_t d0    = (uint32_t)__builtin_amdgcn_readlane((int32_t)D0v, 0);
            uint32_t base4 = (uint32_t)__builtin_amdgcn_readlane((int32_t)D1v, 0);
            uint32_t kind  = d0 & 7u;
            auto advance = [&]() {
                r++;
                k++;
                const uint32_t nd = (uint32_t)__builtin_amdgcn_readlane((int32_t)D0v, k & (kWave - 1));
                base4 = (uint32_t)__builtin_amdgcn_readlane((int32_t)D1v, k & (kWave - 1));
                d0    = k == kWave ? 7u : nd;
                kind  = d0 & 7u;
            };
            for (;;)
            {
                while (kind == 0)
                {
                    reg_row(bs0_tag, std::false_type{}, r, d0, base4);
                    advance();
                }
                if (kind == 7u) break;
                if (kind == 1)
                    reg_row(bs0_tag, std::true_type{}, r, d0, base4);
                else if (kind <= 3)
                {
                    // ===== predecessors from the LDS ring =====
                    const uint32_t bs = ((d0 >> 3) & 0x1ffu) << 2;
                    uint32_t a0[NP];
#pragma unroll
                    for (int p = 0; p < NP; p++)
                    {
                        a1[p] = (2u * bs + (uint32_t)lane8 + 512u * p) & kMask;
                        a0[p] = (a1[p] - 4) & kMask; // dword whose high half is the cell of column c
                    }
                    ga = (a1[0] + guard_off) & kMask;
                    const uint32_t sent16 = (uint32_t)kPkSentinel & 0xffffu;
                    auto rel0_in_slot = [&](uint32_t b) -> int32_t {
                        return (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(b + kSlotBytes - 4) >> 16));
                    };
                    // the predecessors of the row: ring slot base and rows up; slots 0..2 from the descriptor, 3..5 from the side
                    // table. One straight-line instantiation per count 1 / 2 / 3, one with run-time tests for 4..6.
                    const uint32_t cnt3 = d0 >> 30; // 1 (kind 2), 2, 3, or 0 = more than three
                    auto ring_row = [&](auto n_tag) {
                        constexpr int N = decltype(n_tag)::value; // 1, 2, 3: exactly N predecessors; 6: four to six
                        int32_t cnt_all = N;
                        uint64_t xe     = 0;
                        if constexpr (N == 6)
                        {
                            xe      = wave_first64(xpred[r & 255]);
                            cnt_all = (int32_t)((xe >> 13) & 63u);
                        }
                        uint32_t pb[N], pd[N];
#pragma unroll
                        for (int q = 0; q < N; q++)
                        {
                            if (q < 3)
                            {
                                pb[q] = ring_base + (((d0 >> (12 + 3 * q)) & 7u) * kSlotBytes);
                                pd[q] = (d0 >> (21 + 3 * q)) & 7u;
                            }
                            else
                            {
                                const uint32_t prow = q < cnt_all ? (uint32_t)xpred_row(xe, q) : 0u;
                                pb[q] = ring_base + ((prow & (kSlots - 1)) * kSlotBytes);
                                pd[q] = ((uint32_t)r - prow) & 7u;
                            }
                        }
                        // all loads first (one LDS round trip), then the arithmetic
                        uint32_t xq[N][NP];
                        uint2 qq[N][NP];
#pragma unroll
                        for (int q = 0; q < N; q++)
                            if (q < 4 || q < cnt_all)
                            {
#pragma unroll
                                for (int p = 0; p < NP; p++)
                                {
                                    xq[q][p] = lds_load_u32(pb[q] + a0[p]);
                                    qq[q][p] = lds_load_u64(pb[q] + a1[p]);
                                }
                            }
#pragma unroll
                        for (int p = 0; p < NP; p++)
                        {
                            rd4[p]  = lds_load_u32(read_base + bs + 256 * p + lane4);
                            rd4n[p] = lds_load_u32(read_base + bs + 256 * p + lane4 + 4);
                        }
                        // boundary :362-434: the first strict maximum over the predecessors' boundary slots names the boundary trace
                        int32_t fe = min_score + gap_score, rel0_val = min_score;
                        uint32_t mv_boundary = 2u * pd[0];
                        if constexpr (BS0)
                        {
                            int32_t pen = max(min_score, rel0_in_slot(pb[0]));
#pragma unroll
                            for (int q = 1; q < N; q++)
                                if (q < 4 || q < cnt_all)
                                {
                                    const int32_t st = rel0_in_slot(pb[q]);
                                    mv_boundary      = pen < st ? 2u * pd[q] : mv_boundary;
                                    pen              = max(pen, st);
                                }
                            fe       = pen + gap_score;
                            rel0_val = fe;
                        }
                        else if (N > 1 && (int32_t)bs >= stride && lane == 0)
                        {
                            // the stray boundary store of a row with several predecessors (see general_row)
                            const int64_t idx = (int64_t)bs + (int64_t)(r % H) * stride;
                            if ((size_t)idx < tb.ring_elems) tb.ring[idx] = (int16_t)fe;
                        }
                        uint32_t s01[NP], s23[NP], W01[NP], W23[NP];
                        uint32_t M01[N][NP], M23[N][NP], w01[N][NP], w23[N][NP];
#pragma unroll
                        for (int p = 0; p < NP; p++)
                        {
                            uint32_t c01, c23;
                            costs(rd4[p], base4, c01, c23);
                            s01[p] = MIN2;
                            s23[p] = MIN2;
#pragma unroll
                            for (int q = 0; q < N; q++)
                                if (q < 4 || q < cnt_all)
                                {
                                    from_pred(xq[q][p], qq[q][p].x, qq[q][p].y, c01, c23, pd[q], M01[q][p], M23[q][p], w01[q][p], w23[q][p]);
                                    // a chunk beyond the predecessor's band is skipped (only the last pass can be: the band starts of a
                                    // row and its ring predecessors differ by at most 60 columns)
                                    if (p == NP - 1)
                                    {
                                        const bool outside = (qq[q][p].x & 0xffffu) == sent16;
                                        M01[q][p] = outside ? MIN2 : M01[q][p];
                                        M23[q][p] = outside ? MIN2 : M23[q][p];
                                    }
                                    s01[p] = pk_max(s01[p], M01[q][p]);
                                    s23[p] = pk_max(s23[p], M23[q][p]);
                                }
                            // the first predecessor that attains the maximum: apply the candidates last to first
                            W01[p] = ONE2;
                            W23[p] = ONE2;
#pragma unroll
                            for (int q = N - 1; q >= 0; q--)
                                if (q < 4 || q < cnt_all)
                                {
                                    W01[p] = bfi(eqm(M01[q][p], s01[p]), w01[q][p], W01[p]);
                                    W23[p] = bfi(eqm(M23[q][p], s23[p]), w23[q][p], W23[p]);
                                }
                        }
                        scan_row(s01, s23, fe + gap_score);
                        uint32_t mv4[NP];
#pragma unroll
                        for (int p = 0; p < NP; p++) mv4[p] = final_moves(p, s01[p], s23[p], W01[p], W23[p]);
                        store_row(r, rel0_val, mv4, mv_boundary);
                    };
                    if (cnt3 == 1) ring_row(std::integral_constant<int, 1>{});
                    else if (cnt3 == 2) ring_row(std::integral_constant<int, 2>{});
                    else if (cnt3 == 3) ring_row(std::integral_constant<int, 3>{});
                    else ring_row(std::integral_constant<int, 6>{});
                }
                else
                {
                    general_row(r);
                }
                advance();
            }
        }
    };

    const int32_t bs0_end = min(first_moved - 1, graph_count);
    run_rows(std::true_type{}, 1, bs0_end);
    run_rows(std::false_type{}, bs0_end + 1, graph_count);
    return __ballot(unsupported) == 0;
}

} // namespace gwhip
