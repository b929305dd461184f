// poa_forward_moves_tb.h -- the packed forward pass of poa_forward_moves.h for the traceback-buffer modes (round 4):
// cudapoa_nw_tb_banded.cuh:264-531 (restated in oracle/poa_nw_tb.inc) for the 256- and 128-column band with int16 scores.
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
//     transitions, predecessors more than 7 rows up), whose plane-0 bytes are 0 so that the walk takes them one step
//     at a time.
//   * Anything the byte planes cannot hold (a predecessor 128 or more rows up) or that depends on ring-slot aliasing (a
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
    static_assert(BW == 128 || BW == 256, "band widths of the packed pass");
    constexpr int32_t band_width = BW;
    constexpr int32_t stride     = band_width + kRightPad;
    constexpr int kBandLanes     = BW / kCellsPerLane;
    const int lane               = threadIdx.x & (kWave - 1);
    const bool band_lane         = lane < kBandLanes;
    const int32_t lane4 = lane * 4, lane8 = lane * 8;
    const int32_t min_score = Limits<int16_t>::min / 2;
    const int32_t H         = tb.H;

    const int32_t first_moved = classify_kinds<kPkMaxDist, true>(rowinfo, graph_count, lane, xpred, dbg);
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
    const uint32_t K01 = pk_make((lane4 + 0) * gap_score, (lane4 + 1) * gap_score);
    const uint32_t K23 = pk_make((lane4 + 2) * gap_score, (lane4 + 3) * gap_score);
    const uint32_t ring_base = lds_addr(ring);
    const uint32_t read_base = lds_addr(lds_read);
    const uint32_t guard_off  = lane < 16 ? (uint32_t)(2 * BW) : (uint32_t)-136;
    const bool is_lane16      = lane == 16;
    const bool is_last        = lane == kBandLanes - 1; // the band's last lane
    const uint64_t last_mask  = 1ull << (kBandLanes - 1);
    const uint32_t S0X_MIN      = (uint32_t)min_score << 16;

    uint32_t P01 = pk_make((lane4 + 1) * gap_score, (lane4 + 2) * gap_score); // row 0: H[0][x] = x * gap
    uint32_t P23 = pk_make((lane4 + 3) * gap_score, (lane4 + 4) * gap_score);
    int32_t prev_rel0 = 0; // content of the boundary slot (relative index 0) of the row in P
    uint32_t rd4  = lds_load_u32(read_base + lane4);
    uint32_t rd4n = lds_load_u32(read_base + lane4 + 4);
    uint32_t a1   = (uint32_t)lane8;
    uint32_t ga   = (a1 + guard_off) & (kPkSlotBytes - 1);
    // per-lane pointers of the CURRENT row: its aligned quad in the HBM ring (slot row % H), its four move bytes in plane 0
    uint8_t* const ring_lane0 = reinterpret_cast<uint8_t*>(tb.ring) + lane8;
    uint8_t* ring_ptr         = ring_lane0;
    int32_t slot              = 0;
    uint8_t* move_ptr         = tb.plane0 + lane4 + (1 + kRelShift);
    bool unsupported          = false;

    if (BW == 256 || band_lane) lds_store_u64(ring_base + a1, P01, P23);
    lds_store_u64_lanes17(ring_base + ga, SENT2, is_lane16 ? (((uint32_t)kPkSentinel & 0xffffu) | (0u << 16)) : SENT2);

    auto scan_row = [&](uint32_t s01, uint32_t s23, int32_t cu) {
        const uint32_t u01 = pk_sub(s01, K01), u23 = pk_sub(s23, K23);
        const uint32_t pm01 = pk_max(u01, (u01 << 16) | 0x8000u);
        const uint32_t pm23 = pk_max(u23, (u23 << 16) | 0x8000u);
        const int32_t m3    = (int32_t)pk_max(pm01, pm23) >> 16;
        const int32_t incl  = wave_inclusive_max(m3);
        const int32_t excl  = max(wave_shr1(incl, cu), cu);
        const uint32_t ex2  = __builtin_amdgcn_perm((uint32_t)excl, (uint32_t)excl, 0x01000100u);
        const uint32_t m1b  = __builtin_amdgcn_perm(pm01, pm01, 0x03020302u);
        P01 = pk_add(pk_max(pm01, ex2), K01);
        P23 = pk_add(pk_max(pk_max(pm23, m1b), ex2), K23);
    };
    auto nz = [&](uint32_t a, uint32_t b) -> uint32_t { return pk_min_u16(pk_sub(a, b), ONE2); };
    // 0xffff where the halves are equal, 0 where they differ
    auto eqm = [&](uint32_t a, uint32_t b) -> uint32_t { return pk_sub(nz(a, b), ONE2); };
    auto bfi = [&](uint32_t m, uint32_t a, uint32_t b) -> uint32_t { return (m & a) | (~m & b); };
    auto costs = [&](uint32_t base4, uint32_t& c01, uint32_t& c23) {
        const uint32_t x   = rd4 ^ base4;
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
    // the stored move of a cell: the winner's move where the winner stands (H == s and s above min_score), else horizontal
    auto final_moves = [&](uint32_t s01, uint32_t s23, uint32_t w01, uint32_t w23) -> uint32_t {
        const uint32_t k01 = eqm(P01, s01) & pk_sub(0u, nz(s01, MIN2));
        const uint32_t k23 = eqm(P23, s23) & pk_sub(0u, nz(s23, MIN2));
        return __builtin_amdgcn_perm(bfi(k23, w23, ONE2), bfi(k01, w01, ONE2), 0x06040200u); // low byte of each half
    };
    // the finished row: LDS ring slot r & 7 with its guard quad, HBM ring slot r % H, plane-0 bytes (mv4) and the
    // boundary cell's move byte (mv_boundary; rel0_val = content of the boundary slot)
    auto store_row = [&](int32_t r, int32_t rel0_val, uint32_t mv4, uint32_t mv_boundary) {
        ring_ptr += stride * 2;
        slot++;
        if (slot == H)
        {
            slot     = 0;
            ring_ptr = ring_lane0;
        }
        move_ptr += stride;
        const uint32_t sbase = ring_base + (((uint32_t)r & (kPkSlots - 1)) * kPkSlotBytes);
        if (BW == 256 || band_lane) lds_store_u64(sbase + a1, P01, P23);
        {
            // the boundary slot's real content behind the guard cells (a row without predecessors keeps gap_score there)
            const uint32_t rel0pk = ((uint32_t)kPkSentinel & 0xffffu) | ((uint32_t)rel0_val << 16);
            lds_store_u64_lanes17(sbase + ga, SENT2, is_lane16 ? rel0pk : SENT2);
        }
        prev_rel0 = rel0_val;
        // HBM ring, reference indices: the aligned quad rel 4 l .. 4 l + 3, then the band's last cell
        const uint32_t X   = (uint32_t)wave_shr1((int32_t)P23, (int32_t)((uint32_t)rel0_val << 16));
        const uint32_t G01 = __builtin_amdgcn_alignbit(P01, X, 16);
        const uint32_t G23 = __builtin_amdgcn_alignbit(P23, P01, 16);
        if (BW == 256 || band_lane) *reinterpret_cast<uint2*>(ring_ptr) = make_uint2(G01, G23);
        gstore_hi16_lane_above8(last_mask, ring_ptr, P23);
        if (BW == 256 || band_lane) *reinterpret_cast<uint32_t*>(move_ptr) = mv4;
        gstore_u8_lane0_below(move_ptr, mv_boundary);
    };

    // ---------------- general row (kind 4): the reference's arithmetic in 32 bits, predecessors from registers / the HBM ring ----------------
    auto general_row = [&](int32_t r) {
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
            rel0_val = gap_score; // scores[index] = gap_score, whatever the band start
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
        int32_t s0 = min_score, s1 = min_score, s2 = min_score, s3 = min_score;
        int32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        const int32_t np = max(pred_count, 1);
        for (int32_t p = 0; p < np; p++)
        {
            const int32_t prow = pred_row(p);
            if (p > 0 && !((r - prow) < H)) continue; // :463
            const int32_t delta = r - prow;
            if (pred_count != 0 && delta > kTbMaxDelta) unsupported = true;
            const int32_t t_diag = pred_count == 0 ? kTbDiagToRow0 : delta, t_vert = pred_count == 0 ? kTbVertToRow0 : -delta;
            int32_t S0, S1, S2, S3, S4;
            bool valid;
            if (prow == r - 1)
            {
                const int32_t q    = (bs - prev_bs) >> 2;
                const int32_t pend = min(prev_bs + band_width - kCellsPerLane, max_column);
                const int src      = lane + q;
                const int32_t from_left = __shfl(R3, src - 1); // every lane active
                S0 = (q == 0 && lane == 0) ? prev_rel0 : from_left;
                S1 = __shfl(R0, src); S2 = __shfl(R1, src); S3 = __shfl(R2, src); S4 = __shfl(R3, src);
                valid = c <= pend;
            }
            else
            {
                const int32_t pbs  = prow == 0 ? 0 : uniform_row(rowinfo[prow]).bs();
                const int32_t pend = min(pbs + band_width - kCellsPerLane, max_column);
                valid              = !(c > pend || c < pbs);
                if (!synced) { wave_sync(); synced = true; }
                const int16_t* ps = tb.ring + (int64_t)(prow % H) * stride + (valid ? (c - pbs) : 0);
                S0 = ps[0]; S1 = ps[1]; S2 = ps[2]; S3 = ps[3]; S4 = ps[4];
            }
            auto cell = [&](int32_t Sa, int32_t Sb, int32_t cp, int32_t& s, int32_t& t) {
                const int32_t d = Sa + cp, v = Sb + gap_score;
                const bool take_d = valid && d >= v && d > s, take_v = valid && d < v && v > s;
                t = take_d ? t_diag : (take_v ? t_vert : t);
                s = take_d ? d : (take_v ? v : s);
            };
            cell(S0, S1, cp0, s0, t0); cell(S1, S2, cp1, s1, t1); cell(S2, S3, cp2, s2, t2); cell(S3, S4, cp3, s3, t3);
        }
        const uint32_t s01 = pk_make(s0, s1), s23 = pk_make(s2, s3);
        scan_row(s01, s23, fe + gap_score);
        // horizontal strictly greater: trace 0 (:476-513)
        t0 = pk_lo(P01) != s0 ? 0 : t0; t1 = pk_hi(P01) != s1 ? 0 : t1; t2 = pk_lo(P23) != s2 ? 0 : t2; t3 = pk_hi(P23) != s3 ? 0 : t3;
        const uint32_t t4 = ((uint32_t)t0 & 0xffu) | (((uint32_t)t1 & 0xffu) << 8) | (((uint32_t)t2 & 0xffu) << 16) | ((uint32_t)t3 << 24);
        // stores: plane 0 all zero (the walk takes this row step by step), plane 1 exact
        store_row(r, rel0_val, 0u, 0u);
        uint8_t* p1 = tb.plane1 + (move_ptr - tb.plane0);
        if (BW == 256 || band_lane) *reinterpret_cast<uint32_t*>(p1) = t4;
        gstore_u8_lane0_below(p1, (uint32_t)trace0 & 0xffu);
    };

    // ---------------- kinds 0 / 1: one predecessor, the previous row, in registers ----------------
    auto reg_row = [&](auto bs0_tag, auto moved_tag, int32_t r, uint32_t d0, uint32_t base4) {
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
        uint32_t c01, c23, M01, M23, w01, w23;
        costs(base4, c01, c23);
        from_pred(s0x, q01, q23, c01, c23, 1u, M01, M23, w01, w23);
        if constexpr (MOVED) // the band's last lane: the chunk beyond the previous row's band
        {
            M01 = is_last ? MIN2 : M01;
            M23 = is_last ? MIN2 : M23;
        }
        const uint32_t s01 = pk_max(M01, MIN2), s23 = pk_max(M23, MIN2);
        scan_row(s01, s23, cu);
        store_row(r, rel0_val, final_moves(s01, s23, w01, w23), 2u /* vertical, one row up */);
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
                const uint32_t slots = (p0 & 7u) | ((p1 & 7u) << 3) | ((p2 & 7u) << 6);
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
                    a1 = (2u * bs + (uint32_t)lane8) & (kPkSlotBytes - 1);
                    ga = (a1 + guard_off) & (kPkSlotBytes - 1);
                    const uint32_t a0 = (a1 - 4) & (kPkSlotBytes - 1);
                    const uint32_t sent16 = (uint32_t)kPkSentinel & 0xffffu;
                    auto rel0_in_slot = [&](uint32_t b) -> int32_t {
                        return (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(b + kPkSlotBytes - 4) >> 16));
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
                                pb[q] = ring_base + (((d0 >> (12 + 3 * q)) & 7u) * kPkSlotBytes);
                                pd[q] = (d0 >> (21 + 3 * q)) & 7u;
                            }
                            else
                            {
                                const uint32_t prow = q < cnt_all ? (uint32_t)xpred_row(xe, q) : 0u;
                                pb[q] = ring_base + ((prow & 7u) * kPkSlotBytes);
                                pd[q] = ((uint32_t)r - prow) & 7u;
                            }
                        }
                        // all loads first (one LDS round trip), then the arithmetic
                        uint32_t xq[N];
                        uint2 qq[N];
#pragma unroll
                        for (int q = 0; q < N; q++)
                            if (q < 4 || q < cnt_all)
                            {
                                xq[q] = lds_load_u32(pb[q] + a0);
                                qq[q] = lds_load_u64(pb[q] + a1);
                            }
                        rd4  = lds_load_u32(read_base + bs + lane4);
                        rd4n = lds_load_u32(read_base + bs + lane4 + 4);
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
                        uint32_t c01, c23;
                        costs(base4, c01, c23);
                        uint32_t M01[N], M23[N], w01[N], w23[N];
                        uint32_t s01 = MIN2, s23 = MIN2;
#pragma unroll
                        for (int q = 0; q < N; q++)
                            if (q < 4 || q < cnt_all)
                            {
                                from_pred(xq[q], qq[q].x, qq[q].y, c01, c23, pd[q], M01[q], M23[q], w01[q], w23[q]);
                                const bool outside = (qq[q].x & 0xffffu) == sent16; // chunk beyond the predecessor's band: skipped
                                M01[q] = outside ? MIN2 : M01[q];
                                M23[q] = outside ? MIN2 : M23[q];
                                s01    = pk_max(s01, M01[q]);
                                s23    = pk_max(s23, M23[q]);
                            }
                        // the first predecessor that attains the maximum: apply the candidates last to first
                        uint32_t W01 = ONE2, W23 = ONE2;
#pragma unroll
                        for (int q = N - 1; q >= 0; q--)
                            if (q < 4 || q < cnt_all)
                            {
                                W01 = bfi(eqm(M01[q], s01), w01[q], W01);
                                W23 = bfi(eqm(M23[q], s23), w23[q], W23);
                            }
                        scan_row(s01, s23, fe + gap_score);
                        store_row(r, rel0_val, final_moves(s01, s23, W01, W23), mv_boundary);
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
